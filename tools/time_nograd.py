"""Forward-only (VSL_FLAG_NO_GRAD) against the full evaluation: tile-kernel time for all heads of a launch (GPU box).
usage: time_nograd.py [B H W heads]   (default 12 192 640 4)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import mde_amd._abi as abi

B, H, W, heads = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (12, 192, 640, 4)))
fids = [0, -1, 1]
dev = torch.device("cuda:0")
data, t = bench.build_inputs(B, H, W, fids, heads, dev)
step = bench.LossStep(t, B, H, W, fids, heads, dev)
stream_obj = torch.cuda.current_stream()
stream = stream_obj.cuda_stream
bench.settle(step, stream, stream_obj, dev, 0.6)          # clocks of a cold process
for flags in (abi.FLAG_AUTOMASK, abi.FLAG_AUTOMASK | abi.FLAG_NO_GRAD, abi.FLAG_AUTOMASK, abi.FLAG_AUTOMASK | abi.FLAG_NO_GRAD):
    step.descs[0].flags = flags
    ms = ctypes.c_float()
    abi.check(abi.lib.vsl_debug_time_fused(ctypes.byref(step.descs[0]), step.heads, heads, 100, ctypes.byref(ms), stream))
    print("%dx%d b%d, %d heads, flags %d: tile kernel %.1f us" % (W, H, B, heads, flags, ms.value * 1e3))
