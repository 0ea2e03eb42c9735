"""Forward-only (VSL_FLAG_NO_GRAD) against the full evaluation: fused-kernel time per head (GPU box)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import mde_amd._abi as abi

B, H, W, fids = 12, 192, 640, [0, -1, 1]
dev = torch.device("cuda:0")
data, t = bench.build_inputs(B, H, W, fids, 4, dev)
step = bench.LossStep(t, B, H, W, fids, 4, dev)
stream = torch.cuda.current_stream().cuda_stream
step(stream)
for flags in (abi.FLAG_AUTOMASK, abi.FLAG_AUTOMASK | abi.FLAG_NO_GRAD):
    step.descs[0].flags = flags
    ms = ctypes.c_float()
    abi.check(abi.lib.vsl_debug_time_fused(ctypes.byref(step.descs[0]), step.heads, 4, 50, ctypes.byref(ms), stream))
    print("flags %d: fused kernel %.1f us for 4 heads" % (flags, ms.value * 1e3))
