"""Diagnostic (GPU box): per-phase wave time of the fused kernel from in-kernel clock stamps.
Needs tools/libvsloss_stamps.so (built with -DVSL_STAMPS)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["VSL_LIBRARY"] = os.path.join(ROOT, "tools", "libvsloss_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import mde_amd._abi as abi

B, H, W = [int(v) for v in os.environ.get('VSL_STAMP_SHAPE', '12,192,640').split(',')]
fids = [0, -1, 1, 's'] if os.environ.get('VSL_STAMP_STEREO') else [0, -1, 1]
dev = torch.device("cuda:0")
data, t = bench.build_inputs(B, H, W, fids, 1, dev)
step = bench.LossStep(t, B, H, W, fids, 1, dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    step(stream)
torch.cuda.synchronize()
TH = int(os.environ.get('VSL_TILE_H', '12')); NW = (TH + 4) // 4
blocks = ((W + 27) // 28) * ((H + TH - 1) // TH) * B
buf = np.zeros((blocks, 8, 16), np.uint64)      # the library reserves 8 wave slots per workgroup
abi.lib.vsl_debug_read_stamps.restype = ctypes.c_longlong
n = abi.lib.vsl_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(buf.nbytes))
print("bytes", n)
buf = buf[:, :NW]
s = buf.astype(np.int64)
names = ["phase0+bar", "phase1", "bar1", "phase2", "bar2", "phase3", "bar3", "phase4"]
d = np.diff(s[:, :, :9], axis=2)
print("clock ticks per wave (mean over blocks), by wave index 0..4")
for i, nm in enumerate(names):
    print("%-12s" % nm, " ".join("%8.0f" % d[:, w, i].mean() for w in range(NW)), "   all: %8.0f" % d[:, :, i].mean())
tot = (s[:, :, 8] - s[:, :, 0])
print("total       ", " ".join("%8.0f" % tot[:, w].mean() for w in range(NW)))
# global timeline from the constant-rate counter (100 MHz) and the hardware ids
rt0 = s[:, :, 9].min(axis=1); rt1 = s[:, :, 10].max(axis=1)
span = rt1.max() - rt0.min()
print("kernel span %.1f us; mean block life %.1f us; mean blocks in flight %.1f (of %d CUs)" % (
    span / 100.0, (rt1 - rt0).mean() / 100.0, (rt1 - rt0).sum() / span, 256))
hw = buf[:, 0, 11]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 0x1; se = (hwid >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ids, counts = np.unique(cuid, return_counts=True)
print("distinct CU ids %d; blocks per CU min %d max %d" % (len(ids), counts.min(), counts.max()))
# max concurrency on one CU: sweep
conc = []
for c in ids[:64]:
    m = cuid == c
    ev = sorted([(t, 1) for t in rt0[m]] + [(t, -1) for t in rt1[m]])
    cur = mx = 0
    for _, dlt in ev:
        cur += dlt; mx = max(mx, cur)
    conc.append(mx)
print("max concurrent blocks on a CU (first 64 CUs):", sorted(set(conc)), "mean", np.mean(conc))
# which SIMD of its CU each wave of a workgroup runs on (HW_ID bits 5:4)
simd = ((buf[:, :, 11] & np.uint64(0xFFFFFFFF)).astype(np.int64) >> 4) & 0x3
for w in range(NW):
    print("wave %d on SIMD 0..3: %s" % (w, " ".join("%5d" % int((simd[:, w] == i).sum()) for i in range(4))))
print("workgroups whose four waves sit on four different SIMDs: %d of %d" % (
    int((np.sort(simd[:, :4], axis=1) == np.arange(4)).all(axis=1).sum()), simd.shape[0]))
