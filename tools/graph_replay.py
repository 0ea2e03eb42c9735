"""Diagnostic (GPU box): one loss step (identity pass, fused kernel, finalize, gradient assembly of all heads) captured
in a HIP graph through torch.cuda.CUDAGraph and replayed; results against the eager calls, time per step both ways."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import mde_amd._abi as abi

B, H, W = [int(v) for v in os.environ.get("VSL_SHAPE", "12,192,640").split(",")]
heads = int(os.environ.get("VSL_HEADS", "4"))
fids = [0, -1, 1]
dev = torch.device("cuda:0")
data, t = bench.build_inputs(B, H, W, fids, heads, dev)
step = bench.LossStep(t, B, H, W, fids, heads, dev)
cur = torch.cuda.current_stream(dev)
for _ in range(3):
    step(cur.cuda_stream)
torch.cuda.synchronize()
eager = [x.clone() for x in step.losses + step.gdisp + [g for gs in step.gT for g in gs] + step.ident]

side = torch.cuda.Stream(dev)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    step(torch.cuda.current_stream(dev).cuda_stream)
for x in step.losses + step.gdisp:
    x.zero_()
graph.replay()
torch.cuda.synchronize()
replayed = step.losses + step.gdisp + [g for gs in step.gT for g in gs] + step.ident
same = all(torch.equal(a, b) for a, b in zip(eager, replayed))
print("graph replay equals eager calls bit for bit:", same)

def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

print("eager  %.1f us per step" % timed(lambda: step(cur.cuda_stream)))
print("replay %.1f us per step" % timed(graph.replay))
sys.exit(0 if same else 1)
