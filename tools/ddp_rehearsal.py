"""Diagnostic: three DistributedDataParallel modules over gloo with both ranks on GPU 0 and a plain torch loss
(separates the process-group behaviour of the box from this library)."""
import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import synthetic_train as ex
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.distributed.init_process_group("gloo")
torch.manual_seed(0)
models = {"encoder": ex.TinyEncoder().cuda(), "depth": ex.TinyDepthDecoder([0, 1]).cuda(), "pose": ex.TinyPoseDecoder(3).cuda()}
params = [p for m in models.values() for p in m.parameters()]
models = {k: nn.parallel.DistributedDataParallel(m, device_ids=[0]) for k, m in models.items()}
opt = torch.optim.Adam(params, 1e-3)
BATCH, HH, WW = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (2, 32, 96)))
x = torch.rand(3 * BATCH, 3, HH, WW).cuda()
for i in range(4):
    if i == 1:
        torch.distributed.barrier()
    feats = models["encoder"](x)
    feats = [torch.split(f, BATCH) for f in feats]
    features = {k: [f[j] for f in feats] for j, k in enumerate([0, -1, 1])}
    out = models["depth"](features[0])
    aa, tr = models["pose"]([features[k] for k in [0, -1, 1]])
    loss = out[("disp", 0)].mean() + (aa[:, 0] ** 2).sum() + (tr[:, 1] ** 2).sum()
    opt.zero_grad(); loss.backward(); opt.step()
    print("rank", rank, "step", i, float(loss), flush=True)
torch.distributed.destroy_process_group()
