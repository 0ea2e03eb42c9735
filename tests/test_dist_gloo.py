"""world_size-2 gloo check of the data-parallel plumbing: sharded losses / gradients of the
CPU oracle, averaged with dist.all_reduce_mean_, equal the global-batch loss / gradient."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_step(data, lo, hi, fids, H, W):
    from oracle import vsl_oracle as O
    cfg = O.LossConfig(frame_ids=fids, height=H, width=W, automask=True)
    inputs = {"K": data["K"][lo:hi], "inv_K": data["inv_K"][lo:hi]}
    for f in fids:
        inputs[("image", f)] = data[("image", f)][lo:hi].double()
    disp = data[("disp", 0)][lo:hi].clone().requires_grad_(True)
    outputs = {("disp", 0): disp}
    poses = []
    for f in fids[1:]:
        T = data[("cam_T_cam", f)][lo:hi].clone().requires_grad_(True)
        outputs[("cam_T_cam", f)] = T
        poses.append(T)
    losses = O.loss_forward_backward(inputs, outputs, cfg, data["noise"][lo:hi])
    return losses["loss"].detach(), disp.grad, [p.grad for p in poses]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import mde_amd.dist as D
    import mde_amd.synth as synth
    assert D.init("gloo")
    B, H, W, fids = 4, 16, 40, [0, -1, 1]
    data = synth.make_batch(B, H, W, fids, seed=3)
    lo, hi = D.shard_range(B, rank, world)
    loss, gdisp, gposes = _oracle_step(data, lo, hi, fids, H, W)
    # a per-"parameter" gradient every rank holds (here: the sum of pose gradients over its shard, as a
    # stand-in for a shared network weight's gradient) plus the scalar loss
    shared = [g.sum(0).float() for g in gposes] + [loss.float().reshape(1)]
    D.all_reduce_mean_(shared, bucket_bytes=64)       # tiny buckets: exercises several collectives
    t = D.max_over_ranks(float(rank + 1), torch.device("cpu"))
    D.barrier()
    q.put((rank, [s.numpy() for s in shared], t, gdisp.numpy(), lo, hi))
    torch.distributed.destroy_process_group()


def test_sharded_loss_and_gradients_match_global_batch():
    import mde_amd.synth as synth
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    B, H, W, fids = 4, 16, 40, [0, -1, 1]
    data = synth.make_batch(B, H, W, fids, seed=3)
    g_loss, g_gdisp, g_gposes = _oracle_step(data, 0, B, fids, H, W)
    for rank, shared, t, gdisp, lo, hi in results:
        assert t == 2.0                                        # max over ranks
        # mean of the ranks' losses == global loss (equal shards)
        np.testing.assert_allclose(shared[-1][0], g_loss.item(), rtol=2e-6)
        # averaged shard gradients == global gradients / 1 (the global mean already divides by B*H*W):
        # sum over all images of the global gradient equals world * averaged shard-sums / world ... per image
        # the local gradient is world times the global one, so the rank average of shard sums equals the
        # global sum over images.
        for i, g in enumerate(g_gposes):
            np.testing.assert_allclose(shared[i], g.sum(0).numpy(), rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(gdisp / world, g_gdisp.numpy()[lo:hi], rtol=2e-4, atol=1e-9)


def test_shard_range_requires_even_split():
    import mde_amd.dist as D
    assert D.shard_range(96, 3, 8) == (36, 48)
    with pytest.raises(ValueError):
        D.shard_range(10, 0, 4)
