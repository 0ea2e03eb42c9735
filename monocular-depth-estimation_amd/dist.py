"""Data-parallel plumbing around the loss: one process per GPU, batches sharded over ranks.

The view-synthesis loss is per-image up to its final means (Trainer.py:605, layers.py:226,
and the per-image mean disparity of Trainer.py:609), so with equal local batches the mean of
the ranks' losses is the global-batch loss and the average of their gradients is the
global-batch gradient: no exchange happens inside the loss, only the usual gradient
all-reduce of data-parallel training (RCCL on MI355X: backend "nccl"; "gloo" in CPU tests).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Tuple

import torch
import torch.distributed as td


def env_rank() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1 process = defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str, device: torch.device = None) -> bool:
    """Initialise torch.distributed when WORLD_SIZE > 1; returns whether it did."""
    _, world, _ = env_rank()
    if world <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kwargs = {}
    if backend == "nccl" and device is not None:
        kwargs["device_id"] = device
    td.init_process_group(backend, **kwargs)
    return True


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous equal shards; the global batch must divide evenly (equal shards are what makes
    the mean of local means equal the global mean)."""
    if n_items % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (n_items, world))
    per = n_items // world
    return rank * per, (rank + 1) * per


def all_reduce_mean_(tensors: Iterable[torch.Tensor], bucket_bytes: int = 64 << 20) -> None:
    """In-place average over ranks, flattened into buckets so that each collective is large
    (xGMI rings are per-link bound: few big all-reduces beat many small ones)."""
    if not (td.is_available() and td.is_initialized()):
        return
    world = td.get_world_size()
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        td.all_reduce(flat, op=td.ReduceOp.SUM)
        flat /= world
        off = 0
        for t in bucket:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        bucket, size = [], 0

    for t in tensors:
        if bucket and (size + t.numel() * t.element_size() > bucket_bytes or t.dtype != bucket[0].dtype):
            flush()
        bucket.append(t)
        size += t.numel() * t.element_size()
    flush()


def max_over_ranks(value: float, device: torch.device) -> float:
    """Slowest rank's value (bench timing)."""
    if not (td.is_available() and td.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if td.is_available() and td.is_initialized():
        td.barrier()
