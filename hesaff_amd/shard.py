"""Image-level sharding of a batch across the GPUs of one node.

Images are independent (the reference has no cross-image state except two counters,
hesaff.cpp:38-39), so the data path has NO collective: rank r processes the contiguous
block shard_range(n, r, world) and writes its own results.  The only exchange is one
all-gather of per-rank feature counts (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous block of [0, n_items) owned by `rank`: image i -> rank floor(i*world/n)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    lo = (n_items * rank + world - 1) // world
    hi = (n_items * (rank + 1) + world - 1) // world
    return lo, hi


def gather_counts(local_counts, device=None):
    """all_gather of a small int64 vector (e.g. [count_hessian, count_desc, images]).

    Returns an array [world, len(local_counts)].  Without an initialised process group it is
    the identity (single process)."""
    import torch
    import torch.distributed as dist

    t = torch.as_tensor(np.asarray(local_counts, dtype=np.int64))
    if not (dist.is_available() and dist.is_initialized()):
        return t.numpy()[None, :].copy()
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()
