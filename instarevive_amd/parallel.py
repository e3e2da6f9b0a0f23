"""One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

The path shards by independent units — images (test_scripts/inference.py:261, one process() per file) or, inside one
large image, tiles (inference.py:128-134,139-152) — so the data path needs NO collective: every rank runs the full
four-stage path on its own units with replicated weights. The only communication is optional and off the hot path:
gathering finished uint8 images on rank 0 (gather_uint8) and reducing timings (max_over_ranks)."""
import os
from typing import List, Sequence

import torch


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_distributed(backend: str = None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT). No-op for one process."""
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world == 1 or dist.is_initialized():
        return rank, world, local
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return rank, world, local


def shard(items: Sequence, rank: int, world: int) -> List:
    """Round-robin unit assignment: rank r takes items r, r+world, ...."""
    return list(items[rank::world])


def unshard_order(n_items: int, world: int) -> List[int]:
    """Position in the rank-major concatenation of every original item index (inverse of `shard` after a gather)."""
    order = [i for r in range(world) for i in range(r, n_items, world)]
    inv = [0] * n_items
    for pos, i in enumerate(order):
        inv[i] = pos
    return inv


def max_over_ranks(value: float, device=None) -> float:
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_uint8(local: torch.Tensor, dst: int = 0):
    """Gather per-rank uint8 image batches [n_r, H, W, 3] (n_r may differ by one) on rank `dst`; returns the rank-major
    concatenation there and None elsewhere. One RCCL gather of <= 12.6 MB per 2048x2048 image: latency-, not bandwidth-bound."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device))
    counts = [int(c) for c in counts]
    nmax = max(counts)
    padded = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=torch.uint8, device=local.device)
    padded[: local.shape[0]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
