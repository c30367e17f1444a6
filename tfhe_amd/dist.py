"""Multi-GPU plumbing: one process per GPU, independent ciphertext products sharded by batch index.

The contraction needs no collective (every product fits one GPU); the only exchange step is the
gather of results to rank 0 -- 64-bit digests of every product plus a sampled full product per
rank -- over RCCL (`backend="nccl"` on ROCm) or gloo on CPU for tests.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` batch entries owned by `rank` (sizes differ by at most 1)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_root(t: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Gather equally shaped tensors to rank `dst`; returns the list there, None elsewhere."""
    if not dist.is_initialized():
        return [t]
    world = dist.get_world_size()
    if t.is_cuda and dist.get_backend() == "gloo":
        t = t.cpu()  # gloo gathers host tensors only (CPU tests, single-GPU rehearsals of the multi-rank path)
    if dist.get_rank() == dst:
        out = [torch.empty_like(t) for _ in range(world)]
        dist.gather(t, gather_list=out, dst=dst)
        return out
    dist.gather(t, gather_list=None, dst=dst)
    return None


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
