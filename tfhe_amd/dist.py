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


def world_size_seen() -> int:
    """The process group's own count of ranks (1 without a group): what actually took part, whatever was asked for."""
    return dist.get_world_size() if dist.is_initialized() else 1


def backend_name() -> str:
    return dist.get_backend() if dist.is_initialized() else "none (single process)"


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


class ResultDrain:
    """Gathers every rank's batch of results to host memory on rank `dst`, in chunks.

    A whole batch does not fit the root GPU next to its own work (16 384 products of n = 512 are 534 GB),
    so the ranks send `chunk` entries at a time: one `gather` per chunk into one of two receive buffers on
    the root (RCCL: every peer sends over its own xGMI link), while the previous chunk drains from the
    other buffer to pinned host memory on a side stream.  Root ingress bounds the rate (7 links), so this
    is an exchange step to run beside compute on a dedicated stream, never inside a kernel-bound loop.
    """

    def __init__(self, entry_shape, dtype, world: int, rank: int, entries: int, chunk: int, device, dst: int = 0):
        self.world, self.rank, self.dst, self.entries = world, rank, dst, entries
        self.chunk = max(1, min(chunk, entries)) if entries else 1
        self.device = torch.device(device)
        self.root = rank == dst
        self.cuda = self.device.type == "cuda"
        shape = (self.chunk,) + tuple(entry_shape)
        if self.root:
            self.recv = [[torch.empty(shape, dtype=dtype, device=self.device) for _ in range(world)] for _ in range(2)]
            self.host = torch.empty((world, entries) + tuple(entry_shape), dtype=dtype, pin_memory=self.cuda)
            self.side = torch.cuda.Stream(self.device) if self.cuda else None
            self.done = [None, None]

    def run(self, results: torch.Tensor) -> Optional[torch.Tensor]:
        """results: this rank's (entries, *entry_shape) tensor on `device`.  Returns, on the root, the
        (world, entries, *entry_shape) host tensor; None elsewhere."""
        if results.shape[0] != self.entries:
            raise ValueError(f"expected {self.entries} entries, got {results.shape[0]}")
        for k, lo in enumerate(range(0, self.entries, self.chunk)):
            hi = min(lo + self.chunk, self.entries)
            part = results[lo:hi]
            if hi - lo < self.chunk:  # equal shapes on every rank and every call: pad the last chunk
                pad = torch.zeros((self.chunk,) + tuple(results.shape[1:]), dtype=results.dtype, device=results.device)
                pad[: hi - lo] = part
                part = pad
            part = part.contiguous()
            if dist.is_initialized() and part.is_cuda and dist.get_backend() == "gloo":
                part = part.cpu()
            if not self.root:
                dist.gather(part, gather_list=None, dst=self.dst)
                continue
            bufs = self.recv[k & 1]
            if self.done[k & 1] is not None:
                self.done[k & 1].synchronize()  # the drain of chunk k - 2 has left this buffer
            if dist.is_initialized():
                if part.device != bufs[0].device:
                    staged = [torch.empty_like(part) for _ in range(self.world)]
                    dist.gather(part, gather_list=staged, dst=self.dst)
                    for b, s_ in zip(bufs, staged):
                        b.copy_(s_)
                else:
                    dist.gather(part, gather_list=bufs, dst=self.dst)
            else:
                bufs[0].copy_(part)
            if self.cuda:
                self.side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self.side):
                    for r in range(self.world):
                        self.host[r, lo:hi].copy_(bufs[r][: hi - lo], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                self.done[k & 1] = ev
            else:
                for r in range(self.world):
                    self.host[r, lo:hi].copy_(bufs[r][: hi - lo])
        if not self.root:
            return None
        if self.cuda:
            self.side.synchronize()
        return self.host


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
