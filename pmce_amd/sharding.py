"""Clip sharding across the GPUs of one node and the single collective of the path.

Clips are independent (each stride-1 window is its own forward, reference lib/_img_utils.py:74-78), weights are
replicated, so rank r simply owns a contiguous block of the clip index range (contiguity keeps consecutive
middle frames of a sequence on one rank for the acceleration error).  The only communication is the final
metric reduction: an all_reduce / all_gather of a few floats per rank over RCCL (backend "nccl" on ROCm) —
or gloo in the CPU tests."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world_size: int):
    """Contiguous, balanced [lo, hi) block of rank (first n % world ranks get one extra item)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend: str | None = None):
    """One process per GPU, launched by torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in env).
    Returns (rank, local_rank, world_size).  No-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:   # PMCE_DIST_BACKEND=gloo: plumbing tests of the N>1 path without N GPUs
            backend = os.environ.get("PMCE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            # bind the communicator to this rank's device up front: barrier() then needs no device guess
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def barrier():
    if dist.is_initialized():
        dist.barrier()


def _comm_device(device):
    """RCCL reduces device tensors in place; gloo (CPU tests / plumbing runs) gets host copies."""
    if dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return device


def reduce_max(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(torch.device(device)))
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_metric_sums(partial: torch.Tensor) -> torch.Tensor:
    """SUM-reduce a small vector of per-rank metric partials ([sum_err..., count]) over all ranks."""
    t = partial.detach().to(_comm_device(partial.device)).clone()
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.to(partial.device)


def gather_rows(local: torch.Tensor) -> torch.Tensor:
    """all_gather of per-sample rows (e.g. 14x3 regressed joints) with ragged per-rank counts -> [N_total, ...]
    in rank order (= clip order for contiguous shards)."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()
    out_device = local.device
    local = local.detach().to(_comm_device(local.device))
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0).to(out_device)
