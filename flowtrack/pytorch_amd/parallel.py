"""Multi-GPU sharding of the pose / flow batch: one process per GPU, RCCL over xGMI.

The reference's only multi-GPU mechanism is single-process nn.DataParallel (scatter on dim 0,
replicate the module every forward, gather to GPU 0 — tools/flownet/main.py:133-134,186,197,
tools/flownet/demo.py:75).  Here crops / frame pairs are independent units (eval-mode BN), so each
rank owns a contiguous slice of the batch, weights are replicated once at start-up and the only
exchange is one all-gather of the per-rank OUTPUT rows (keypoints, or heatmaps / flow when asked).
Payloads are small (3.3 kB/rank of keypoints, 1.6 MB per flow pair), so a single
ncclAllGather on the default RCCL communicator is latency-bound; no bucketing is needed.
backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of n units for `rank`; sizes differ by at most one, lower ranks first."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's environment; initialises the default group if world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def all_gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Gather per-rank row blocks (dim 0, possibly ragged by one row) into the full [n_total, ...] tensor,
    in rank order, on every rank.  Single collective: rows are padded to the largest shard."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    max_rows = -(-n_total // world)
    pad = max_rows - local.shape[0]
    if pad:
        local = torch.cat((local, local.new_zeros((pad,) + tuple(local.shape[1:]))), 0)
    out = local.new_empty((world * max_rows,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, local.contiguous())
    pieces: List[torch.Tensor] = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        pieces.append(out[r * max_rows:r * max_rows + (hi - lo)])
    return torch.cat(pieces, 0)


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
