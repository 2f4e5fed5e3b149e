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
            # FT_DIST_BACKEND=gloo: a dry run of the N > 1 code on ONE GPU (two ranks sharing device 0; RCCL refuses that)
            backend = os.environ.get("FT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


class RowGatherer:
    """Pre-allocated, double-buffered all-gather of equal-sized per-rank row blocks, issued on a communication stream.

    The per-step exchange of the N > 1 path (bench.py, tools/): `start(rows)` copies the rank's rows into a send slot on
    the compute stream and launches ONE all_gather_into_tensor on a side stream behind an event; the compute stream is
    not blocked, so the next step's HIP graph replays while the previous step's rows travel over xGMI.  `finish(handle)`
    makes the current stream wait for that exchange and returns the gathered [world * n_local, ...] view (rank-major =
    batch order for contiguous equal shards).  Nothing is allocated, padded or concatenated per call; `depth` slots
    allow that many exchanges in flight.  On CPU tensors / gloo (the tests) the same calls run synchronously."""

    def __init__(self, n_local: int, row_shape, dtype, device, depth: int = 2, force_stream: bool = False):
        """force_stream (single-process GPU tests): take the communication-stream path at world size 1 too, with a device
        copy standing in for the collective — the wait_stream -> side-stream exchange -> event -> lagged finish() sequence of
        the N > 1 path then runs on one GPU."""
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.n_local, self.depth = n_local, depth
        shape = (n_local,) + tuple(row_shape)
        self.send = [torch.empty(shape, dtype=dtype, device=device) for _ in range(depth)]
        self.recv = [torch.empty((self.world * n_local,) + tuple(row_shape), dtype=dtype, device=device) for _ in range(depth)]
        self.cuda = torch.device(device).type == "cuda"
        self.stream = torch.cuda.Stream(device) if self.cuda and (self.world > 1 or force_stream) else None
        self.ready = [None] * depth      # per slot: event recorded on the comm stream once the exchange is queued
        self.slot = 0

    def _exchange(self, k: int) -> None:
        if self.world > 1:
            dist.all_gather_into_tensor(self.recv[k], self.send[k])
        else:
            self.recv[k].copy_(self.send[k])

    def start(self, rows: torch.Tensor) -> int:
        k = self.slot
        self.slot = (k + 1) % self.depth
        if self.stream is None:
            if self.world == 1:
                self.recv[k].copy_(rows)     # same contract (a stable buffer the caller may read later), no collective
            else:
                self.send[k].copy_(rows)
                self._exchange(k)
            return k
        cur = torch.cuda.current_stream(rows.device)
        if self.ready[k] is not None:
            # more than `depth` exchanges outstanding: the one that still owns this slot must be done reading send[k] /
            # writing recv[k] before the slot is refilled (the caller never finish()ed it; its result is dropped)
            cur.wait_event(self.ready[k])
            self.ready[k] = None
        self.send[k].copy_(rows)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self._exchange(k)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.ready[k] = ev
        return k

    def finish(self, k: int) -> torch.Tensor:
        if self.stream is not None and self.ready[k] is not None:
            torch.cuda.current_stream(self.recv[k].device).wait_event(self.ready[k])
            self.ready[k] = None
        return self.recv[k]


def verify_gather(gather: "RowGatherer", rank: int, device=None) -> dict:
    """Evidence that the exchange of the N > 1 step really moved every rank's rows (bench.py prints it as `rccl`): each rank
    stamps its row block with (rank + 1) + a per-row ramp, ONE RowGatherer round (start -> finish, the same code path the
    timed steps use), then every rank checks all N blocks of what it received and the verdicts are MIN-reduced.  Outside
    any timed region.  Returns {"world", "ranks_verified" (on the worst rank), "backend"}."""
    world, n = gather.world, gather.n_local
    send = gather.send[0]
    ramp = torch.arange(n, dtype=torch.float32, device=send.device).view((n,) + (1,) * (send.dim() - 1)) / 1024.0
    rows = (torch.full(send.shape, float(rank + 1), dtype=torch.float32, device=send.device) + ramp).to(send.dtype)
    got = gather.finish(gather.start(rows))
    if gather.cuda:
        torch.cuda.current_stream(send.device).synchronize()
    ok = 0
    for r in range(world):
        want = (torch.full(send.shape, float(r + 1), dtype=torch.float32, device=send.device) + ramp).to(send.dtype)
        ok += int(torch.equal(got[r * n:(r + 1) * n], want))
    if dist.is_initialized() and world > 1:
        t = torch.tensor([ok], dtype=torch.int64, device=send.device if gather.cuda else None)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item())
    return {"world": world, "ranks_verified": ok, "backend": dist.get_backend() if dist.is_initialized() else "none",
            "bytes_per_rank": send.numel() * send.element_size()}


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
