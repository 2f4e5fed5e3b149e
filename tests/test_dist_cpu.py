"""world_size-2 gloo test of the batch sharding + output all-gather used by the N>1 path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flowtrack.pytorch_amd import parallel


def test_shard_ranges_partition_the_batch():
    for n in (0, 1, 7, 8, 64, 129):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env(backend="gloo")
    full = torch.arange(n_total * 17 * 3, dtype=torch.float32).view(n_total, 17, 3)   # stand-in keypoint rows
    lo, hi = parallel.shard_range(n_total, rank, world)
    got = parallel.all_gather_rows(full[lo:hi].clone(), n_total)
    ok = torch.equal(got, full)
    t = parallel.max_over_ranks(float(rank + 1))
    parallel.barrier()
    q.put((rank, ok, t))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7])
def test_all_gather_rows_world2(n_total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results) and all(t == 2.0 for _, _, t in results)
