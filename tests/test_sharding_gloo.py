"""The N>1 path on CPU: world_size-2 gloo run of the clip sharding + the final metric reduction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from pmce_amd import sharding
    r, _, w = sharding.init_from_env(backend="gloo")
    lo, hi = sharding.shard_range(n_items, r, w)
    # stand-in for per-clip results of this rank's shard: row i = [i, 2i, 3i]
    rows = torch.arange(lo, hi, dtype=torch.float32)[:, None] * torch.tensor([1.0, 2.0, 3.0])
    partial = torch.tensor([rows.sum().item(), float(hi - lo)], dtype=torch.float64)
    total = sharding.reduce_metric_sums(partial)
    allrows = sharding.gather_rows(rows)
    tmax = sharding.reduce_max(float(r + 1), torch.device("cpu"))
    sharding.barrier()
    q.put((r, lo, hi, total.tolist(), allrows[:, 0].tolist(), tmax))
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    world, n = 2, 37
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, tot0, rows0, m0), (r1, lo1, hi1, tot1, rows1, m1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 19, 19, 37)                      # contiguous, balanced, covering
    expect = 6.0 * sum(range(n))
    assert tot0 == tot1 == [expect, float(n)]                            # SUM all_reduce of metric partials
    assert rows0 == rows1 == [float(i) for i in range(n)]                # ragged all_gather in clip order
    assert m0 == m1 == 2.0                                               # MAX over ranks (timing reduction)


def test_shard_range_properties():
    from pmce_amd.sharding import shard_range
    for n in (0, 1, 7, 64, 35515):
        for w in (1, 2, 4, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
