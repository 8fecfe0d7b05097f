"""The N>1 path of bench.py on CPU: two gloo ranks shard rooms with no data-path collective; the only
communication is the barrier / max-time / unit-sum of the timing contract (disco_amd/dist.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from disco_amd import dist as dd


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dd.init('gloo', rank, world)
    lo, hi = dd.room_range(rank, world, 5)
    # stand-in workload: each rank "processes" its rooms; rank 1 is slower
    units = (hi - lo) * 4 * 626
    seconds = 0.010 * (1 + rank)
    dist.barrier()
    value, tmax = dd.whole_job_throughput(units, seconds, world)
    # room ids must tile [0, world*5) without overlap
    ids = torch.zeros(world * 5, dtype=torch.int64)
    ids[lo:hi] = 1
    dist.all_reduce(ids)
    q.put((rank, lo, hi, value, tmax, ids.tolist()))
    dist.destroy_process_group()


def test_two_rank_room_sharding_and_timing_contract():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 5), (5, 10)]
    for r in res:
        assert r[5] == [1] * 10                                 # disjoint cover
        assert r[4] == pytest.approx(0.020)                     # max over ranks
        assert r[3] == pytest.approx(2 * 5 * 4 * 626 / 0.020)   # whole-job units / max time
    assert res[0][3] == res[1][3]


def test_split_rooms_balanced():
    for total, world in [(1000, 8), (1001, 8), (7, 8), (256, 3)]:
        parts = dd.split_rooms(total, world)
        assert parts[0][0] == 0 and parts[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        sizes = [b - a for a, b in parts]
        assert max(sizes) - min(sizes) <= 1
