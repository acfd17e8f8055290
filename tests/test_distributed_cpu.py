"""World-size-2 gloo test of the sharding + final all-gather (the N > 1 path of bench.py / solve_sharded)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nemo_amd.distributed import gather_records, shard_range, solve_sharded


def _fake_solve(start, stop):
    i = torch.arange(start, stop, dtype=torch.float32)
    return torch.stack([i, i * 2, i * 3, -i, i + 0.5, i % 7, torch.ones_like(i), torch.zeros_like(i)], dim=1)


def _worker(rank, world, port, n, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    out = solve_sharded(_fake_solve, n)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (0, 1, 7, 1024, 1025):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1


def test_two_rank_gather_equals_single_rank():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    n = 13  # odd: uneven shards
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    ref = _fake_solve(0, n).numpy()
    assert np.array_equal(res[0], ref) and np.array_equal(res[1], ref)


def test_gather_single_process_is_identity():
    x = _fake_solve(0, 5)
    assert gather_records(x, 5) is x
