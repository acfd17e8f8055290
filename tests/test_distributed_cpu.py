"""World-size-2 gloo tests of everything the N > 1 path runs outside the kernels: sharding, the final all-gather of
the pose records, bench.py's fence / max-over-ranks timing, the runner's shard + gather with a stub solver, and
bench.py's own rank launcher."""
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nemo_amd.distributed import fence, gather_records, max_over_ranks, shard_range, solve_sharded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_solve(start, stop):
    i = torch.arange(start, stop, dtype=torch.float32)
    return torch.stack([i, i * 2, i * 3, -i, i + 0.5, i % 7, torch.ones_like(i), torch.zeros_like(i)], dim=1)


def _fake_solve10(start, stop):
    return torch.cat([_fake_solve(start, stop), torch.full((stop - start, 2), 0.25)], dim=1)


def _worker(rank, world, port, n, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nemo_amd.solve_all import run_sharded
    out = solve_sharded(_fake_solve, n)
    fence(None)                                             # bench.py's bracket: barrier on a CPU group
    slowest = max_over_ranks(1.0 + rank)                    # bench.py's timing: the slowest rank's clock
    runner = run_sharded(n, _fake_solve10, torch.device('cpu'))   # solve_all.py: shard, stub solve, one gather
    few = run_sharded(1, _fake_solve10, torch.device('cpu'))      # fewer images than ranks: rank 1's shard is empty
    q.put((rank, out.numpy(), slowest, runner.numpy(), few.numpy()))
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
    got = [q.get(timeout=120) for _ in range(2)]
    [p.join(timeout=60) for p in procs]
    ref, ref10 = _fake_solve(0, n).numpy(), _fake_solve10(0, n).numpy()
    assert sorted(g[0] for g in got) == [0, 1]
    for _, out, slowest, runner, few in got:
        assert np.array_equal(out, ref) and np.array_equal(runner, ref10)
        assert slowest == 2.0
        assert np.array_equal(few, ref10[:1])


def test_gather_single_process_is_identity():
    x = _fake_solve(0, 5)
    assert gather_records(x, 5) is x
    assert max_over_ranks(3.5) == 3.5
    fence(None)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (a child torch.distributed.run);
    on this GPU-less box each RANK then stops at "no GPU visible" -- not the argument parser."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    if torch.cuda.is_available():
        env['HIP_VISIBLE_DEVICES'] = ''   # same experiment on a GPU box: hide the card
        env['CUDA_VISIBLE_DEVICES'] = ''
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert 'no GPU visible' in p.stderr        # printed by a rank (the launcher stops the other one once the first has failed)
    assert 'needs torch.distributed.run' not in p.stderr
