"""world_size-2 gloo tests (CPU): the instance sharder + the single all-reduce of the multi-GPU path.  The per-shard
solve is the CPU oracle here (the checker standing in for the GPU forward), so the test pins the property the design
relies on: sharding by instances gives the same per-instance results and the same totals as the unsharded run."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import REPO
from pdp import parallel
from pdp.factorgraph import dataset


def test_shard_bounds_cover_and_balance():
    rng = np.random.RandomState(0)
    for world in (1, 2, 3, 8):
        for n in (1, 2, 7, 100):
            edges = rng.randint(10, 500, size=n)
            b = parallel.shard_bounds(edges, world)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1)) and all(lo <= hi for lo, hi in b)
            if n >= world:
                assert all(hi > lo for lo, hi in b)
            loads = [int(edges[lo:hi].sum()) for lo, hi in b]
            if n >= 4 * world:
                assert max(loads) <= edges.sum() / world + edges.max()


def _oracle_solve(items):
    sys.path.insert(0, REPO)
    from oracle import binding
    b = dataset.collate_segment(items)
    p = binding.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    res = p.forward('p-d-p', 25, local_search_iterations=20, tolerance=0.05, t_max=8, seed=3)
    solved, unsat = p.cnf_eval(res['prediction'])
    offs = np.concatenate(([0], np.cumsum([it[0] for it in items])))
    rows = [(it[5][0], int(solved[i]), res['prediction'][offs[i]:offs[i + 1]].astype(int).tolist()) for i, it in enumerate(items)]
    return solved, unsat, rows


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    items = dataset.random_ksat_items(12, 30, 3, m=100, seed=4000)
    stats, rows, offset = parallel.solve_sharded(items, _oracle_solve)
    if rank == 0:
        q.put((stats, rows))
    dist.destroy_process_group()


def test_sharded_solve_equals_unsharded():
    items = dataset.random_ksat_items(12, 30, 3, m=100, seed=4000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    stats, rows = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # unsharded: every instance on its own (instance-local computation; Philox random numbers are indexed by the
    # instance-local variable position only through the batch offset, so compare solved counts and per-instance solved flags
    # of runs that see each instance with the same offsets: run the two shards here again, unsharded in-process)
    lo_hi = parallel.shard_bounds([it[2].shape[1] for it in items], 2)
    exp_rows, exp_solved, exp_unsat = [], 0, 0
    for lo, hi in lo_hi:
        s, u, r = _oracle_solve(items[lo:hi])
        exp_rows += r; exp_solved += int(np.sum(s)); exp_unsat += int(np.sum(u))
    assert stats == dict(instances=12, solved=exp_solved, unsat_clauses=exp_unsat, solved_fraction=exp_solved / 12.0)
    assert rows == exp_rows


def _metrics_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, REPO)
    from oracle import binding
    items = dataset.random_ksat_items(10, 30, 3, m=100, seed=5000)
    mine, _ = parallel.shard_items(items, rank, world)
    b = dataset.collate_segment(mine)
    p = binding.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    pred = np.random.RandomState(7 + rank).rand(p.V).astype(np.float32)
    solved, _ = p.cnf_eval((pred > 0.5).astype(np.float32))
    label = (np.arange(len(mine)) % 2).astype(np.float32)
    n = float(len(mine))
    sums = np.array([[n * np.mean(np.abs(solved - label))], [n * np.sum(label * np.abs(solved - label)) / max(label.sum(), 1e-8)],
                     [n * p.sat_loss(pred, 2.0, 1e-8, 5)]])
    mean, total = parallel.reduce_test_metrics(sums, len(mine))
    q.put((rank, sums, len(mine), mean, total))
    dist.destroy_process_group()


def test_test_mode_metrics_reduce_over_ranks():
    "test mode across ranks: one all-reduce(sum) of the per-example-weighted metric sums [accuracy, recall, loss] and the example count"
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_metrics_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = (got[0][1] + got[1][1]) / float(got[0][2] + got[1][2])
    for rank, sums, n, mean, total in got:
        assert total == 10
        np.testing.assert_allclose(mean, exp, rtol=1e-12)


def test_shard_bounds_small_inputs():
    assert parallel.shard_bounds([5, 5], 3) == [(0, 1), (1, 2), (2, 2)]           # fewer instances than ranks: the first ranks get one each
    assert parallel.shard_bounds([], 2) == [(0, 0), (0, 0)]
    assert parallel.shard_bounds([1, 1, 1, 100], 2) == [(0, 3), (3, 4)]             # one heavy instance at the end still leaves the last rank one
    assert parallel.shard_bounds([100, 1, 1, 1], 2) == [(0, 1), (1, 4)]
    assert parallel.shard_bounds([10] * 8, 8) == [(i, i + 1) for i in range(8)]


def _run_bench(extra_env, *argv):
    import subprocess
    env = dict(os.environ); env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          universal_newlines=True, env=env, timeout=600)


def test_bench_launcher_starts_the_ranks_it_is_asked_for():
    """`python bench.py --gpus 2` outside torch.distributed.run is a launcher: it starts 2 ranks through torch.distributed.run (gloo here:
    no GPU in the CPU suite, --selftest-collective skips the device work but runs the rendezvous, the barrier, the MAX / SUM reductions
    and rank 0's single line), forwards one JSON line with n_gpus == rccl_ranks == 2, and fails when a rank fails."""
    import json
    r = _run_bench({'PDP_DIST_BACKEND': 'gloo'}, '--gpus', '2', '--steps', '4', '--warmup', '1', '--selftest-collective')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split('\n') if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['steps'] == 4 and line['warmup'] == 1
    assert line['instances'] == 2000.0 and line['rank_sum'] == 3.0 and abs(line['max_elapsed_s'] - 0.02) < 1e-12     # SUM over ranks, MAX of the times
    assert abs(line['value'] - 2 * 400.0 / 0.02) < 1e-6                                                                # whole-job aggregate / max time
    bad = _run_bench({'PDP_DIST_BACKEND': 'gloo', 'PDP_BENCH_FAIL_RANK': '1'}, '--gpus', '2', '--selftest-collective')
    assert bad.returncode != 0 and not [l for l in bad.stdout.split('\n') if l.startswith('{')]
    # a rank count that does not match --gpus is refused, not silently run as one rank
    mism = _run_bench({'WORLD_SIZE': '1', 'RANK': '0'}, '--gpus', '2', '--selftest-collective')
    assert mism.returncode != 0 and 'WORLD_SIZE' in (mism.stderr + mism.stdout)
