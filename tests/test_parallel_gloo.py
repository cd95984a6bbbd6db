"""world_size-2 gloo tests (CPU): dealing (loader batch, segment) units to ranks + the single all-reduce of the multi-GPU path.  The
per-unit solve is the CPU oracle here (the checker standing in for the GPU forward; its forward has the reference's couplings between
the instances of one call), so the test pins the property the design relies on: an N-rank run writes the rows and totals of the 1-rank
run."""
import json
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


T_RUN, TOL_RUN, TMAX_RUN, W_RUN, SEED_RUN = 120, 0.05, 8, 20, 3


def _run_items():
    "60 instances = 5 loader batches of 12 (-z 12); batches 1, 2 and 3 hold instances whose surveys become NaN (SURVEY App. B-6)"
    return dataset.random_ksat_items(60, 60, 3, seed=7000)


def _run_batches(items, z=12):
    return [items[s:s + z] for s in range(0, len(items), z)]


def _oracle_solve(items, batch_index, segment_index=0, want_nan=None):
    """one unit (segment of a loader batch) through the oracle's strict (coupled) p-d-p forward with the Philox key of that unit"""
    sys.path.insert(0, REPO)
    from oracle import binding
    b = dataset.collate_segment(items)
    p = binding.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    res = p.forward('p-d-p', T_RUN, local_search_iterations=W_RUN, tolerance=TOL_RUN, t_max=TMAX_RUN,
                    seed=parallel.batch_seed(SEED_RUN, batch_index, segment_index))
    if want_nan is not None:
        want_nan.append(bool(np.isnan(res['fs']).any()))
    solved, unsat = p.cnf_eval(res['prediction'])
    offs = np.concatenate(([0], np.cumsum([it[0] for it in items])))
    rows = [(it[5][0], int(solved[i]), int(unsat[i]), res['prediction'][offs[i]:offs[i + 1]].astype(int).tolist()) for i, it in enumerate(items)]
    return solved, unsat, rows


def _worker(rank, world, port, q, z=12, limit=None):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    stats, rows, mine = parallel.solve_sharded(_run_batches(_run_items(), z), _oracle_solve, limit=limit)
    q.put((rank, stats, rows, mine))
    dist.destroy_process_group()


def _spawn(world, port, *args):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_sharded_run_writes_the_rows_of_the_unsharded_run():
    """The property BASELINE's "identical solved-fraction at 8 GPUs" rests on: a run on N ranks = the run on one rank, row for row.
    Five loader batches, three of them NaN-poisoned (there the reference's batch-wide couplings decide which variables are still
    decimated, so the rows depend on which instances share a forward call); two ranks deal whole units and key the random numbers by the
    global (batch, segment) index.  The expected rows come from one process that solves all five batches in order."""
    items = _run_items()
    batches = _run_batches(items)
    nan_seen, exp_rows, exp_solved, exp_unsat = [], [], 0, 0
    for j, b in enumerate(batches):
        s, u, r = _oracle_solve(b, j, 0, nan_seen)
        exp_rows += r; exp_solved += int(np.sum(s)); exp_unsat += int(np.sum(u))
    assert sum(nan_seen) >= 2, "the test input no longer poisons a batch: pick other seeds"
    got = _spawn(2, 29500 + (os.getpid() % 2000))
    units = [g[3] for g in got]
    assert sorted(units[0] + units[1]) == [(j, 0) for j in range(len(batches))] and all(len(u) >= 2 for u in units)
    for rank, stats, rows, mine in got:
        assert stats == dict(instances=60, solved=exp_solved, unsat_clauses=exp_unsat, solved_fraction=exp_solved / 60.0)
        # the writer (rank 0) holds every row in single-process order; the other ranks keep their own units' rows, in unit order
        assert rows == (exp_rows if rank == 0 else [r for j in range(len(batches)) if (j, 0) in mine for r in exp_rows[12 * j:12 * j + 12]])
    # the test is sensitive to the coupling domain: cutting the instance list per rank first (what a per-instance sharder does) and
    # batching afterwards changes the rows of this input
    lo, hi = parallel.shard_bounds([it[2].shape[1] for it in items], 2)[0]
    cut_rows = []
    for part in (items[lo:hi], items[hi:]):
        for j, b in enumerate(_run_batches(part)):
            cut_rows += _oracle_solve(b, j)[2]
    assert [r[0] for r in cut_rows] == [r[0] for r in exp_rows] and cut_rows != exp_rows


def test_one_loader_batch_cut_into_segments_keeps_both_ranks_busy():
    """configs[4]'s shape in small: ONE loader batch (-z 60) that the dynamic-batching budget cuts into >= 4 segments.  The segments are
    the reference's forward calls, so they are what is dealt: every one of two (and of three) ranks gets work, and the rows equal the
    single-process rows of the same segments in segment order."""
    items = _run_items()
    edges = [it[2].shape[1] for it in items]
    limit = 14 * max(edges)
    segments = dataset.divide(edges, limit, 1)
    assert len(segments) >= 4
    exp_rows, exp_solved, exp_unsat = [], 0, 0
    for i, seg in enumerate(segments):
        s, u, r = _oracle_solve([items[k] for k in seg], 0, i)
        exp_rows += r; exp_solved += int(np.sum(s)); exp_unsat += int(np.sum(u))
    for world, port in ((2, 33500), (3, 35500)):
        got = _spawn(world, port + (os.getpid() % 2000), 60, limit)
        units = [g[3] for g in got]
        assert sorted(sum(units, [])) == [(0, i) for i in range(len(segments))] and all(len(u) >= 1 for u in units)
        for rank, stats, rows, _ in got:
            assert stats == dict(instances=60, solved=exp_solved, unsat_clauses=exp_unsat, solved_fraction=exp_solved / 60.0)
            if rank == 0:
                assert rows == exp_rows                                    # (the writer; the other ranks keep their own units' rows)
            else:
                assert rows and all(r in exp_rows for r in rows) and len(rows) < len(exp_rows)


def _oracle_solve_isolated(items, batch_index, segment_index=0, first_variable=0, first_instance=0):
    """a contiguous part of a segment with every instance solved on its own (a forward of one instance has none of the reference's
    cross-instance couplings: that IS the isolated semantics) -- the instance's Philox counters are those of its place in the segment"""
    sys.path.insert(0, REPO)
    from oracle import binding
    solved, unsat, rows = [], [], []
    v0 = int(first_variable)
    for k, it in enumerate(items):
        b = dataset.collate_segment([it])
        p = binding.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
        p.set_rng_base(v0, int(first_instance) + k)
        res = p.forward('p-d-p', T_RUN, local_search_iterations=W_RUN, tolerance=TOL_RUN, t_max=TMAX_RUN,
                        seed=parallel.batch_seed(SEED_RUN, batch_index, segment_index))
        s, u = p.cnf_eval(res['prediction'])
        solved.append(int(s[0])); unsat.append(int(u[0]))
        rows.append((it[5][0], int(s[0]), int(u[0]), res['prediction'].astype(int).tolist()))
        v0 += int(it[0])
    return np.asarray(solved), np.asarray(unsat), rows


def _isolated_worker(rank, world, port, q, z, limit):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    stats, rows, mine = parallel.solve_sharded(_run_batches(_run_items()[:24], z), _oracle_solve_isolated, limit=limit, split_instances=True)
    q.put((rank, stats, rows, mine))
    dist.destroy_process_group()


def test_isolated_instances_are_dealt_as_instance_ranges():
    """--isolated: ONE forward (one loader batch, one segment) is spread over all ranks as contiguous instance ranges, and the rows are
    those of the single process that solves the segment whole -- including the random fill and the Walk-SAT draws, whose Philox counters
    are the variable's / instance's index inside the SEGMENT (a part starts counting at its first variable / instance).  Also with the
    batch cut into dynamic segments: every segment is spread."""
    items = _run_items()[:24]
    edges = [it[2].shape[1] for it in items]
    for z, limit, world, port in ((24, None, 2, 37500), (24, None, 3, 39500), (24, 7 * max(edges), 2, 41500)):
        segments = [list(range(24))] if limit is None else dataset.divide(edges, limit, 1)
        exp_rows, exp_solved, exp_unsat = [], 0, 0
        for i, seg in enumerate(segments):
            s, u, r = _oracle_solve_isolated([items[k] for k in seg], 0, i)
            exp_rows += r; exp_solved += int(np.sum(s)); exp_unsat += int(np.sum(u))
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        procs = [ctx.Process(target=_isolated_worker, args=(r, world, port + (os.getpid() % 2000), q, z, limit)) for r in range(world)]
        for p in procs:
            p.start()
        got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for rank, stats, rows, mine in got:
            assert mine == [(0, i, rank) for i in range(len(segments))]                   # a part of every segment on every rank
            assert stats == dict(instances=24, solved=exp_solved, unsat_clauses=exp_unsat, solved_fraction=exp_solved / 24.0)
            if rank == 0:
                assert rows == exp_rows
            else:
                assert rows and all(r in exp_rows for r in rows)
    # the counters matter: without the part's base the still-undecided variables are filled with other numbers
    lo, hi = parallel.shard_bounds(edges, 2)[1]
    shifted = _oracle_solve_isolated(items[lo:hi], 0, 0, sum(it[0] for it in items[:lo]), lo)[2]
    unshifted = _oracle_solve_isolated(items[lo:hi], 0, 0)[2]
    assert shifted == _oracle_solve_isolated(items, 0, 0)[2][lo:hi] and shifted != unshifted


def _exchange_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ex = parallel.make_exchange()
    out = []
    for rnd in range(3):                                                    # several exchanges in a row, as a chunked forward makes them
        mins = np.array([100 + 7 * rank + rnd, 0xffffffff if rank else 5], dtype=np.uint32)
        maxs = np.array([rank * 3 + rnd], dtype=np.uint32)
        ors = np.array([1 << rank, 0x80000000 if rank == world - 1 else 0, 0], dtype=np.uint32)
        ex(mins, maxs, ors)
        out.append((mins.tolist(), maxs.tolist(), ors.tolist()))
    ex(np.zeros(0, np.uint32), np.array([rank], np.uint32), np.zeros(0, np.uint32))      # empty arrays are fine
    q.put((rank, out))
    dist.destroy_process_group()


def test_the_exchange_of_a_coupled_forward_reduces_min_max_or():
    "parallel.make_exchange: element-wise min / max / bit-wise OR over the ranks, in place, full 32-bit range (what --split-forward hands to the native solver)"
    for world, port in ((2, 43500), (3, 45500)):
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        procs = [ctx.Process(target=_exchange_worker, args=(r, world, port + (os.getpid() % 2000), q)) for r in range(world)]
        for p in procs:
            p.start()
        got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for rank, out in got:
            for rnd, (mins, maxs, ors) in enumerate(out):
                assert mins == [100 + rnd, 5] and maxs == [(world - 1) * 3 + rnd] and ors == [(1 << world) - 1, 0x80000000, 0]


def _idle_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    stats, rows, mine = parallel.solve_sharded([_run_items()[:6]], _oracle_solve)          # one unit, two ranks
    q.put((rank, stats, rows, mine))
    dist.destroy_process_group()


def test_a_rank_without_units_still_meets_the_collectives():
    "fewer units than ranks: the idle rank contributes zeros to the all-reduce and an empty list to the gather; the writer (rank 0) sees the full result"
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 37500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_idle_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s, u, r = _oracle_solve(_run_items()[:6], 0, 0)
    assert got[0][3] == [(0, 0)] and got[1][3] == []
    for rank, stats, rows, _ in got:
        assert stats['instances'] == 6 and stats['solved'] == int(np.sum(s)) and rows == (r if rank == 0 else [])


def test_batch_seed():
    assert parallel.batch_seed(42, 0, 0) == 42
    keys = {parallel.batch_seed(42, j, i) for j in range(64) for i in range(8)}
    assert len(keys) == 512 and all(0 <= k < 2 ** 64 for k in keys)


def _metrics_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, REPO)
    from oracle import binding
    items = dataset.random_ksat_items(10, 30, 3, m=100, seed=5000)
    mine, _ = parallel.shard_items(items, rank, world)           # (metrics are per-example sums: any partition reduces to the same means)
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
    # the driver's largest launch: 8 ranks (gloo on CPU here), same plumbing
    r8 = _run_bench({'PDP_DIST_BACKEND': 'gloo'}, '--gpus', '8', '--steps', '4', '--warmup', '1', '--selftest-collective')
    assert r8.returncode == 0, r8.stderr[-2000:]
    l8 = json.loads([l for l in r8.stdout.split('\n') if l.strip()][-1])
    assert l8['n_gpus'] == 8 and l8['rccl_ranks'] == 8 and l8['instances'] == 8000.0 and l8['rank_sum'] == 36.0 and abs(l8['max_elapsed_s'] - 0.08) < 1e-12
    # a rank count that does not match --gpus is refused, not silently run as one rank
    mism = _run_bench({'WORLD_SIZE': '1', 'RANK': '0'}, '--gpus', '2', '--selftest-collective')
    assert mism.returncode != 0 and 'WORLD_SIZE' in (mism.stderr + mism.stdout)


def _gather_worker(rank, world, port, q, rows_per_rank, values):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import time
    rng = np.random.RandomState(rank)
    # what a rank of configs[3] holds after its forward: one unit of 5 000 result rows, each the reference's JSON line with a 400-value solution
    text = "".join(json.dumps({"ID": "r%d_%d" % (rank, i), "label": 0, "solved": 0, "unsat_clauses": 3, "solution": rng.randint(0, 2, values).tolist()}) + "\n"
                   for i in range(rows_per_rank))
    units = [((rank, 0), text)]
    dist.barrier()
    t0 = time.perf_counter()
    parts = parallel.gather_units(units)
    dt = time.perf_counter() - t0
    q.put((rank, dt, len(parts), sum(len(p) for p in parts), [p[:40] for p in parts]))
    dist.destroy_process_group()


def test_gather_of_eight_ranks_of_5000_rows_to_the_writer():
    """configs[3] at eight ranks: 8 x 5 000 result rows of 400 values (~1.2 KB of JSON each, 6.5 MB per rank) reach the writer as one padded
    byte tensor per rank in ONE gather; the other ranks receive nothing (all_gather_object used to hand every rank all 52 MB).  Pins the
    order (unit order = rank order here), the byte counts, and a generous bound on the time of the host path before it meets eight GPUs."""
    world, rows_per_rank, values = 8, 5000, 400
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 39500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q, rows_per_rank, values)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    per_rank = got[1][3]
    assert per_rank > rows_per_rank * values * 3                              # "0, " per value
    rank0 = got[0]
    assert rank0[2] == world and abs(rank0[3] - world * per_rank) < 0.01 * world * per_rank
    assert [h.split('"')[3].split('_')[0] for h in rank0[4]] == ['r%d' % r for r in range(world)]     # rank-major = unit order
    for rank, dt, n_parts, n_bytes, _ in got[1:]:
        assert n_parts == 1 and n_bytes == per_rank or abs(n_bytes - per_rank) < 0.01 * per_rank      # the others keep their own unit only
    assert max(g[1] for g in got) < 20.0, [g[1] for g in got]                 # seconds, eight processes on the CPU box's cores (measured: well under 2)
