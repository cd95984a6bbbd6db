"""N ranks = 1 rank, row for row (BASELINE's "identical solved-fraction at 8 GPUs"), on the GPU through the CLI.

The box has one GPU, so the two ranks share it (gloo instead of RCCL: RCCL wants one GPU per rank; the data path has no collective, the
single all-reduce of the counters and the gather of the rows work the same on both back ends).  RCCL itself runs at world size 1
(PDP_DIST_FORCE=1: test_rccl_runs_the_collectives_at_world_size_one).  The unsharded run is the same satyr.py in one process.
Inputs have at least three loader batches; the middle batch of the p-d-p input is the NaN-poisoned one of
tests/golden/headline_n200_poison.npz (reference run: first NaN at sweep 81, nothing of the batch is decimated afterwards), so the
rows depend on which instances share a batch -- the coupling domain the dealer must keep (reference: base.py:252-278, dataset.py:189-211).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import REPO, load_golden

pytestmark = pytest.mark.gpu

SATYR = os.path.join(REPO, 'pdp-solver_amd', 'satyr.py')


def _lines(items):
    out = []
    for vn, fn, gm, ef, label, misc in items:
        signed = ((gm[0] + 1) * ef.astype(np.int64)).tolist()
        out.append(json.dumps([[int(vn), int(fn)], [int(x) for x in signed], [int(x) + 1 for x in gm[1]], int(label), misc]))
    return out


def _run(argv, ranks, out, port, force_env=None):
    env = dict(os.environ)
    if force_env:
        env.update(force_env)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr', '127.0.0.1',
               '--master-port', str(port), SATYR] + argv + ['-o', out]
    elif ranks > 1:
        env['PDP_DIST_BACKEND'] = 'gloo'
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr', '127.0.0.1',
               '--master-port', str(port), SATYR] + argv + ['-o', out]
    else:
        cmd = [sys.executable, SATYR] + argv + ['-o', out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=env, timeout=900, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return [l for l in open(out).read().split('\n') if l.strip()], r.stderr


def test_two_ranks_write_the_rows_of_one_rank_with_a_poisoned_batch(tmp_path):
    from pdp.factorgraph import dataset
    d = load_golden('headline_n200_poison')
    n, mcl, T, seed, sweeps = [int(x) for x in d['meta']]
    poison = []
    for sd in d['seeds']:
        poison += dataset.random_ksat_items(1, n, 3, m=mcl, seed=int(sd))
    z = len(poison)                                                        # 50: the golden batch is loader batch 1 of 4
    items = dataset.random_ksat_items(z, n, 3, m=mcl, seed=91000) + poison + dataset.random_ksat_items(z, n, 3, m=mcl, seed=92000) \
        + dataset.random_ksat_items(17, 120, 3, seed=93000)
    path = tmp_path / 'in.json'
    path.write_text("\n".join(_lines(items)) + "\n")
    argv = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-sp-pytorch.yaml'), str(path), str(T), '-z', str(z), '-s', '11', '-w', '60',
            '--rng', 'philox', '-v']
    one, _ = _run(argv, 1, str(tmp_path / 'one.jsonl'), 0)
    two, log = _run(argv, 2, str(tmp_path / 'two.jsonl'), 29731)
    assert len(one) == len(items) and one == two
    # the counters of the single all-reduce = the totals of the rows
    rows = [json.loads(l) for l in one]
    tail = [l for l in log.split('\n') if 'instances %d' % len(items) in l]
    assert tail and ('solved %d,' % sum(r['solved'] for r in rows)) in tail[-1] and ('clauses %d' % sum(r['unsat_clauses'] for r in rows)) in tail[-1]
    # loader batch 1 is the golden's poisoned batch, in its order
    assert [r['ID'] for r in rows[z:2 * z]] == [it[5][0] for it in poison]
    # the default --rng torch cannot be dealt to ranks (one sequential host stream): a model that draws from it is switched to philox
    # with a warning, and writes the philox rows -- a torch.distributed.run command without --rng keeps working
    dflt, log = _run([a for a in argv if a not in ('--rng', 'philox')], 2, str(tmp_path / 'dflt.jsonl'), 29733)
    assert dflt == one and 'switching to --rng philox' in log


def test_two_ranks_share_one_loader_batch_cut_into_segments(tmp_path):
    """configs[4]'s shape: ONE loader batch (-z >= the instance count) that the batch limit cuts into >= 4 dynamic segments, -b 4.  The unit
    that is dealt is the segment (= one forward call of the reference, base.py:252-278), so both ranks work on the one batch, and the
    rows are the single-process rows.  p-nd-np draws no random number without Walk-SAT: it runs on two ranks with the default --rng."""
    import re
    from pdp.factorgraph import dataset
    items = []
    for i in range(40):
        items += dataset.random_ksat_items(1, 60 + 9 * (i % 7), 3 + (i % 3), seed=97000 + i)
    path = tmp_path / 'one_batch.json'
    path.write_text("\n".join(_lines(items)) + "\n")
    edges = [it[2].shape[1] for it in items]
    limit = 4 * 3 * max(edges) * 4                      # -b 4, hidden_dim 3 (p-d-p): about four of the largest instances per segment
    assert len(dataset.divide(edges, limit // 4, 3)) >= 4
    base = [str(path), '30', '-z', '64', '-s', '5', '-b', '4', '-l', str(limit), '-v']
    pdp = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-sp-pytorch.yaml')] + base + ['-w', '40', '--rng', 'philox']
    one, _ = _run(pdp, 1, str(tmp_path / 'one.jsonl'), 0)
    two, log = _run(pdp, 2, str(tmp_path / 'two.jsonl'), 29741)
    assert len(one) == len(items) and one == two
    units = {int(m.group(1)): int(m.group(2)) for m in re.finditer(r'rank (\d) of 2 solved (\d+) units', log)}
    assert set(units) == {0, 1} and min(units.values()) >= 2 and sum(units.values()) == len(dataset.divide(edges, limit // 4, 3))
    # the hybrid model of configs[4] with the weights trained here, default --rng (nothing is drawn with -w 0): also equal, no switch
    hyb = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-nd-np-demo-h128.yaml'), str(path), '12', '-z', '64', '-s', '5', '-b', '4', '-w', '0',
           '-l', str(4 * 128 * max(edges) * 4), '-v']
    one, _ = _run(hyb, 1, str(tmp_path / 'h_one.jsonl'), 0)
    two, log = _run(hyb, 2, str(tmp_path / 'h_two.jsonl'), 29743)
    assert len(one) == len(items) and one == two and 'switching to --rng philox' not in log
    units = {int(m.group(1)): int(m.group(2)) for m in re.finditer(r'rank (\d) of 2 solved (\d+) units', log)}
    assert set(units) == {0, 1} and min(units.values()) >= 1


def test_isolated_run_spreads_one_forward_over_the_ranks(tmp_path):
    """--isolated removes the couplings inside a forward, so the instance is the unit: ONE loader batch that is ONE segment (config 2's
    shape: fewer forwards than GPUs) is cut into one contiguous instance range per rank, and two / three ranks write the rows of the
    single process -- random fill and Walk-SAT draws included (a part's Philox counters start at its first variable / instance inside
    the segment).  The batch holds the golden poisoned instances: with the couplings on, their NaN would decide the other rows."""
    from pdp.factorgraph import dataset
    d = load_golden('headline_n200_poison')
    n, mcl, T, seed, sweeps = [int(x) for x in d['meta']]
    items = dataset.random_ksat_items(21, 120, 3, seed=94000)
    for sd in d['seeds'][:10]:
        items += dataset.random_ksat_items(1, n, 3, m=mcl, seed=int(sd))
    items += dataset.random_ksat_items(30, n, 3, m=mcl, seed=95000)
    path = tmp_path / 'in.json'
    path.write_text("\n".join(_lines(items)) + "\n")
    argv = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-sp-pytorch.yaml'), str(path), str(T), '-z', '5000', '-s', '11', '-w', '60',
            '--rng', 'philox', '--isolated', '-v']
    one, _ = _run(argv, 1, str(tmp_path / 'one.jsonl'), 0)
    assert len(one) == len(items)
    for ranks, port in ((2, 29751), (3, 29753)):
        many, log = _run(argv, ranks, str(tmp_path / ('r%d.jsonl' % ranks)), port)
        assert many == one
        assert 'one per rank' in log
        for r in range(ranks):
            assert ('rank %d of %d solved 1 units (forward calls): [(0, 0, %d)]' % (r, ranks, r)) in log
    # the strict run of the same batch differs (the NaN instances stop everybody's decimation), so the test sees the semantics it is about
    strict, _ = _run([a for a in argv if a != '--isolated'], 1, str(tmp_path / 'strict.jsonl'), 0)
    assert strict != one


def test_split_forward_keeps_the_couplings_across_ranks(tmp_path):
    """--split-forward: ONE coupled forward (one loader batch, one segment: config 2's shape) is spread over the ranks as contiguous instance
    ranges, and the reference's batch-wide reductions are completed across them -- per chunk of sweeps one exchange of the persistent solver's
    control words (first NaN sweep, exact-zero record, executed sweeps), one more after the poison replay, one for the Walk-SAT record.  The
    batch holds the golden NaN-producing instances: their NaN at sweep 81 stops the decimation of EVERY instance, also of those on the other
    ranks, so the rows depend on the exchange -- and two / three ranks write the rows of the single process.  (--isolated gives other rows.)"""
    from pdp.factorgraph import dataset
    d = load_golden('headline_n200_poison')
    n, mcl, T, seed, sweeps = [int(x) for x in d['meta']]
    items = dataset.random_ksat_items(20, n, 3, m=mcl, seed=96000)
    for sd in d['seeds'][:8]:
        items += dataset.random_ksat_items(1, n, 3, m=mcl, seed=int(sd))          # (the NaN instances land in the LAST part)
    path = tmp_path / 'in.json'
    path.write_text("\n".join(_lines(items)) + "\n")
    argv = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-sp-pytorch.yaml'), str(path), str(T), '-z', '5000', '-s', '11', '-w', '60',
            '--rng', 'philox', '-v']
    one, _ = _run(argv, 1, str(tmp_path / 'one.jsonl'), 0)
    iso, _ = _run(argv + ['--isolated'], 1, str(tmp_path / 'iso.jsonl'), 0)
    assert len(one) == len(items) and iso != one, "the couplings do not act on this input: pick other seeds"
    for ranks, port in ((2, 29761), (3, 29763)):
        many, log = _run(argv + ['--split-forward'], ranks, str(tmp_path / ('r%d.jsonl' % ranks)), port)
        assert 'coupled forwards spread over the ranks' in log
        for r in range(ranks):
            assert ('rank %d of %d solved 1 units (forward calls): [(0, 0, %d)]' % (r, ranks, r)) in log
        assert many == one
    # -w 0: no local search.  The zero-step Walk-SAT call is init / copy-out on every part (no batch-global reduction): the split stays a split,
    # nothing is reported as a failed speculation and solved again whole
    w0 = [a for a in argv]; w0[w0.index('-w') + 1] = '0'
    one_w0, _ = _run(w0, 1, str(tmp_path / 'one_w0.jsonl'), 0)
    two_w0, log = _run(w0 + ['--split-forward'], 2, str(tmp_path / 'two_w0.jsonl'), 29771)
    assert two_w0 == one_w0 and len(one_w0) == len(items) and 'solved whole' not in log and 'coupled forwards spread over the ranks' in log
    # dynamic segments: the batch limit cuts the loader batch into three forwards (10 + 10 + 8 instances, sorted by size), each spread over both ranks
    seg_argv = argv + ['-l', str(10 * 3 * max(it[2].shape[1] for it in items))]
    one_s, _ = _run(seg_argv, 1, str(tmp_path / 'one_s.jsonl'), 0)
    two_s, log = _run(seg_argv + ['--split-forward'], 2, str(tmp_path / 'two_s.jsonl'), 29767)
    assert two_s == one_s and '[(0, 0, 1), (0, 1, 1), (0, 2, 1)]' in log
    # the pure Walk-SAT solver (model type walk-sat: random fill + local search, no message passing): its record of the batch-global minimum is
    # completed across the parts the same way
    ws_argv = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-walksat-pytorch.yaml'), str(path), '300', '-z', '5000', '-s', '11', '--rng', 'philox', '-v']
    ws_one, _ = _run(ws_argv, 1, str(tmp_path / 'ws_one.jsonl'), 0)
    ws_two, _ = _run(ws_argv + ['--split-forward'], 2, str(tmp_path / 'ws_two.jsonl'), 29769)
    assert ws_two == ws_one and len(ws_one) == len(items)
    # a segment with fewer instances than ranks cannot be spread (every rank takes part in every exchange): refused on every rank
    env = dict(os.environ, PDP_DIST_BACKEND='gloo')
    small = tmp_path / 'small.json'
    small.write_text("\n".join(_lines(items[:2])) + "\n")
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '3', '--master-addr', '127.0.0.1', '--master-port', '29765',
                        SATYR, argv[0], str(small)] + argv[2:] + ['--split-forward', '-o', str(tmp_path / 'x.jsonl')],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=env, timeout=600, cwd=REPO)
    assert r.returncode != 0 and 'has 2 instances for 3 ranks' in r.stderr


def test_split_forward_falls_back_when_one_part_cannot_take_the_resident_loop(tmp_path):
    """Whether a part of a coupled forward can take the LDS-resident loops is a local fact -- here the LAST part holds an instance past the
    LDS limit -- and the other parts are about to wait in the first chunk's exchange.  The parts agree on it before anything else
    (pdp_sp_solve / pdp_local_search: one OR across the parts), every part reports the failed speculation, and the rank of part 0 solves
    the segment whole: the rows of the single process, no hang, no abort.  Same for the pure Walk-SAT model with the routing switched off."""
    from pdp.factorgraph import dataset
    items = dataset.random_ksat_items(13, 200, 3, m=840, seed=97000) + dataset.random_ksat_items(1, 1000, 3, seed=97100)
    path = tmp_path / 'in.json'
    path.write_text("\n".join(_lines(items)) + "\n")
    argv = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-sp-pytorch.yaml'), str(path), '40', '-z', '5000', '-s', '11', '-w', '40',
            '--rng', 'philox', '-v', '-l', '4000000000']
    one, _ = _run(argv, 1, str(tmp_path / 'one.jsonl'), 0)
    two, log = _run(argv + ['--split-forward'], 2, str(tmp_path / 'two.jsonl'), 29775)
    assert len(one) == len(items) and two == one
    assert 'solved whole on rank of part 0' in log
    ws_argv = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-walksat-pytorch.yaml'), str(path), '60', '-z', '5000', '-s', '11', '--rng', 'philox', '-v', '-l', '4000000000']
    env = {'PDP_WALKSAT_NO_ROUTING': '1'}
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ws_one, _ = _run(ws_argv, 1, str(tmp_path / 'ws_one.jsonl'), 0)
        ws_two, log = _run(ws_argv + ['--split-forward'], 2, str(tmp_path / 'ws_two.jsonl'), 29777)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert ws_two == ws_one and len(ws_one) == len(items)
    assert 'solved whole on rank of part 0' in log


def test_split_soak_short():
    """tools/split_soak.py for a few seconds: random batches (NaN-producing instances at random places, 30-130 sweeps, with and without
    Walk-SAT) cut into 2-4 parts that run in threads of one process -- the coupled form with an in-memory exchange, the isolated form without --
    equal the batch solved whole.  (The long form ran 22 787 batches / 102 204 exchanges without a mismatch.)"""
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'split_soak.py'), '12', '7'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [l for l in r.stdout.split('\n') if l.startswith('split soak:')][-1]
    coupled, isolated = int(line.split()[2]), int(line.split()[5])
    assert coupled >= 20 and isolated >= 20 and 'MISMATCH' not in r.stdout


def test_rccl_runs_the_collectives_at_world_size_one(tmp_path):
    """All the collective evidence a one-GPU box can give: under ``torch.distributed.run --nproc-per-node 1`` with PDP_DIST_FORCE=1 the CLI
    and bench.py join an ``nccl`` (= RCCL) process group of one rank; the device-side all-reduce of the counters, the object gather of
    the rows, bench.py's barrier + MAX / SUM all-reduces complete on the GPU back end, and the rows equal the run without a group."""
    from pdp.factorgraph import dataset
    items = dataset.random_ksat_items(30, 60, 3, seed=98000)
    path = tmp_path / 'in.json'
    path.write_text("\n".join(_lines(items)) + "\n")
    argv = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-sp-pytorch.yaml'), str(path), '40', '-z', '12', '-s', '3', '-w', '30', '-v']
    plain, _ = _run(argv, 1, str(tmp_path / 'plain.jsonl'), 0)
    forced, log = _run(argv, 1, str(tmp_path / 'forced.jsonl'), 29745, force_env={'PDP_DIST_FORCE': '1'})
    assert len(plain) == 30 and plain == forced
    assert '(1 ranks, nccl)' in log and 'rank 0 of 1 solved 3 units' in log
    # --split-forward in the group of one: the exchange of the coupled forward (an all-gather of a few words per chunk) runs through RCCL too
    # (on a batch large enough for the persistent solver's speculation to hold; the dozen-instance batches of the first input: below)
    big = dataset.random_ksat_items(40, 200, 3, m=840, seed=98100)
    bpath = tmp_path / 'big.json'
    bpath.write_text("\n".join(_lines(big)) + "\n")
    bargv = [argv[0], str(bpath), '40', '-z', '5000', '-s', '3', '-w', '30', '--rng', 'philox', '-v']
    bplain, _ = _run(bargv, 1, str(tmp_path / 'bplain.jsonl'), 0)
    split, log = _run(bargv + ['--split-forward'], 1, str(tmp_path / 'split.jsonl'), 29749, force_env={'PDP_DIST_FORCE': '1'})
    assert split == bplain and 'coupled forwards spread over the ranks' in log and '(1 ranks, nccl)' in log and '[(0, 0, 0)]' in log
    # the small batches of the first input fail the speculation: every part agrees on that, and the rank of the segment's first part solves the
    # segment whole with the single-process loops -- same rows again
    pplain, _ = _run(argv + ['--rng', 'philox'], 1, str(tmp_path / 'pplain.jsonl'), 0)
    fell, log = _run(argv + ['--rng', 'philox', '--split-forward'], 1, str(tmp_path / 'fell.jsonl'), 29771, force_env={'PDP_DIST_FORCE': '1'})
    assert fell == pplain and 'needs the single-process loop' in log
    fell2, log = _run(argv + ['--rng', 'philox', '--split-forward'], 2, str(tmp_path / 'fell2.jsonl'), 29773)
    assert fell2 == pplain and 'needs the single-process loop' in log
    env = dict(os.environ, PDP_DIST_FORCE='1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                        '--master-port', '29747', os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '400',
                        '--no-cpu-baseline', '--no-secondary'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=env,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads([l for l in r.stdout.split('\n') if l.startswith('{') and '"metric"' in l][-1])
    assert line['rccl_ranks'] == 1 and line['n_gpus'] == 1 and line['collective_backend'] == 'nccl' and line['value'] > 0


def test_two_ranks_equal_one_rank_with_dynamic_segments_and_replication(tmp_path):
    """mixed sizes, a batch limit that cuts every loader batch into several segments, batch replication 2, Walk-SAT post-processing: the
    segment index is part of the random key and the replicas stay on one rank"""
    from pdp.factorgraph import dataset
    items = []
    for i in range(46):
        items += dataset.random_ksat_items(1, 40 + 7 * (i % 9), 3 + (i % 3), seed=94000 + i)
    path = tmp_path / 'mixed.json'
    path.write_text("\n".join(_lines(items)) + "\n")
    argv = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-sp-pytorch.yaml'), str(path), '40', '-z', '12', '-s', '5', '-w', '80', '-b', '2',
            '-l', str(2 * 3 * 500 * 5), '--rng', 'philox']
    one, _ = _run(argv, 1, str(tmp_path / 'one.jsonl'), 0)
    two, _ = _run(argv, 2, str(tmp_path / 'two.jsonl'), 29735)
    assert len(one) == len(items) and one == two


def test_two_ranks_equal_one_rank_on_a_dimacs_directory(tmp_path):
    """-d: every rank lists the directory like the converter does, forms the same loader batches from the file list (balanced by file size)
    and parses only the files of its own batches; the output equals the single-process run's byte for byte"""
    from pdp import generator
    ddir = tmp_path / 'cnf'
    ddir.mkdir()
    rng = np.random.RandomState(4)
    for k in range(37):
        n = int(rng.randint(20, 90)); m = int(rng.uniform(2.5, 4.2) * n)
        clauses = generator.uniform_ksat(n, m, 3, np.random.RandomState(600 + k))
        (ddir / ('f%02d_%d.cnf' % (k, k % 2))).write_text('p cnf %d %d\n' % (n, len(clauses)) + ''.join(' '.join(str(x) for x in c) + ' 0\n' for c in clauses))
    argv = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-sp-pytorch.yaml'), str(ddir), '40', '-d', '-z', '8', '-s', '9', '-w', '50', '--rng', 'philox']
    one, _ = _run(argv, 1, str(tmp_path / 'one.jsonl'), 0)
    two, _ = _run(argv, 2, str(tmp_path / 'two.jsonl'), 29737)
    assert len(one) == 37 and one == two


def _timing_lines(log):
    import re
    return [(int(m.group(1)), float(m.group(2)), int(m.group(3)), float(m.group(4)))
            for m in re.finditer(r'rank (\d+) of \d+: gather of the result rows ([0-9.]+) ms; (\d+) exchanges of the coupled forwards, ([0-9.]+) ms each', log)]


def test_eight_ranks_on_the_shapes_of_configs_3_and_4(tmp_path):
    """The rank count BASELINE names, on this box's one GPU (eight gloo ranks share it: RCCL wants one GPU per rank).
    (a) configs[3]'s shape -- 8 loader batches, one forward each, np-nd-np hidden 128 with Walk-SAT: every rank solves exactly one batch and
        rank 0 writes the rows of the single process;
    (b) configs[4]'s shape under the strict semantics -- ONE loader batch of mixed 3- / 4-SAT in one segment, p-d-p with --split-forward: the
        coupled forward is spread over the eight ranks (one exchange of control words per chunk of sweeps), rows = the single process's.
    The wall time of the row gather and of one exchange is logged per rank (gloo here; RCCL on a node) and printed with -s."""
    import re
    from pdp.factorgraph import dataset
    # ---- (a) -------------------------------------------------------------------------------------------------------------------------
    items = []
    for i in range(8 * 12):
        items += dataset.random_ksat_items(1, 40 + (i % 5) * 8, 3, seed=88000 + i)
    path = tmp_path / 'c3.json'
    path.write_text("\n".join(_lines(items)) + "\n")
    argv = [os.path.join(REPO, 'config', 'Predict', 'PDP-np-nd-np-demo-h128.yaml'), str(path), '12', '-z', '12', '-s', '3', '-w', '50', '--rng', 'philox',
            '-l', '4000000000', '-v']
    one, _ = _run(argv, 1, str(tmp_path / 'c3_one.jsonl'), 0)
    eight, log = _run(argv, 8, str(tmp_path / 'c3_eight.jsonl'), 29781)
    assert len(one) == len(items) and eight == one
    units = {int(m.group(1)): int(m.group(2)) for m in re.finditer(r'rank (\d) of 8 solved (\d+) units', log)}
    assert units == {r: 1 for r in range(8)}
    t3 = _timing_lines(log)
    assert len(t3) == 8 and all(n_ex == 0 for _, _, n_ex, _ in t3)
    print('configs[3] shape, 8 gloo ranks on one GPU: gather of %d rows %.1f ms on the writer' % (len(items), [g for r, g, _, _ in t3 if r == 0][0]))
    # ---- (b) -------------------------------------------------------------------------------------------------------------------------
    items = []
    for i in range(64):
        # (mixed 3- / 4-SAT; every instance fits the LDS-resident solver -- a coupled forward runs on that one only, a part that cannot take it
        #  sends the whole segment to one rank: test_split_forward_falls_back_...)
        k = 4 if i % 3 == 0 else 3
        items += dataset.random_ksat_items(1, (40 + 3 * (i % 7)) if k == 4 else (60 + 7 * (i % 9)), k, seed=89000 + i)
    path = tmp_path / 'c4.json'
    path.write_text("\n".join(_lines(items)) + "\n")
    argv = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-sp-pytorch.yaml'), str(path), '60', '-z', '5000', '-s', '11', '-w', '40', '--rng', 'philox',
            '-l', '4000000000', '-v']
    one, _ = _run(argv, 1, str(tmp_path / 'c4_one.jsonl'), 0)
    eight, log = _run(argv + ['--split-forward'], 8, str(tmp_path / 'c4_eight.jsonl'), 29783)
    assert len(one) == len(items) and eight == one
    assert 'coupled forwards spread over the ranks' in log
    t4 = _timing_lines(log)
    assert len(t4) == 8
    assert 'solved whole' not in log and all(n_ex >= 3 for _, _, n_ex, _ in t4)         # 60 sweeps = 3 chunks of 25 (+ the Walk-SAT record)
    print('configs[4] shape (strict, --split-forward), 8 gloo ranks on one GPU: %d exchanges per rank, %.3f ms each (max over ranks); gather %.1f ms'
          % (max(n for _, _, n, _ in t4), max(ms for _, _, _, ms in t4), [g for r, g, _, _ in t4 if r == 0][0]))


def test_split_forward_of_a_neural_triple(tmp_path):
    """configs[2]'s shape on several GPUs: ONE loader batch, one segment, np-nd-np (the weights trained here) -- with segment dealing one rank would
    do all the work.  A neural triple couples its instances in one place, the end of the loop (`active_mask.sum() <= 0`, solver.py:383-384), so
    --split-forward spreads the forward over the ranks as instance ranges and completes that one bit per sweep across them (OR over the parts);
    the Walk-SAT behind it completes its record as for p-d-p.  Two and three ranks write the rows of the single process; the run where every
    part would stop on its own count (--isolated is refused for neural triples, so: no exchange = a single rank per part) is not what is tested
    here -- the parts must run the SAME number of sweeps, which the log shows."""
    import re
    from pdp.factorgraph import dataset
    items = []
    rng = np.random.RandomState(4)
    for i in range(45):
        n = int(rng.randint(10, 41))
        items += dataset.random_ksat_items(1, n, 3, m=int(round(rng.uniform(2.0, 4.0) * n)), seed=98000 + i)
    path = tmp_path / 'in.json'
    path.write_text("\n".join(_lines(items)) + "\n")
    for cfg_name, w in (('PDP-np-nd-np-demo-h128.yaml', '30'), ('PDP-p-nd-np-demo-h128.yaml', '0')):
        argv = [os.path.join(REPO, 'config', 'Predict', cfg_name), str(path), '25', '-z', '5000', '-s', '9', '-w', w, '--rng', 'philox', '-l', '4000000000', '-v']
        one, _ = _run(argv, 1, str(tmp_path / 'one.jsonl'), 0)
        assert len(one) == len(items)
        solved = sum(json.loads(l)['solved'] for l in one)
        assert 0 < solved < len(items), "the batch should hold instances solved along the way and unsolved ones"
        for ranks, port in ((2, 29791), (3, 29793)):
            many, log = _run(argv + ['--split-forward'], ranks, str(tmp_path / ('r%d.jsonl' % ranks)), port)
            assert many == one and 'coupled forwards spread over the ranks' in log and 'solved whole' not in log
            for r in range(ranks):
                assert ('rank %d of %d solved 1 units (forward calls): [(0, 0, %d)]' % (r, ranks, r)) in log
            ex = _timing_lines(log)
            assert len(ex) == ranks and len({n_ex for _, _, n_ex, _ in ex}) == 1 and ex[0][2] >= 1      # every part took part in every exchange
