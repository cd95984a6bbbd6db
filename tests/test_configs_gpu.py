"""GPU tests on the workloads BASELINE.json names (configs[0], configs[2]'s width over many sweeps, configs[3]'s per-GPU shape,
configs[4]) -- each against outputs of the reference itself (tests/golden/generate_golden.py: config0, config4, neural_long) or,
at sizes the reference cannot run, against the CPU oracle on a sub-batch (instances are independent: no message crosses an
instance boundary, so the big batch must give on those instances what the oracle computes for them alone)."""
import io
import json
import logging
import os

import numpy as np
import pytest
import torch

from helpers import load_golden, REPO
from test_hip_ops import t, npy

pytestmark = pytest.mark.gpu
LOG = logging.getLogger('test')
GOLD = os.path.join(REPO, 'tests', 'golden')
RTOL, ATOL = 3e-4, 3e-5          # fp32 states against torch (another sgemm summation order, Sleef exp/log); same as test_api_forward.py


def _rows(path):
    return [l for l in open(path).read().split('\n') if l.strip()]


def _listdir_in_order(monkeypatch, directory, names):
    "os.listdir order is a property of the file system; the reference's rows were produced in the order its converter listed the files"
    real = os.listdir
    directory = os.path.realpath(str(directory))

    def fake(path='.'):
        if os.path.realpath(str(path)) == directory:
            assert sorted(names) == sorted(real(path))
            return list(names)
        return real(path)

    monkeypatch.setattr(os, 'listdir', fake)


# ---- configs[0] and the committed 20-file directory through `satyr.py -d` -------------------------------------------------------
def test_cli_dimacs_directory_equals_reference_rows(tmp_path, monkeypatch):
    """`satyr.py <p-d-p yaml> <dir> 50 -d -z 8 -s 7 -w 40` on the 20 committed DIMACS files, read by the native parser (no temporary
    JSON file): exactly the rows the reference CLI wrote for the same command with -c (cli_pdp_dimacs20.out.jsonl)."""
    import shutil
    import satyr
    ddir = tmp_path / 'cnf'
    shutil.copytree(os.path.join(GOLD, 'dimacs20'), str(ddir))
    ref = _rows(os.path.join(GOLD, 'cli_pdp_dimacs20.out.jsonl'))
    _listdir_in_order(monkeypatch, ddir, [json.loads(r)['ID'] for r in ref])
    out = tmp_path / 'out.jsonl'
    satyr.main([os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-sp-pytorch.yaml'), str(ddir), '50', '-d', '-z', '8', '-s', '7', '-w', '40',
                '-o', str(out)])
    assert _rows(str(out)) == ref
    assert not os.path.exists(str(ddir / 'temp_problem_file.json'))


def test_cli_directory_with_a_big_instance(tmp_path):
    """`satyr.py -d` on a directory that mixes 30 small files with one instance far past the LDS limit (n = 3000, 10 500 clauses), and on
    the big file alone: the persistent loop (per-instance routing: the big one as a workgroup team; alone: exact single-instance mode)
    writes the rows of the strict step-wise loop, Walk-SAT pass included (--rng philox: the same device-side numbers on both)."""
    import satyr
    from pdp import generator
    yaml_ = os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-walksat-pytorch.yaml')
    mixed = tmp_path / 'mixed'; mixed.mkdir()
    alone = tmp_path / 'alone'; alone.mkdir()
    for i in range(30):
        generator.write_dimacs(str(mixed / ('s_%03d.cnf' % i)), 60, generator.uniform_ksat(60, 240, 3, np.random.RandomState(700 + i)))
    big = generator.uniform_ksat(3000, 10500, 3, np.random.RandomState(31337))
    generator.write_dimacs(str(mixed / 'big.cnf'), 3000, big)
    generator.write_dimacs(str(alone / 'big.cnf'), 3000, big)
    for ddir in (mixed, alone):
        rows = []
        for extra in ([], ['--stepwise']):
            out = tmp_path / ('out_%s_%d.jsonl' % (ddir.name, len(extra)))
            satyr.main([yaml_, str(ddir), '60', '-d', '-z', '100', '-s', '3', '-w', '50', '--rng', 'philox', '-o', str(out)] + extra)
            rows.append(_rows(str(out)))
        assert rows[0] == rows[1] and len(rows[0]) == (31 if ddir is mixed else 1)


@pytest.mark.parametrize('extra', [[], ['--stepwise']])
def test_config0_cli_equals_reference_rows(tmp_path, monkeypatch, extra):
    """BASELINE configs[0]: 'p-d-p' on 100 random 3-SAT DIMACS files n=50 m=210, batch_size=100, T=50 (default -w 100 -e 0.5), the
    reference run with --cpu_mode: the 100 rows (solved, unsat_clauses, the whole assignment) equal the reference's, on the
    persistent one-launch loop and on the step-wise loop."""
    import satyr
    from pdp import generator
    ref = _rows(os.path.join(GOLD, 'cli_config0.out.jsonl'))
    ddir = tmp_path / 'cfg0'
    ddir.mkdir()
    for i in range(100):
        generator.write_dimacs(str(ddir / ('c0_%03d_%d.cnf' % (i, i % 2))), 50, generator.uniform_ksat(50, 210, 3, np.random.RandomState(9000 + i)))
    _listdir_in_order(monkeypatch, ddir, [json.loads(r)['ID'] for r in ref])
    out = tmp_path / 'out.jsonl'
    satyr.main([os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-sp-pytorch.yaml'), str(ddir), '50', '-d', '-z', '100', '-s', '7', '-o', str(out)] + extra)
    got = _rows(str(out))
    assert len(got) == 100 and got == ref
    assert sum(json.loads(r)['solved'] for r in got) == 13


# ---- configs[4] at the reference's size: mixed k, replication 4, dynamic batching, p-nd-np + Walk-SAT --------------------------------
def _neural_cfg(model_type, H, **kw):
    c = dict(model_type=model_type, model_name='t-' + model_type, verbose=False, local_search_iteration=0, epsilon=0.5, tolerance=0.02, t_max=100,
             pi=0.01, decimation_probability=0.5, rng='torch', random_seed=0, hidden_dim=H, edge_feature_dim=1, meta_feature_dim=0,
             prediction_dim=1, mem_hidden_dim=100, agg_hidden_dim=100, mem_agg_hidden_dim=50, classifier_dim=50, test_batch_limit=40000000,
             batch_size=5000, test_recurrence_num=1, max_cache_size=100000)
    c.update(kw)
    return c


def _load_weights(m, d, alias_file):
    alias = json.load(open(os.path.join(GOLD, alias_file)))
    sd = {}
    for key, canon in alias.items():
        k = 'w__' + canon.replace('.', '__')
        if key == '_global_step':
            sd[key] = torch.zeros(1)
        elif k in d.files:
            sd[key] = torch.from_numpy(d[k])
    m.load_state_dict(sd, strict=True)


def test_config4_predict_equals_reference():
    """BASELINE configs[4] at a size the reference runs: 'p-nd-np' (hidden 32), 14 instances of mixed random k-SAT (k in {3,4,5}, n in [20,60]),
    `-b 4`, a test_batch_limit that cuts the loader batch into five dynamic segments, 25 Walk-SAT steps.  `predict()` (loader, divider,
    replication, forward, Walk-SAT on the reference's torch stream, de-duplication, post-processor) must write the reference's rows --
    the de-duplicated assignments bit for bit -- with per-segment per-sweep predictions within RTOL / ATOL and equal active masks."""
    from pdp.trainer import SatFactorGraphTrainer
    d = load_golden('config4_mixed')
    T, H, w, R, seed, limit, nseg = [int(x) for x in d['meta']]
    tr = SatFactorGraphTrainer(_neural_cfg('p-nd-np', H, local_search_iteration=w, test_recurrence_num=T, batch_size=14, test_batch_limit=limit),
                               use_cuda=True, logger=LOG)
    m = tr._model_list[0]
    _load_weights(m, d, 'state_dict_alias_map_config4.json')
    rec, seg = {}, {'i': -1, 'sweep': 0, 'edges': []}
    orig_check, orig_batch = tr._check_recurrence_termination, tr._predict_batch

    def check(active, prediction, sp):
        rec['seg%d_pred_%d' % (seg['i'], seg['sweep'])] = npy(prediction[0]).reshape(-1)
        orig_check(active, prediction, sp)
        rec['seg%d_active_%d' % (seg['i'], seg['sweep'])] = npy(active).reshape(-1)
        seg['sweep'] += 1

    def predict_batch(graph_map, *a, **k):
        seg['i'] += 1; seg['sweep'] = 0; seg['edges'].append(int(graph_map.size(1)))
        return orig_batch(graph_map, *a, **k)

    tr._check_recurrence_termination = check
    tr._predict_batch = predict_batch
    torch.manual_seed(seed); np.random.seed(seed)
    buf = io.StringIO()
    tr.predict(os.path.join(GOLD, 'config4_mixed.json'), buf, import_path_base=None, post_processor=tr._post_process_predictions, batch_replication=R)
    assert seg['edges'] == [int(x) for x in d['segment_edges']] and len(seg['edges']) == nseg
    keys = [k for k in d.files if k.startswith('seg') and k[3].isdigit()]
    assert sorted(keys) == sorted(rec)
    for k in keys:
        if '_pred_' in k:
            np.testing.assert_allclose(rec[k], d[k], rtol=RTOL, atol=ATOL, err_msg=k)
        else:
            np.testing.assert_array_equal(rec[k].astype(np.int64), d[k].astype(np.int64), err_msg=k)
    assert _rows_of(buf.getvalue()) == _rows(os.path.join(GOLD, 'config4_mixed.out.jsonl'))
    # the global generator advanced exactly as far as the reference's (DataLoader base seed + Walk-SAT draws)
    nxt = torch.rand(4)
    torch.manual_seed(seed); torch.empty((), dtype=torch.int64).random_(); torch.rand(int(d['rand_sizes'].sum()))
    np.testing.assert_array_equal(nxt.numpy(), torch.rand(4).numpy())


def _rows_of(text):
    return [l for l in text.split('\n') if l.strip()]


# ---- hidden 128 over 24 sweeps against the reference -------------------------------------------------------------------------------
@pytest.mark.parametrize('mt', ['np-nd-np', 'p-nd-np'])
def test_neural_long_equals_reference(mt):
    """configs[2]'s width (hidden 128) on 8 instances of bench.py's family (n=200 m=840, 20 160 edges = 315 edge tiles), 24 sweeps from the
    test mode's random initial state (same torch seed as the reference run): every per-sweep prediction, the active masks and samples of
    the GRU states within RTOL / ATOL of the reference, thresholded final assignment identical."""
    from pdp.trainer import SatFactorGraphTrainer
    from pdp.factorgraph import dataset
    d = load_golden('neural_long_' + mt.replace('-', '_'))
    n, mcl, T, H, sweeps = [int(x) for x in d['meta']]
    tr = SatFactorGraphTrainer(_neural_cfg(mt, H), use_cuda=True, logger=LOG)
    m = tr._model_list[0]
    if mt == 'np-nd-np':
        _load_weights(m, load_golden('trace_neural_h128'), 'state_dict_alias_map.json')
    else:
        _load_weights(m, d, 'state_dict_alias_map_pndnp_h128.json')
    items = []
    for sd in d['seeds']:
        items += dataset.random_ksat_items(1, n, 3, m=mcl, seed=int(sd))
    dev = torch.device('cuda:0')
    b = dataset.to_torch(dataset.collate_segment(items), dev)
    gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
    rec = []

    def check(active, prediction, sp):
        p = npy(prediction[0]).reshape(-1)
        tr._check_recurrence_termination(active, prediction, sp)
        rec.append((p, npy(active).reshape(-1).copy()))

    dec = []
    hook = m._decimator.register_forward_hook(lambda mod, inp, outp: dec.append((npy(outp[0][::997]), npy(outp[1][::997]))))
    torch.manual_seed(3)
    with torch.no_grad():
        st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=True, batch_replication=1)
        for a, k in zip((st[0][0], st[0][1], st[1][0], st[1][1]), ('init_prop_0', 'init_prop_1', 'init_dec_0', 'init_dec_1')):
            np.testing.assert_array_equal(npy(a[::997]), d[k + '_sample'], err_msg=k)
        pred, (ps, ds) = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                           is_training=False, iteration_num=T, check_termination=check, batch_replication=1)
    hook.remove()
    assert len(rec) == sweeps == len(dec) and m.last_run['iterations'] == sweeps
    for i, (p, am) in enumerate(rec):
        np.testing.assert_allclose(p, d['pred_%d' % i], rtol=RTOL, atol=ATOL, err_msg='pred %d' % i)
        np.testing.assert_array_equal(am.astype(np.int64), d['active_mask_%d' % i].astype(np.int64))
        np.testing.assert_allclose(dec[i][0], d['dec_v_sample_%d' % i], rtol=RTOL, atol=ATOL, err_msg='dec_v %d' % i)
        np.testing.assert_allclose(dec[i][1], d['dec_f_sample_%d' % i], rtol=RTOL, atol=ATOL, err_msg='dec_f %d' % i)
    np.testing.assert_array_equal(npy(pred[0])[:, 0], d['final_prediction'])


# ---- configs[3]'s per-GPU shape: 5 000 instances of n=400 m=1680 -------------------------------------------------------------------------
N3, M3, B3, H3 = 400, 1680, 5000, 128


@pytest.fixture(scope='module')
def big400():
    from pdp.factorgraph import dataset
    items = dataset.random_ksat_items(B3, N3, 3, m=M3, seed=31)
    host = dataset.collate_segment(items)
    return items, host


def _offsets(items):
    v0 = np.concatenate(([0], np.cumsum([it[0] for it in items])))
    e0 = np.concatenate(([0], np.cumsum([it[2].shape[1] for it in items])))
    return v0, e0


def test_config3_shape_neural_forward_equals_oracle_on_a_sub_batch(big400, oracle):
    """np-nd-np, hidden 128, T=8 sweeps through the Python API on configs[3]'s per-GPU batch (25.2 M edges; [E,128] states of 12.9 GB, past
    32-bit element offsets): for 16 instances spread over the batch -- the first, the last, and the ones around the 2^31-element mark --
    every per-sweep prediction and the final decimator states equal, bit for bit, what the oracle computes for those 16 instances as a
    batch of their own with the same seeded random weights (the neural operators are row- / instance-local)."""
    from pdp.trainer import SatFactorGraphTrainer
    from pdp.factorgraph import dataset
    items, host = big400
    dev = torch.device('cuda:0')
    T = 8
    torch.manual_seed(1234)
    tr = SatFactorGraphTrainer(_neural_cfg('np-nd-np', H3, batch_size=B3, test_batch_limit=1 << 62), use_cuda=True, logger=LOG)
    m = tr._model_list[0]
    v0, e0 = _offsets(items)
    mark = int(np.searchsorted(e0, (1 << 31) // H3))           # the instance whose rows cross element 2^31 of an [E,128] state
    pick = sorted(set([0, 1, 2, mark - 1, mark, mark + 1, B3 // 2, B3 - 3, B3 - 2, B3 - 1] + [517, 1033, 2999, 3777, 4242, 4821]))
    assert len(pick) == 16
    vsel = np.concatenate([np.arange(v0[i], v0[i + 1]) for i in pick])
    esel = np.concatenate([np.arange(e0[i], e0[i + 1]) for i in pick])
    tv, te = torch.from_numpy(vsel).to(dev), torch.from_numpy(esel).to(dev)
    b = dataset.to_torch(host, dev)
    gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
    preds = []

    def check(active, prediction, sp):
        preds.append(npy(prediction[0].reshape(-1)[tv]))
        tr._check_recurrence_termination(active, prediction, sp)

    with torch.no_grad():
        st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1)
        pred, (ps, ds) = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                           is_training=False, iteration_num=T, check_termination=check, batch_replication=1)
    assert m.last_run['iterations'] == T and gm.size(1) * H3 > 2 ** 31
    got_final = npy(pred[0].reshape(-1)[tv])
    got_dec = [npy(ds[0][te]), npy(ds[1][te])]
    got_prop = [npy(ps[0][te]), npy(ps[1][te])]
    del pred, ps, ds, st
    torch.cuda.empty_cache()
    # the oracle on the 16 instances, with the model's weights
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    w = oracle.neural_weights({k.replace('.', '__'): v for k, v in sd.items()}, prefix='')
    sub = dataset.collate_segment([items[i] for i in pick])
    op = oracle.Problem(sub['graph_map'], sub['batch_variable_map'], sub['batch_function_map'], sub['edge_feature'])
    op.simplify()
    z = lambda: np.zeros((op.E, H3), np.float32)
    trace = []
    final, ost = oracle.neural_forward(op, w, (z(), z(), z(), z()), T, trace=trace)
    assert ost['iterations'] == T == len(preds)
    for i, trc in enumerate(trace):
        np.testing.assert_array_equal(preds[i], trc['pred'], err_msg='sweep %d' % i)
    np.testing.assert_array_equal(got_final, final)
    np.testing.assert_array_equal(got_dec[0], ost['dec_v']); np.testing.assert_array_equal(got_dec[1], ost['dec_f'])
    np.testing.assert_array_equal(got_prop[0], ost['prop_v']); np.testing.assert_array_equal(got_prop[1], ost['prop_f'])


def test_config3_shape_walksat_1000_steps_equals_oracle_on_a_sub_batch(big400, oracle):
    """Walk-SAT w=1000 (configs[3]) on the full per-GPU batch of 5 000 instances of n=400 (2 M variables) from a random assignment, the
    per-step random numbers handed over as device arrays (8 GB: every instance sees the same numbers in the big batch and in the
    oracle's sub-batch): for 16 instances the final assignment equals the oracle's strict step-by-step search bit for bit; at full size
    the unsatisfied-clause counts of the result are re-derived on the CPU, and the search never increases an instance's energy flag
    incorrectly (solved instances stay solved: cnf_eval of the result agrees with the energy)."""
    from pdp import native
    from pdp.factorgraph import dataset
    items, host = big400
    dev = torch.device('cuda:0')
    w, eps = 1000, 0.5
    b = dataset.to_torch(host, dev)
    hp = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    hp.simplify()
    V, B = hp.V, hp.B
    g = torch.Generator(device=dev); g.manual_seed(77)
    pred0 = (torch.rand(V, device=dev, generator=g) > 0.5).float()
    var_rand = torch.rand(w, V, device=dev, generator=g)
    coin = torch.rand(w, B, device=dev, generator=g)
    out, steps = hp.local_search(pred0, w, eps, var_rand, coin)
    assert steps == w                                               # half of the instances are unsatisfiable at alpha = 4.2: the search never stops early
    final = hp.update_solution(out.reshape(-1).contiguous())
    v0, e0 = _offsets(items)
    pick = [0, 1, 7, 499, 500, 1234, 2047, 2048, 2500, 3333, 4095, 4096, 4500, 4997, 4998, 4999]
    vsel = np.concatenate([np.arange(v0[i], v0[i + 1]) for i in pick])
    tv = torch.from_numpy(vsel).to(dev)
    ti = torch.tensor(pick, device=dev)
    # the oracle's stream: per step rand(V_sub) then rand(B_sub) (solver.py:457,460)
    stream = torch.cat((var_rand[:, tv], coin[:, ti]), dim=1).contiguous().cpu().numpy().reshape(-1)
    sub = dataset.collate_segment([items[i] for i in pick])
    op = oracle.Problem(sub['graph_map'], sub['batch_variable_map'], sub['batch_function_map'], sub['edge_feature'])
    op.simplify()
    ref, rsteps, cur = op.local_search(npy(pred0[tv]), w, eps, stream=stream)
    assert rsteps == w and cur == stream.size
    np.testing.assert_array_equal(npy(out.reshape(-1)[tv]), ref)
    np.testing.assert_array_equal(npy(final.reshape(-1)[tv]), op.update_solution(ref))
    # full size: clause counts of the result re-derived on the CPU from the assignment
    solved, unsat = hp.cnf_eval(final.reshape(-1).contiguous())
    x = npy(final).reshape(-1)
    gmh, sgn = host['graph_map'], host['edge_feature'].reshape(-1)
    lit_true = (sgn * x[gmh[0]] + (1.0 - sgn) / 2.0) > 0.5
    sat_clause = np.zeros(host['batch_function_map'].shape[0], bool)
    np.logical_or.at(sat_clause, gmh[1], lit_true)
    per_inst_unsat = np.bincount(host['batch_function_map'][~sat_clause], minlength=B)
    np.testing.assert_array_equal(npy(unsat).reshape(-1).astype(np.int64), per_inst_unsat)
    # Walk-SAT made progress at scale: fewer unsatisfied clauses than the random start
    _, unsat0 = hp.cnf_eval(hp.update_solution(pred0).reshape(-1).contiguous())
    assert float(unsat.sum()) < 0.5 * float(unsat0.sum())


# ---- configs[4]'s real shape on one GPU against the oracle on a sub-batch ---------------------------------------------------------------------
def test_config4_shape_equals_oracle_on_a_sub_batch(oracle):
    """configs[4]'s family at its real instance sizes: 'p-nd-np', hidden 128, mixed random k-SAT with k in {3,4,5}, n in [100,500],
    m = 0.9 alpha_k n, batch_replication 4, the loader batch cut by the reference's edge x hidden limit (dynamic batching), T=4 sweeps and
    30 Walk-SAT steps on the torch stream.  For every dynamic segment, a few of its instances (the smallest ones: the oracle is a scalar
    CPU program) are run through the oracle as a replicated batch of their own with the slice of the random stream the big batch gave
    them: de-duplicated predictions equal bit for bit."""
    from pdp.trainer import SatFactorGraphTrainer
    from pdp.factorgraph import dataset
    dev = torch.device('cuda:0')
    H, T, w, R, Bn = 128, 4, 30, 4, 120
    rng = np.random.RandomState(0)
    alpha = {3: 0.9 * 4.27, 4: 0.9 * 9.93, 5: 0.9 * 21.12}
    items = []
    for i in range(Bn):
        k = int(rng.choice([3, 4, 5])); n = int(rng.randint(100, 501))
        items += dataset.random_ksat_items(1, n, k, m=int(round(alpha[k] * n)), seed=1000 + i)
    edges = [it[2].shape[1] for it in items]
    limit = R * H * (sum(edges) // 2 + 1)                      # two or three segments, the largest instances first
    segs = dataset.divide(edges, limit // R, H)
    assert len(segs) >= 2
    torch.manual_seed(4321)
    tr = SatFactorGraphTrainer(_neural_cfg('p-nd-np', H, local_search_iteration=w, batch_size=Bn, test_batch_limit=limit), use_cuda=True, logger=LOG)
    m = tr._model_list[0]
    sd = {k.replace('.', '__'): v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    weights = oracle.pnd_weights(sd, prefix='')
    for si, seg in enumerate(segs):
        its = [items[j] for j in seg]
        hb = dataset.collate_segment(its)
        b = dataset.to_torch(hb, dev)
        gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
        V0, B0 = int(bvm.numel()), len(its)
        torch.manual_seed(100 + si)
        with torch.no_grad():
            st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=R)
            pred, _ = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                        is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination, batch_replication=R)
        assert m.last_run['iterations'] == T and m.last_run['walksat_steps'] == w and pred[0].numel() == V0
        torch.manual_seed(100 + si)
        draws = torch.rand(w * R * (V0 + B0)).view(w, R * (V0 + B0))            # what _local_search drew (one piece: w (V+B) < 2^26)
        v0, _ = _offsets(its)
        pick, tot = [], 0                                       # segments are sorted by descending edge count: take from the small end
        for i in range(len(its) - 1, -1, -1):
            if len(pick) >= 2 and (tot + its[i][2].shape[1] > 25000 or len(pick) >= 6):
                break
            pick.append(i); tot += its[i][2].shape[1]
        pick = sorted(pick)
        vloc = np.concatenate([np.arange(v0[i], v0[i + 1]) for i in pick])
        cols = np.concatenate([r * V0 + vloc for r in range(R)] + [R * V0 + r * B0 + np.asarray(pick) for r in range(R)])
        stream = draws[:, torch.from_numpy(cols)].contiguous().numpy().reshape(-1)
        sub = dataset.collate_segment([its[i] for i in pick])
        op = oracle.Problem(sub['graph_map'], sub['batch_variable_map'], sub['batch_function_map'], sub['edge_feature'], replication=R)
        op.simplify()
        E = op.E
        q = np.full((E, 3), 1.0, np.float32) / np.float32(3.0)
        fs = np.zeros((E, 2), np.float32); fs[:, 0] = 0.5
        ws = dict(steps=w, epsilon=0.5, stream=stream, cursor=0)
        final, ost = oracle.pnd_forward(op, weights, (q, fs, np.zeros((E, H), np.float32), np.zeros((E, H), np.float32)), T, walksat=ws)
        assert ost['iterations'] == T and ost['walksat_steps'] == w and ws['cursor'] == stream.size
        np.testing.assert_array_equal(npy(pred[0]).reshape(-1)[vloc], final, err_msg='segment %d' % si)
