"""Pins the neural half of the CPU oracle (oracle/pdp_oracle_neural.c) against traces of the reference's
np-nd-np solver (seeded random weights, tests/golden/trace_neural_*.npz).  Dense products use another summation order
than torch's MKL sgemm, so floats are compared with a tolerance; the thresholded final assignment must match exactly."""
import numpy as np
import pytest

from helpers import load_golden

RTOL, ATOL = 3e-4, 3e-5


@pytest.mark.parametrize('name', ['trace_neural_h32', 'trace_neural_h128'])
def test_neural_forward_matches_reference(oracle, name):
    d = load_golden(name)
    T, H = [int(x) for x in d['meta']]
    p = oracle.Problem(d['graph_map'], d['batch_variable_map'], d['batch_function_map'], d['edge_feature'])
    p.simplify()
    w = oracle.neural_weights(d)
    trace = []
    final, st = oracle.neural_forward(p, w, (d['init_prop_v'], d['init_prop_f'], d['init_dec_v'], d['init_dec_f']), T, trace=trace)
    assert st['iterations'] == T
    for i, tr in enumerate(trace):
        np.testing.assert_allclose(tr['pred'], d['pred_%d' % i], rtol=RTOL, atol=ATOL, err_msg='pred %d' % i)
        np.testing.assert_array_equal(tr['active_mask'], d['active_mask_%d' % i])
        for k in ('prop_v', 'prop_f', 'dec_v', 'dec_f'):
            key = '%s_%d' % (k, i)
            if key in d.files:
                np.testing.assert_allclose(tr[k], d[key], rtol=RTOL, atol=ATOL, err_msg=key)
    np.testing.assert_array_equal(final, d['final_prediction'])


def test_p_nd_np_forward_matches_reference(oracle):
    """model type p-nd-np (SP propagator with adaptors + neural decimator + neural predictor).  The reference cannot construct this
    model as written; the golden trace comes from the reference with the one-word shim in
    tests/golden/generate_golden.py::gen_p_nd_np (function-message width 2), everything else unmodified."""
    d = load_golden('trace_p_nd_np')
    T, H, iters = [int(x) for x in d['meta']]
    p = oracle.Problem(d['graph_map'], d['batch_variable_map'], d['batch_function_map'], d['edge_feature'])
    p.simplify()
    w = oracle.pnd_weights(d)
    trace = []
    final, st = oracle.pnd_forward(p, w, (d['init_prop_q'], d['init_prop_fs'], d['init_dec_v'], d['init_dec_f']), T, trace=trace)
    assert st['iterations'] == iters
    for i, tr in enumerate(trace):
        np.testing.assert_allclose(tr['pred'], d['pred_%d' % i], rtol=RTOL, atol=ATOL, err_msg='pred %d' % i)
        np.testing.assert_array_equal(tr['active_mask'], d['active_mask_%d' % i])
        for k in ('prop_q', 'prop_fs', 'dec_v', 'dec_f'):
            np.testing.assert_allclose(tr[k], d['%s_%d' % (k, i)], rtol=RTOL, atol=ATOL, err_msg='%s %d' % (k, i))
    np.testing.assert_array_equal(final, d['final_prediction'])


@pytest.mark.parametrize('mt', ['np-nd-np', 'p-nd-np'])
def test_neural_long_pins(oracle, mt):
    """Hidden 128 (configs[2]'s width), 8 instances of bench.py's family (n=200 m=840: 20 160 edges), 24 sweeps from the test mode's random
    initial state (torch seed 3, regenerated here; a sample is pinned): every per-sweep prediction, the active mask and a sample of the
    decimator states of the reference within RTOL / ATOL, thresholded final assignment identical.  This is the long tie between the
    oracle's MFMA-ordered fmaf chains and torch's sgemm."""
    import torch
    from pdp.factorgraph import dataset
    d = load_golden('neural_long_' + mt.replace('-', '_'))
    n, mcl, T, H, sweeps = [int(x) for x in d['meta']]
    items = []
    for sd in d['seeds']:
        items += dataset.random_ksat_items(1, n, 3, m=mcl, seed=int(sd))
    b = dataset.collate_segment(items)
    p = oracle.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    p.simplify()
    E = p.E
    torch.manual_seed(3)
    if mt == 'np-nd-np':
        w = oracle.neural_weights(load_golden('trace_neural_h128'))
        init = [(2.0 * torch.rand(E, H) - 1.0).numpy() for _ in range(4)]           # pdp_propagate.py:97-108, pdp_decimate.py:89-100
        run, keys = oracle.neural_forward, ('dec_v', 'dec_f')
    else:
        w = oracle.pnd_weights(d)
        q = torch.rand(E, 3); q = q / torch.sum(q, 1).unsqueeze(1)                  # pdp_propagate.py:223-237
        fs = torch.rand(E, 2); fs[:, 1] = 0
        init = [q.numpy(), fs.numpy()] + [(2.0 * torch.rand(E, H) - 1.0).numpy() for _ in range(2)]
        run, keys = oracle.pnd_forward, ('dec_v', 'dec_f')
    for a, k in zip(init, ('init_prop_0', 'init_prop_1', 'init_dec_0', 'init_dec_1')):
        np.testing.assert_array_equal(a[::997], d[k + '_sample'], err_msg=k)
    trace = []
    final, st = run(p, w, init, T, trace=trace)
    assert st['iterations'] == sweeps == len(trace)
    for i, tr in enumerate(trace):
        np.testing.assert_allclose(tr['pred'], d['pred_%d' % i], rtol=RTOL, atol=ATOL, err_msg='pred %d' % i)
        np.testing.assert_array_equal(tr['active_mask'], d['active_mask_%d' % i])
        for k in keys:
            np.testing.assert_allclose(tr[k][::997], d['%s_sample_%d' % (k, i)], rtol=RTOL, atol=ATOL, err_msg='%s %d' % (k, i))
    np.testing.assert_array_equal(final, d['final_prediction'])


def config4_segments(d):
    "the loader batch of tests/golden/config4_mixed.json cut into the reference's dynamic segments (dataset.py:24-74 with limit // replication)"
    import os
    from helpers import REPO
    from pdp.factorgraph import dataset
    T, H, w, R, seed, limit, nseg = [int(x) for x in d['meta']]
    lines = [l for l in open(os.path.join(REPO, 'tests', 'golden', 'config4_mixed.json')).read().split('\n') if l.strip()]
    items = [dataset.parse_line(l) for l in lines]
    segs = dataset.divide([it[2].shape[1] for it in items], limit // R, H)
    assert len(segs) == nseg and [sum(items[j][2].shape[1] for j in s_) for s_ in segs] == [int(x) for x in d['segment_edges']]
    return items, segs


def test_config4_mixed_k_replicated_dynamic_batches(oracle):
    """BASELINE configs[4] at test size through the oracle: 'p-nd-np' on mixed random k-SAT (k in {3,4,5}, n in [20,60]), batch_replication 4,
    five dynamic segments, 25 Walk-SAT steps on the reference's torch stream, de-duplication -- against the reference's own predict() run
    (+ App. B-4 / B-5 shims): per-segment per-sweep predictions within RTOL / ATOL, active masks equal, and the rows the reference wrote
    (solved flag, unsatisfied clauses, the de-duplicated assignment) reproduced exactly."""
    import json
    import os
    import torch
    from helpers import REPO
    from pdp.factorgraph import dataset
    d = load_golden('config4_mixed')
    T, H, w, R, seed, limit, nseg = [int(x) for x in d['meta']]
    items, segs = config4_segments(d)
    weights = oracle.pnd_weights(d)
    ref_rows = [json.loads(l) for l in open(os.path.join(REPO, 'tests', 'golden', 'config4_mixed.out.jsonl')) if l.strip()]
    torch.manual_seed(seed)
    torch.empty((), dtype=torch.int64).random_()                    # the DataLoader iterator's base seed (base.py:258)
    stream = torch.rand(int(d['rand_sizes'].sum()) + 16).numpy()
    ws = dict(steps=w, epsilon=0.5, stream=stream, cursor=0)
    row = 0
    for si, seg in enumerate(segs):
        b = dataset.collate_segment([items[j] for j in seg])
        p = oracle.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'], replication=R)
        p1 = oracle.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
        p.simplify()
        E = p.E
        q = np.full((E, 3), 1.0, np.float32) / np.float32(3.0)
        fs = np.zeros((E, 2), np.float32); fs[:, 0] = 0.5
        trace = []
        final, st = oracle.pnd_forward(p, weights, (q, fs, np.zeros((E, H), np.float32), np.zeros((E, H), np.float32)), T, trace=trace, walksat=ws)
        for i, tr in enumerate(trace):
            np.testing.assert_allclose(tr['pred'], d['seg%d_pred_%d' % (si, i)], rtol=RTOL, atol=ATOL, err_msg='segment %d sweep %d' % (si, i))
            np.testing.assert_array_equal(tr['active_mask'], d['seg%d_active_%d' % (si, i)])
        assert 'seg%d_pred_%d' % (si, len(trace)) not in d.files
        solved, unsat = p1.cnf_eval(final)
        off = np.concatenate(([0], np.cumsum([items[j][0] for j in seg])))
        for k, j in enumerate(seg):
            r = ref_rows[row]; row += 1
            assert r['ID'] == items[j][5][0]
            assert (r['solved'], r['unsat_clauses']) == (int(solved[k]), int(unsat[k])), r['ID']
            assert r['solution'] == (final[off[k]:off[k + 1]] > 0.5).astype(int).tolist(), r['ID']
    assert row == len(ref_rows) and ws['cursor'] == int(d['rand_sizes'].sum())
