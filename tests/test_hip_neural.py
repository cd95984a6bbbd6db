"""GPU parity tests of the neural operators (fp32 MFMA kernels in csrc/pdp_neural.hip) against the CPU oracle: the
matrix-core products are k-ordered fmaf chains, the oracle computes the same chains, so results must be bit-exact."""
import numpy as np
import pytest
import torch

from helpers import random_batch, load_golden
from test_hip_ops import t, npy, make_pair

pytestmark = pytest.mark.gpu


def rand_agg(rng, din, m1, a, g, out, fd):
    s = lambda *sh: (rng.randn(*sh) * 0.3).astype(np.float32)
    return dict(W1m=s(m1, din), b1m=s(m1), W2m=s(a, m1), W1a=s(g, a + fd), b1a=s(g), W2a=s(out, g))


def dev_agg(w, fd):
    from pdp import native
    return native.AggregatorWeights(t(w['W1m']), t(w['b1m']), t(w['W2m']), t(w['W1a']), t(w['b1a']), t(w['W2a']), fd)


@pytest.mark.parametrize('H,m1,a,g,grid', [(32, 100, 50, 100, None), (128, 100, 50, 100, None), (20, 36, 17, 40, None),
                                           (128, 100, 50, 100, 3), (32, 100, 50, 100, 2), (150, 100, 50, 100, None), (150, 100, 50, 100, 2),
                                           (128, 100, 50, 100, -2), (150, 100, 50, 100, -1)])
def test_aggregator_gru_predict_bit_exact(oracle, monkeypatch, H, m1, a, g, grid):
    from pdp import native
    if grid and grid < 0:   # the generic tile kernels on the shapes that have specialised ones (PDP_NEURAL_GENERIC: the cross-check switch)
        monkeypatch.setenv('PDP_NEURAL_GENERIC', '1')
        grid = -grid
    if grid:          # persistent kernels: many tiles per workgroup (cross-tile prefetch and the pipelined GRU's carried epilogue)
        monkeypatch.setenv('PDP_NEURAL_GRID', str(grid))
    b = random_batch(batch=9, n=25, mixed=True, seed=77)
    hp, op = make_pair(oracle, b)
    hp.simplify(); op.simplify()
    rng = np.random.RandomState(H)
    assign = np.zeros(op.V, np.float32); pick = rng.choice(op.V, size=op.V // 6, replace=False); assign[pick] = rng.randint(0, 2, len(pick)) * 2 - 1
    hp.set_variables(t(assign)); op.set_variables(assign)
    hp.refresh_edge_mask(); em, _ = op.refresh_edge_mask()
    ev, ec, es, vi, fi = op.graph()
    E, V, F, B = op.E, op.V, op.F, op.B
    state = (rng.randn(E, H) * 0.5).astype(np.float32); old = (rng.randn(E, H) * 0.5).astype(np.float32)
    am = (rng.rand(B) > 0.3).astype(np.uint8)
    mask = am[vi[ev]].astype(np.float32)
    w = rand_agg(rng, H + 1, m1, a, g, H, 1)
    for by_var, rows, nrows in ((True, ev, V), (False, ec, F)):
        for use_em in (True, False):
            ref = oracle.aggregator(rows, nrows, state, es, em if use_em else None, False, w)
            ref = mask[:, None] * ref + (1.0 - mask[:, None]) * old
            got = hp.neural_aggregate_edges(dev_agg(w, 1), by_var, t(state), hp.edge_mask if use_em else None, t(am), t(old))
            np.testing.assert_array_equal(npy(got), ref.astype(np.float32), err_msg='agg by_var=%s em=%s' % (by_var, use_em))
    # GRU
    s = lambda *sh: (rng.randn(*sh) * 0.2).astype(np.float32)
    gw = dict(W_ih=s(3 * H, H + 1), W_hh=s(3 * H, H), b_ih=s(3 * H), b_hh=s(3 * H))
    hprev = (rng.randn(E, H) * 0.5).astype(np.float32)
    ref = oracle.gru(state, es, hprev, mask=mask, **gw)
    got = hp.neural_gru(native.GruWeights(t(gw['W_ih']), t(gw['W_hh']), t(gw['b_ih']), t(gw['b_hh'])), t(state), t(hprev), t(am))
    np.testing.assert_array_equal(npy(got), ref)
    # predictor (include_self aggregator + perceptron head)
    wp = rand_agg(rng, H + 1, m1, a, g, H, 0)
    hw = dict(W1=s(50, H), b1=s(50), W2=s(1, 50))
    agg = oracle.aggregator(ev, V, state, es, em, True, wp)
    ref = oracle.perceptron(agg, hw['W1'], hw['b1'], hw['W2'])
    got = hp.neural_predict(dev_agg(wp, 0), native.HeadWeights(t(hw['W1']), t(hw['b1']), t(hw['W2']), 'sigmoid'), t(state), hp.edge_mask)
    np.testing.assert_array_equal(npy(got), ref)


@pytest.mark.parametrize('H,generic', [(128, False), (150, False), (32, False), (128, True)])
def test_neural_operators_at_the_edges_of_the_activation_range(oracle, monkeypatch, H, generic):
    """Pre-activations far outside the range the random-weight tests reach: |x| up to a few thousand (logsigmoid's exponential underflows at
    -104.5, the sigmoids clamp at 87 / 89, tanh at 10), exact zeros, and rows holding NaN, +inf, -inf and 1e30 -- the select-free scalar forms
    of include/pdp_math.h (the oracle) and the device's pair-wise packed forms must agree on every bit, NaN for NaN (pdp_neural.hip:
    pk_logsigmoid, k_gru_pipe / k_gru_wave activation slices; reference: F.logsigmoid in util.py:56-74, nn.GRUCell in pdp_decimate.py:51-87)."""
    from pdp import native
    if generic:
        monkeypatch.setenv('PDP_NEURAL_GENERIC', '1')
    b = random_batch(batch=7, n=30, mixed=True, seed=91)
    hp, op = make_pair(oracle, b)
    hp.simplify(); op.simplify()
    hp.refresh_edge_mask(); em, _ = op.refresh_edge_mask()
    ev, ec, es, vi, fi = op.graph()
    E, V, F, B = op.E, op.V, op.F, op.B
    rng = np.random.RandomState(1000 + H)
    state = (rng.randn(E, H) * 0.5).astype(np.float32)
    state[rng.rand(E) < 0.3] *= 400.0                      # rows whose pre-activations run into every clamp
    state[rng.rand(E) < 0.1] = 0.0
    special = rng.choice(E, size=8, replace=False)
    for row, val in zip(special, (np.nan, np.inf, -np.inf, 1e30, -1e30, np.nan, 3e38, -3e38)):
        state[row, rng.randint(0, H)] = val
    old = (rng.randn(E, H) * 0.5).astype(np.float32)
    am = np.ones(B, np.uint8); am[rng.randint(0, B)] = 0
    mask = am[vi[ev]].astype(np.float32)
    w = rand_agg(rng, H + 1, 100, 50, 100, H, 1)
    with np.errstate(all='ignore'):
        for by_var, rows, nrows in ((True, ev, V), (False, ec, F)):
            ref = oracle.aggregator(rows, nrows, state, es, em, False, w)
            ref = mask[:, None] * ref + (1.0 - mask[:, None]) * old
            got = npy(hp.neural_aggregate_edges(dev_agg(w, 1), by_var, t(state), hp.edge_mask, t(am), t(old)))
            np.testing.assert_array_equal(got, ref.astype(np.float32), err_msg='aggregator by_var=%s' % by_var)
            assert np.isnan(got).any() and np.isfinite(got).any()
        s = lambda *sh: (rng.randn(*sh) * 0.2).astype(np.float32)
        gw = dict(W_ih=s(3 * H, H + 1), W_hh=s(3 * H, H), b_ih=s(3 * H), b_hh=s(3 * H))
        hprev = (rng.randn(E, H) * 0.5).astype(np.float32)
        hprev[rng.rand(E) < 0.2] *= 300.0
        ref = oracle.gru(state, es, hprev, mask=mask, **gw)
        got = npy(hp.neural_gru(native.GruWeights(t(gw['W_ih']), t(gw['W_hh']), t(gw['b_ih']), t(gw['b_hh'])), t(state), t(hprev), t(am)))
        np.testing.assert_array_equal(got, ref)
        assert np.isnan(got).any() and np.isfinite(got).any()


@pytest.mark.parametrize('H,grid', [(32, None), (128, None), (20, None), (128, 2)])
def test_sp_adaptors_and_adapted_propagate_bit_exact(oracle, monkeypatch, H, grid):
    """adaptor form of the SP propagator (model type p-nd-np): the projections are k-ascending fmaf chains on both sides, so the
    log-domain inputs and the propagated surveys equal the oracle's bit for bit; small-GRU shapes (3 + 1 and 2 + 1 inputs) too."""
    from pdp import native
    if grid:
        monkeypatch.setenv('PDP_NEURAL_GRID', str(grid))
    b = random_batch(batch=7, n=22, mixed=True, seed=91)
    hp, op = make_pair(oracle, b)
    hp.simplify(); op.simplify()
    rng = np.random.RandomState(H + 5)
    assign = np.zeros(op.V, np.float32); pick = rng.choice(op.V, size=op.V // 7, replace=False); assign[pick] = rng.randint(0, 2, len(pick)) * 2 - 1
    hp.set_variables(t(assign)); op.set_variables(assign)
    hp.refresh_edge_mask(); em, _ = op.refresh_edge_mask()
    ev, ec, es, vi, fi = op.graph()
    E, B = op.E, op.B
    dv = (rng.randn(E, H) * 0.7).astype(np.float32); df = (rng.randn(E, H) * 0.7).astype(np.float32)
    dv[3, 1] = np.nan; df[5, H - 1] = np.nan; df[7, 0] = np.inf          # NaN / inf propagate like torch (sign(NaN) = NaN)
    w_f = (rng.randn(1, H) * 0.4).astype(np.float32); W_v = (rng.randn(2, H) * 0.4).astype(np.float32)
    xlog_ref, fs2_ref = oracle.sp_adaptors(dv, df, w_f, W_v)
    xlog, fs2 = hp.sp_adaptors(t(dv), t(df), t(w_f.reshape(-1)), t(W_v))
    np.testing.assert_array_equal(npy(xlog), xlog_ref)
    np.testing.assert_array_equal(npy(fs2), fs2_ref)
    am = (rng.rand(B) > 0.3).astype(np.uint8)
    iq = rng.rand(E, 3).astype(np.float32); ifs = rng.rand(E, 2).astype(np.float32)
    for use_em in (True, False):
        q_ref, fs_ref = op.sp_propagate_adapted(xlog_ref, fs2_ref, em if use_em else None, am, iq, ifs, 0.0)
        q, fs = hp.sp_propagate_adapted(xlog, fs2, hp.edge_mask if use_em else None, t(am), t(iq), t(ifs), 0.0)
        np.testing.assert_array_equal(npy(q), q_ref); np.testing.assert_array_equal(npy(fs), fs_ref)
    # the decimator's two GRU cells of this model type: [E,3] + sign and [E,2] + sign inputs
    mask = am[vi[ev]].astype(np.float32)
    s = lambda *sh: (rng.randn(*sh) * 0.3).astype(np.float32)
    for width, inp in ((3, q_ref), (2, fs_ref)):
        gw = dict(W_ih=s(3 * H, width + 1), W_hh=s(3 * H, H), b_ih=s(3 * H), b_hh=s(3 * H))
        ref = oracle.gru(inp, es, dv, mask=mask, **gw)
        got = hp.neural_gru(native.GruWeights(t(gw['W_ih']), t(gw['W_hh']), t(gw['b_ih']), t(gw['b_hh'])), t(inp), t(dv), t(am))
        np.testing.assert_array_equal(npy(got), ref)


def test_neural_predictor_with_both_classifiers_equals_reference(oracle):
    """NeuralPredictor built with a variable AND a function classifier (pdp_predict.py:49-91).  The variable branch is the fused native
    predictor; the function branch -- no solver of the reference's factory builds it -- runs on the generic layer / row-sum operators.  Both
    against the reference's predictions (with and without an edge mask), the function branch also against the oracle's per-clause form."""
    from pdp.nn import pdp_predict
    from pdp.nn.solver import SATProblem
    from pdp.trainer import Perceptron
    d = load_golden('predictor_function_branch')
    dev = torch.device('cuda:0')
    gm, bvm, bfm, ef = [t(d[k]) for k in ('graph_map', 'batch_variable_map', 'batch_function_map', 'edge_feature')]
    sp = SATProblem((gm, bvm, bfm, ef, None, None), dev, 1)
    H = d['dec_v'].shape[1]
    pr = pdp_predict.NeuralPredictor(dev, H, 1, 1, 0, 12, 10, 6, variable_classifier=Perceptron(H, 5, 1), function_classifier=Perceptron(H, 7, 1)).to(dev)
    # canonical names only: the `_module_list` entries are the same Parameter objects under a second name
    sd = {k: torch.from_numpy(d['w__' + k.replace('.', '__')]) for k in pr.state_dict().keys() if '_module_list' not in k}
    missing, unexpected = pr.load_state_dict(sd, strict=False)
    assert not unexpected and all('_module_list' in k for k in missing) and len(sd) == 18
    dv, df, em = t(d['dec_v']), t(d['dec_f']), t(d['edge_mask']).reshape(-1, 1)
    with torch.no_grad():
        pv, pf = pr((dv, df, em), sp)
        pv2, pf2 = pr((dv, df), sp)
    assert tuple(pv.shape) == (sp._variable_num, 1) and tuple(pf.shape) == (sp._function_num, 1)
    for got, key in ((pv, 'pred_v_masked'), (pf, 'pred_f_masked'), (pv2, 'pred_v'), (pf2, 'pred_f')):
        np.testing.assert_allclose(npy(got)[:, 0], d[key], rtol=3e-5, atol=3e-6, err_msg=key)
    # the oracle's per-clause aggregator + head (the same restatement the variable branch is pinned with)
    op = oracle.Problem(d['graph_map'], d['batch_variable_map'], d['batch_function_map'], d['edge_feature'], 1)
    ev, ec, es, vi, fi = op.graph()
    g = lambda name: d['w__' + name]
    w = dict(W1m=g('_function_aggregator___W1_m__weight'), b1m=g('_function_aggregator___W1_m__bias'), W2m=g('_function_aggregator___W2_m__weight'),
             W1a=g('_function_aggregator___W1_a__weight'), b1a=g('_function_aggregator___W1_a__bias'), W2a=g('_function_aggregator___W2_a__weight'))
    agg = oracle.aggregator(ec, op.F, d['dec_f'], es, d['edge_mask'], True, w)
    ref = oracle.perceptron(agg, g('_function_classifier___layer1__weight'), g('_function_classifier___layer1__bias'), g('_function_classifier___layer2__weight'))
    np.testing.assert_allclose(npy(pf)[:, 0], np.asarray(ref).reshape(-1), rtol=3e-6, atol=3e-7)
