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
