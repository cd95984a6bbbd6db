"""Pins the CPU oracle (oracle/pdp_oracle.c) against vectors captured from the unmodified
reference (tests/golden/generate_golden.py).  Integer-valued outputs must match exactly; fp32
messages within FP_RTOL/FP_ATOL (the oracle's exp/log are <1.5 ulp, torch's Sleef kernels 1 ulp)."""
import os

import numpy as np
import pytest

FP_RTOL = 2e-5
FP_ATOL = 2e-6


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def make_problem(oracle, d, replication=1):
    return oracle.Problem(d['graph_map'], d['batch_variable_map'], d['batch_function_map'], d['edge_feature'], replication)


def test_math_accuracy(oracle):
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.uniform(-100, 30, 200000), rng.uniform(-1, 1, 50000), [-92.1034, -103.9, 0.0, 30.0]]).astype(np.float32)
    y = oracle.math_apply('exp', x)
    ref = np.exp(x.astype(np.float64))
    ok = ref > 1e-37
    assert np.max(np.abs(y[ok] - ref[ok]) / ref[ok]) < 2.0e-7
    assert np.max(np.abs(y[~ok] - ref[~ok])) < 1.5e-45 + 2e-7 * 1e-37  # denormal grid
    assert abs(float(oracle.math_apply('exp', [-92.1034])[0]) - 1e-40) < 3e-45
    xl = np.concatenate([np.exp(rng.uniform(-95, 5, 200000)), rng.uniform(0.5, 2.0, 100000), [1e-40, 1.0, 1e-10]]).astype(np.float32)
    yl = oracle.math_apply('log', xl)
    refl = np.log(xl.astype(np.float64))
    err = np.abs(yl - refl) / np.maximum(np.abs(refl), 1e-30)
    assert np.max(err[np.abs(refl) > 1e-3]) < 2.0e-7
    assert np.max(np.abs(yl - refl)[np.abs(refl) <= 1e-3]) < 1e-9
    assert oracle.math_apply('log', [1.0])[0] == 0.0
    xs = rng.uniform(-40, 40, 100000).astype(np.float32)
    ls = oracle.math_apply('logsigmoid', xs)
    refs = -np.logaddexp(0, -xs.astype(np.float64))
    assert np.max(np.abs(ls - refs) / np.maximum(np.abs(refs), 1e-30)) < 4e-7
    th = oracle.math_apply('tanh', xs / 4)
    assert np.max(np.abs(th - np.tanh(xs.astype(np.float64) / 4))) < 2e-7
    tha = oracle.math_apply('tanh_abs', xs / 4)           # the GRU's candidate gate: absolute accuracy only
    assert np.max(np.abs(tha - np.tanh(xs.astype(np.float64) / 4))) < 1.5e-7
    assert np.all(np.sign(tha) * np.sign(xs) >= 0) and np.all(np.abs(tha) <= 1.0)
    sg = oracle.math_apply('sigmoid', xs)
    assert np.max(np.abs(sg - 1 / (1 + np.exp(-xs.astype(np.float64))))) < 1.5e-7
    u = oracle.math_apply('philox', np.zeros(100000))
    assert u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.01


def test_simplify_and_set_variables(oracle, golden_dir):
    d = load(golden_dir, 'problem_simplify')
    p = make_problem(oracle, d)
    p.simplify()
    av, af, sol, sat = p.state()
    np.testing.assert_array_equal(av, d['simplify_active_variables'])
    np.testing.assert_array_equal(af, d['simplify_active_functions'])
    np.testing.assert_array_equal(sol, d['simplify_solution'])
    np.testing.assert_array_equal(sat, d['simplify_is_sat'])
    p.set_variables(d['setvar_assignment'])
    av, af, sol, sat = p.state()
    np.testing.assert_array_equal(av, d['setvar_active_variables'])
    np.testing.assert_array_equal(af, d['setvar_active_functions'])
    np.testing.assert_array_equal(sol, d['setvar_solution'])
    np.testing.assert_array_equal(sat, d['setvar_is_sat'])


def test_replication_layout(oracle, golden_dir):
    d = load(golden_dir, 'problem_simplify')
    p = make_problem(oracle, d, replication=3)
    ev, ec, es, vi, fi = p.graph()
    np.testing.assert_array_equal(np.stack([ev, ec]), d['rep3_graph_map'])
    np.testing.assert_array_equal(vi, d['rep3_batch_variable_map'])
    np.testing.assert_array_equal(fi, d['rep3_batch_function_map'])
    np.testing.assert_array_equal(es, d['rep3_edge_feature'][:, 0])


def test_scorer_with_adaptors_equals_reference(oracle, golden_dir):
    "SurveyScorer(include_adaptors=True) (pdp_predict.py:145-152, 161-208): projector -> sigmoid / sign -> the score, oracle against the reference"
    d = load(golden_dir, 'scorer_adaptors')
    p = make_problem(oracle, d)
    p.simplify(); p.set_variables(d['assign'])
    np.testing.assert_array_equal(p.state()[1], d['active_functions'])
    W = d['weight']
    _, fs2 = oracle.sp_adaptors(d['message'], d['message'], W[0], W)
    for tag, pi in (('pi0', 0.0), ('pi01', 0.1)):
        np.testing.assert_allclose(p.survey_score(fs2, pi), d['score_' + tag], rtol=FP_RTOL, atol=FP_ATOL)


@pytest.fixture(scope='module')
def ops(oracle, golden_dir):
    d = load(golden_dir, 'ops_classical')
    p = make_problem(oracle, d)
    p.set_state(d['active_variables'], d['active_functions'], d['solution'])
    p.set_edge_mask(d['edge_mask'])
    return d, p


def test_edge_mask(ops):
    d, p = ops
    m, s = p.refresh_edge_mask()
    np.testing.assert_array_equal(m, d['edge_mask'])


def test_smooth_max_and_instance_max(ops):
    d, p = ops
    np.testing.assert_allclose(p.smooth_max(d['smax_in']), d['smax_out'], rtol=FP_RTOL, atol=FP_ATOL)
    np.testing.assert_array_equal(p.instance_max(d['vmax_in']), d['vmax_out'])
    np.testing.assert_array_equal(p.instance_argmax(d['vmax_in']), d['vargmax_out'])
    np.testing.assert_array_equal(p.instance_max(d['vmax_neg_in']), d['vmax_neg_out'])
    np.testing.assert_array_equal(p.instance_argmax(d['vmax_neg_in']), d['vargmax_neg_out'])


def test_cnf_eval(ops):
    d, p = ops
    s, u = p.cnf_eval(d['cnf_pred'])
    np.testing.assert_array_equal(s, d['cnf_solved'])
    np.testing.assert_array_equal(u, d['cnf_unsat'])


@pytest.mark.parametrize('tag,pi', [('pi0', 0.0), ('pi1', 0.1)])
def test_sp_propagate_and_score(ops, tag, pi):
    d, p = ops
    g = lambda k: d['sp_%s_%s' % (tag, k)]
    q, fs = p.sp_propagate(g('q'), g('fs'), d['edge_mask'], g('active_mask'), g('init_q'), g('init_fs'), pi)
    np.testing.assert_allclose(q, g('masked_q'), rtol=FP_RTOL, atol=FP_ATOL)
    np.testing.assert_allclose(fs, g('masked_fs'), rtol=FP_RTOL, atol=FP_ATOL)
    q, fs = p.sp_propagate(g('q'), g('fs'), None, None, g('init_q'), g('init_fs'), pi)
    np.testing.assert_allclose(q, g('plain_q'), rtol=FP_RTOL, atol=FP_ATOL)
    np.testing.assert_allclose(fs, g('plain_fs'), rtol=FP_RTOL, atol=FP_ATOL)
    np.testing.assert_allclose(p.survey_score(g('fs'), pi), d['score_%s' % tag], rtol=1e-4, atol=2e-6)


def test_energy(ops):
    d, p = ops
    en, uf = p.energy(d['energy_assignment'])
    np.testing.assert_array_equal(en, d['energy_per_instance'])
    np.testing.assert_array_equal(uf, d['energy_unsat_functions'])
    np.testing.assert_array_equal(p.energy_diff(d['energy_assignment']), d['energy_delta'])


TRACES = [('trace_pdp_n50', 'p-d-p', dict(tolerance=0.02, t_max=100)),
          ('trace_pdp_easy_ws', 'p-d-p', dict(tolerance=0.05, t_max=10)),
          ('trace_pdp_mixed', 'p-d-p', dict(tolerance=0.05, t_max=8)),
          ('trace_walksat_easy', 'walk-sat', {}),
          ('trace_pdp_rep3', 'p-d-p', dict(tolerance=0.05, t_max=6)),
          ('trace_reinforce_easy', 'reinforce', dict(pi=0.01, decimation_probability=0.5)),
          # an instance leaves through the gate and the reference's mask blend (mask * new + (1 - mask) * old) turns its messages and force into NaN
          ('trace_reinforce_nan_leak', 'reinforce', dict(pi=0.1, decimation_probability=0.6))]


@pytest.mark.parametrize('name,model,kw', TRACES)
def test_full_forward_trace(oracle, golden_dir, name, model, kw):
    """End-to-end: same inputs, same recorded torch.rand stream -> identical integer trajectory
    (active flags, solution, active mask per iteration), identical final assignment, same number of
    random numbers consumed, and fp32 states within tolerance."""
    d = load(golden_dir, name)
    T, w, seed, R = [int(x) for x in d['meta']]
    p = make_problem(oracle, d, replication=R)
    res = p.forward(model, T, local_search_iterations=w, epsilon=0.5, stream=d['rand_stream'], trace=True,
                    trace_float=True, **kw)
    it = int(d['iterations_run'][0])
    assert res['iterations_run'] == it
    if it:
        np.testing.assert_array_equal(res['trace_active_mask'][:it], d['trace_active_mask'])
        np.testing.assert_array_equal(res['trace_active_var'][:it], d['trace_active_variables'])
        np.testing.assert_array_equal(res['trace_active_fn'][:it], d['trace_active_functions'])
        np.testing.assert_array_equal(res['trace_solution'][:it], d['trace_solution'])
        for k in d.files:
            if k.startswith('prop_q_'):
                i = int(k.split('_')[-1])
                if name == 'trace_reinforce_nan_leak' and i > 2:
                    # 75 sweeps at the threshold amplify the 1-ulp differences between torch's kernels and pdp_math.h (the integer
                    # trajectories above are equal); what this trace pins is WHERE the NaNs are, sweep by sweep: instance 1 turns NaN
                    # while active, instance 0 (inactive from sweep 75) gets its first 33 NaN surveys at sweep 76 through the mask blend
                    assert np.array_equal(np.isnan(res['trace_q'][i]), np.isnan(d[k])), k
                    assert np.array_equal(np.isnan(res['trace_fs'][i]), np.isnan(d['prop_fs_%d' % i])), k
                    continue
                np.testing.assert_allclose(res['trace_q'][i], d[k], rtol=2e-4, atol=2e-6, err_msg=k)
    assert res['rand_consumed'] == int(d['rand_sizes'].sum())
    np.testing.assert_array_equal(res['prediction'], d['final_prediction'])
    if name == 'trace_reinforce_nan_leak':
        # final state: instance 0 is NaN throughout (1 083 surveys, 3 249 entries of q) although it was inactive from sweep 75 on; the force
        # column stays finite (torch.sign(NaN) = 0)
        assert np.isnan(d['final_dec_1'][:, 0]).sum() == 1367 and not np.isnan(d['final_dec_1'][:, 1]).any()
        assert np.array_equal(np.isnan(res['fs']), np.isnan(d['final_dec_1'])) and np.array_equal(np.isnan(res['q']), np.isnan(d['final_dec_0']))
        np.testing.assert_array_equal(res['fs'][:, 1], d['final_dec_1'][:, 1])
    elif 'final_prop_0' in d.files and R == 1:
        np.testing.assert_allclose(res['q'], d['final_prop_0'], rtol=5e-4, atol=5e-6)
        np.testing.assert_allclose(res['fs'], d['final_prop_1'], rtol=5e-4, atol=5e-6)
    if 'final_solved' in d.files:
        s, u = p.cnf_eval(res['prediction'])
        np.testing.assert_array_equal(s, d['final_solved'])
        np.testing.assert_array_equal(u, d['final_unsat'])


def test_sat_loss_and_test_metrics(oracle, golden_dir):
    """test mode (base.py:223-250, trainer.py:108-123): accuracy / recall errors from the clause check and the energy loss of
    SatLossEvaluator (util.py:178-197) against the reference's own _compute_evaluation_metrics on a labelled batch.  The loss is a
    mean of logs in another summation order than torch.mean: rtol 2e-6; the infinite case (a clause with zero weighted value) is inf."""
    d = load(golden_dir, 'test_metrics')
    p = make_problem(oracle, d)
    alpha, max_coeff, eps, sharp = [float(x) for x in d['params']]
    label = d['label'].reshape(-1)
    for k in range(3):
        pred = d['pred_%d' % k]
        coeff = np.float32(min(np.float32(d['global_step'][k]) ** np.float32(alpha), np.float32(max_coeff)))
        loss = p.sat_loss(pred, coeff, eps, int(sharp))
        solved, _ = p.cnf_eval(pred)
        out = (solved > 0.5).astype(np.float32)
        acc_err = np.mean(np.abs(out - label))
        rec_err = np.sum(label * np.abs(out - label)) / max(np.sum(label), 1e-8)
        ref = d['metrics'][k]
        assert abs(acc_err - ref[0]) < 1e-6 and abs(rec_err - ref[1]) < 1e-6
        if np.isinf(ref[2]):
            assert np.isinf(loss) and loss > 0
        else:
            assert abs(loss - ref[2]) <= 2e-6 * abs(ref[2]), (loss, ref[2])


def config0_items(golden_dir):
    "BASELINE configs[0]'s 100 instances (n=50 m=210, instance i = RandomState(9000 + i)) in the order of the reference CLI's rows"
    import json
    from pdp import generator
    from pdp.factorgraph import dataset
    rows = [json.loads(l) for l in open(os.path.join(golden_dir, 'cli_config0.out.jsonl')) if l.strip()]
    items = []
    for r in rows:
        i = int(r['ID'].split('_')[1])
        items.append(dataset.instance_from_clauses(50, generator.uniform_ksat(50, 210, 3, np.random.RandomState(9000 + i)), label=i % 2, name=r['ID']))
    return rows, items


def test_config0_cli_rows(oracle, golden_dir):
    """BASELINE configs[0] ('p-d-p' on 100 random 3-SAT DIMACS n=50, --cpu_mode, batch_size=100, T=50): the oracle, fed with the torch CPU
    stream the reference CLI consumes for `-s 7` (DataLoader base seed, random fill, 100 Walk-SAT steps), reproduces every row the
    reference wrote: solved flag, unsatisfied-clause count and the whole assignment of all 100 instances."""
    import torch
    from pdp.factorgraph import dataset
    rows, items = config0_items(golden_dir)
    b = dataset.collate_segment(items)
    p = oracle.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    torch.manual_seed(7)
    torch.empty((), dtype=torch.int64).random_()
    stream = torch.rand(p.V + 100 * (p.V + p.B)).numpy()
    res = p.forward('p-d-p', 50, local_search_iterations=100, epsilon=0.5, tolerance=0.02, t_max=100, stream=stream)
    solved, unsat = p.cnf_eval(res['prediction'])
    off = np.concatenate(([0], np.cumsum([it[0] for it in items])))
    for i, r in enumerate(rows):
        assert r['solved'] == int(solved[i]) and r['unsat_clauses'] == int(unsat[i]), r['ID']
        assert r['solution'] == (res['prediction'][off[i]:off[i + 1]] > 0.5).astype(int).tolist(), r['ID']
    assert sum(r['solved'] for r in rows) == 13


def test_torch_sparse_port_equals_oracle(oracle):
    """bench.py's second CPU baseline -- the PyTorch-CPU restatement of the reference's sparse-mm formulation (oracle/torch_sparse_port.py,
    SURVEY.md 8(d)(2)) -- runs the same algorithm as the C oracle: identical integer trajectory (active flags, solution, active mask per
    sweep, decimations included) and messages within the tolerance that separates torch's exp / log from include/pdp_math.h."""
    import torch
    from oracle import torch_sparse_port as port
    from pdp.factorgraph import dataset
    items = dataset.random_ksat_items(12, 40, 3, m=140, seed=900) + dataset.random_ksat_items(6, 30, 3, m=126, seed=950)
    b = dataset.collate_segment(items)
    T = 40
    p = oracle.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    res = p.forward('p-d-p', T, local_search_iterations=0, tolerance=0.05, t_max=10, seed=1, trace=True, trace_float=True)
    P = port.SparseBatch(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    trace = []
    with torch.no_grad():
        q, fs, done = port.forward_loop(P, T, tolerance=0.05, t_max=10, trace=trace)
    assert done == res['iterations_run'] and done > 10
    decimated = 0
    for i, tr in enumerate(trace):
        np.testing.assert_array_equal(tr['active_var'], res['trace_active_var'][i], err_msg='sweep %d' % i)
        np.testing.assert_array_equal(tr['active_fn'], res['trace_active_fn'][i])
        np.testing.assert_array_equal(tr['solution'], res['trace_solution'][i])
        np.testing.assert_array_equal(tr['active_mask'], res['trace_active_mask'][i])
        np.testing.assert_allclose(tr['q'], res['trace_q'][i], rtol=5e-4, atol=5e-6)
        decimated += int(i > 0 and (tr['active_var'] != trace[i - 1]['active_var']).any())
    assert decimated >= 3                       # the comparison covered decimation events (scorer, arg-max, set_variables, simplify)
