"""Full-size checks on the headline workload (BASELINE.json configs[1]: random 3-SAT n=200 m=840, batch 5000, T=100).

The workload bench.py times is compared with the ORACLE as a whole batch first (test_timed_workload_equals_the_oracle_on_the_whole_batch:
one oracle process, about two minutes; Reinforce at B=1500), then through properties that do not depend on the size:
  * two independent implementations agree bit for bit: the persistent LDS-resident solver (one workgroup per instance,
    speculation + device-side NaN-poison replay) and the strict step-wise kernels (batch-wide semantics with global flag words);
    this batch is NaN-poisoned at iteration 81, so the replay machinery is exercised at scale;
  * instance independence: below the poison iteration every instance of the big batch ends in exactly the state the ORACLE
    computes for a small sub-batch containing it;
  * the clause-satisfaction integers of the result are re-derived on the CPU from the assignment;
  * determinism: a second run reproduces every bit."""
import logging

import numpy as np
import pytest
import torch

from test_hip_ops import npy

pytestmark = pytest.mark.gpu
LOG = logging.getLogger('test')
N, M, B, T = 200, 840, 5000, 100


@pytest.fixture(scope='module')
def big():
    from pdp.factorgraph import dataset
    items = dataset.random_ksat_items(B, N, 3, m=M, seed=0)          # the benchmark's rank-0 batch
    host = dataset.collate_segment(items)
    return items, host, dataset.to_torch(host, torch.device('cuda:0'))


def _solve(tb, T_, tol=0.02, t_max=100):
    from pdp import native
    hp = native.Problem(tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'])
    hp.simplify()
    dev = torch.device('cuda:0')
    q = torch.full((hp.E, 3), 1.0, device=dev) / 3.0
    fs = torch.zeros(hp.E, 2, device=dev); fs[:, 0] = 0.5
    am = torch.ones(hp.B, dtype=torch.uint8, device=dev)
    dec = native.Decimator(hp)
    iters, lds = hp.sp_solve(q, fs, am, dec, T_, tol, t_max, inputs_disposable=True)      # exactly the call bench.py times
    return hp, q, fs, am, iters, lds


def test_timed_workload_equals_the_oracle_on_the_whole_batch(big, oracle, monkeypatch):
    """bench.py's rank-0 batch (B=5000, n=200, m=840, T=100, tolerance 0.02, t_max 100) through pdp_sp_solve -- the call the headline number
    times: chunked launches, NaN poison at sweep 81, the poison adopted by the workgroups that can and the device-side replay of the others --
    against the oracle's forward (reference loop solver.py:355-386 with its batch-wide couplings) on the WHOLE batch in one process.
    Everything bit for bit: messages, active flags, solution, per-instance mask, executed sweeps; then the random fill with the same Philox key
    and the final prediction.  Twice: as shipped (few or no instances left to replay) and with the looks for a recorded NaN switched off
    (round 4's form: half of the batch is replayed)."""
    items, host, tb = big
    op = oracle.Problem(host['graph_map'], host['batch_variable_map'], host['batch_function_map'], host['edge_feature'])
    res = op.forward('p-d-p', T, local_search_iterations=0, tolerance=0.02, t_max=100, seed=3, trace='mask')
    av, af, sol, _ = op.state()                                        # after the loop and the random fill (key 3)
    assert np.isnan(res['fs']).any()
    fixed = av == 0                                                     # decimated / simplified variables: their value is the loop's result
    for looks in (False, True):
        for name in ('PDP_SOLVE_NO_ADOPT', 'PDP_SOLVE_NO_RISK_ORDER', 'PDP_SOLVE_NO_EVENT_LOOK'):
            if looks: monkeypatch.delenv(name, raising=False)
            else: monkeypatch.setenv(name, '1')
        hp, q, fs, am, iters, lds = _solve(tb, T)
        assert lds and iters == res['iterations_run'] == T
        stats = hp.last_solve_stats
        assert stats['hbm_instances'] == 0, stats                       # every instance on the LDS-resident kernel
        assert looks or stats['replays'] > 0, stats                     # without the looks the poison replay ran
        np.testing.assert_array_equal(npy(q), res['q'])
        np.testing.assert_array_equal(npy(fs), res['fs'])
        np.testing.assert_array_equal(npy(am), res['trace_active_mask'][T - 1])
        np.testing.assert_array_equal(npy(hp.active_variables).reshape(-1), av)
        np.testing.assert_array_equal(npy(hp.active_functions).reshape(-1), af)
        np.testing.assert_array_equal(npy(hp.solution).reshape(-1)[fixed], sol[fixed])
    hp.random_fill(seed=3)                                              # the final predictor call (pdp_predict.py:118-128), same Philox key
    out, _ = hp.local_search(hp.solution.clone(), 0, 0.5, seed=3)
    pred = hp.update_solution(out.reshape(-1).contiguous())
    np.testing.assert_array_equal(npy(pred).reshape(-1), res['prediction'])
    solved, unsat = hp.cnf_eval(pred.reshape(-1).contiguous())
    o_solved, o_unsat = op.cnf_eval(res['prediction'])
    np.testing.assert_array_equal(npy(solved).reshape(-1), o_solved.reshape(-1))
    np.testing.assert_array_equal(npy(unsat).reshape(-1), o_unsat.reshape(-1))


def test_reinforce_instantiation_equals_the_oracle_at_b1500(big, oracle):
    """the persistent kernel's Reinforce instantiation (bench.py's secondary: pi 0.1, decimation probability 0.5) on the first 1 500
    instances of the headline batch, T=100, against the oracle's loop with the same coins"""
    from pdp import native
    from pdp.factorgraph import dataset
    items, _, _ = big
    Br = 1500
    host = dataset.collate_segment(items[:Br])
    tb = dataset.to_torch(host, torch.device('cuda:0'))
    coins = np.random.RandomState(11).rand(T).astype(np.float32)
    op = oracle.Problem(host['graph_map'], host['batch_variable_map'], host['batch_function_map'], host['edge_feature'])
    res = op.forward('reinforce', T, local_search_iterations=0, pi=0.1, decimation_probability=0.5, stream=coins, trace='mask')
    av, af, sol, _ = op.state()
    hp = native.Problem(tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'])
    hp.simplify()
    dev = torch.device('cuda:0')
    q = torch.full((hp.E, 3), 1.0, device=dev) / 3.0
    fs = torch.zeros(hp.E, 2, device=dev); fs[:, 0] = 0.5
    am = torch.ones(hp.B, dtype=torch.uint8, device=dev)
    iters, lds = hp.sp_solve(q, fs, am, native.Decimator(hp), T, 0.01, 0.0, pi=0.1, model=native.MODEL_REINFORCE,
                             coins=torch.from_numpy(coins).to(dev), decimation_probability=0.5)
    it = res['iterations_run']
    assert lds and iters == it and res['rand_consumed'] == it
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_variables).reshape(-1), av)
    np.testing.assert_array_equal(npy(hp.active_functions).reshape(-1), af)
    np.testing.assert_array_equal(npy(hp.solution).reshape(-1), sol)
    assert np.abs(res['fs'][:, 1]).sum() > 0


def test_persistent_equals_stepwise_at_full_size(big):
    from pdp.trainer import SatFactorGraphTrainer
    items, host, tb = big
    outs = []
    for persistent in (True, False):
        tr = SatFactorGraphTrainer(dict(model_type='p-d-p', model_name='full', verbose=False, local_search_iteration=0, epsilon=0.5,
                                        tolerance=0.02, t_max=100, rng='philox', random_seed=3, hidden_dim=3, persistent=persistent,
                                        test_batch_limit=40000000, batch_size=B, test_recurrence_num=1), use_cuda=True, logger=LOG)
        m = tr._model_list[0]
        ef = tb['edge_feature']
        with torch.no_grad():
            st = m.get_init_state(tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], ef, None, randomized=False, batch_replication=1)
            pred, (ps, ds) = m(init_state=st, graph_map=tb['graph_map'], batch_variable_map=tb['batch_variable_map'],
                               batch_function_map=tb['batch_function_map'], edge_feature=ef, meta_data=None, is_training=False,
                               iteration_num=T, check_termination=tr._check_recurrence_termination, batch_replication=1)
        assert m.last_run['path'] == ('persistent-lds' if persistent else 'stepwise'), m.last_run
        assert m.last_run['iterations'] == T
        sp = m._last_problem
        outs.append((npy(pred[0]), npy(ps[0]), npy(ps[1]), npy(sp._active_variables), npy(sp._active_functions), npy(sp._solution)))
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    # the poison really happened (NaN messages exist), i.e. the comparison covered the replay path
    assert np.isnan(outs[0][1]).any()


def test_instances_are_independent_below_the_poison_iteration(big, oracle):
    items, host, tb = big
    T_ = 60
    hp, q, fs, am, iters, lds = _solve(tb, T_)
    assert iters == T_ and lds
    from pdp.factorgraph import dataset
    # a sub-batch for the oracle: 24 instances, at least one with a variable removed by the initial simplify (it supplies
    # the exact zeros that keep the reference's batch-global min at 0, like the big batch does)
    av0 = npy(hp.active_variables).reshape(-1)
    v0 = np.concatenate(([0], np.cumsum([it[0] for it in items])))
    has_inactive = [i for i in range(B) if (av0[v0[i]:v0[i + 1]] == 0).any()]
    assert has_inactive, "no instance with an inactive variable in the batch"
    pick = sorted(set(has_inactive[:2] + list(range(7, 7 + 22))))
    sub = dataset.collate_segment([items[i] for i in pick])
    op = oracle.Problem(sub['graph_map'], sub['batch_variable_map'], sub['batch_function_map'], sub['edge_feature'])
    res = op.forward('p-d-p', T_, local_search_iterations=0, tolerance=0.02, t_max=100, seed=1, trace=True)
    e0 = np.concatenate(([0], np.cumsum([it[2].shape[1] for it in items])))
    qh, fsh, solh = npy(q), npy(fs), npy(hp.solution).reshape(-1)
    eo = vo = 0
    for i in pick:
        ne, nv = e0[i + 1] - e0[i], v0[i + 1] - v0[i]
        np.testing.assert_array_equal(qh[e0[i]:e0[i + 1]], res['q'][eo:eo + ne], err_msg='q of instance %d' % i)
        np.testing.assert_array_equal(fsh[e0[i]:e0[i + 1]], res['fs'][eo:eo + ne], err_msg='fs of instance %d' % i)
        np.testing.assert_array_equal(solh[v0[i]:v0[i + 1]], res['trace_solution'][T_ - 1][vo:vo + nv], err_msg='solution of instance %d' % i)
        eo += ne; vo += nv


def test_clause_counts_rederived_and_deterministic(big):
    items, host, tb = big
    hp, q, fs, am, iters, lds = _solve(tb, T)
    hp.random_fill(seed=5)
    pred = hp.update_solution(hp.solution.clone().reshape(-1))
    solved, unsat = hp.cnf_eval(pred.reshape(-1).contiguous())
    x = npy(pred).reshape(-1)
    gm, sgn = host['graph_map'], host['edge_feature'].reshape(-1)
    lit_true = (sgn * x[gm[0]] + (1.0 - sgn) / 2.0) > 0.5
    sat_clause = np.zeros(host['batch_function_map'].shape[0], bool)
    np.logical_or.at(sat_clause, gm[1], lit_true)
    per_inst_unsat = np.bincount(host['batch_function_map'][~sat_clause], minlength=B)
    np.testing.assert_array_equal(npy(unsat).reshape(-1).astype(np.int64), per_inst_unsat)
    np.testing.assert_array_equal(npy(solved).reshape(-1) == 1, per_inst_unsat == 0)
    hp2, q2, fs2, am2, iters2, _ = _solve(tb, T)
    assert iters2 == iters
    for a, b in ((q, q2), (fs, fs2), (hp.active_variables, hp2.active_variables), (am, am2)):
        np.testing.assert_array_equal(npy(a), npy(b))


def test_poison_adopted_by_later_workgroups_changes_the_replay_not_the_results(big, monkeypatch):
    """A workgroup of pass 1 that starts after another one has recorded the batch's first NaN sweep runs poisoned from there on and needs no
    replay, and pass 1 dispatches the instances closest to a NaN first so that this happens early.  How many workgroups adopt depends
    on the dispatch order -- the results must not: the batch (poisoned at sweep 81) ends in the same bits with the look switched off, in block
    order, and with other chunk lengths that move the poison inside its launch."""
    items, host, tb = big
    runs = []
    for adopt, chunk, risk in ((True, None, True), (False, None, True), (True, '7', True), (False, '7', False), (True, None, False), (True, '40', True)):
        if adopt: monkeypatch.delenv('PDP_SOLVE_NO_ADOPT', raising=False)
        else: monkeypatch.setenv('PDP_SOLVE_NO_ADOPT', '1')
        if risk: monkeypatch.delenv('PDP_SOLVE_NO_RISK_ORDER', raising=False)       # pass 1 in the order of k_order_by_risk / in block order
        else: monkeypatch.setenv('PDP_SOLVE_NO_RISK_ORDER', '1')
        if chunk == '7': monkeypatch.setenv('PDP_SOLVE_NO_EVENT_LOOK', '1')          # (the look in front of an event off in two of the runs)
        else: monkeypatch.delenv('PDP_SOLVE_NO_EVENT_LOOK', raising=False)
        if chunk: monkeypatch.setenv('PDP_SOLVE_CHUNK', chunk)
        else: monkeypatch.delenv('PDP_SOLVE_CHUNK', raising=False)
        hp, q, fs, am, iters, lds = _solve(tb, T)
        assert lds and iters == T
        runs.append([npy(x).copy() for x in (q, fs, am, hp.active_variables, hp.active_functions, hp.solution, hp.is_sat)])
    assert np.isnan(runs[0][1]).any()
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            np.testing.assert_array_equal(a, b)


def test_neural_operator_forms_agree_at_full_size(big, oracle, monkeypatch):
    """configs[2] shapes (hidden 128, 12.6 M edges): the specialised kernels (in-wave pipelined GRU, wave-private / prefetched aggregator
    halves, many tiles per persistent workgroup) give bit for bit what the plain tile kernels give, an active mask with holes included;
    a sample of GRU rows is checked against the oracle (the cell is row-wise, so a sample of rows is an exact check)."""
    from pdp import native
    _, host, tb = big
    dev = torch.device('cuda:0')
    p = native.Problem(tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'])
    E, H = p.E, 128
    g = torch.Generator(device='cpu'); g.manual_seed(5)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.2).to(dev)
    gw_host = [r(3 * H, H + 1), r(3 * H, H), r(3 * H), r(3 * H)]
    gw = native.GruWeights(*gw_host)
    aw = native.AggregatorWeights(r(100, H + 1), r(100), r(50, 100), r(100, 51), r(100), r(H, 100), 1)
    state = torch.randn(E, H, device=dev) * 0.5
    h = torch.randn(E, H, device=dev) * 0.5
    am = (torch.rand(p.B, device=dev) > 0.2).to(torch.uint8)
    new_gru = p.neural_gru(gw, state, h, am)
    new_agg = [p.neural_aggregate_edges(aw, bv, state, None, am, h) for bv in (True, False)]
    monkeypatch.setenv('PDP_NEURAL_GENERIC', '1')            # every operator on its generic tile kernel
    old_gru = p.neural_gru(gw, state, h, am)
    assert torch.equal(new_gru, old_gru)
    del old_gru
    for bv, new in zip((True, False), new_agg):
        assert torch.equal(new, p.neural_aggregate_edges(aw, bv, state, None, am, h))
    # oracle on a sample of edges
    rng = np.random.RandomState(3)
    idx = np.sort(rng.choice(E, 4096, replace=False))
    ti = torch.from_numpy(idx).to(dev)
    gm = npy(tb['graph_map']); vi = npy(tb['batch_variable_map'])
    mask = npy(am)[vi[gm[0][idx]]].astype(np.float32)
    es = npy(tb['edge_feature']).reshape(-1)[idx]
    ref = oracle.gru(npy(state[ti]), es, npy(h[ti]), *[npy(w) for w in gw_host], mask=mask)
    np.testing.assert_array_equal(npy(new_gru[ti]), ref)


def test_neural_operators_past_2G_elements():
    """configs[3]'s per-GPU shape (5 000 x n=400 m=1680: 25.2 M edges, [E, 128] states of 3.2e9 elements = 12.9 GB -- past 32-bit element
    and byte offsets).  Instances are independent, so the whole batch must give, bit for bit, what its two halves (1.6e9 elements each,
    the range the other full-size test pins against the plain kernels and the oracle) give when they are run as batches of their own."""
    from pdp.factorgraph import dataset
    from pdp import native
    dev = torch.device('cuda:0')
    Bq, Nq, Mq, H = 5000, 400, 1680, 128
    items = dataset.random_ksat_items(Bq, Nq, 3, m=Mq, seed=11)
    mk = lambda its: dataset.to_torch(dataset.collate_segment(its), dev)
    prob = lambda tb: native.Problem(tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'])
    g = torch.Generator(device='cpu'); g.manual_seed(9)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.2).to(dev)
    gw = native.GruWeights(r(3 * H, H + 1), r(3 * H, H), r(3 * H), r(3 * H))
    aw = native.AggregatorWeights(r(100, H + 1), r(100), r(50, 100), r(100, 51), r(100), r(H, 100), 1)
    pw = native.AggregatorWeights(r(100, H + 1), r(100), r(50, 100), r(100, 50), r(100), r(H, 100), 0)
    hw = native.HeadWeights(r(50, H), r(50), r(1, 50), 'sigmoid')
    tb = mk(items)
    p = prob(tb)
    E, V = p.E, p.V
    assert E * H > 2 ** 31 and E == Bq * Mq * 3
    gs = torch.Generator(device=dev); gs.manual_seed(4)
    state = torch.randn(E, H, device=dev, generator=gs) * 0.5
    h = torch.randn(E, H, device=dev, generator=gs) * 0.5
    am = (torch.rand(Bq, device=dev, generator=gs) > 0.2).to(torch.uint8)
    whole = dict(gru=p.neural_gru(gw, state, h, am),
                 agg_v=p.neural_aggregate_edges(aw, True, state, None, am, h),
                 agg_f=p.neural_aggregate_edges(aw, False, state, None, am, h),
                 pred=p.neural_predict(pw, hw, state, None))
    torch.cuda.synchronize()
    del p
    half = Bq // 2
    e0 = v0 = 0
    for k in range(2):
        tbh = mk(items[k * half:(k + 1) * half])
        ph = prob(tbh)
        e1, v1 = e0 + ph.E, v0 + ph.V                      # instance-major ids: a half is a contiguous range of edges and variables
        s, hh, a = state[e0:e1].contiguous(), h[e0:e1].contiguous(), am[k * half:(k + 1) * half].contiguous()
        assert torch.equal(whole['gru'][e0:e1], ph.neural_gru(gw, s, hh, a))
        assert torch.equal(whole['agg_v'][e0:e1], ph.neural_aggregate_edges(aw, True, s, None, a, hh))
        assert torch.equal(whole['agg_f'][e0:e1], ph.neural_aggregate_edges(aw, False, s, None, a, hh))
        assert torch.equal(whole['pred'][v0:v1], ph.neural_predict(pw, hw, s, None))
        del ph, s, hh, a
        e0, v0 = e1, v1
    assert e0 == E and v0 == V
