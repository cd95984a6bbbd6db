"""GPU parity tests: every step-wise C-ABI entry point (include/pdp_hip.h) against the CPU oracle on the
same inputs.  The kernels and the oracle share only include/pdp_math.h, so results must be BIT-EXACT
(floats compared with array_equal), and the oracle itself is pinned to the reference by
tests/test_oracle_golden.py."""
import numpy as np
import pytest
import torch

from helpers import random_batch, load_golden

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        x = x.to(dtype)
    return x.to(dev())


def npy(x):
    return x.detach().cpu().numpy()


def make_pair(oracle, batch, replication=1):
    from pdp import native
    hp = native.Problem(t(batch['graph_map']), t(batch['batch_variable_map']), t(batch['batch_function_map']),
                        t(batch['edge_feature']), replication=replication)
    op = oracle.Problem(batch['graph_map'], batch['batch_variable_map'], batch['batch_function_map'], batch['edge_feature'], replication)
    return hp, op


def assert_state_equal(hp, op):
    av, af, sol, sat = op.state()
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], av)
    np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], af)
    np.testing.assert_array_equal(npy(hp.solution), sol)
    np.testing.assert_array_equal(npy(hp.is_sat), sat)


BATCHES = [dict(batch=8, n=20, mixed=True, seed=1), dict(batch=64, n=40, mixed=True, seed=100),
           dict(batch=16, n=50, k=3, seed=7), dict(batch=3, n=300, k=3, seed=9), dict(batch=1, n=30, k=3, seed=3),
           dict(batch=40, n=12, mixed=True, seed=1000)]


def test_device_math_bit_exact(oracle):
    from pdp import native
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.uniform(-110, 35, 300000), rng.uniform(-1, 1, 100000), [-92.1034, 0.0, 30.0, np.nan, np.inf, -np.inf]]).astype(np.float32)
    for fn in ('exp', 'safe_exp', 'safe_exp_fast', 'logsigmoid', 'sigmoid', 'tanh', 'tanh_abs', 'philox'):
        np.testing.assert_array_equal(npy(native.math_apply(fn, t(x))), oracle.math_apply(fn, x), err_msg=fn)
    xl = np.concatenate([np.exp(rng.uniform(-100, 10, 300000)), rng.uniform(0, 2, 100000), [1e-40, 1e-45, 0.0, -1.0, np.nan, np.inf]]).astype(np.float32)
    for fn in ('log', 'safe_log', 'safe_log_fin', 'safe_log_fin_scorer', 'rcp'):
        np.testing.assert_array_equal(npy(native.math_apply(fn, t(xl))), oracle.math_apply(fn, xl), err_msg=fn)
    # the select-free hot-loop forms against the GENERAL functions of the oracle on their domain (finite or NaN)
    xe = np.concatenate([x[np.isfinite(x) & (x <= 30)], np.arange(np.float32(-100).view(np.uint32), np.float32(-110).view(np.uint32), 7, dtype=np.uint32).view(np.float32),
                         [np.nan]]).astype(np.float32)
    np.testing.assert_array_equal(npy(native.math_apply('exp_fin', t(xe))), oracle.math_apply('safe_exp', xe))
    xf = np.concatenate([xl[np.isfinite(xl)], np.arange(0, 0x00800000, 5, dtype=np.uint32).view(np.float32), [np.nan, -3.0, -1e-30]]).astype(np.float32)
    np.testing.assert_array_equal(npy(native.math_apply('safe_log_fin', t(xf))), oracle.math_apply('safe_log', xf))


@pytest.mark.parametrize('spec', BATCHES)
def test_layout_and_simplify(oracle, spec):
    b = random_batch(**spec)
    hp, op = make_pair(oracle, b)
    assert (hp.E, hp.V, hp.F, hp.B) == (op.E, op.V, op.F, op.B)
    hp.simplify(); op.simplify()
    assert_state_equal(hp, op)
    rng = np.random.RandomState(5)
    assign = np.zeros(op.V, np.float32)
    pick = rng.choice(op.V, size=max(1, op.V // 10), replace=False)
    assign[pick] = rng.randint(0, 2, size=len(pick)) * 2 - 1
    ta = t(assign)
    hp.set_variables(ta); a2 = op.set_variables(assign)
    assert_state_equal(hp, op)
    np.testing.assert_array_equal(npy(ta), a2)
    all_active = hp.refresh_edge_mask()
    m, s = op.refresh_edge_mask()
    np.testing.assert_array_equal(npy(hp.edge_mask)[:, 0], m)
    assert all_active == (s == op.E)


@pytest.mark.parametrize('spec', [BATCHES[1], BATCHES[3], dict(batch=50, n=30, k=3, m=60, seed=21)])
def test_simplify_of_a_rebound_problem(oracle, spec):
    """A problem whose state is bound afresh (pdp_problem_bind_state, as bench.py does per step) is simplified again: the LDS-resident kernel
    gathers the slot topology on the first call, leaves it in HBM on the second and reads it back from the third on -- the same state each time,
    also after variables were set in between."""
    from pdp import native
    b = random_batch(**spec)
    hp, op = make_pair(oracle, b)
    hp.simplify(); op.simplify()
    assert_state_equal(hp, op)
    av0, af0, sol0, sat0 = [np.array(x) for x in op.state()]
    rng = np.random.RandomState(3)
    for rep in range(4):
        native.check(native.lib().pdp_problem_bind_state(hp._h, native.ptr(hp.active_variables), native.ptr(hp.active_functions), native.ptr(hp.solution),
                                                         native.ptr(hp.is_sat), native.ptr(hp.edge_mask), native._stream()))
        hp.simplify()
        np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], av0)
        np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], af0)
        np.testing.assert_array_equal(npy(hp.solution), sol0)
        np.testing.assert_array_equal(npy(hp.is_sat), sat0)
    # a state that is not the fresh one: fix some variables on both sides, simplify again (the kept topology serves any state)
    assign = np.zeros(op.V, np.float32)
    pick = rng.choice(op.V, size=max(1, op.V // 8), replace=False)
    assign[pick] = rng.randint(0, 2, size=len(pick)) * 2 - 1
    hp.set_variables(t(assign)); op.set_variables(assign)
    hp.simplify(); op.simplify()
    assert_state_equal(hp, op)


def test_replicated_layout(oracle):
    b = random_batch(batch=5, n=15, mixed=True, seed=11)
    hp, op = make_pair(oracle, b, replication=3)
    gm, bvm, bfm, ef = hp.export_graph()
    ev, ec, es, vi, fi = op.graph()
    np.testing.assert_array_equal(npy(gm), np.stack([ev, ec]))
    np.testing.assert_array_equal(npy(bvm), vi)
    np.testing.assert_array_equal(npy(bfm), fi)
    np.testing.assert_array_equal(npy(ef)[:, 0], es)
    hp.simplify(); op.simplify()
    assert_state_equal(hp, op)


def test_bad_layout_is_rejected():
    from pdp import native
    b = random_batch(batch=4, n=10, seed=1)
    bvm = b['batch_variable_map'].copy(); bvm[0] = 3
    with pytest.raises(native.NativeError):
        native.Problem(t(b['graph_map']), t(bvm), t(b['batch_function_map']), t(b['edge_feature']))


def prepared_pair(oracle, spec, seed=0):
    """Problem pair after simplify + a few fixed variables, with a refreshed edge mask."""
    b = random_batch(**spec)
    hp, op = make_pair(oracle, b)
    hp.simplify(); op.simplify()
    rng = np.random.RandomState(seed)
    assign = np.zeros(op.V, np.float32)
    pick = rng.choice(op.V, size=max(1, op.V // 8), replace=False)
    assign[pick] = rng.randint(0, 2, size=len(pick)) * 2 - 1
    hp.set_variables(t(assign)); op.set_variables(assign)
    hp.refresh_edge_mask(); op.refresh_edge_mask()
    return hp, op, rng


@pytest.mark.parametrize('spec', BATCHES[:4])
def test_reductions(oracle, spec):
    hp, op, rng = prepared_pair(oracle, spec)
    x = rng.rand(op.E).astype(np.float32); x[rng.rand(op.E) < 0.1] = 0
    np.testing.assert_array_equal(npy(hp.smooth_max(t(x)))[:, 0], op.smooth_max(x))
    for xv in (rng.rand(op.V).astype(np.float32) * (rng.rand(op.V) > 0.2), rng.randn(op.V).astype(np.float32),
               0.3 + rng.rand(op.V).astype(np.float32), np.round(rng.randn(op.V) * 3).astype(np.float32)):
        xv = xv.astype(np.float32)
        np.testing.assert_array_equal(npy(hp.instance_max(t(xv))), op.instance_max(xv))
        np.testing.assert_array_equal(npy(hp.instance_argmax(t(xv))), op.instance_argmax(xv))
    xn = rng.rand(op.V).astype(np.float32); xn[op.V // 2] = np.nan
    np.testing.assert_array_equal(npy(hp.instance_max(t(xn))), op.instance_max(xn))
    np.testing.assert_array_equal(npy(hp.instance_argmax(t(xn))), op.instance_argmax(xn))


@pytest.mark.parametrize('spec', BATCHES[:4])
@pytest.mark.parametrize('pi', [0.0, 0.1])
def test_sp_propagate_score_cnf(oracle, spec, pi):
    hp, op, rng = prepared_pair(oracle, spec)
    E, V, B = op.E, op.V, op.B
    q = rng.rand(E, 3).astype(np.float32); q /= q.sum(1, keepdims=True); q[rng.rand(E) < 0.05, 0] = 0
    fs = rng.rand(E, 2).astype(np.float32); fs[rng.rand(E) < 0.05, 0] = 1.0
    fs[:, 1] = rng.randint(-1, 2, size=E) if pi > 0 else 0
    iq = rng.rand(E, 3).astype(np.float32); ifs = rng.rand(E, 2).astype(np.float32)
    am = (rng.rand(B) > 0.3).astype(np.uint8)
    em, _ = op.refresh_edge_mask()
    for use_mask in (True, False):
        hq, hfs = hp.sp_propagate(t(q), t(fs), hp.edge_mask if use_mask else None, t(am) if use_mask else None, t(iq), t(ifs), pi)
        oq, ofs = op.sp_propagate(q, fs, em if use_mask else None, am if use_mask else None, iq, ifs, pi)
        np.testing.assert_array_equal(npy(hq), oq)
        np.testing.assert_array_equal(npy(hfs), ofs)
    np.testing.assert_array_equal(npy(hp.survey_score(t(fs), pi))[:, 0], op.survey_score(fs, pi))
    pred = rng.rand(V).astype(np.float32); pred[rng.rand(V) < 0.3] = 0.5; pred[rng.rand(V) < 0.2] = 1; pred[rng.rand(V) < 0.2] = 0
    hs, hu = hp.cnf_eval(t(pred)); os_, ou = op.cnf_eval(pred)
    np.testing.assert_array_equal(npy(hs)[:, 0], os_); np.testing.assert_array_equal(npy(hu)[:, 0], ou)
    np.testing.assert_array_equal(npy(hp.update_solution(t(pred)))[:, 0], op.update_solution(pred))
    assert_state_equal(hp, op)
    ham = t(np.ones(B, np.uint8)); hp.check_termination(ham, t(pred))
    np.testing.assert_array_equal(npy(ham), op.check_termination(np.ones(B, np.uint8), pred))


def test_sp_sweep_in_three_phases_equals_the_fused_launch(oracle, monkeypatch):
    """A batch that cannot fill the chip with one workgroup per instance (the small dynamic segments of configs[4] under the reference's
    default memory limit) takes the step-wise SP sweep as three launches with several workgroups per instance; same statements per edge and
    row, so the results are those of the one-launch form -- and the oracle's -- bit for bit, in both input forms (surveys / log-domain)."""
    from pdp import native
    hp, op, rng = prepared_pair(oracle, dict(batch=5, n=400, k=3, seed=21))
    E, B = op.E, op.B
    q = rng.rand(E, 3).astype(np.float32); q /= q.sum(1, keepdims=True); q[rng.rand(E) < 0.05, 0] = 0
    fs = rng.rand(E, 2).astype(np.float32); fs[rng.rand(E) < 0.05, 0] = 1.0
    fs[:, 1] = rng.randint(-1, 2, size=E)
    iq = rng.rand(E, 3).astype(np.float32); ifs = rng.rand(E, 2).astype(np.float32)
    am = (rng.rand(B) > 0.3).astype(np.uint8)
    xlog = (-rng.rand(E) * 5).astype(np.float32)
    em, _ = op.refresh_edge_mask()
    got = {}
    for fused in (False, True):
        if fused:
            monkeypatch.setenv('PDP_SP_SWEEP_FUSED', '1')
        a = hp.sp_propagate(t(q), t(fs), hp.edge_mask, t(am), t(iq), t(ifs), 0.1)
        name = native.kernel_name('sp_sweep')
        b = hp.sp_propagate_adapted(t(xlog), t(fs), hp.edge_mask, t(am), t(iq), t(ifs), 0.1)
        assert ('three phases' in name) == (not fused) and ('three phases' in native.kernel_name('sp_sweep')) == (not fused)
        got[fused] = [npy(x) for x in a + b]
    for x, y in zip(got[False], got[True]):
        np.testing.assert_array_equal(x, y)
    oq, ofs = op.sp_propagate(q, fs, em, am, iq, ifs, 0.1)
    np.testing.assert_array_equal(got[False][0], oq); np.testing.assert_array_equal(got[False][1], ofs)


@pytest.mark.parametrize('spec', BATCHES[:4])
def test_energy_and_walksat_pieces(oracle, spec):
    hp, op, rng = prepared_pair(oracle, spec)
    av = op.state()[0]
    a = ((rng.randint(0, 2, size=op.V) * 2 - 1) * av).astype(np.float32)
    he, hu = hp.energy(t(a)); oe, ou = op.energy(a)
    np.testing.assert_array_equal(npy(he)[:, 0], oe); np.testing.assert_array_equal(npy(hu)[:, 0], ou)
    np.testing.assert_array_equal(npy(hp.energy_diff(t(a)))[:, 0], op.energy_diff(a))


@pytest.mark.parametrize('spec', BATCHES[:5])
@pytest.mark.parametrize('mode', ['stream', 'philox'])
def test_random_fill_and_local_search(oracle, spec, mode):
    hp, op, rng = prepared_pair(oracle, spec)
    w = 25
    n_active = int((op.state()[0] > 0).sum())
    if mode == 'stream':
        stream = rng.rand(n_active + w * (op.V + op.B)).astype(np.float32)
        hp.random_fill(values=t(stream[:n_active]))
        cur = op.random_fill(stream=stream)
        assert cur == n_active
        rest = stream[n_active:].reshape(w, op.V + op.B)
        var_rand = np.ascontiguousarray(rest[:, :op.V]); coin = np.ascontiguousarray(rest[:, op.V:])
        assert_state_equal(hp, op)
        pred = op.state()[2]
        hout, hsteps = hp.local_search(t(pred), w, 0.5, t(var_rand), t(coin))
        oout, osteps, cur2 = op.local_search(pred, w, 0.5, stream=stream, cursor=cur)
        assert cur2 == n_active + osteps * (op.V + op.B)
    else:
        hp.random_fill(seed=1234); op.random_fill(seed=1234)
        assert_state_equal(hp, op)
        pred = op.state()[2]
        hout, hsteps = hp.local_search(t(pred), w, 0.5, seed=99)
        oout, osteps, _ = op.local_search(pred, w, 0.5, seed=99)
    assert hsteps == osteps
    np.testing.assert_array_equal(npy(hout)[:, 0], oout)


@pytest.mark.parametrize('mode', ['stream', 'philox'])
@pytest.mark.parametrize('n_big,with_small,form', [(1, True, 'team'), (2, True, 'team'), (2, False, 'team'), (2, False, 'wide'), (1, True, 'one-workgroup')])
def test_local_search_routes_big_instances(oracle, monkeypatch, mode, n_big, with_small, form):
    """Persistent Walk-SAT with per-instance routing: instances past the LDS limit (n = 3000 / 3500: ~40 000 edges) run the HBM-resident
    form of the same kernel next to the LDS-resident launch of the small ones (or alone) -- as a team of workgroups on one XCD, as a
    chip-wide team (threshold lowered for the test) or on one workgroup each -- same assignments and step count as the oracle."""
    from pdp.factorgraph import dataset
    if form == 'wide':
        monkeypatch.setenv('PDP_SOLVE_TEAM_WIDE_EDGES', '10000')
    if form == 'one-workgroup':
        monkeypatch.setenv('PDP_WALKSAT_NO_TEAM', '1')
    items = []
    if with_small:
        items += dataset.random_ksat_items(24, 50, 3, m=200, seed=40)
    items += [dataset.random_ksat_items(1, 3000 + 500 * i, 3, m=int(3.8 * (3000 + 500 * i)), seed=60 + i)[0] for i in range(n_big)]
    if with_small:
        items += dataset.random_ksat_items(8, 40, 3, m=150, seed=41)
    b = dataset.collate_segment(items)
    hp, op = make_pair(oracle, b)
    hp.simplify(); op.simplify()
    w = 60
    rng = np.random.RandomState(5)
    n_active = int((op.state()[0] > 0).sum())
    if mode == 'stream':
        stream = rng.rand(n_active + w * (op.V + op.B)).astype(np.float32)
        hp.random_fill(values=t(stream[:n_active]))
        cur = op.random_fill(stream=stream)
        rest = stream[n_active:].reshape(w, op.V + op.B)
        var_rand = np.ascontiguousarray(rest[:, :op.V]); coin = np.ascontiguousarray(rest[:, op.V:])
        pred = op.state()[2]
        hout, hsteps = hp.local_search(t(pred), w, 0.5, t(var_rand), t(coin))
        oout, osteps, _ = op.local_search(pred, w, 0.5, stream=stream, cursor=cur)
    else:
        hp.random_fill(seed=1234); op.random_fill(seed=1234)
        pred = op.state()[2]
        hout, hsteps = hp.local_search(t(pred), w, 0.5, seed=99)
        oout, osteps, _ = op.local_search(pred, w, 0.5, seed=99)
    assert hsteps == osteps
    assert (oout != pred).any()
    np.testing.assert_array_equal(npy(hout)[:, 0], oout)


@pytest.mark.parametrize('form', ['persistent', 'persistent-big-team', 'persistent-big-one-workgroup', 'strict'])
def test_philox_counters_start_at_the_parts_base(oracle, monkeypatch, form):
    """pdp_problem_set_rng_base: the batch is a contiguous part of a larger forward, and the in-kernel Philox draws of the random fill and of
    Walk-SAT count variables / instances from the part's place in it.  Every form of the search takes the base -- the LDS-resident persistent
    kernel, its HBM-resident form for instances past the LDS limit (a team and one workgroup), the strict step-wise loop -- and equals the
    oracle with the same base; without the base the draws are others."""
    from pdp.factorgraph import dataset
    if form == 'strict':
        monkeypatch.setenv('PDP_WALKSAT_STRICT', '1')
    if form == 'persistent-big-one-workgroup':
        monkeypatch.setenv('PDP_WALKSAT_NO_TEAM', '1')
    items = dataset.random_ksat_items(12, 60, 3, m=250, seed=70)
    if form.startswith('persistent-big'):
        items = items[:6] + [dataset.random_ksat_items(1, 3000, 3, m=11400, seed=71)[0]] + items[6:]
    b = dataset.collate_segment(items)
    w = 50
    outs = []
    for base in ((123457, 4321), (0, 0)):
        hp, op = make_pair(oracle, b)
        hp.simplify(); op.simplify()
        hp.set_rng_base(*base); op.set_rng_base(*base)
        hp.random_fill(seed=777); op.random_fill(seed=777)
        assert_state_equal(hp, op)
        pred = op.state()[2]
        hout, hsteps = hp.local_search(t(pred), w, 0.5, seed=555)
        oout, osteps, _ = op.local_search(pred, w, 0.5, seed=555)
        assert hsteps == osteps and (oout != pred).any()
        np.testing.assert_array_equal(npy(hout)[:, 0], oout)
        outs.append((pred.copy(), oout.copy()))
    assert not np.array_equal(outs[0][0], outs[1][0]) and not np.array_equal(outs[0][1], outs[1][1])


def _planted_instance(n, m, k, seed, name):
    "random k-SAT whose clauses all hold under the all-TRUE assignment (one literal of every clause is made positive)"
    from pdp.factorgraph import dataset
    rng = np.random.RandomState(seed)
    clauses = []
    for _ in range(m):
        vs = rng.choice(n, size=k, replace=False) + 1
        sg = rng.randint(0, 2, size=k) * 2 - 1
        sg[rng.randint(k)] = 1
        clauses.append([int(a * b) for a, b in zip(vs, sg)])
    return dataset.instance_from_clauses(n, clauses, label=1, name=name)


@pytest.mark.parametrize('form', ['one-workgroup', 'team'])
@pytest.mark.parametrize('R,w', [(2, 400), (4, 300)])
def test_local_search_replicated_batch_with_big_instances(oracle, monkeypatch, form, R, w):
    """configs[4]'s situation: batch replication AND instances past the LDS limit of the Walk-SAT kernel.  The search of an original
    instance ends for all its replicas at the step its first replica is satisfied, and the whole call at the step the last original is
    (solver.py:446-449): replicas that were still searching at that global stop are run again with it as their cap -- the LDS-resident ones
    from a list, the big ones (HBM-resident form / teams) all together.  Planted instances started a few flips away from their planted
    assignment, so that the stop comes before the step limit.  Same assignments and step count as the oracle; before, such a batch took the
    strict 3-launch loop."""
    from pdp import native
    from pdp.factorgraph import dataset
    if form == 'one-workgroup':
        monkeypatch.setenv('PDP_WALKSAT_NO_TEAM', '1')
    items = [_planted_instance(40, 120, 3, 140 + i, 's%d' % i) for i in range(10)]
    items += [_planted_instance(2800 + 300 * i, int(3.0 * (2800 + 300 * i)), 3, 160 + i, 'big%d' % i) for i in range(2)]
    items += [_planted_instance(30, 90, 3, 180 + i, 't%d' % i) for i in range(5)]
    b = dataset.collate_segment(items)
    hp, op = make_pair(oracle, b, replication=R)
    hp.simplify(); op.simplify()
    rng = np.random.RandomState(9)
    pred = np.ones(op.V, np.float32)
    pred[rng.choice(op.V, size=op.V // 150, replace=False)] = 0.0          # every replica starts from its own few wrong variables
    native.kernel_timing(True)
    hout, hsteps = hp.local_search(t(pred), w, 0.5, seed=31)
    launches = native.kernel_timing_read()['walksat'][1]; native.kernel_timing(False)
    oout, osteps, _ = op.local_search(pred, w, 0.5, seed=31)
    assert hsteps == osteps and 0 < osteps < w, (hsteps, osteps)          # the global stop came first: the truncation was exercised
    assert launches >= 1                                                  # the persistent kernels ran (the strict loop has none of them)
    np.testing.assert_array_equal(npy(hout)[:, 0], oout)


@pytest.mark.parametrize('spec', BATCHES[:4])
def test_sequential_decimator_steps(oracle, spec):
    """Drive propagate + decimate for a number of iterations through the step-wise entry points and compare
    the complete state with the oracle after every step (tolerance chosen so decimation fires)."""
    from pdp import native
    b = random_batch(**spec)
    hp, op = make_pair(oracle, b)
    hp.simplify(); op.simplify()
    E, B = op.E, op.B
    q = np.full((E, 3), 1.0 / 3.0, np.float32); fs = np.zeros((E, 2), np.float32); fs[:, 0] = 0.5
    hq, hfs = t(q), t(fs)
    ham = t(np.ones(B, np.uint8)); oam = np.ones(B, np.uint8)
    hd = native.Decimator(hp); od = op.new_decimator()
    use_mask = False
    for it in range(25):
        hq, hfs = hp.sp_propagate(hq, hfs, hp.edge_mask if use_mask else None, ham, hq, hfs, 0.0)
        em = op.refresh_edge_mask()[0] if use_mask else None
        q, fs = op.sp_propagate(q, fs, em, oam, q, fs, 0.0)
        np.testing.assert_array_equal(npy(hq), q, err_msg='q it %d' % it)
        hp.sequential_decimate(hd, hfs, ham, 0.05, 6, 0.0)
        oam, _ = op.sequential_decimate(od, fs, oam, 0.05, 6, 0.0)
        np.testing.assert_array_equal(npy(ham), oam, err_msg='active mask it %d' % it)
        assert_state_equal(hp, op)
        all_active = hp.refresh_edge_mask()
        m, s = op.refresh_edge_mask()
        assert all_active == (s == E)
        if not all_active:
            use_mask = True
        pred = op.state()[2]
        hpred = hp.update_solution(hp.solution.clone())
        opred = op.update_solution(pred)
        hp.check_termination(ham, hpred.view(-1)); oam = op.check_termination(oam, opred)
        np.testing.assert_array_equal(npy(ham), oam)
    prev, cnt, fl = op.decimator_get(od)
    op.free_decimator(od)


def test_deduplicate(oracle):
    b = random_batch(batch=6, n=20, mixed=True, seed=21)
    hp, op = make_pair(oracle, b, replication=4)
    hp.simplify(); op.simplify()
    rng = np.random.RandomState(1)
    pred = (rng.rand(op.V) > 0.5).astype(np.float32)
    hout, hch = hp.deduplicate(t(pred)); oout, och = op.deduplicate(pred)
    np.testing.assert_array_equal(npy(hout)[:, 0], oout)
    np.testing.assert_array_equal(npy(hch), och)


@pytest.mark.parametrize('sharp', [1, 3, 5])
def test_sat_loss_bit_exact(oracle, sharp):
    "energy loss of a prediction (SatLossEvaluator.forward, util.py:178-197): same nesting of sums on both sides, so the float is identical"
    b = random_batch(batch=11, n=30, mixed=True, seed=123)
    hp, op = make_pair(oracle, b)
    rng = np.random.RandomState(sharp)
    for pred, coeff in ((rng.rand(op.V).astype(np.float32), 1.0), ((rng.rand(op.V) > 0.5).astype(np.float32), 3.5),
                        ((0.5 + 0.2 * rng.randn(op.V)).astype(np.float32), 10.0)):
        ref = np.float32(op.sat_loss(pred, np.float32(coeff), 1e-8, sharp))
        got = npy(hp.sat_loss(t(pred), float(np.float32(coeff)), 1e-8, sharp))[0]
        assert (np.isinf(ref) and np.isinf(got)) or ref == got, (ref, got)


def test_reciprocal_exhaustive():
    """pdp_rcp_ge1 (v_rcp_f32 + Newton step + residual correction, no scaling / fix-up instructions) against the IEEE division for EVERY
    float in [1, 2^126]: the range of the denominators 1 + e^-x and e^2|x| + 1 of the GRU gates.  Compared on the device with the
    compiler's own correctly rounded `1.0f / x` (math probe 'rcp', itself checked against the CPU in test_math_functions_bit_exact)."""
    from pdp import native
    lo, hi = 0x3f800000, 0x7e800000                      # bit patterns of 1.0f and 2^126
    step = 1 << 26
    bad = 0
    for start in range(lo, hi + 1, step):
        n = min(step, hi + 1 - start)
        bits = torch.arange(start, start + n, dtype=torch.int64, device='cuda:0').to(torch.int32)
        x = bits.view(torch.float32)
        a = native.math_apply('rcp_ge1', x)
        b = native.math_apply('rcp', x)
        bad += int((a.view(torch.int32) != b.view(torch.int32)).sum().item())
    assert bad == 0


def test_survey_scorer_with_adaptors_equals_oracle_and_reference(oracle):
    """The plug-in ``SurveyScorer(include_adaptors=True)`` (pdp_predict.py:145-152, 161-208) through its module interface: projector ->
    sigmoid / sign (k_sp_adaptors) -> k_survey_score.  Bit for bit the oracle's value; the reference's within the fp tolerance of the
    classical operators (its projector is an sgemm, ours a k-ascending fmaf chain)."""
    from pdp.nn import pdp_predict
    from pdp.nn.solver import SATProblem
    d = load_golden('scorer_adaptors')
    gm, bvm, bfm, ef = [t(d[k]) for k in ('graph_map', 'batch_variable_map', 'batch_function_map', 'edge_feature')]
    sp = SATProblem((gm, bvm, bfm, ef, None, None), dev(), 1)
    sp.simplify(); sp.set_variables(t(d['assign']).reshape(-1, 1))
    np.testing.assert_array_equal(npy(sp._active_functions).reshape(-1), d['active_functions'])
    op = oracle.Problem(d['graph_map'], d['batch_variable_map'], d['batch_function_map'], d['edge_feature'], 1)
    op.simplify(); op.set_variables(d['assign'])
    W = d['weight']
    _, fs2 = oracle.sp_adaptors(d['message'], d['message'], W[0], W)
    for tag, pi in (('pi0', 0.0), ('pi01', 0.1)):
        sc = pdp_predict.SurveyScorer(dev(), message_dimension=W.shape[1], include_adaptors=True, pi=pi)
        sc.load_state_dict({'_projector.weight': torch.from_numpy(W), '_module_list.0.weight': torch.from_numpy(W)}, strict=True)
        got, _ = sc((None, t(d['message'])), sp)
        assert tuple(got.shape) == (sp._variable_num, 1)
        np.testing.assert_array_equal(npy(got)[:, 0], op.survey_score(fs2, pi))
        np.testing.assert_allclose(npy(got)[:, 0], d['score_' + tag], rtol=2e-5, atol=2e-6)
