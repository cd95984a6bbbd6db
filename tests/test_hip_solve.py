"""GPU parity tests of the persistent solver (pdp_sp_solve): one launch for the whole PDP loop must leave
exactly the state the oracle's iteration loop leaves (bit-exact floats, identical integer trajectories)."""
import numpy as np
import pytest
import torch

from helpers import random_batch, load_golden
from test_hip_ops import t, npy, make_pair

pytestmark = pytest.mark.gpu


def run_pair(oracle, batch, T, tol, t_max):
    from pdp import native
    hp, op = make_pair(oracle, batch)
    res = op.forward('p-d-p', T, local_search_iterations=0, tolerance=tol, t_max=t_max, seed=5, trace=True)
    hp.simplify()
    E, B = hp.E, hp.B
    q = torch.full((E, 3), 1.0, device='cuda:0') / 3.0
    fs = torch.zeros(E, 2, device='cuda:0'); fs[:, 0] = 0.5
    am = torch.ones(B, dtype=torch.uint8, device='cuda:0')
    dec = native.Decimator(hp)
    try:
        iters, used_lds = hp.sp_solve(q, fs, am, dec, T, tol, t_max)
        spec_ok = True
    except native.SpeculationFailed:
        spec_ok = False
        iters, used_lds = -1, None
    return hp, res, q, fs, am, iters, used_lds, spec_ok


SPECS = [(dict(batch=16, n=50, k=3, seed=7), 40, 0.02, 100),
         (dict(batch=64, n=40, mixed=True, seed=100), 30, 0.05, 6),
         (dict(batch=40, n=40, k=3, m=140, seed=900), 60, 0.05, 10),
         (dict(batch=200, n=30, k=3, m=100, seed=300), 50, 0.05, 8),
         (dict(batch=6, n=200, k=3, seed=11), 30, 0.02, 100)]


@pytest.mark.parametrize('spec,T,tol,t_max', SPECS)
def test_persistent_solve_matches_oracle_loop(oracle, spec, T, tol, t_max):
    b = random_batch(**spec)
    hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, T, tol, t_max)
    assert spec_ok, "speculation unexpectedly failed on a benign batch"
    assert used_lds
    it = res['iterations_run']
    assert iters == it
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], res['trace_active_fn'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])


@pytest.mark.parametrize('odd', [False, True])
@pytest.mark.parametrize('spec,T,tol,t_max', [SPECS[0], SPECS[3]])
def test_persistent_solve_with_an_external_force_column(oracle, spec, T, tol, t_max, odd):
    """The SP triple with pi > 0 and a caller-supplied external force in fs[:, 1] (pdp_propagate.py:197,201; SurveyScorer pi terms): the call
    first enqueues the force-free instantiation, k_solve_import meets the force, every launch returns untouched and the call runs again on
    k_sp_solve_lds<true, false, false, *> with the force column -- against the oracle's loop built from its step-wise operators (the
    statements of solver.py:355-386), bit for bit."""
    from pdp import native
    pi = 0.1
    b = random_batch(**spec)
    hp, op = make_pair(oracle, b)
    hp.simplify(); op.simplify()
    E, B = op.E, op.B
    rng = np.random.RandomState(3)
    q = np.full((E, 3), 1.0 / 3.0, np.float32)
    fs = np.zeros((E, 2), np.float32); fs[:, 0] = 0.5
    # (the LDS-resident kernel keeps the force as a 2-bit code in the slot word -- 0, +1, -1, NaN, the values a force ever has; `odd`: other
    #  values make k_force_import fail the speculation, and the call is served by the lock-step launch or the caller's step-wise loop)
    fs[:, 1] = rng.choice([-1.0, 0.0, 1.0, 0.5, -2.0] if odd else [-1.0, 0.0, 1.0], size=E).astype(np.float32)
    hq, hfs = t(q), t(fs)
    ham = torch.ones(B, dtype=torch.uint8, device='cuda:0')
    try:
        iters, used_lds = hp.sp_solve(hq, hfs, ham, native.Decimator(hp), T, tol, t_max, pi=pi)
    except native.SpeculationFailed:
        assert odd
        np.testing.assert_array_equal(npy(hq), q); np.testing.assert_array_equal(npy(hfs), fs)      # nothing was touched
        return
    assert odd or (used_lds and native.kernel_name('sp_solve').startswith('k_sp_solve_lds<true, false, false'))
    # the oracle's loop
    oam = np.ones(B, np.uint8)
    od = op.new_decimator()
    use_mask, it = False, 0
    for _ in range(T):
        em = op.refresh_edge_mask()[0] if use_mask else None
        q, fs = op.sp_propagate(q, fs, em, oam, q, fs, pi)
        oam, _n = op.sequential_decimate(od, fs, oam, tol, t_max, pi)
        _, s_ = op.refresh_edge_mask()
        if s_ < E:
            use_mask = True
        pred = op.update_solution(op.state()[2])
        oam = op.check_termination(oam, pred)
        it += 1
        if int(oam.sum()) <= 0:
            break
    op.free_decimator(od)
    assert iters == it
    np.testing.assert_array_equal(npy(ham), oam)
    assert_state = op.state()
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], assert_state[0])
    np.testing.assert_array_equal(npy(hp.solution), assert_state[2])
    np.testing.assert_array_equal(npy(hq), q)
    np.testing.assert_array_equal(npy(hfs), fs)
    assert (assert_state[0] == 0).sum() > 0                                            # variables were fixed along the way


def test_nan_force_on_instances_that_leave_early(oracle):
    """Under-constrained instances go inactive after a few sweeps (every survey vanishes) with sweeps still to come; a NaN in their external
    force column then reaches the frozen state through the reference's masked update (0 * NaN), which the persistent loop does not model:
    its ghost sweep (lds_ghost_bad, on the LDS image as the instance leaves) must notice -- the call fails with every array untouched -- or
    the call's result is the oracle's loop, bit for bit.  Never a silent difference."""
    from pdp import native
    pi, T, tol, t_max = 0.1, 30, 0.05, 8
    b = random_batch(batch=60, n=30, k=3, m=18, seed=4242)
    hp, op = make_pair(oracle, b)
    hp.simplify(); op.simplify()
    E, B = op.E, op.B
    rng = np.random.RandomState(9)
    q = np.full((E, 3), 1.0 / 3.0, np.float32)
    fs = np.zeros((E, 2), np.float32); fs[:, 0] = 0.5
    fs[:, 1] = rng.choice([-1.0, 0.0, 1.0], size=E).astype(np.float32)
    fs[rng.choice(E, size=max(1, E // 40), replace=False), 1] = np.nan
    q0, fs0 = q.copy(), fs.copy()
    hq, hfs = t(q), t(fs)
    ham = torch.ones(B, dtype=torch.uint8, device='cuda:0')
    try:
        iters, used_lds = hp.sp_solve(hq, hfs, ham, native.Decimator(hp), T, tol, t_max, pi=pi)
    except native.SpeculationFailed:
        np.testing.assert_array_equal(npy(hq), q0); np.testing.assert_array_equal(npy(hfs), fs0)      # nothing was touched
        return
    oam = np.ones(B, np.uint8)
    od = op.new_decimator()
    use_mask, it = False, 0
    for _ in range(T):
        em = op.refresh_edge_mask()[0] if use_mask else None
        q, fs = op.sp_propagate(q, fs, em, oam, q, fs, pi)
        oam, _n = op.sequential_decimate(od, fs, oam, tol, t_max, pi)
        _, s_ = op.refresh_edge_mask()
        if s_ < E:
            use_mask = True
        pred = op.update_solution(op.state()[2])
        oam = op.check_termination(oam, pred)
        it += 1
        if int(oam.sum()) <= 0:
            break
    op.free_decimator(od)
    assert iters == it
    np.testing.assert_array_equal(npy(ham), oam)
    np.testing.assert_array_equal(npy(hp.solution), op.state()[2])
    np.testing.assert_array_equal(npy(hq), q)
    np.testing.assert_array_equal(npy(hfs), fs)


def test_bad_ghost_sweep_fails_the_call(oracle, monkeypatch):
    """The verdict of the in-kernel ghost sweep travels: an instance that leaves inactive with sweeps to come and whose frozen state is not
    finite under the reference's masked sweep (here: a NaN q_u planted by PDP_DEBUG_GHOST_INJECT in the slot the instance writes back) marks
    its flag, the last chunk's k_solve_finish collects the flags, the call fails and every array is what it was at call entry; without the
    plant the same batch runs through and equals the oracle."""
    from pdp import native
    b = random_batch(batch=60, n=30, k=3, m=18, seed=4242)          # under-constrained: instances leave after a few sweeps, at different sweeps
    T, tol, t_max = 30, 0.05, 8
    hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, T, tol, t_max)
    assert spec_ok and used_lds and (npy(am) == 0).any()             # (instances left along the way)
    np.testing.assert_array_equal(npy(q), res['q'])
    monkeypatch.setenv('PDP_DEBUG_GHOST_INJECT', '1')
    monkeypatch.setenv('PDP_SOLVE_NO_LOCKSTEP', '1')                 # (a small batch whose speculation fails is otherwise served by the lock-step launch)
    hp2, _ = make_pair(oracle, b)
    hp2.simplify()
    av0, sol0 = npy(hp2.active_variables).copy(), npy(hp2.solution).copy()
    q2 = torch.full((hp2.E, 3), 1.0, device='cuda:0') / 3.0
    fs2 = torch.zeros(hp2.E, 2, device='cuda:0'); fs2[:, 0] = 0.5
    am2 = torch.ones(hp2.B, dtype=torch.uint8, device='cuda:0')
    q0, fs0 = npy(q2).copy(), npy(fs2).copy()
    with pytest.raises(native.SpeculationFailed):
        hp2.sp_solve(q2, fs2, am2, native.Decimator(hp2), T, tol, t_max)
    np.testing.assert_array_equal(npy(q2), q0); np.testing.assert_array_equal(npy(fs2), fs0)
    np.testing.assert_array_equal(npy(am2), np.ones(hp2.B, np.uint8))
    np.testing.assert_array_equal(npy(hp2.active_variables), av0); np.testing.assert_array_equal(npy(hp2.solution), sol0)


def test_persistent_solve_golden_trace():
    """Against the reference itself (golden trace): identical integer trajectory end state."""
    from pdp import native
    d = load_golden('trace_pdp_easy_ws')
    T = int(d['meta'][0])
    hp = native.Problem(t(d['graph_map']), t(d['batch_variable_map']), t(d['batch_function_map']), t(d['edge_feature']))
    hp.simplify()
    q = torch.full((hp.E, 3), 1.0, device='cuda:0') / 3.0
    fs = torch.zeros(hp.E, 2, device='cuda:0'); fs[:, 0] = 0.5
    am = torch.ones(hp.B, dtype=torch.uint8, device='cuda:0')
    dec = native.Decimator(hp)
    iters, _ = hp.sp_solve(q, fs, am, dec, T, 0.05, 10)
    it = int(d['iterations_run'][0])
    assert iters == it
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], d['trace_active_variables'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], d['trace_active_functions'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), d['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(am), d['trace_active_mask'][it - 1])
    np.testing.assert_allclose(npy(q), d['final_prop_0'], rtol=5e-4, atol=5e-6)


def test_speculation_detects_coupling(oracle, monkeypatch):
    """Two instances without any inactive variable: the batch-global min is > 0 and no instance supplies an exact zero, so the
    persistent solver must refuse (PDP_ERR_SPECULATION) instead of returning a result that differs from the reference.  (A batch of
    ONE such instance is solved exactly instead: test_single_instance_batches_run_exact.)"""
    from pdp import native
    from pdp.factorgraph import dataset
    n = 12
    items = []
    for shift in (4, 5):
        clauses = []
        for v in range(1, n + 1):
            a, b2 = (v % n) + 1, ((v + shift) % n) + 1
            clauses.append([v, -a, b2]); clauses.append([-v, a, -b2]); clauses.append([v, a, -b2]); clauses.append([-v, -a, b2])
        items.append(dataset.instance_from_clauses(n, clauses))
    b = dataset.collate_segment(items)
    hp, op = make_pair(oracle, b)
    hp.simplify(); op.simplify()
    assert op.state()[0].min() == 1.0     # nothing got de-activated: no exact zero in the batch
    q = torch.full((hp.E, 3), 1.0, device='cuda:0') / 3.0
    fs = torch.zeros(hp.E, 2, device='cuda:0'); fs[:, 0] = 0.5
    am = torch.ones(hp.B, dtype=torch.uint8, device='cuda:0')
    monkeypatch.setenv('PDP_SOLVE_NO_LOCKSTEP', '1')
    with pytest.raises(native.SpeculationFailed):
        hp.sp_solve(q, fs, am, native.Decimator(hp), 5, 0.02, 100)
    # a caller that declares q / fs disposable (the solver class: its init_state stays intact) gets everything ELSE back -- the problem's
    # flags and solution, the instance mask -- and the messages are its own business
    state0 = [npy(x).copy() for x in (hp.active_variables, hp.active_functions, hp.solution, am)]
    q2, fs2 = q.clone(), fs.clone()
    with pytest.raises(native.SpeculationFailed):
        hp.sp_solve(q2, fs2, am, native.Decimator(hp), 5, 0.02, 100, inputs_disposable=True)
    for a0, x in zip(state0, (hp.active_variables, hp.active_functions, hp.solution, am)):
        np.testing.assert_array_equal(npy(x), a0)
    # every array is back at its call-entry state; with the lock-step launch (the default for small batches) the same call succeeds
    # and equals the oracle's strict semantics
    monkeypatch.delenv('PDP_SOLVE_NO_LOCKSTEP')
    res = op.forward('p-d-p', 25, local_search_iterations=0, tolerance=0.02, t_max=100, seed=5, trace=True)
    iters, used_lds = hp.sp_solve(q, fs, am, native.Decimator(hp), 25, 0.02, 100)
    it = res['iterations_run']
    assert iters == it and not used_lds
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])


@pytest.mark.parametrize('spec,T,tol,t_max', [(dict(batch=400, n=60, k=3, seed=7000), 120, 0.05, 8),
                                              (dict(batch=200, n=100, k=3, seed=5000), 150, 0.05, 10)])
def test_persistent_solve_reproduces_nan_poisoning(oracle, spec, T, tol, t_max):
    """At the SAT threshold some instance produces 0/0 in the survey normalisation; in the reference the NaN then
    disables gating / convergence / decimation for the WHOLE batch.  The persistent solver must
    reproduce that exactly (pass 1 finds the poison iteration, the affected instances are replayed)."""
    b = random_batch(**spec)
    hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, T, tol, t_max)
    assert np.isnan(res['fs']).any(), "test batch no longer produces a NaN: pick another seed"
    assert spec_ok and used_lds
    it = res['iterations_run']
    assert iters == it
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], res['trace_active_fn'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])


@pytest.mark.parametrize('first,chunk', [('1', '7'), ('13', '25'), ('25', '25'), ('90', '10'), ('400', '25'), (None, '64')])
def test_chunk_schedule_does_not_change_the_result(oracle, monkeypatch, first, chunk):
    """The loop runs as launches of PDP_SOLVE_FIRST_CHUNK (default 2 C), C, C, ... sweeps: where the chunk boundaries fall -- before, on or
    behind the batch's first NaN, one sweep per launch, the whole loop in one launch, a first chunk past the LDS-local speculation record --
    decides what is replayed and in which order instances are dispatched, never the result (the NaN-poisoned batch of
    test_persistent_solve_reproduces_nan_poisoning against the oracle's loop, bit for bit)."""
    monkeypatch.setenv('PDP_SOLVE_CHUNK', chunk)
    if first: monkeypatch.setenv('PDP_SOLVE_FIRST_CHUNK', first)
    else: monkeypatch.delenv('PDP_SOLVE_FIRST_CHUNK', raising=False)
    b = random_batch(batch=400, n=60, k=3, seed=7000)
    hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, 120, 0.05, 8)
    assert np.isnan(res['fs']).any() and spec_ok and used_lds
    it = res['iterations_run']
    assert iters == it
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], res['trace_active_fn'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])


@pytest.mark.parametrize('chunk', [None, '5'])
def test_persistent_solve_resumes_across_calls(oracle, monkeypatch, chunk):
    """Two consecutive calls (the second one enters with decimator state: previous surveys, counters, edge mask) end exactly
    where one call of the total length ends -- also with a chunk length that does not divide either call."""
    from pdp import native
    if chunk:
        monkeypatch.setenv('PDP_SOLVE_CHUNK', chunk)
    b = random_batch(batch=40, n=40, k=3, m=140, seed=900)
    T1, T2, tol, t_max = 23, 37, 0.05, 10
    hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, T1 + T2, tol, t_max)
    assert spec_ok and used_lds
    hp2, op2 = make_pair(oracle, b)
    hp2.simplify()
    q2 = torch.full((hp2.E, 3), 1.0, device='cuda:0') / 3.0
    fs2 = torch.zeros(hp2.E, 2, device='cuda:0'); fs2[:, 0] = 0.5
    am2 = torch.ones(hp2.B, dtype=torch.uint8, device='cuda:0')
    dec2 = native.Decimator(hp2)
    it1, _ = hp2.sp_solve(q2, fs2, am2, dec2, T1, tol, t_max)
    it2, _ = hp2.sp_solve(q2, fs2, am2, dec2, T2, tol, t_max)
    assert it1 == T1 and it1 + it2 == iters == res['iterations_run']
    for a, c in ((q, q2), (fs, fs2), (am, am2), (hp.active_variables, hp2.active_variables), (hp.active_functions, hp2.active_functions),
                 (hp.solution, hp2.solution)):
        np.testing.assert_array_equal(npy(a), npy(c))
    np.testing.assert_array_equal(npy(q2), res['q'])


@pytest.mark.parametrize('spec,T,tol,t_max', [SPECS[1], SPECS[3], (dict(batch=400, n=60, k=3, seed=7000), 120, 0.05, 8)])
def test_hbm_resident_kernel_matches_oracle_loop(oracle, monkeypatch, spec, T, tol, t_max):
    """Instances that do not fit the LDS run the same algorithm on HBM-resident arrays (host-driven chunk loop with snapshots);
    forced here for small instances, including a batch that is NaN-poisoned."""
    monkeypatch.setenv('PDP_SOLVE_FORCE_HBM', '1')
    b = random_batch(**spec)
    hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, T, tol, t_max)
    if not spec_ok:
        pytest.skip("speculation failed on this batch (the caller would rerun step-wise)")
    assert not used_lds
    it = res['iterations_run']
    assert iters == it
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])


def _tiny_batch(clause_lists):
    from pdp.factorgraph import dataset
    items = [dataset.instance_from_clauses(max(abs(l) for c in cl for l in c), cl, label=-1, name="t%d" % i) for i, cl in enumerate(clause_lists)]
    return dataset.collate_segment(items)


@pytest.mark.parametrize('T', [1, 7])
def test_persistent_solve_degenerate_instances(oracle, T):
    """Edge cases next to ordinary instances in one batch: an instance that simplify() solves completely (unit propagation), a
    contradictory one (x and not x: conflict, solver.py:248-262), a single clause, one long clause, one variable in 30 clauses."""
    from pdp import generator
    rng = np.random.RandomState(4)
    ordinary = generator.uniform_ksat(30, 100, 3, rng)
    b = _tiny_batch([[[1], [-1, 2], [-2, 3]],
                     [[1], [-1]],
                     [[1, -2, 3]],
                     [[i + 1 for i in range(40)]],
                     [[1, (i % 6) + 2, -((i % 5) + 8)] for i in range(30)],
                     ordinary])
    hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, T, 0.05, 10)
    assert spec_ok and used_lds
    it = res['iterations_run']
    assert iters == it
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], res['trace_active_fn'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])


def test_persistent_solve_with_nothing_to_do(oracle):
    "every instance enters inactive: no iteration runs and no array changes"
    from pdp import native
    b = random_batch(batch=5, n=30, k=3, seed=77)
    hp, op = make_pair(oracle, b)
    hp.simplify()
    q = torch.rand(hp.E, 3, device='cuda:0'); fs = torch.rand(hp.E, 2, device='cuda:0')
    q0, fs0, av0 = q.clone(), fs.clone(), hp.active_variables.clone()
    am = torch.zeros(hp.B, dtype=torch.uint8, device='cuda:0')
    iters, _ = hp.sp_solve(q, fs, am, native.Decimator(hp), 9, 0.02, 100)
    assert iters == 0 and int(am.sum()) == 0
    assert torch.equal(q, q0) and torch.equal(fs, fs0) and torch.equal(hp.active_variables, av0)


@pytest.mark.parametrize('spec,T,pi,dprob', [(dict(batch=32, n=50, k=3, seed=70), 40, 0.1, 0.5),
                                             (dict(batch=64, n=40, mixed=True, seed=170), 30, 0.01, 0.3),
                                             (dict(batch=100, n=30, k=3, m=100, seed=310), 50, 0.2, 0.7),
                                             (dict(batch=6, n=200, k=3, seed=19), 30, 0.1, 1.0)])
@pytest.mark.parametrize('chunk', [None, '7'])
def test_persistent_reinforce_matches_oracle_loop(oracle, monkeypatch, spec, T, pi, dprob, chunk):
    """The Reinforce triple (pdp_decimate.py:202-234, pdp_predict.py:221-226) in one persistent call against the oracle's iteration loop
    with the same recorded coins: messages and the force column bit for bit, same active mask, solution and executed iterations."""
    if chunk:
        monkeypatch.setenv('PDP_SOLVE_CHUNK', chunk)
    res = _reinforce_pair(oracle, random_batch(**spec), T, pi, dprob, spec['seed'])
    assert np.abs(res['fs'][:, 1]).sum() > 0


def _reinforce_pair(oracle, b, T, pi, dprob, seed, expect_lds=True):
    from pdp import native
    hp, op = make_pair(oracle, b)
    coins = np.random.RandomState(seed).rand(T).astype(np.float32)
    res = op.forward('reinforce', T, local_search_iterations=0, pi=pi, decimation_probability=dprob, stream=coins, trace=True)
    hp.simplify()
    E, B = hp.E, hp.B
    q = torch.full((E, 3), 1.0, device='cuda:0') / 3.0
    fs = torch.zeros(E, 2, device='cuda:0'); fs[:, 0] = 0.5
    am = torch.ones(B, dtype=torch.uint8, device='cuda:0')
    dec = native.Decimator(hp)
    iters, used_lds = hp.sp_solve(q, fs, am, dec, T, 0.01, 0.0, pi=pi, model=native.MODEL_REINFORCE, coins=t(coins), decimation_probability=dprob)
    it = res['iterations_run']
    assert (used_lds or not expect_lds) and iters == it and res['rand_consumed'] == it
    res['hbm_instances'] = hp.last_solve_stats['hbm_instances']
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])
    return res


@pytest.mark.parametrize('spec,T,pi,dprob', [(dict(batch=32, n=50, k=3, seed=70), 40, 0.1, 0.5),
                                             (dict(batch=64, n=40, mixed=True, seed=170), 30, 0.01, 0.3),
                                             (dict(batch=200, n=100, k=3, seed=5000), 60, 0.1, 0.5)])
def test_reinforce_on_the_hbm_resident_kernel(oracle, monkeypatch, spec, T, pi, dprob):
    """The Reinforce triple on the HBM-resident kernel (instances past the LDS limit; forced here for small ones, host-driven chunk loop
    with snapshots -- the third batch is NaN-poisoned and replayed): same comparison with the oracle's loop, bit for bit."""
    monkeypatch.setenv('PDP_SOLVE_FORCE_HBM', '1')
    res = _reinforce_pair(oracle, random_batch(**spec), T, pi, dprob, spec['seed'], expect_lds=False)
    assert np.nansum(np.abs(res['fs'][:, 1])) > 0 and res['hbm_instances'] == spec['batch']


def test_reinforce_mixed_batch_and_single_instance(oracle):
    """Reinforce with an instance past the LDS limit: (a) next to small ones -- the small ones LDS-resident, the big one as a workgroup team
    of the HBM-resident kernel in the same chunk loop; (b) alone in its batch: exact mode (its own minimum in the gate, one launch)."""
    from pdp.factorgraph import dataset
    big = dataset.random_ksat_items(1, 2500, 3, m=int(4.0 * 2500), seed=78)
    items = dataset.random_ksat_items(20, 60, 3, m=240, seed=300) + big + dataset.random_ksat_items(19, 50, 3, m=200, seed=900)
    res = _reinforce_pair(oracle, dataset.collate_segment(items), 40, 0.1, 0.5, 5)
    assert res['hbm_instances'] == 1 and np.nansum(np.abs(res['fs'][:, 1])) > 0
    res = _reinforce_pair(oracle, dataset.collate_segment(big), 30, 0.1, 0.5, 6, expect_lds=False)
    assert res['hbm_instances'] == 1
    for seed in range(4):
        res = _reinforce_pair(oracle, dataset.collate_segment(dataset.random_ksat_items(1, 60, 3, m=250, seed=500 + seed)), 40, 0.1, 0.6, seed, expect_lds=False)


@pytest.mark.parametrize('spec,T,tol,t_max', [(dict(batch=500, n=30, k=3, m=100, seed=300), 50, 0.05, 8), (dict(batch=400, n=60, k=3, seed=7000), 120, 0.05, 8),
                                              (dict(batch=64, n=40, mixed=True, seed=100), 30, 0.05, 6)])
def test_lockstep_launch_on_larger_batches(oracle, monkeypatch, spec, T, tol, t_max):
    """The lock-step launch forced on batches that would normally run speculatively (several workgroups per CU, every one resident; mailbox
    reads folded over ranks r, r + 256, ...): 500 instances, the NaN-poisoned batch of 400 (the poison is exchanged inline, no replay), a
    mixed-k batch -- the oracle's strict semantics bit for bit."""
    monkeypatch.setenv('PDP_SOLVE_FORCE_LOCKSTEP', '1')
    b = random_batch(**spec)
    hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, T, tol, t_max)
    assert spec_ok and not used_lds and hp.last_solve_stats['hbm_instances'] == spec['batch']
    it = res['iterations_run']
    assert iters == it
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], res['trace_active_fn'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])


@pytest.mark.parametrize('B,n,T', [(2, 50, 40), (6, 40, 40), (12, 30, 50)])
def test_reinforce_small_batches_lockstep(oracle, B, n, T):
    """Reinforce on batches of a few instances: the gate's batch-global minimum is rarely an exact zero there (nothing is ever decimated), the
    speculation fails and the call reruns in the lock-step launch -- same comparison with the oracle's loop, for several batches each."""
    from pdp.factorgraph import dataset
    lock = 0
    for seed in range(6):
        b = dataset.collate_segment(dataset.random_ksat_items(B, n, 3, m=int(4.0 * n), seed=7700 + 10 * seed + B))
        res = _reinforce_pair(oracle, b, T, 0.1, 0.5, seed, expect_lds=False)
        lock += 1 if res['hbm_instances'] == B else 0
    assert lock > 0


def test_isolated_mode_on_the_hbm_resident_kernel(monkeypatch):
    """Isolated instances (no batch-wide couplings, a NaN stays inside its instance) on the HBM-resident kernel: the same end state as the
    LDS-resident kernel gives for the same batch, bit for bit -- a NaN-poisoned batch, so the two modes differ from the strict one."""
    from pdp import native
    b = random_batch(batch=400, n=60, k=3, seed=7000)
    outs = []
    for force_hbm in (False, True):
        if force_hbm:
            monkeypatch.setenv('PDP_SOLVE_FORCE_HBM', '1')
        hp = native.Problem(t(b['graph_map']), t(b['batch_variable_map']), t(b['batch_function_map']), t(b['edge_feature']))
        hp.simplify()
        q = torch.full((hp.E, 3), 1.0, device='cuda:0') / 3.0
        fs = torch.zeros(hp.E, 2, device='cuda:0'); fs[:, 0] = 0.5
        am = torch.ones(hp.B, dtype=torch.uint8, device='cuda:0')
        iters, used_lds = hp.sp_solve(q, fs, am, native.Decimator(hp), 120, 0.05, 8, isolate_instances=True)
        assert used_lds != force_hbm
        outs.append((iters, npy(q), npy(fs), npy(am), npy(hp.active_variables), npy(hp.active_functions), npy(hp.solution)))
    assert np.isnan(outs[0][2]).any(), "test batch no longer produces a NaN: pick another seed"
    assert outs[0][0] == outs[1][0]
    for x, y in zip(outs[0][1:], outs[1][1:]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize('T', [1, 2, 9])
def test_persistent_reinforce_degenerate_instances(oracle, T):
    "the edge-case batch of test_persistent_solve_degenerate_instances under the Reinforce triple"
    from pdp import generator
    rng = np.random.RandomState(4)
    ordinary = generator.uniform_ksat(30, 100, 3, rng)
    b = _tiny_batch([[[1], [-1, 2], [-2, 3]],
                     [[1], [-1]],
                     [[1, -2, 3]],
                     [[i + 1 for i in range(40)]],
                     [[1, (i % 6) + 2, -((i % 5) + 8)] for i in range(30)],
                     ordinary])
    _reinforce_pair(oracle, b, T, 0.1, 0.6, 3)


@pytest.mark.parametrize('big_n,T,tol,t_max', [(2500, 40, 0.05, 8), (4000, 26, 0.02, 100)])
def test_mixed_batch_routes_per_instance(oracle, big_n, T, tol, t_max):
    """Per-instance routing: a batch of small instances plus ONE instance far past the LDS limit (31 500 / 50 400 edges) no longer moves
    the whole batch to the HBM-resident kernel -- the small ones run LDS-resident, the big one on the HBM-resident kernel inside the same
    device-driven chunk loop (shared speculation record, NaN-poison decision and replay, early exit).  End state of every instance = the
    oracle's strict batch semantics, bit for bit."""
    from pdp import native
    from pdp.factorgraph import dataset
    items = dataset.random_ksat_items(20, 60, 3, m=240, seed=300) + dataset.random_ksat_items(1, big_n, 3, m=int(4.2 * big_n), seed=77) + \
        dataset.random_ksat_items(19, 50, 3, m=200, seed=900)
    b = dataset.collate_segment(items)
    hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, T, tol, t_max)
    assert spec_ok and used_lds and hp.last_solve_stats['hbm_instances'] == 1
    it = res['iterations_run']
    assert iters == it
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], res['trace_active_fn'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])


@pytest.mark.parametrize('n_big,with_small', [(3, True), (10, True), (3, False)])
def test_big_instances_run_as_workgroup_teams(oracle, n_big, with_small):
    """An instance past the LDS limit is spread over a TEAM of workgroups of the HBM-resident kernel (device-scope barriers and
    reductions, the team on one XCD): several big instances at once (ten = two teams on some XCDs), next to LDS-resident ones in the
    device-driven loop and alone in the host-driven loop; alpha = 3.5, so the big instances converge and are decimated (scorer,
    arg-max and simplification run through the team primitives too).  End state = the oracle's, bit for bit."""
    from pdp.factorgraph import dataset
    items = []
    if with_small:
        items += dataset.random_ksat_items(30, 60, 3, m=240, seed=310)
    items += [dataset.random_ksat_items(1, 1500 + 100 * i, 3, m=int(3.5 * (1500 + 100 * i)), seed=4100 + i)[0] for i in range(n_big)]
    if with_small:
        items += dataset.random_ksat_items(10, 50, 3, m=200, seed=910)
    b = dataset.collate_segment(items)
    hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, 60, 0.05, 8)
    if not spec_ok:
        pytest.skip("speculation failed on this batch (the caller would rerun step-wise)")
    assert used_lds == with_small and hp.last_solve_stats['hbm_instances'] == (n_big if with_small else len(items))
    it = res['iterations_run']
    assert iters == it
    # the big instances did get decimated
    assert (res['trace_active_var'][it - 1] == 0).sum() > 0
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], res['trace_active_fn'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])


@pytest.mark.parametrize('n,alpha,T,seeds', [(50, 4.2, 60, range(0, 12)), (200, 4.2, 100, range(20, 26)), (60, 3.5, 120, range(40, 52)), (3000, 3.5, 40, range(60, 62))])
def test_single_instance_batches_run_exact(oracle, n, alpha, T, seeds):
    """A batch of ONE instance: nobody else can supply the exact zero the speculation counts on (it fails in the first sweep for most
    random instances), but the batch-global minima of util.sparse_max / sparse_argmax are then the instance's own.  pdp_sp_solve computes
    them in the HBM-resident kernel (one launch, a team of workgroups for the n=3000 instance), a NaN survey poisons the instance from
    that sweep on without a replay, and the call never fails.  End state = the oracle's strict semantics, bit for bit."""
    from pdp.factorgraph import dataset
    saw_decimation = False
    for seed in seeds:
        items = dataset.random_ksat_items(1, n, 3, m=int(alpha * n), seed=9000 + seed)
        b = dataset.collate_segment(items)
        hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, T, 0.05, 8)
        assert spec_ok and not used_lds
        it = res['iterations_run']
        assert iters == it, seed
        saw_decimation |= bool((res['trace_active_var'][it - 1] == 0).any())
        np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1], err_msg=str(seed))
        np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1], err_msg=str(seed))
        np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], res['trace_active_fn'][it - 1], err_msg=str(seed))
        np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1], err_msg=str(seed))
        np.testing.assert_array_equal(npy(q), res['q'], err_msg=str(seed))
        np.testing.assert_array_equal(npy(fs), res['fs'], err_msg=str(seed))
    assert saw_decimation


@pytest.mark.parametrize('n,seed', [(50, 3), (60, 9), (2600, 4)])
def test_replicas_of_a_single_instance_run_exact(oracle, n, seed):
    """One instance with batch replication 3 from the deterministic initial state (the predict path with -b 3 on one file): the replicas are
    identical, so each replica's own minimum is the batch's -- exact mode, no speculation.  Equal to the oracle's replicated batch."""
    from pdp import native
    from pdp.factorgraph import dataset
    b = dataset.collate_segment(dataset.random_ksat_items(1, n, 3, m=int(3.6 * n), seed=9100 + seed))
    hp, op = make_pair(oracle, b, replication=3)
    res = op.forward('p-d-p', 50, local_search_iterations=0, tolerance=0.05, t_max=8, seed=5, trace=True)
    hp.simplify()
    q = torch.full((hp.E, 3), 1.0, device='cuda:0') / 3.0
    fs = torch.zeros(hp.E, 2, device='cuda:0'); fs[:, 0] = 0.5
    am = torch.ones(hp.B, dtype=torch.uint8, device='cuda:0')
    iters, used_lds = hp.sp_solve(q, fs, am, native.Decimator(hp), 50, 0.05, 8, replicas_identical=True)
    it = res['iterations_run']
    assert not used_lds and iters == it and hp.last_solve_stats['hbm_instances'] == 3
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])


@pytest.mark.parametrize('B,n,alpha,T,seeds', [(2, 50, 4.2, 60, range(0, 8)), (10, 50, 4.2, 60, range(0, 20)), (10, 60, 3.5, 100, range(30, 36)), (40, 30, 3.6, 80, range(50, 53))])
def test_small_batches_fall_back_to_lockstep(oracle, B, n, alpha, T, seeds):
    """Batches of a few instances: when no instance supplies the exact zero the speculation fails (4 of 20 random batches of 10, most
    batches of 2) and pdp_sp_solve reruns the batch in ONE lock-step launch -- one workgroup per instance, the batch-global minima, the NaN
    flag and the termination exchanged through mailboxes in every iteration -- instead of handing it to the step-wise loop.  Either way the
    end state equals the oracle's strict semantics bit for bit, and the call never fails."""
    from pdp.factorgraph import dataset
    lock = 0
    for seed in seeds:
        b = dataset.collate_segment(dataset.random_ksat_items(B, n, 3, m=int(alpha * n), seed=1000 * seed + 7))
        hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, T, 0.05, 8)
        assert spec_ok
        lock += 0 if used_lds else 1
        it = res['iterations_run']
        assert iters == it, seed
        np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1], err_msg=str(seed))
        np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1], err_msg=str(seed))
        np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], res['trace_active_fn'][it - 1], err_msg=str(seed))
        np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1], err_msg=str(seed))
        np.testing.assert_array_equal(npy(q), res['q'], err_msg=str(seed))
        np.testing.assert_array_equal(npy(fs), res['fs'], err_msg=str(seed))
    if B <= 10 and alpha > 4:
        assert lock > 0          # some of these batches did take the lock-step launch


@pytest.mark.parametrize('threads', [None, '256', '512'])
@pytest.mark.parametrize('n_big', [1, 3])
def test_wide_teams_across_xcds(oracle, monkeypatch, n_big, threads):
    """Big instances with nothing LDS-resident next to them get chip-wide teams (workgroups on all XCDs, agent-scope barriers issued by
    one wave per workgroup; 1 024 threads per workgroup by default, the other sizes through PDP_SOLVE_TEAM_THREADS); the threshold is lowered
    so that instances the oracle finishes in seconds qualify.  One instance alone (exact mode) and three in a batch (speculative mode,
    host-driven loop): end state = the oracle's, bit for bit."""
    from pdp.factorgraph import dataset
    monkeypatch.setenv('PDP_SOLVE_TEAM_WIDE_EDGES', '10000')
    if threads:
        monkeypatch.setenv('PDP_SOLVE_TEAM_THREADS', threads)
    items = [dataset.random_ksat_items(1, 2600 + 150 * i, 3, m=int(3.5 * (2600 + 150 * i)), seed=4300 + i)[0] for i in range(n_big)]
    b = dataset.collate_segment(items)
    hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, 60, 0.05, 8)
    if not spec_ok:
        pytest.skip("speculation failed on this batch (the caller would rerun step-wise)")
    assert not used_lds and hp.last_solve_stats['hbm_instances'] == n_big
    it = res['iterations_run']
    assert iters == it
    assert (res['trace_active_var'][it - 1] == 0).sum() > 0
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], res['trace_active_fn'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])


def test_mixed_batch_routing_under_nan_poison(oracle, monkeypatch):
    """The NaN-poisoned batch of test_persistent_solve_reproduces_nan_poisoning with a 21 000-edge instance added: the big instance runs on
    the HBM-resident kernel, is saved at chunk entry and replayed from there once some (small) instance poisons the batch; the small
    instances are replayed selectively as before.  Everything equals the oracle's strict semantics bit for bit -- with decimations of the
    big instance before the poison and none after it."""
    from pdp.factorgraph import dataset
    from helpers import random_batch as rb
    monkeypatch.setenv('PDP_SOLVE_NO_ADOPT', '1')          # every instance with an event behind the poison is replayed (the replay is what this test is about)
    small = rb(batch=400, n=60, k=3, seed=7000)
    # rebuild the same small instances as loader items and append the big one
    from pdp import generator
    items = []
    for i in range(400):
        rng = np.random.RandomState(7000 + i)
        items.append(dataset.instance_from_clauses(60, generator.uniform_ksat(60, generator.clause_count(60, 3), 3, rng), label=-1, name="u%d" % i))
    b0 = dataset.collate_segment(items)
    np.testing.assert_array_equal(b0['graph_map'], small['graph_map'])
    items.append(dataset.random_ksat_items(1, 2000, 3, m=7000, seed=123)[0])           # alpha = 3.5: converges and gets decimated
    b = dataset.collate_segment(items)
    hp, res, q, fs, am, iters, used_lds, spec_ok = run_pair(oracle, b, 120, 0.05, 8)
    assert np.isnan(res['fs']).any() and spec_ok and used_lds and hp.last_solve_stats['hbm_instances'] == 1 and hp.last_solve_stats['replays'] >= 1
    it = res['iterations_run']
    assert iters == it
    big_v = slice(int(b['batch_variable_map'].size) - 2000, None)
    assert (res['trace_active_var'][it - 1][big_v] == 0).sum() >= 3                     # the big instance was decimated (before the poison)
    np.testing.assert_array_equal(npy(am), res['trace_active_mask'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1])
    np.testing.assert_array_equal(npy(hp.active_functions)[:, 0], res['trace_active_fn'][it - 1])
    np.testing.assert_array_equal(npy(hp.solution), res['trace_solution'][it - 1])
    np.testing.assert_array_equal(npy(q), res['q'])
    np.testing.assert_array_equal(npy(fs), res['fs'])
