#!/usr/bin/env python3
"""Randomised parity soak of pdp_sp_solve and pdp_local_search against the CPU oracle (strict reference semantics, bit for bit): random batch
compositions (tiny to mid-size instances, mixed k, optionally one or two instances past the LDS limit, single-instance and very small
batches), both model types, random T / tolerance / t_max / alpha, so that every routing decision of the library is hit: LDS-resident,
per-instance routing with workgroup teams, exact single-instance mode, lock-step launch, Walk-SAT routing.  Test infrastructure (uses oracle/).
usage: python tools/parity_soak.py [seconds] [seed]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd')); sys.path.insert(0, REPO)
import numpy as np, torch
from pdp import native, generator
from pdp.factorgraph import dataset
from oracle import binding
binding.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device('cuda:0')
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
npy = lambda x: x.detach().cpu().numpy()
paths = {}
t_end = time.time() + budget
runs = 0
while time.time() < t_end:
    kind = rng.choice(['small', 'tiny', 'single', 'mixed', 'bigs'])
    items = []
    def inst(n, alpha, k=None):
        kk = int(k or rng.choice([3, 3, 3, 4, 5, 2]))
        m = max(1, int(alpha * n))
        return dataset.instance_from_clauses(n, generator.uniform_ksat(n, m, kk, np.random.RandomState(rng.randint(1 << 30))), label=-1, name='s')
    alpha = float(rng.choice([3.0, 3.5, 3.8, 4.2, 4.4]))
    if kind == 'small':
        items = [inst(int(rng.randint(8, 90)), alpha) for _ in range(int(rng.randint(20, 300)))]
    elif kind == 'tiny':
        items = [inst(int(rng.randint(8, 70)), alpha) for _ in range(int(rng.randint(2, 12)))]
    elif kind == 'single':
        items = [inst(int(rng.choice([20, 60, 200, 900, 2500])), alpha, 3)]
    elif kind == 'mixed':
        items = [inst(int(rng.randint(10, 80)), alpha) for _ in range(int(rng.randint(10, 60)))]
        for _ in range(int(rng.randint(1, 3))):
            items.insert(int(rng.randint(0, len(items) + 1)), inst(int(rng.randint(1500, 3500)), float(rng.choice([3.5, 3.8, 4.2])), 3))
    else:
        items = [inst(int(rng.randint(1500, 3000)), float(rng.choice([3.5, 4.0])), 3) for _ in range(int(rng.randint(2, 5)))]
    b = dataset.collate_segment(items)
    model = 'reinforce' if rng.rand() < 0.3 else 'p-d-p'
    T = int(rng.choice([1, 7, 13, 30, 60, 100]))
    if kind in ('single', 'mixed', 'bigs'):
        T = min(T, 40)
    tol = float(rng.choice([0.02, 0.05, 0.1])); t_max = float(rng.choice([4, 8, 100]))
    hp = native.Problem(t(b['graph_map']), t(b['batch_variable_map']), t(b['batch_function_map']), t(b['edge_feature']))
    op = binding.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'], 1)
    hp.simplify()
    q = torch.full((hp.E, 3), 1.0, device=dev) / 3.0; fs = torch.zeros(hp.E, 2, device=dev); fs[:, 0] = 0.5
    am = torch.ones(hp.B, dtype=torch.uint8, device=dev); dec = native.Decimator(hp)
    desc = '%s %s B=%d E=%d T=%d tol=%g tmax=%g alpha=%g' % (kind, model, hp.B, hp.E, T, tol, t_max, alpha)
    try:
        if model == 'p-d-p':
            res = op.forward('p-d-p', T, local_search_iterations=0, tolerance=tol, t_max=t_max, seed=5, trace=True)
            iters, lds = hp.sp_solve(q, fs, am, dec, T, tol, t_max)
        else:
            coins = rng.rand(T).astype(np.float32); pi = float(rng.choice([0.01, 0.1])); dprob = float(rng.choice([0.3, 0.6, 1.0]))
            res = op.forward('reinforce', T, local_search_iterations=0, pi=pi, decimation_probability=dprob, stream=coins, trace=True)
            iters, lds = hp.sp_solve(q, fs, am, dec, T, 0.01, 0.0, pi=pi, model=native.MODEL_REINFORCE, coins=t(coins), decimation_probability=dprob)
    except native.SpeculationFailed:
        paths['speculation failed -> caller'] = paths.get('speculation failed -> caller', 0) + 1
        continue
    it = res['iterations_run']
    ok = iters == it and np.array_equal(npy(am), res['trace_active_mask'][it - 1]) and np.array_equal(npy(hp.solution), res['trace_solution'][it - 1]) \
        and np.array_equal(npy(q), res['q'], equal_nan=True) and np.array_equal(npy(fs), res['fs'], equal_nan=True) \
        and np.array_equal(npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1])
    key = ('lds' if lds else 'hbm') + ' / %d on the HBM-resident kernel' % (0 if hp.last_solve_stats['hbm_instances'] == 0 else (1 if hp.last_solve_stats['hbm_instances'] < hp.B else 2))
    paths[kind + ' ' + model + ' ' + key] = paths.get(kind + ' ' + model + ' ' + key, 0) + 1
    if not ok:
        bad = [name for name, x, y in (('active_mask', npy(am), res['trace_active_mask'][it - 1]), ('solution', npy(hp.solution), res['trace_solution'][it - 1]),
                                       ('q', npy(q), res['q']), ('fs', npy(fs), res['fs']), ('active_var', npy(hp.active_variables)[:, 0], res['trace_active_var'][it - 1]))
               if not np.array_equal(x, y, equal_nan=True)]
        print('MISMATCH solve:', desc, 'iters', iters, it, 'fields', bad, 'path', key)
        os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
        extra = dict(coins=coins, pi=pi, dprob=dprob) if model != 'p-d-p' else {}
        np.savez(os.path.join(REPO, 'gpurun_out', 'soak_fail.npz'), graph_map=b['graph_map'], batch_variable_map=b['batch_variable_map'],
                 batch_function_map=b['batch_function_map'], edge_feature=b['edge_feature'], T=T, tol=tol, t_max=t_max, model=model, **extra)
        sys.exit(1)
    # Walk-SAT on the state the solve left, Philox numbers
    w = int(rng.choice([5, 40]))
    hp.random_fill(seed=77); op.random_fill(seed=77)
    pred = op.state()[2]
    hout, hsteps = hp.local_search(t(pred), w, 0.5, seed=9)
    oout, osteps, _ = op.local_search(pred, w, 0.5, seed=9)
    if hsteps != osteps or not np.array_equal(npy(hout)[:, 0], oout):
        print('MISMATCH walksat:', desc, hsteps, osteps); sys.exit(1)
    runs += 1
print('parity soak: %d runs, all equal to the oracle' % runs)
for k in sorted(paths):
    print('  %-70s %d' % (k, paths[k]))
