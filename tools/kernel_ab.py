#!/usr/bin/env python3
"""Debug helper: runs the persistent solver on two of its paths -- 'lds' (default routing), 'hbm' (PDP_SOLVE_FORCE_HBM: the HBM-resident kernel,
host-driven chunk loop) and 'lock' (PDP_SOLVE_FORCE_LOCKSTEP: the lock-step launch, batches of up to 1 024 instances) -- from the same initial state
for T = 1, 2, ... and reports the first T at which any output differs, with the instance / element it happens in.
usage: python tools/kernel_ab.py [batch] [n] [Tmax] [alpha] [pathA] [pathB]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import numpy as np, torch
from pdp import native
from pdp.factorgraph import dataset
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
Tmax = int(sys.argv[3]) if len(sys.argv) > 3 else 40
alpha = float(sys.argv[4]) if len(sys.argv) > 4 else 4.2
PATH_A = sys.argv[5] if len(sys.argv) > 5 else 'lds'
PATH_B = sys.argv[6] if len(sys.argv) > 6 else 'hbm'
dev = torch.device('cuda:0')
items = dataset.random_ksat_items(B, n, 3, m=int(round(alpha * n)), seed=7)
b = dataset.to_torch(dataset.collate_segment(items), dev)
e0 = np.concatenate(([0], np.cumsum([it[2].shape[1] for it in items]))); v0 = np.concatenate(([0], np.cumsum([it[0] for it in items])))

def run(kernel, T, tol=0.02, tmax=100):
    for k in ('PDP_SOLVE_FORCE_HBM', 'PDP_SOLVE_FORCE_LOCKSTEP'):
        os.environ.pop(k, None)
    if kernel == 'hbm':
        os.environ['PDP_SOLVE_FORCE_HBM'] = '1'
    elif kernel == 'lock':
        os.environ['PDP_SOLVE_FORCE_LOCKSTEP'] = '1'
    hp = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    hp.simplify()
    q = torch.full((hp.E, 3), 1.0, device=dev) / 3.0; fs = torch.zeros(hp.E, 2, device=dev); fs[:, 0] = 0.5
    am = torch.ones(hp.B, dtype=torch.uint8, device=dev); dec = native.Decimator(hp)
    try:
        it, lds = hp.sp_solve(q, fs, am, dec, T, tol, tmax)
    except native.SpeculationFailed as ex:
        return None
    g = lambda x: x.detach().cpu().numpy().copy()
    return dict(q=g(q), fs=g(fs), am=g(am), av=g(hp.active_variables).reshape(-1), af=g(hp.active_functions).reshape(-1), sol=g(hp.solution), it=it)

for T in list(range(1, 14)) + list(range(14, Tmax + 1, 3)):
    a, c = run(PATH_A, T), run(PATH_B, T)
    if a is None or c is None:
        print('T=%d: speculation failure v3=%s w=%s' % (T, a is None, c is None)); break
    bad = [k for k in ('q', 'fs', 'am', 'av', 'af', 'sol') if not np.array_equal(a[k], c[k], equal_nan=True)]
    print('T=%d iters v3=%d w=%d %s' % (T, a['it'], c['it'], 'DIFF ' + ','.join(bad) if bad else 'equal'))
    if bad:
        for k in bad:
            x, y = a[k].reshape(len(a[k]), -1), c[k].reshape(len(c[k]), -1)
            rows = np.where(~np.all((x == y) | (np.isnan(x) & np.isnan(y)), axis=1))[0]
            off = e0 if k in ('q', 'fs') else (v0 if k in ('av', 'sol') else None)
            inst = sorted(set(int(np.searchsorted(off, r, side='right') - 1) for r in rows[:2000])) if off is not None else []
            print('   %s: %d rows differ, first %s, instances %s' % (k, len(rows), rows[:6], inst[:10]))
            for r in rows[:3]:
                print('      row %d: v3 %s  w %s' % (r, x[r], y[r]))
        break
