#!/usr/bin/env python3
"""Per-instance routing on the headline batch: 5 000 instances of n=200 plus ONE instance far past the LDS limit (default 50 400 edges).
Prints iterations/s of the persistent solve for the plain batch, the mixed batch with routing, and the mixed batch with routing disabled
(PDP_SOLVE_NO_ROUTING=1: the whole batch on the HBM-resident kernel, the behaviour before).
usage: python tools/mixed_batch_time.py [big_n] [T]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import torch
from pdp import native
from pdp.factorgraph import dataset
big_n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda:0')
items = dataset.random_ksat_items(5000, 200, 3, m=840, seed=0)
big = dataset.random_ksat_items(1, big_n, 3, m=int(round(4.2 * big_n)), seed=99)


def run(its, reps=5):
    b = dataset.to_torch(dataset.collate_segment(its), dev)
    hp = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    L = native.lib()
    q = torch.empty(hp.E, 3, device=dev); fs = torch.empty(hp.E, 2, device=dev); am = torch.empty(hp.B, dtype=torch.uint8, device=dev)
    dec = native.Decimator(hp)
    best = None
    for r in range(reps):
        native.check(L.pdp_problem_bind_state(hp._h, native.ptr(hp.active_variables), native.ptr(hp.active_functions), native.ptr(hp.solution),
                                              native.ptr(hp.is_sat), native.ptr(hp.edge_mask), native._stream()))
        q.fill_(1.0); q.div_(3.0); fs.zero_(); fs[:, 0] = 0.5; am.fill_(1); dec.reset(); hp.simplify()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        it, lds = hp.sp_solve(q, fs, am, dec, T, 0.02, 100)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return it / best, hp.last_solve_stats, lds


if os.environ.get('PDP_MIXED_SMALL'):
    # the big instance next to ONE small one: what the big instance costs by itself
    alone, st, _ = run(items[:int(os.environ['PDP_MIXED_SMALL'])] + big)
    print('%s small + the big instance: %8.0f iterations/s = %.1f us per iteration (%d on the HBM-resident kernel)'
          % (os.environ['PDP_MIXED_SMALL'], alone, 1e6 / alone, st['hbm_instances']))
    sys.exit(0)
plain, st0, _ = run(items)
print('5000 x n=200:                          %8.0f iterations/s' % plain)
mixed, st1, lds1 = run(items + big)
print('+ 1 instance of n=%d (%d edges): %8.0f iterations/s = %.0f %% of the plain batch; LDS-resident %s, %d instance(s) on the HBM-resident kernel'
      % (big_n, big[0][2].shape[1], mixed, 100.0 * mixed / plain, lds1, st1['hbm_instances']))
os.environ['PDP_SOLVE_NO_ROUTING'] = '1'
old, st2, lds2 = run(items + big, reps=2)
print('same batch without routing:            %8.0f iterations/s (LDS-resident %s, %d instances on the HBM-resident kernel)' % (old, lds2, st2['hbm_instances']))
