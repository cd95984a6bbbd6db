"""How often does pdp_sp_solve succeed by batch size (20 random batches each)?  By default never-failing paths are on: single-instance batches are
solved exactly and small batches whose speculation failed rerun in the lock-step launch.  PDP_SOLVE_NO_EXACT=1 PDP_SOLVE_NO_LOCKSTEP=1 shows the raw
speculation: it holds for 20/20 batches at B >= 100, 16/20 at B = 10, 1-10/20 at B = 1, failing in the first iteration."""
import os, sys
sys.path.insert(0, '/root/repo/pdp-solver_amd')
import torch, numpy as np
from pdp import native
from pdp.factorgraph import dataset
dev = torch.device('cuda:0')
def run(its, T=50):
    b = dataset.to_torch(dataset.collate_segment(its), dev)
    hp = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    q = torch.full((hp.E, 3), 1.0/3, device=dev); fs = torch.zeros(hp.E, 2, device=dev); fs[:, 0] = 0.5
    am = torch.ones(hp.B, dtype=torch.uint8, device=dev)
    dec = native.Decimator(hp); hp.simplify()
    try:
        it, lds = hp.sp_solve(q, fs, am, dec, T, 0.02, 100)
        return True
    except native.SpeculationFailed:
        return False
for B, n in [(1, 50), (1, 200), (10, 50), (100, 50), (100, 200), (1000, 50)]:
    ok = 0; tot = 20; first = []
    for s in range(tot):
        its = dataset.random_ksat_items(B, n, 3, m=int(4.2 * n), seed=1000 * s + 7)
        held = run(its)
        ok += held
        if not held:            # first iteration count at which the call fails = 1 + the failing iteration
            first.append(next(t for t in range(1, 51) if not run(its, T=t)))
    print('B=%d n=%d: speculation held in %d of %d batches; failing iteration (1-based) of the others: %s' % (B, n, ok, tot, sorted(first)))
