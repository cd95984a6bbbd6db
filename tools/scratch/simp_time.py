import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import torch
from pdp import native
native.LIB_PATH = native.LIB_PATH.replace('.so', '_exp.so')
from pdp.factorgraph import dataset
dev = torch.device('cuda:0')
for B in (5000, 4999):
    b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(B, 200, 3, seed=0)), dev)
    prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    L = native.lib()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for rep in range(12):
        native.check(L.pdp_problem_bind_state(prob._h, native.ptr(prob.active_variables), native.ptr(prob.active_functions), native.ptr(prob.solution), native.ptr(prob.is_sat), native.ptr(prob.edge_mask), native._stream()))
        torch.cuda.synchronize()
        ev[0].record(); prob.simplify(); ev[1].record(); torch.cuda.synchronize()
        ts.append(1e3 * ev[0].elapsed_time(ev[1]))
    print('B=%d: simplify %.1f us (min %.1f)' % (B, sorted(ts)[len(ts) // 2], min(ts)))
