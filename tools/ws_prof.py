#!/usr/bin/env python3
"""Debug helper: per-phase cycle sums of the persistent Walk-SAT kernel (needs libpdp_hip_prof.so built with -DPDP_PHASE_PROF)."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import torch
from pdp import native
native.LIB_PATH = native.LIB_PATH.replace('.so', '_prof.so')
from pdp.factorgraph import dataset
B, n, steps = 5000, 200, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device('cuda:0')
b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(B, n, 3, seed=0)), dev)
prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
prob.simplify(); prob.random_fill(seed=3)
L = native.lib(); out = (C.c_ulonglong * 8)()
for rep in range(2):
    L.pdp_debug_ws_cycles(out, 1)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
    res, st = prob.local_search(prob.solution.clone(), steps, 0.5, seed=999)
    t1.record(); torch.cuda.synchronize()
    L.pdp_debug_ws_cycles(out, 0)
names = ['setup', 'scan', 'wave maxima + barrier', 'join + random-number prefetch', 'update pass + barrier']
tot = sum(out[i] for i in range(5))
for i, nm in enumerate(names):
    print("%-36s %14d cycles %5.1f%%  (%.0f per instance-step)" % (nm, out[i], 100.0 * out[i] / tot, out[i] / (B * st)))
print("call %.2f ms, %d steps" % (t0.elapsed_time(t1), st))
