#!/usr/bin/env python3
"""Debug helper: per-phase shader-cycle sums of the persistent solver (needs a -DPDP_PHASE_PROF build)."""
import ctypes as C, os, sys, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import torch
from pdp import native
if os.path.exists(native.LIB_PATH.replace('.so', '_prof.so')):
    native.LIB_PATH = native.LIB_PATH.replace('.so', '_prof.so')      # built with EXTRA=-DPDP_PHASE_PROF next to the product library
from pdp.factorgraph import dataset
B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000, 200
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda:0')
b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(B, n, 3, seed=0)), dev)
prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
L = native.lib()
out = (C.c_ulonglong * 40)()
for rep in range(2):
    native.check(L.pdp_problem_bind_state(prob._h, native.ptr(prob.active_variables), native.ptr(prob.active_functions), native.ptr(prob.solution), native.ptr(prob.is_sat), native.ptr(prob.edge_mask), native._stream()))
    q = torch.full((prob.E, 3), 1.0, device=dev) / 3.0; fs = torch.zeros(prob.E, 2, device=dev); fs[:, 0] = 0.5
    am = torch.ones(prob.B, dtype=torch.uint8, device=dev); dec = native.Decimator(prob)
    prob.simplify(); torch.cuda.synchronize()
    L.pdp_debug_phase_cycles(out, 1)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
    it, lds = prob.sp_solve(q, fs, am, dec, T, 0.02, 100)
    t1.record(); torch.cuda.synchronize()
    wall_ms = t0.elapsed_time(t1)
    L.pdp_debug_phase_cycles(out, 0)
names = {0: 'load', 1: 'E1 logs', 2: 'R1 row sums', 3: 'E2 exps/div', 4: 'P4 smooth max', 5: 'P5 reduce', 24: 'gate + bookkeeping', 6: 'P6 decimate', 25: 'P7 mask refresh', 30: 'mask pass behind a refresh', 29: 'P8 clause count (the count)', 26: 'P8 clause count (its barrier)', 8: 'write back'}
tot = sum(out[i] for i in names)
for i, nm in names.items():
    print("%-16s %14d cycles  %5.1f%%" % (nm, out[i], 100.0 * out[i] / tot))
print("total WG-cycles %d, solve call %.2f ms (PDP_DEBUG_SKIP=%s)" % (tot, wall_ms, os.environ.get("PDP_DEBUG_SKIP", "0")))
for i, nm in ((16, "scorer edge logs"), (17, "scorer sums+score+flags"), (18, "arg-max"), (19, "fix-point verification"), (20, "set variable"), (21, "unit/pure scans")):
    print("  decimation: %-26s %12d cycles  %5.1f%% of P6" % (nm, out[i], 100.0 * out[i] / max(1, out[6])))
print("  decimations followed by the general unit propagation: %d, by the general peel: %d" % (out[22], out[23]))
print("clause counts %d, mask refreshes %d" % (out[27], out[28]))
print("instance-iterations %d, exact smooth-max passes %d, decimations on the neighbourhood path %d, on the general path %d" % (out[12], out[9], out[10], out[11]))
