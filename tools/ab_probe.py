#!/usr/bin/env python3
"""Same-box A/B of pdp_sp_solve on the headline batch: python tools/ab_probe.py libA.so libB.so [reps] -- alternates the two libraries (each in
its own child process, so that nothing is shared) and prints the step and kernel times of every round."""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, os.path.join(%r, 'pdp-solver_amd'))
import torch
from pdp import native
from pdp.factorgraph import dataset
dev = torch.device('cuda:0')
B = 5000
b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(B, 200, 3, m=840, seed=0)), dev)
prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'], batch_size=B)
E = prob.E; L = native.lib(); ts = []; ks = []; rs = []
for rep in range(8):
    native.check(L.pdp_problem_bind_state(prob._h, native.ptr(prob.active_variables), native.ptr(prob.active_functions), native.ptr(prob.solution), native.ptr(prob.is_sat), native.ptr(prob.edge_mask), native._stream()))
    q = torch.full((E, 3), 1.0, device=dev) / 3.0; fs = torch.zeros(E, 2, device=dev); fs[:, 0] = 0.5
    am = torch.ones(B, dtype=torch.uint8, device=dev); dec = native.Decimator(prob)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prob.simplify()
    it, lds = prob.sp_solve(q, fs, am, dec, 100, 0.02, 100, time_kernels=True, inputs_disposable=True)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0)); ks.append(prob.last_solve_stats['solve_kernel_ms']); rs.append(prob.last_solve_stats['replay_kernel_ms'])
print(json.dumps(dict(step_ms=sorted(ts)[len(ts) // 2], kernel_ms=sorted(ks)[len(ks) // 2], replay_ms=sorted(rs)[len(rs) // 2])))
''' % REPO
libs = sys.argv[1:3]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for r in range(reps):
    for lib in libs:
        parts = lib.split(':')                       # lib.so[:ENV=VALUE ...]: extra environment for that arm
        env = dict(os.environ, PDP_HIP_LIB=os.path.join(REPO, 'pdp-solver_amd', 'csrc', parts[0]))
        env.update(kv.split('=', 1) for kv in parts[1:])
        out = subprocess.run([sys.executable, '-c', CHILD], env=env, stdout=subprocess.PIPE, universal_newlines=True).stdout.strip().split('\n')[-1]
        print(r, lib, out, flush=True)
