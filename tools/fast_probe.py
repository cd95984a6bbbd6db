import os, sys, time, json
sys.path.insert(0, 'pdp-solver_amd')
import numpy as np, torch
from pdp import native
from pdp.factorgraph import dataset
dev = torch.device('cuda:0')
B, n = 5000, 200
items = dataset.random_ksat_items(B, n, 3, m=840, seed=0)
hb = dataset.collate_segment(items)
b = dataset.to_torch(hb, dev)
out = {}
res = {}
for build in ('parity', 'fast'):
    native.use_build(build)
    prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'], batch_size=B)
    E = prob.E
    ts = []
    for rep in range(6):
        L = native.lib()
        native.check(L.pdp_problem_bind_state(prob._h, native.ptr(prob.active_variables), native.ptr(prob.active_functions), native.ptr(prob.solution), native.ptr(prob.is_sat), native.ptr(prob.edge_mask), native._stream()))
        prob.simplify()
        q = torch.full((E, 3), 1.0, device=dev) / 3.0; fs = torch.zeros(E, 2, device=dev); fs[:, 0] = 0.5
        am = torch.ones(B, dtype=torch.uint8, device=dev); dec = native.Decimator(prob)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        it, lds = prob.sp_solve(q, fs, am, dec, 100, 0.02, 100, time_kernels=True, inputs_disposable=True)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        st = dict(prob.last_solve_stats)
    prob.random_fill(seed=1); o, ws = prob.local_search(prob.solution.clone(), 100, 0.5, seed=2)
    pred = prob.update_solution(o.reshape(-1).contiguous()); solved, unsat = prob.cnf_eval(pred.reshape(-1).contiguous())
    res[build] = dict(q=q.cpu().numpy(), av=prob.active_variables.cpu().numpy(), sol=prob.solution.cpu().numpy())
    out[build] = dict(call_ms=ts, iters=it, stats=st, solved=float(solved.sum()), unsat=float(unsat.sum()), kernel=native.kernel_name('sp_solve'))
    del prob, dec
a, c = res['parity'], res['fast']
nanq = np.isnan(a['q']).any(axis=1), np.isnan(c['q']).any(axis=1)
out['compare'] = dict(av_equal_frac=float((a['av'] == c['av']).mean()), nan_rows=(int(nanq[0].sum()), int(nanq[1].sum())),
                      q_maxabs=float(np.nanmax(np.abs(a['q'] - c['q']))), sol_equal_frac=float((a['sol'] == c['sol']).mean()))
print(json.dumps(out, indent=1))
