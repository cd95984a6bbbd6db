"""Times the persistent Walk-SAT (pdp_local_search) alone.  Usage: python tools/walksat_time.py [n] [steps] [batch] [big_n]
(big_n: one more instance of that size in the batch -- past the LDS limit the whole batch takes the strict loop today)"""
import sys, time, torch
sys.path.insert(0, '/root/repo/pdp-solver_amd')
from pdp.factorgraph import dataset
from pdp import native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
dev = torch.device('cuda:0')
items = dataset.random_ksat_items(B, n, 3, m=int(round(4.2 * n)), seed=0)
if len(sys.argv) > 4:
    bn = int(sys.argv[4]); items = items + dataset.random_ksat_items(1, bn, 3, m=int(round(4.2 * bn)), seed=99); B += 1
tb = dataset.to_torch(dataset.collate_segment(items), dev)
p = native.Problem(tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'])
p.simplify(); p.random_fill(seed=1)
for rep in range(3):
    sol = p.solution.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, done = p.local_search(sol, steps, 0.5, seed=5 + rep)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    solved, unsat = p.cnf_eval(out.reshape(-1).contiguous())
    print('n=%d batch=%d: %d Walk-SAT steps in %.2f ms (%.1f M instance-steps/s), solved %d, unsat clauses %d'
          % (n, B, done, 1e3 * dt, B * done / dt / 1e6, int(solved.sum().item()), int(unsat.sum().item())))
