import os, sys, time
sys.path.insert(0, '/root/repo/pdp-solver_amd')
import torch
from pdp import native
from pdp.factorgraph import dataset
dev = torch.device('cuda:0')
n = int(sys.argv[1]); T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
its = dataset.random_ksat_items(1, n, 3, m=int(3.5 * n), seed=11)
b = dataset.to_torch(dataset.collate_segment(its), dev)
hp = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
L = native.lib()
q = torch.empty(hp.E, 3, device=dev); fs = torch.empty(hp.E, 2, device=dev); am = torch.empty(hp.B, dtype=torch.uint8, device=dev)
dec = native.Decimator(hp)
best = None
for r in range(3):
    native.check(L.pdp_problem_bind_state(hp._h, native.ptr(hp.active_variables), native.ptr(hp.active_functions), native.ptr(hp.solution),
                                          native.ptr(hp.is_sat), native.ptr(hp.edge_mask), native._stream()))
    q.fill_(1.0); q.div_(3.0); fs.zero_(); fs[:, 0] = 0.5; am.fill_(1); dec.reset(); hp.simplify()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    it, lds = hp.sp_solve(q, fs, am, dec, T, 0.02, 100)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    best = dt if best is None or dt < best else best
print('n=%d E=%d: %d iterations, %.3f ms per iteration, %.3g edge updates/s' % (n, hp.E, it, 1e3 * best / it, 2.0 * hp.E * it / best))
