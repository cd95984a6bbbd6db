import sys, time, torch
sys.path.insert(0, '/root/repo/pdp-solver_amd')
from pdp.factorgraph import dataset
from pdp import native
dev = torch.device('cuda:0')
tb = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(5000, 200, 3, m=840, seed=0)), dev)
def t(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = t()
    hp = native.Problem(tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'])
    t1 = t()
    hp.simplify()
    t2 = t()
    q = torch.full((hp.E, 3), 1.0, device=dev) / 3.0; fs = torch.zeros(hp.E, 2, device=dev); fs[:, 0] = 0.5
    am = torch.ones(hp.B, dtype=torch.uint8, device=dev); dec = native.Decimator(hp)
    t3 = t()
    hp.sp_solve(q, fs, am, dec, 100, 0.02, 100)
    t4 = t()
    hp.sp_solve(q, fs, am, dec, 1, 0.02, 100)
    t5 = t()
    del hp, dec
    t6 = t()
    print("create %.2f  simplify %.2f  state %.2f  first solve %.2f  (second call, 1 iteration: %.2f)  destroy %.2f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)))
