"""Where a neural forward spends its wall time outside the kernels: one np-nd-np forward at the configs[3] per-GPU shape (n=400, 5 000 instances,
25.2 M edges), every host-visible phase bracketed by torch.cuda.synchronize.  Usage: python tools/neural_forward_phases.py [n] [batch] [T]"""
import sys, time, logging
import torch
sys.path.insert(0, '/root/repo/pdp-solver_amd')
from pdp.factorgraph import dataset
from pdp.trainer import SatFactorGraphTrainer
from pdp.nn import solver as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda:0')
cfg = dict(model_type='np-nd-np', model_name='m', verbose=False, local_search_iteration=0, epsilon=0.5, rng='philox', random_seed=1, hidden_dim=128,
           edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1, mem_hidden_dim=100, agg_hidden_dim=100, mem_agg_hidden_dim=50, classifier_dim=50,
           test_batch_limit=1 << 62, batch_size=B, test_recurrence_num=T, tolerance=0.02, t_max=100)
torch.manual_seed(1234)
tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=logging.getLogger('t'))
m = tr._model_list[0]
b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(B, n, 3, m=int(round(4.2 * n)), seed=7000001)), dev)
gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']


class Clock:
    def __init__(self): self.rows = []
    def __call__(self, name, fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        self.rows.append((name, 1e3 * (time.perf_counter() - t0))); return r


for rep in range(3):
    ck = Clock()
    with torch.no_grad():
        st = ck('get_init_state', lambda: m.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1))
        sp = ck('SATProblem', lambda: S.SATProblem((gm, bvm, bfm, ef, None, None), dev, 1))
        ck('simplify', sp.simplify)
        ps, ds = st
        am = torch.ones(sp._batch_size, 1, dtype=torch.uint8, device=dev)
        for t in range(T):
            ps = ck('propagator %d' % t, lambda: m._propagator(ps, ds, sp, False, am))
            ds = ck('decimator %d' % t, lambda: m._decimator(ds, ps, sp, False, am))
            ck('refresh_edge_mask %d' % t, sp.refresh_edge_mask)
            pred = ck('predictor %d' % t, lambda: m._predictor(ds, sp))
            pred = ck('update_solution %d' % t, lambda: m._update_solution(pred, sp))
            ck('check_termination %d' % t, lambda: tr._check_recurrence_termination(am, pred, sp))
            ck('active sum %d' % t, lambda: int(am.sum().item()))
        whole = ck('WHOLE forward (fresh)', lambda: m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                                                    is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination, batch_replication=1))
    print('--- repetition %d (E = %d)' % (rep, gm.size(1)))
    for name, ms in ck.rows:
        print('  %-28s %9.2f ms' % (name, ms))
    print('  torch allocator: reserved %.1f GB, allocated %.1f GB' % (torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_allocated() / 1e9))
