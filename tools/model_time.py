"""Times a whole `forward` of the Python API for one of the classical model types on config 2's batch (5 000 x n=200 m=840).
Usage: python tools/model_time.py [model_type: p-d-p | reinforce | walk-sat] [iterations] [alpha] [walk-sat steps]"""
import sys, time, logging
import torch
sys.path.insert(0, '/root/repo/pdp-solver_amd')
from pdp.factorgraph import dataset
from pdp.trainer import SatFactorGraphTrainer
MODEL = sys.argv[1] if len(sys.argv) > 1 else 'reinforce'
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ALPHA = float(sys.argv[3]) if len(sys.argv) > 3 else 4.2
B, N = 5000, 200
WS = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cfg = dict(model_type=MODEL, model_name='m', verbose=False, local_search_iteration=WS, epsilon=0.5, rng='torch', random_seed=1,
           pi=0.01, decimation_probability=0.5, tolerance=0.02, t_max=100, batch_size=B, test_recurrence_num=T,
           hidden_dim=128, edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1, mem_hidden_dim=100, agg_hidden_dim=100,
           mem_agg_hidden_dim=50, classifier_dim=50, test_batch_limit=int(4e9))
tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=logging.getLogger('t'))
model = tr._model_list[0]
dev = torch.device('cuda:0')
b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(B, N, 3, m=int(round(ALPHA * N)), seed=0)), dev)
gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
for rep in range(3):
    torch.manual_seed(7)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        st = model.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1)
        pred, _ = model(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                        is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination, batch_replication=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    solved, unsat = tr._cnf_evaluator(pred[0], gm, bvm, bfm, ef, None, sat_problem=model._last_problem)
    print('%s: forward %.1f ms, %d iterations (%s) = %.3f ms per iteration; solved %s' % (
        MODEL, dt * 1e3, model.last_run['iterations'], model.last_run['path'], dt * 1e3 / max(model.last_run['iterations'], 1),
        '%d of %d, %d unsatisfied clauses' % (int(solved.sum().item()), B, int(unsat.sum().item()))))
