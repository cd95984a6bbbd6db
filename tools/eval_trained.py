"""How the weights trained on n in [10, 40] (models/, tools/train_demo.py) do on larger instances: solved fraction of 2 000 uniform random 3-SAT
instances per (n, alpha), T sweeps of the network, with and without the Walk-SAT post-process (Philox numbers).
Usage: python tools/eval_trained.py [model_type] [T] [walksat steps]"""
import logging, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
from pdp.factorgraph import dataset
from pdp.trainer import SatFactorGraphTrainer
MT = sys.argv[1] if len(sys.argv) > 1 else 'np-nd-np'
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
dev = torch.device('cuda:0')
for w in (0, W):
    cfg = dict(model_type=MT, model_name='eval', verbose=False, dropout=0.0, error_dim=3, exploration=0.1, hidden_dim=128, local_search_iteration=w, epsilon=0.5,
               tolerance=0.02, t_max=100, edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1, mem_hidden_dim=100, agg_hidden_dim=100,
               mem_agg_hidden_dim=50, classifier_dim=50, rng='philox', random_seed=3, test_recurrence_num=T)
    for weights in ('random', 'trained'):
        torch.manual_seed(5)
        tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=logging.getLogger('e'))
        m = tr._model_list[0]
        if weights == 'trained':
            m.load_state_dict(torch.load(os.path.join(REPO, 'models', 'demo-%s-h128.pt' % MT), map_location=dev), strict=True)
        for n, alpha in ((40, 3.5), (100, 3.5), (200, 3.0), (200, 3.5), (200, 4.0), (200, 4.2)):
            b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(2000, n, 3, m=int(round(alpha * n)), seed=88_000_000 + 1000 * n)), dev)
            gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
            with torch.no_grad():
                st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1)
                pred, _ = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                            is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination, batch_replication=1)
                solved, unsat = tr._cnf_evaluator(pred[0], gm, bvm, bfm, ef, None, sat_problem=m._last_problem)
            print('%s %-7s weights, T=%d, Walk-SAT %4d: n=%3d alpha=%.1f  solved %.3f  unsatisfied clauses %6d  (sweeps run %d)'
                  % (MT, weights, T, w, n, alpha, float(solved.mean().item()), int(unsat.sum().item()), m.last_run['iterations']), flush=True)
