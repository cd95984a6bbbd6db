#!/usr/bin/env python3
"""One big instance through the whole p-d-p forward of the API (set-up, T sweeps, random fill, w Walk-SAT steps): to be run under
rocprofv3 --kernel-trace --stats to see which kernels a single big instance spends its time in.
usage: python tools/big_forward_profile.py [n] [T] [w]"""
import os, sys, time, logging
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import torch
from pdp.factorgraph import dataset
from pdp.trainer import SatFactorGraphTrainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
w = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device('cuda:0')
tb = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(1, n, 3, m=int(3.5 * n), seed=11)), dev)
tr = SatFactorGraphTrainer(dict(model_type='p-d-p', model_name='t', verbose=False, local_search_iteration=w, epsilon=0.5, tolerance=0.02, t_max=100,
                                rng='philox', random_seed=3, hidden_dim=3, persistent=True, test_batch_limit=1 << 62, batch_size=5000,
                                test_recurrence_num=1), use_cuda=True, logger=logging.getLogger('t'))
m = tr._model_list[0]
for rep in range(2):
    with torch.no_grad():
        st = m.get_init_state(tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'], None, randomized=False, batch_replication=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pred, _ = m(init_state=st, graph_map=tb['graph_map'], batch_variable_map=tb['batch_variable_map'], batch_function_map=tb['batch_function_map'],
                    edge_feature=tb['edge_feature'], meta_data=None, is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination,
                    batch_replication=1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('n=%d: forward with %d sweeps + %d Walk-SAT steps: %.1f ms  %s' % (n, T, w, 1e3 * dt, m.last_run))
