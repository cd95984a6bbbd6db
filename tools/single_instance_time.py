#!/usr/bin/env python3
"""One CNF instance per batch (the "solve this file" use of the CLI): a whole forward of the p-d-p solver through the Python API on the
persistent path (exact single-instance mode of pdp_sp_solve: one launch, a workgroup team when the instance is big) against the strict
step-wise loop (the fallback such batches took before).
usage: python tools/single_instance_time.py [n ...]   (alpha = 3.5, T = 100)"""
import os, sys, time, logging
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import torch
from pdp.factorgraph import dataset
from pdp.trainer import SatFactorGraphTrainer
dev = torch.device('cuda:0')
T = 100
for n in [int(x) for x in sys.argv[1:]] or [200, 4000, 100000]:
    tb = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(1, n, 3, m=int(3.5 * n), seed=11)), dev)
    out = []
    for persistent in ((True,) if os.environ.get('PDP_SKIP_STEPWISE') else (True, False)):
        tr = SatFactorGraphTrainer(dict(model_type='p-d-p', model_name='t', verbose=False, local_search_iteration=0, epsilon=0.5, tolerance=0.02, t_max=100,
                                        rng='philox', random_seed=3, hidden_dim=3, persistent=persistent, test_batch_limit=40000000, batch_size=5000,
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
        out.append((dict(m.last_run), dt, pred[0].clone()))
    if len(out) == 1:
        print('n=%d (%d edges): %s %.2f ms per forward of %d sweeps' % (n, tb['graph_map'].shape[1], out[0][0], 1e3 * out[0][1], T)); continue
    same = bool((out[0][2] == out[1][2]).all().item())
    print('n=%d (%d edges): %s %.2f ms | %s %.2f ms per forward of %d sweeps (problem set-up included); same prediction: %s'
          % (n, tb['graph_map'].shape[1], out[0][0], 1e3 * out[0][1], out[1][0], 1e3 * out[1][1], T, same))
