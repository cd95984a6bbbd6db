import sys, time, logging, torch
sys.path.insert(0, '/root/repo/pdp-solver_amd')
from pdp.factorgraph import dataset
from pdp.trainer import SatFactorGraphTrainer
dev = torch.device('cuda:0')
tb = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(5000, 200, 3, m=840, seed=0)), dev)
for persistent in (True, False):
    tr = SatFactorGraphTrainer(dict(model_type='p-d-p', model_name='t', verbose=False, local_search_iteration=0, epsilon=0.5, tolerance=0.02, t_max=100,
                                    rng='philox', random_seed=3, hidden_dim=3, persistent=persistent, test_batch_limit=40000000, batch_size=5000,
                                    test_recurrence_num=1), use_cuda=True, logger=logging.getLogger('t'))
    m = tr._model_list[0]
    for rep in range(2):
        with torch.no_grad():
            st = m.get_init_state(tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'], None, randomized=False, batch_replication=1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pred, _ = m(init_state=st, graph_map=tb['graph_map'], batch_variable_map=tb['batch_variable_map'], batch_function_map=tb['batch_function_map'],
                        edge_feature=tb['edge_feature'], meta_data=None, is_training=False, iteration_num=100, check_termination=tr._check_recurrence_termination, batch_replication=1)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('persistent' if persistent else 'step-wise', m.last_run, '%.1f ms per forward (100 iterations, includes problem set-up)' % (1e3 * dt))
