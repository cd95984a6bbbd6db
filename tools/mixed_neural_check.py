"""configs[4]-shaped run on one GPU: mixed random k-SAT (k in {3,4,5}, n in [100,500]), batch_replication 4, dynamic batching, np-nd-np
(hidden 128, seeded random weights) + Walk-SAT.  Prints segment sizes, time per iteration and the result-row statistics.
Usage: python tools/mixed_neural_check.py [instances] [iterations] [model_type: np-nd-np | p-nd-np]"""
import sys, time, logging, io, json
import numpy as np, torch
sys.path.insert(0, '/root/repo/pdp-solver_amd')
from pdp.factorgraph import dataset
from pdp.trainer import SatFactorGraphTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 600
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
MODEL = sys.argv[3] if len(sys.argv) > 3 else 'np-nd-np'
rng = np.random.RandomState(0)
alpha = {3: 0.9 * 4.27, 4: 0.9 * 9.93, 5: 0.9 * 21.12}
items = []
for i in range(B):
    k = int(rng.choice([3, 4, 5])); n = int(rng.randint(100, 501))
    items += dataset.random_ksat_items(1, n, k, m=int(round(alpha[k] * n)), seed=1000 + i)
edges = [it[2].shape[1] for it in items]
print('instances %d, edges %d (min %d max %d per instance)' % (B, sum(edges), min(edges), max(edges)))
cfg = dict(model_type=MODEL, model_name='mixed', verbose=False, local_search_iteration=100, epsilon=0.5, rng='philox', random_seed=1,
           hidden_dim=128, edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1, mem_hidden_dim=100, agg_hidden_dim=100,
           mem_agg_hidden_dim=50, classifier_dim=50, test_batch_limit=int(4e9), batch_size=B, test_recurrence_num=T, tolerance=0.02, t_max=100)
torch.manual_seed(1234)
tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=logging.getLogger('t'))
segs = dataset.divide(edges, cfg['test_batch_limit'] // 4, 128)
print('dynamic batching (limit / replication): %d segments, sizes %s' % (len(segs), [len(s) for s in segs][:12]))
dev = torch.device('cuda:0')
model = tr._model_list[0]
rows = 0; solved = 0; t_all = 0.0
for seg in segs:
    b = dataset.to_torch(dataset.collate_segment([items[j] for j in seg]), dev)
    gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        st = model.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=4)
        pred, _ = model(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                        is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination, batch_replication=4)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0; t_all += dt
    out = tr._post_process_predictions(model, pred, gm, bvm, bfm, ef, None, b.get('label'), b.get('misc_data')) if hasattr(tr, '_post_process_predictions') else None
    print('segment of %d instances x4 replicas, %d edges: %.2f s, %d iterations (%s), prediction %s' % (
        len(seg), 4 * gm.size(1), dt, model.last_run['iterations'], model.last_run['path'], tuple(pred[0].shape)))
print('total %.2f s' % t_all)
