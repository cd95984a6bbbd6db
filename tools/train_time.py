"""Times `_train_batch` (base.py:149-182) on a ~1 M-edge batch at hidden 128 (3 outer recurrences, dropout 0.2, clipped Adam step) -- run it under
`rocprofv3 --kernel-trace --stats` to see which kernels the step is made of.  Usage: python tools/train_time.py [model_type] [instances] [n]"""
import sys, time, logging
import numpy as np, torch
import torch.optim as optim
sys.path.insert(0, '/root/repo/pdp-solver_amd')
from pdp.factorgraph import dataset
from pdp.trainer import SatFactorGraphTrainer
MT = sys.argv[1] if len(sys.argv) > 1 else 'np-nd-np'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 400
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device('cuda:0')
b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(B, N, 3, m=int(round(4.2 * N)), seed=555)), dev)
gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
label = torch.ones(B, 1, device=dev)
cfg = dict(model_type=MT, model_name='t', verbose=False, dropout=0.2, error_dim=3, exploration=0.1, hidden_dim=128, local_search_iteration=0, epsilon=0.5,
           tolerance=0.02, t_max=100, edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1, mem_hidden_dim=100, agg_hidden_dim=100,
           mem_agg_hidden_dim=50, classifier_dim=50, loss_sharpness=5, randomized=True, train_inner_recurrence_num=1, train_outer_recurrence_num=3,
           clip_norm=0.65, batch_size=B, rng='philox', random_seed=0, init_rng='device')
cfg['lambda'] = 0.9
torch.manual_seed(99)
tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=logging.getLogger('t'))
opt = optim.Adam(tr.get_parameter_list(), lr=1e-4, weight_decay=1e-10)
total = np.zeros(1, dtype=np.float32)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr._train_batch(total, opt, gm, bvm, bfm, ef, None, label)
    torch.cuda.synchronize()
    print('%s: _train_batch on %d edges: %.1f ms (loss sum so far %.4f)' % (MT, gm.size(1), 1e3 * (time.perf_counter() - t0), float(total[0])))
