#!/usr/bin/env python3
"""How many [E, H] state tensors are alive during a step-wise neural forward?  Prints torch's allocated bytes / (E * H * 4) at the
entry and exit of every plug-in call of a T-sweep np-nd-np forward (B instances of n = 400)."""
import logging, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import torch
from pdp.factorgraph import dataset
from pdp.trainer import SatFactorGraphTrainer
from pdp.nn.solver import OwnedState
B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H = 128
dev = torch.device('cuda:0')
cfg = dict(model_type='np-nd-np', model_name='probe', verbose=False, dropout=0, error_dim=1, exploration=0, hidden_dim=H, local_search_iteration=0, epsilon=0.5,
           tolerance=0.02, t_max=100, edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1, mem_hidden_dim=100, agg_hidden_dim=100, mem_agg_hidden_dim=50,
           classifier_dim=50, loss_sharpness=5, rng='philox', random_seed=0, test_recurrence_num=T, batch_size=B, test_batch_limit=10 ** 12)
tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=logging.getLogger('probe'))
m = tr._model_list[0]
b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(B, 400, 3, m=1680, seed=1)), dev)
gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
unit = gm.size(1) * H * 4
base = torch.cuda.memory_allocated()
say = lambda tag: print('%-28s %5.2f state tensors (peak %5.2f)' % (tag, (torch.cuda.memory_allocated() - base) / unit, (torch.cuda.max_memory_allocated() - base) / unit))
for name, mod in (('propagator', m._propagator), ('decimator', m._decimator), ('predictor', m._predictor)):
    mod.register_forward_pre_hook(lambda mod_, inp, name=name: say(name + ' in'))
    mod.register_forward_hook(lambda mod_, inp, out, name=name: say(name + ' out'))
for rep in range(2):
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        m.forward(init_state=OwnedState(m.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1)), graph_map=gm, batch_variable_map=bvm,
                  batch_function_map=bfm, edge_feature=ef, meta_data=None, is_training=False, iteration_num=T,
                  check_termination=tr._check_recurrence_termination, batch_replication=1)
    say('after forward %d' % rep)
