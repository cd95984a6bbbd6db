"""End-to-end training demonstration on the MI355X ("so that the neural configs have non-random weights"): trains a neural PDP
solver from its random initialisation on freshly generated uniform random k-SAT with the reference's unsupervised energy loss
(`_train_batch`, base.py:149-182; generator as `satyr-train-test.py -g`) and reports, every few steps, the fraction of a fixed held-out
set that the model solves (T sweeps, no Walk-SAT, no random fill: the network's own prediction) -- before training that fraction is what
random weights give.  Saves the trained state dict next to the log.
Usage: python tools/train_demo.py [model_type np-nd-np|p-nd-np] [steps] [batch instances] [hidden] [out dir]"""
import json, logging, os, sys, time
import numpy as np, torch
import torch.optim as optim
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
from pdp.factorgraph import dataset
from pdp.trainer import SatFactorGraphTrainer

MT = sys.argv[1] if len(sys.argv) > 1 else 'np-nd-np'
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
H = int(sys.argv[4]) if len(sys.argv) > 4 else 128
OUT = sys.argv[5] if len(sys.argv) > 5 else os.path.join(REPO, 'gpurun_out', 'train_demo')
os.makedirs(OUT, exist_ok=True)
dev = torch.device('cuda:0')
cfg = dict(model_type=MT, model_name='demo-' + MT, verbose=False, dropout=0.2, error_dim=3, exploration=0.1, hidden_dim=H, local_search_iteration=0,
           epsilon=0.5, tolerance=0.02, t_max=100, edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1, mem_hidden_dim=100, agg_hidden_dim=100,
           mem_agg_hidden_dim=50, classifier_dim=50, loss_sharpness=5, randomized=True, train_inner_recurrence_num=1, train_outer_recurrence_num=10,
           clip_norm=0.65, batch_size=BATCH, rng='philox', random_seed=0, init_rng='device', dropout_rng='device', test_recurrence_num=30)
cfg['lambda'] = 1.0
np.random.seed(2026); torch.manual_seed(2026)
tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=logging.getLogger('demo'))
model = tr._model_list[0]
opt = optim.Adam(tr.get_parameter_list(), lr=3e-4, weight_decay=1e-10)
rng = np.random.RandomState(2026)
_next_seed = [10_000_000]


def draw(count):
    "uniform random 3-SAT, n ~ U{10..40}, alpha ~ U[2, 4): mostly satisfiable (the distribution of `satyr-train-test.py -g` with the uniform generator, drawn by the vectorised generator)"
    items = []
    for _ in range(count):
        n = int(rng.randint(10, 41)); m = max(1, int(rng.uniform(2.0, 4.0) * n))
        items += dataset.random_ksat_items(1, n, 3, m=m, seed=_next_seed[0]); _next_seed[0] += 1
    return dataset.to_torch(dataset.collate_segment(items), dev)


held_out = draw(2000)


def solved_fraction(T=30):
    b = held_out
    gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
    with torch.no_grad():
        st = model.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1)
        pred, _ = model(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                        is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination, batch_replication=1)
        solved, unsat = tr._cnf_evaluator(pred[0], gm, bvm, bfm, ef, None, sat_problem=model._last_problem)
    return float(solved.mean().item()), float(unsat.sum().item())


log = []
t_train = 0.0
frac, unsat = solved_fraction()
print('%s hidden %d: step 0 (random weights): held-out solved %.3f, unsatisfied clauses %d' % (MT, H, frac, unsat), flush=True)
log.append(dict(step=0, solved=frac, unsat=unsat, loss=None, train_seconds=0.0))
total = np.zeros(1, dtype=np.float32)
for step in range(1, STEPS + 1):
    b = draw(BATCH)
    label = torch.ones(BATCH, 1, device=dev)
    total[:] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr._train_batch(total, opt, b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'], None, label)
    torch.cuda.synchronize(); t_train += time.perf_counter() - t0
    model._global_step.data += 1
    if step % 25 == 0 or step == STEPS:
        frac, unsat = solved_fraction()
        print('step %4d: loss %.4f per batch, %.1f ms per step (%d edges); held-out solved %.3f, unsatisfied clauses %d'
              % (step, float(total[0]), 1e3 * t_train / step, b['graph_map'].size(1), frac, unsat), flush=True)
        log.append(dict(step=step, solved=frac, unsat=unsat, loss=float(total[0]), train_seconds=t_train))
torch.save(model.state_dict(), os.path.join(OUT, 'demo-%s-h%d.pt' % (MT, H)))
json.dump(dict(model_type=MT, hidden=H, steps=STEPS, batch=BATCH, log=log), open(os.path.join(OUT, 'train_demo_%s.json' % MT), 'w'), indent=1)
