#!/usr/bin/env python3
"""Randomised soak through the reference-shaped Python API: the persistent one-launch loop (whatever launch the library picks: LDS-resident,
per-instance routing, exact, lock-step, or the fail-over) against the strict step-wise loop, which is checked operator by operator against the
oracle (tests/test_hip_ops.py) and end to end against the reference's golden traces.  Random batch compositions, model types p-d-p / reinforce /
walk-sat, deterministic and random initial states (test mode), batch replication 1-3, with and without the Walk-SAT pass, rng='torch' (so both
forms must also leave the global generator at the same position).  Everything must be equal bit for bit, NaNs included.
usage: python tools/api_soak.py [seconds] [seed]"""
import logging, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import numpy as np, torch
from pdp import generator
from pdp.factorgraph import dataset
from pdp.trainer import SatFactorGraphTrainer
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device('cuda:0')
LOG = logging.getLogger('soak')
paths = {}
t_end = time.time() + budget
runs = 0


def inst(n, alpha, k=None):
    kk = int(k or rng.choice([3, 3, 3, 4, 5, 2]))
    return dataset.instance_from_clauses(n, generator.uniform_ksat(n, max(1, int(alpha * n)), kk, np.random.RandomState(rng.randint(1 << 30))), label=-1, name='s')


def forward(model_type, persistent, b, T, R, randomized, seed, kw):
    cfg = dict(model_type=model_type, model_name='soak', verbose=False, epsilon=0.5, rng='torch', random_seed=0, hidden_dim=3,
               test_batch_limit=40000000, batch_size=5000, test_recurrence_num=1, persistent=persistent)
    cfg.update(kw)
    tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=LOG)
    m = tr._model_list[0]
    gm, bvm, bfm, ef = [torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ('graph_map', 'batch_variable_map', 'batch_function_map', 'edge_feature')]
    torch.manual_seed(seed)
    with torch.no_grad():
        st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=randomized, batch_replication=R)
        pred, states = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                         is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination, batch_replication=R)
    nxt = torch.rand(3).numpy()
    flat = [pred[0]] + [x for s in states if s is not None for x in s if torch.is_tensor(x)]
    return [x.detach().cpu().numpy() for x in flat], dict(m.last_run), nxt


while time.time() < t_end:
    kind = rng.choice(['small', 'tiny', 'single', 'mixed'])
    alpha = float(rng.choice([3.0, 3.5, 3.8, 4.2, 4.4]))
    if kind == 'small':
        items = [inst(int(rng.randint(8, 90)), alpha) for _ in range(int(rng.randint(20, 200)))]
    elif kind == 'tiny':
        items = [inst(int(rng.randint(8, 70)), alpha) for _ in range(int(rng.randint(2, 12)))]
    elif kind == 'single':
        items = [inst(int(rng.choice([20, 60, 200, 900])), alpha, 3)]
    else:
        items = [inst(int(rng.randint(10, 80)), alpha) for _ in range(int(rng.randint(10, 40)))]
        items.insert(int(rng.randint(0, len(items) + 1)), inst(int(rng.randint(1500, 2500)), float(rng.choice([3.5, 3.8, 4.2])), 3))
    b = dataset.collate_segment(items)
    model_type = str(rng.choice(['p-d-p', 'p-d-p', 'reinforce', 'walk-sat']))
    T = int(rng.choice([1, 7, 13, 30, 60])); R = int(rng.choice([1, 1, 2, 3])); randomized = bool(rng.rand() < 0.5)
    kw = dict(tolerance=float(rng.choice([0.02, 0.05, 0.1])), t_max=float(rng.choice([4, 8, 100])), pi=float(rng.choice([0.0, 0.01, 0.1])),
              decimation_probability=float(rng.choice([0.3, 0.6, 1.0])), local_search_iteration=int(rng.choice([0, 0, 15])))
    if kind in ('single', 'mixed'):
        T = min(T, 30)
    seed = int(rng.randint(1 << 20))
    desc = '%s %s B=%d R=%d T=%d randinit=%s %s seed=%d' % (kind, model_type, len(items), R, T, randomized, kw, seed)
    a, ra, na = forward(model_type, True, b, T, R, randomized, seed, kw)
    s, rs, ns = forward(model_type, False, b, T, R, randomized, seed, kw)
    key = '%-9s %-6s R%s %-8s -> %s' % (model_type, kind, '>1' if R > 1 else '=1', 'randinit' if randomized else 'det', ra.get('path'))
    paths[key] = paths.get(key, 0) + 1
    ok = len(a) == len(s) and all(x.shape == y.shape and np.array_equal(x, y, equal_nan=True) for x, y in zip(a, s)) and np.array_equal(na, ns) \
        and ra.get('iterations') == rs.get('iterations')
    if not ok:
        bad = [i for i, (x, y) in enumerate(zip(a, s)) if x.shape != y.shape or not np.array_equal(x, y, equal_nan=True)]
        print('MISMATCH:', desc, 'persistent', ra, 'stepwise', rs, 'tensors', bad, 'generator equal', np.array_equal(na, ns))
        os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
        np.savez(os.path.join(REPO, 'gpurun_out', 'api_soak_fail.npz'), graph_map=b['graph_map'], batch_variable_map=b['batch_variable_map'],
                 batch_function_map=b['batch_function_map'], edge_feature=b['edge_feature'], desc=desc)
        sys.exit(1)
    runs += 1
print('api soak: %d runs, persistent loop == step-wise loop everywhere' % runs)
for k in sorted(paths):
    print('  %-70s %d' % (k, paths[k]))
