"""Randomised check of the persistent Walk-SAT on replicated batches that contain instances past the LDS limit (configs[4]'s situation) against the
CPU oracle: random replication factor, mix of planted (quickly solved: the global stop comes early and replicas are truncated) and uniform
instances, one to three big ones, random step budget.  Test infrastructure (uses oracle/).  usage: python tools/walksat_replica_soak.py [seconds] [seed]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd')); sys.path.insert(0, REPO)
import numpy as np, torch
from pdp import native, generator
from pdp.factorgraph import dataset
from oracle import binding
binding.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device('cuda:0')
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def planted(n, m, k):
    cl = []
    for _ in range(m):
        vs = rng.choice(n, size=k, replace=False) + 1
        sg = rng.randint(0, 2, size=k) * 2 - 1
        sg[rng.randint(k)] = 1
        cl.append([int(a * b) for a, b in zip(vs, sg)])
    return dataset.instance_from_clauses(n, cl, label=1, name='p')


def uniform(n, m, k):
    return dataset.instance_from_clauses(n, generator.uniform_ksat(n, m, k, np.random.RandomState(rng.randint(1 << 30))), label=-1, name='u')


runs = early = 0
t_end = time.time() + budget
while time.time() < t_end:
    R = int(rng.choice([2, 3, 4]))
    easy = rng.rand() < 0.6
    make = planted if easy else uniform
    items = [make(int(rng.randint(15, 60)), int(rng.randint(30, 150)), 3) for _ in range(int(rng.randint(3, 20)))]
    for _ in range(int(rng.randint(1, 4))):
        n = int(rng.randint(2600, 3600))
        items.insert(int(rng.randint(0, len(items) + 1)), make(n, int(rng.uniform(2.5, 3.8) * n), 3))
    if rng.rand() < 0.3:
        os.environ['PDP_WALKSAT_NO_TEAM'] = '1'
    else:
        os.environ.pop('PDP_WALKSAT_NO_TEAM', None)
    b = dataset.collate_segment(items)
    hp = native.Problem(t(b['graph_map']), t(b['batch_variable_map']), t(b['batch_function_map']), t(b['edge_feature']), replication=R)
    op = binding.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'], R)
    hp.simplify(); op.simplify()
    if easy:
        pred = np.ones(op.V, np.float32); pred[rng.choice(op.V, size=max(1, op.V // int(rng.choice([100, 200, 400]))), replace=False)] = 0.0
    else:
        pred = (rng.rand(op.V) > 0.5).astype(np.float32)
    w = int(rng.choice([20, 60, 150, 300]))
    seed = int(rng.randint(1 << 30))
    hout, hsteps = hp.local_search(t(pred), w, 0.5, seed=seed)
    oout, osteps, _ = op.local_search(pred, w, 0.5, seed=seed)
    assert hsteps == osteps, (runs, hsteps, osteps)
    assert np.array_equal(hout.cpu().numpy()[:, 0], oout), ('assignment differs', runs, R, easy, w)
    runs += 1; early += 1 if osteps < w else 0
print('replicated Walk-SAT with big instances: %d runs equal the oracle (%d of them stopped before their step budget: replica truncation exercised)' % (runs, early))
