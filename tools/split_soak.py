#!/usr/bin/env python3
"""Soak of the forwards that are spread over several processes (pdp/parallel.py), on ONE GPU and in ONE process: the parts of a batch run in
threads of their own (own model, own stream), the exchange of the coupled form (--split-forward) is a barrier + an in-memory reduction, and
the concatenated predictions of the parts must equal the prediction of the batch solved whole -- for the coupled form against the strict
forward, for --isolated against the isolated one.  Random batches (the golden NaN-producing instances mixed in at random places, random
sweep counts, with and without Walk-SAT).  usage: python tools/split_soak.py [seconds] [seed]"""
import logging, os, sys, threading, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import torch
from pdp import native, parallel
from pdp.factorgraph import dataset
from pdp.trainer import SatFactorGraphTrainer

LOG = logging.getLogger('soak')
DEV = torch.device('cuda:0')
POISON = None


def cfg(w, isolated):
    return dict(model_type='p-d-p', model_name='soak', verbose=False, local_search_iteration=w, epsilon=0.5, tolerance=0.02, t_max=100, pi=0.01,
                decimation_probability=0.5, rng='philox', random_seed=0, hidden_dim=3, test_batch_limit=40000000, batch_size=5000, test_recurrence_num=1,
                isolated=isolated)


def forward(items, T, w, key, isolated, base=(0, 0), exchange=None):
    tr = SatFactorGraphTrainer(cfg(w, isolated), use_cuda=True, logger=LOG)
    m = tr._model_list[0]
    b = dataset.to_torch(dataset.collate_segment(items), DEV)
    m.set_random_key(key, *base)
    m._exchange = exchange
    with torch.no_grad():
        st = m.get_init_state(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'], None, randomized=False, batch_replication=1)
        pred, _ = m(init_state=st, graph_map=b['graph_map'], batch_variable_map=b['batch_variable_map'], batch_function_map=b['batch_function_map'],
                    edge_feature=b['edge_feature'], meta_data=None, is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination,
                    batch_replication=1)
    return pred[0].reshape(-1).cpu().numpy(), m.last_run['path']


class Exchange(object):
    "min / max / or over the parts: two barrier phases per call (publish, then read before anyone publishes again)"
    def __init__(self, n):
        self.n, self.slots, self.barrier, self.calls = n, [None] * n, threading.Barrier(n), 0

    def part(self, r):
        def ex(mins, maxs, ors):
            self.slots[r] = (mins.copy(), maxs.copy(), ors.copy())
            self.barrier.wait()
            every = list(self.slots)
            self.barrier.wait()
            if mins.size: mins[:] = np.min([e[0] for e in every], axis=0)
            if maxs.size: maxs[:] = np.max([e[1] for e in every], axis=0)
            if ors.size: ors[:] = np.bitwise_or.reduce([e[2] for e in every], axis=0)
            if r == 0: self.calls += 1
        return ex


def parts_forward(items, nparts, T, w, key, coupled):
    edges = [it[2].shape[1] for it in items]
    offs = np.concatenate(([0], np.cumsum([it[0] for it in items])))
    bounds = parallel.shard_bounds(edges, nparts)
    xch = Exchange(nparts) if coupled else None
    out, err = [None] * nparts, [None] * nparts

    def worker(r):
        lo, hi = bounds[r]
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                out[r] = forward(items[lo:hi], T, w, key, not coupled, (int(offs[lo]), lo), xch.part(r) if coupled else None)[0]
                torch.cuda.current_stream().synchronize()
        except native.CoupledForwardFailed as e:
            err[r] = e
        except BaseException as e:           # noqa
            err[r] = e
            if xch is not None:
                xch.barrier.abort()
    th = [threading.Thread(target=worker, args=(r,)) for r in range(nparts)]
    for t in th: t.start()
    for t in th: t.join()
    if any(isinstance(e, native.CoupledForwardFailed) for e in err):
        assert all(isinstance(e, native.CoupledForwardFailed) for e in err), err        # every part agrees on the outcome
        return None, xch.calls
    for e in err:
        if e is not None:
            raise e
    return np.concatenate(out), (xch.calls if xch else 0)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    gold = np.load(os.path.join(REPO, 'tests', 'golden', 'headline_n200_poison.npz'))
    n, mcl = int(gold['meta'][0]), int(gold['meta'][1])
    t0 = time.time()
    runs = dict(coupled=0, isolated=0, refused=0, poisoned=0, exchanges=0)
    while time.time() - t0 < seconds:
        B = int(rng.randint(24, 90))
        items = dataset.random_ksat_items(B, n, 3, m=mcl, seed=int(rng.randint(1 << 30)))
        n_poison = int(rng.randint(0, 4))
        for sd in rng.choice(gold['seeds'][:4], size=n_poison, replace=False):
            items.insert(int(rng.randint(0, len(items) + 1)), dataset.random_ksat_items(1, n, 3, m=mcl, seed=int(sd))[0])
        T = int(rng.choice([30, 60, 100, 130])); w = int(rng.choice([0, 25])); nparts = int(rng.choice([2, 3, 4]))
        key = parallel.batch_seed(int(rng.randint(1 << 30)), int(rng.randint(5)), int(rng.randint(3)))
        coupled = bool(rng.randint(2))
        whole, path = forward(items, T, w, key, isolated=not coupled)
        got, calls = parts_forward(items, nparts, T, w, key, coupled)
        if got is None:
            runs['refused'] += 1
            continue
        if not np.array_equal(got, whole):
            bad = np.nonzero(got != whole)[0]
            print('MISMATCH coupled=%s B=%d parts=%d T=%d w=%d poison=%d path=%s: %d of %d variables differ (first %d)' % (coupled, len(items), nparts, T, w, n_poison, path, bad.size, whole.size, bad[0]))
            sys.exit(1)
        runs['coupled' if coupled else 'isolated'] += 1; runs['poisoned'] += 1 if n_poison else 0; runs['exchanges'] += calls
    print('split soak: %(coupled)d coupled and %(isolated)d isolated batches in 2-4 parts equal the batch solved whole (%(poisoned)d with NaN-producing instances, '
          '%(exchanges)d exchanges); %(refused)d coupled batches failed their speculation on every part alike' % runs)


if __name__ == '__main__':
    main()
