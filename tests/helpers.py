"""Shared helpers for the test-suite."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, 'pdp-solver_amd'), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

from pdp import generator  # noqa: E402
from pdp.factorgraph import dataset  # noqa: E402


def random_batch(batch, n, k=3, m=None, seed=0, mixed=False):
    """numpy batch dict of uniform random k-SAT instances (mixed: varying n / k incl. unit clauses)."""
    items = []
    for i in range(batch):
        rng = np.random.RandomState(seed + i)
        if mixed:
            ni = int(rng.randint(max(4, n // 2), n + 1))
            mi = int(round(rng.uniform(2.5, 4.5) * ni))
            clauses = []
            for _ in range(mi):
                kk = int(rng.choice([1, 2, 3, 3, 3, 4, 5]))
                vs = rng.choice(ni, size=min(kk, ni), replace=False) + 1
                sg = rng.randint(0, 2, size=len(vs)) * 2 - 1
                clauses.append([int(a * b) for a, b in zip(vs, sg)])
            items.append(dataset.instance_from_clauses(ni, clauses, label=-1, name="m%d" % i))
        else:
            mm = m if m is not None else generator.clause_count(n, k)
            items.append(dataset.instance_from_clauses(n, generator.uniform_ksat(n, mm, k, rng), label=-1, name="u%d" % i))
    return dataset.collate_segment(items)


def load_golden(name):
    return np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz'))
