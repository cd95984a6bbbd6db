#!/usr/bin/env python3
"""Throughput of the native DIMACS reader (pdp_dimacs_open, host code) against the pure-Python statement of the same rules."""
import os, sys, time, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import numpy as np
from pdp import native, generator
import dimacs2json

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
m = int(4.2 * n)
variables, signs = generator.uniform_ksat_arrays(n, m, 3, np.random.RandomState(0))
lit = (variables + 1) * signs
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, 'big.cnf')
    with open(path, 'w') as f:
        f.write("p cnf %d %d\n" % (n, m))
        f.write("\n".join("%d %d %d 0" % tuple(r) for r in lit.tolist()) + "\n")
    size = os.path.getsize(path)
    t0 = time.perf_counter(); vn, cn, sv, ci = native.dimacs_parse(path); t1 = time.perf_counter()
    t2 = time.perf_counter(); nn, clauses = dimacs2json.parse_dimacs(path); pv = generator.compact_instance(nn, clauses); t3 = time.perf_counter()
    assert (vn, cn) == (pv[0], pv[1]) and np.array_equal(sv, pv[2]) and np.array_equal(ci, pv[3])
    print("file %.1f MB, %d clauses: native %.3f s (%.0f MB/s), python %.3f s (%.1f MB/s), x%.0f" %
          (size / 1e6, cn, t1 - t0, size / 1e6 / (t1 - t0), t3 - t2, size / 1e6 / (t3 - t2), (t3 - t2) / (t1 - t0)))
