#!/usr/bin/env python3
"""End-to-end timing of the command line on a synthetic DIMACS directory (host side included): where does the wall time go?"""
import cProfile, io, os, pstats, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import numpy as np
from pdp import generator
import satyr

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
with tempfile.TemporaryDirectory() as d:
    t0 = time.perf_counter()
    for i in range(N):
        variables, signs = generator.uniform_ksat_arrays(200, 840, 3, np.random.RandomState(i))
        lit = (variables + 1) * signs
        with open(os.path.join(d, 'inst_%05d_1.cnf' % i), 'w') as f:
            f.write("p cnf 200 840\n" + "\n".join("%d %d %d 0" % tuple(r) for r in lit.tolist()) + "\n")
    print("wrote %d DIMACS files in %.1f s" % (N, time.perf_counter() - t0))
    out = os.path.join(d, 'out.jsonl')
    argv = [os.path.join(REPO, 'config', 'Predict', 'PDP-p-d-p-walksat-pytorch.yaml'), d, '100', '-d', '-z', '5000', '-s', '7', '--rng', 'philox', '-o', out]
    satyr.main(argv)                      # warm-up (library load, allocator cache)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable(); satyr.main(argv); pr.disable()
    dt = time.perf_counter() - t0
    rows = sum(1 for _ in open(out))
    print("satyr.py -d on %d instances: %.2f s wall (%.0f instances/s), %d result rows" % (N, dt, N / dt, rows))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3500])
