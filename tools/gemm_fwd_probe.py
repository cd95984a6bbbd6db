#!/usr/bin/env python3
"""forward time of the training GEMM shapes (pdp_train_linear, act none / logsigmoid) and the dX product: python tools/gemm_fwd_probe.py [rows]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import torch
from pdp.nn import train_ops as T
dev = torch.device('cuda:0')
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
def tm(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
out = []
for K, N in ((129, 100), (100, 50), (51, 100), (100, 128), (129, 384), (128, 384)):
    x = torch.randn(R, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
    with torch.no_grad():
        t0 = tm(lambda: T.LinearAct.apply(x, w, b, 'none')); t1 = tm(lambda: T.LinearAct.apply(x, w, b, 'logsigmoid'))
    fl = 2.0 * R * K * N
    out.append("%dx%d: %.3f ms %.0f TF | logsig %.3f ms %.0f TF" % (K, N, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9))
print(os.environ.get('PDP_HIP_LIB', 'default')[-14:], ' ; '.join(out))
