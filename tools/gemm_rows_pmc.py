#!/usr/bin/env python3
"""One training GEMM shape in a loop (for rocprofv3 counter passes): python tools/gemm_rows_pmc.py K N [rows] [reps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import torch
from pdp.nn import train_ops as T
K, N = int(sys.argv[1]), int(sys.argv[2])
R = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device('cuda:0')
x = torch.randn(R, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
with torch.no_grad():
    for _ in range(reps):
        y = T.LinearAct.apply(x, w, b, 'none')
torch.cuda.synchronize()
print('done', float(y[0, 0]))
