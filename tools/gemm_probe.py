#!/usr/bin/env python3
"""The training GEMM shapes at 1 M rows: the library's k_gemm (pdp_train_linear / _backward) against torch.mm (rocBLAS / hipBLASLt fp32)."""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
import torch
from pdp import native
from pdp.nn import train_ops as T
dev = torch.device('cuda:0')
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
def tm(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for K, N in ((129, 100), (100, 50), (51, 100), (100, 128), (129, 384), (128, 384)):
    x = torch.randn(R, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
    dy = torch.randn(R, N, device=dev)
    with torch.no_grad():
        t_nat = tm(lambda: T.LinearAct.apply(x, w, b, 'none'))
        t_lib = tm(lambda: torch.addmm(b, x, w.t()))
        t_dx_lib = tm(lambda: torch.mm(dy, w)); t_dw_lib = tm(lambda: torch.mm(dy.t(), x))
    xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True); bg = b.clone().requires_grad_(True)
    def bwd():
        y = T.LinearAct.apply(xg, wg, bg, 'none'); y.backward(dy)
    t_fb = tm(bwd, 3)
    fl = 2.0 * R * K * N
    print("K=%3d N=%3d  forward: native %.3f ms (%.1f TF)  lib %.3f ms (%.1f TF) | native fwd+bwd %.3f ms; lib dX %.3f + dW %.3f ms"
          % (K, N, t_nat, fl / t_nat / 1e9, t_lib, fl / t_lib / 1e9, t_fb, t_dx_lib, t_dw_lib))
