#!/usr/bin/env python3
"""Fast build: one MessageAggregator call (hidden 128, inner widths 100 / 50) with its halves on three-term bf16 products against the
same build's fp32 chains (PDP_AGG_NO_BF16X3=1 in a child process): largest deviation relative to the largest |out|, library kernel times."""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, time
sys.path.insert(0, os.path.join(%r, 'pdp-solver_amd'))
os.environ['PDP_BUILD'] = 'fast'
import numpy as np, torch
from pdp import native
from pdp.factorgraph import dataset
B = int(sys.argv[1])
dev = torch.device('cuda:0')
b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(B, 200, 3, m=840, seed=0)), dev)
prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
E = prob.E
torch.manual_seed(5)
H = 128
lin = lambda i, o, bias=True: torch.nn.Linear(i, o, bias=bias).to(dev)
l1m, l2m, l1a, l2a = lin(H + 1, 100), lin(100, 50, False), lin(51, 100), lin(100, H, False)
w = native.AggregatorWeights(l1m.weight, l1m.bias, l2m.weight, l1a.weight, l1a.bias, l2a.weight, 1)
state = torch.randn(E, H, device=dev) * 0.7
old = torch.randn(E, H, device=dev) * 0.5
am = torch.ones(prob.B, dtype=torch.uint8, device=dev); am[::7] = 0
res = {}
for byv in (1, 0):
    out = prob.neural_aggregate_edges(w, byv, state, None, am, old)
    torch.cuda.synchronize()
    native.kernel_timing(True)
    for _ in range(3):
        out = prob.neural_aggregate_edges(w, byv, state, None, am, old)
    torch.cuda.synchronize()
    t = native.kernel_timing_read(); native.kernel_timing(False)
    res['by_variable' if byv else 'by_clause'] = {k: round(t[k][0] / max(1, t[k][1]), 3) for k in ('agg_pre', 'row_sum', 'agg_post')}
    np.save(sys.argv[2] + ('_v' if byv else '_c') + '.npy', np.concatenate([out[:100000].cpu().numpy(), out[-100000:].cpu().numpy()]))
print(json.dumps(dict(pre=native.kernel_name('agg_pre'), post=native.kernel_name('agg_post'), ms=res)))
''' % REPO
B = sys.argv[1] if len(sys.argv) > 1 else '5000'
for name, env in (('bf16x3', {}), ('f32', {'PDP_AGG_NO_BF16X3': '1'})):
    r = subprocess.run([sys.executable, '-c', CHILD, B, '/tmp/bf3agg_%s' % name], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    if r.returncode != 0:
        print(r.stderr[-3000:]); sys.exit(1)
    print(name, r.stdout.strip().split('\n')[-1])
import numpy as np
for d in ('_v', '_c'):
    a, b = np.load('/tmp/bf3agg_bf16x3%s.npy' % d), np.load('/tmp/bf3agg_f32%s.npy' % d)
    print(d, 'max |diff| = %.3e, max |out| = %.3f, relative %.3e, nan %d / %d' % (np.abs(a - b).max(), np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max(), int(np.isnan(a).sum()), int(np.isnan(b).sum())))
