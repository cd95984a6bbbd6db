#!/usr/bin/env python3
"""Fast build: the hidden-128 GRU on three-term bf16 products (k_gru_bf3) against the same build's fp32 chain (PDP_GRU_NO_BF16X3=1 in a
child process) -- largest deviation relative to the largest |h'|, and the time of a call.  python tools/bf3_check.py [instances]"""
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
DX = int(sys.argv[3])
cell = torch.nn.GRUCell(DX + 1, 128).to(dev)
w = native.GruWeights(cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
state = torch.randn(E, DX, device=dev) * 0.7
h = torch.randn(E, 128, device=dev) * 0.5
am = torch.ones(prob.B, dtype=torch.uint8, device=dev); am[::7] = 0
out = prob.neural_gru(w, state, h, am)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = prob.neural_gru(w, state, h, am)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
np.save(sys.argv[2], np.concatenate([out[:100000].cpu().numpy(), out[-100000:].cpu().numpy()]))
print(json.dumps(dict(kernel=native.kernel_name('gru'), ms=ms, E=E, lib=native.LIB_PATH)))
''' % REPO
B = sys.argv[1] if len(sys.argv) > 1 else '5000'
for DX in ('128', '3', '2'):
  res = {}
  for name, env in (('bf16x3', {}), ('f32', {'PDP_GRU_NO_BF16X3': '1'})):
    r = subprocess.run([sys.executable, '-c', CHILD, B, '/tmp/bf3_%s.npy' % name, DX], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    if r.returncode != 0:
        print(r.stderr[-3000:]); sys.exit(1)
    res[name] = json.loads(r.stdout.strip().split('\n')[-1]); print('dx', DX, name, res[name])
  import numpy as np
  a, b = np.load('/tmp/bf3_bf16x3.npy'), np.load('/tmp/bf3_f32.npy')
  print('max |diff| = %.3e, max |h| = %.3f, relative %.3e, nan %d / %d' % (np.abs(a - b).max(), np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max(), int(np.isnan(a).sum()), int(np.isnan(b).sum())))
