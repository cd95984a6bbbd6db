#!/usr/bin/env python3
"""Time of pdp_neural_predict's head kernel on the headline graph (hidden 128), specialised against generic: python tools/predict_time.py"""
import os, sys, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, os.path.join(%r, 'pdp-solver_amd'))
import torch
from pdp import native
from pdp.factorgraph import dataset
dev = torch.device('cuda:0')
B, H = 5000, 128
b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(B, 200, 3, m=840, seed=0)), dev)
prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'], batch_size=B)
g = torch.Generator(device='cpu').manual_seed(1)
r = lambda *sh: (torch.randn(*sh, generator=g) * 0.3).to(dev)
w = native.AggregatorWeights(r(100, H + 1), r(100), r(50, 100), r(100, 50), r(100), r(H, 100), 0)
hw = native.HeadWeights(r(50, H), r(50), r(1, 50), 'sigmoid')
state = r(prob.E, H)
outs = []
for rep in range(4):
    out = prob.neural_predict(w, hw, state, None)
torch.cuda.synchronize()
t = native.kernel_times() if hasattr(native, 'kernel_times') else {}
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for rep in range(10): out = prob.neural_predict(w, hw, state, None)
e1.record(); torch.cuda.synchronize()
print(json.dumps(dict(predict_call_ms=e0.elapsed_time(e1) / 10, checksum=float(out.double().sum()), kernel=native.kernel_name('predict_head') if hasattr(native, 'kernel_name') else '')))
''' % REPO
for env in ({}, {'PDP_PREDICT_GENERIC': '1'}):
    out = subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    print(env, out.stdout.strip().split('\n')[-1] if out.stdout.strip() else out.stderr[-800:], flush=True)
