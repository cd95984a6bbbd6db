"""Times the GRU cell entry point alone on the config-2 graph (5 000 x n=200 m=840) for a hidden width (default 150, the reference's shipped
np-nd-np predict config) and checks the specialised kernel against the generic tile kernel (PDP_NEURAL_GENERIC) bit for bit at full size.
Usage: python tools/gru_time.py [hidden [reps]]; PDP_HIP_LIB selects another build of the library for same-box A/B runs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pdp-solver_amd'))
from pdp.factorgraph import dataset
from pdp import native
dev = torch.device('cuda:0')
H = int(sys.argv[1]) if len(sys.argv) > 1 else 150
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tb = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(5000, 200, 3, m=840, seed=0)), dev)
p = native.Problem(tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'])
E = p.E
g = torch.Generator(device='cpu'); g.manual_seed(1)
r = lambda *s: (torch.randn(*s, generator=g) * 0.2).to(dev)
gw = native.GruWeights(r(3 * H, H + 1), r(3 * H, H), r(3 * H), r(3 * H))
torch.manual_seed(5)
state = torch.randn(E, H, device=dev) * 0.5
h = torch.randn(E, H, device=dev) * 0.5
am = (torch.rand(p.B, device=dev) > 0.1).to(torch.uint8)
out = p.neural_gru(gw, state, h, am); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for i in range(reps):
    out = p.neural_gru(gw, state, h, am); ev[i + 1].record()
torch.cuda.synchronize()
ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2]
flop = 2.0 * E * 3 * H * (2 * H + 1)
print('%s: gru hidden %d, E = %d: %.2f ms per call = %.1f TFLOP/s' % (os.path.basename(os.environ.get('PDP_HIP_LIB', 'product')), H, E, ms, flop / ms * 1e-9), flush=True)
if len(sys.argv) > 3:
    os.environ['PDP_NEURAL_GENERIC'] = '1'
    ref = p.neural_gru(gw, state, h, am); torch.cuda.synchronize()
    print('equal to the generic tile kernel at full size:', bool(torch.equal(out, ref)), ' NaN:', int(torch.isnan(out).sum()))
