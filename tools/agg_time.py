"""Times the edge aggregator entry point (pre-transform, row sums, post-transform) on the config-2 graph with the library's per-kernel HIP events.
Usage: python tools/agg_time.py [hidden [reps]]; PDP_HIP_LIB selects another build of the library for same-box A/B runs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pdp-solver_amd'))
from pdp.factorgraph import dataset
from pdp import native
dev = torch.device('cuda:0')
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tb = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(5000, 200, 3, m=840, seed=0)), dev)
p = native.Problem(tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'])
E = p.E
g = torch.Generator(device='cpu'); g.manual_seed(1)
r = lambda *s: (torch.randn(*s, generator=g) * 0.2).to(dev)
aw = native.AggregatorWeights(r(100, H + 1), r(100), r(50, 100), r(100, 51), r(100), r(H, 100), 1)
torch.manual_seed(5)
state = torch.randn(E, H, device=dev) * 0.5
old = torch.randn(E, H, device=dev) * 0.5
am = (torch.rand(p.B, device=dev) > 0.1).to(torch.uint8)
p.refresh_edge_mask()
em = p.edge_mask if os.environ.get('AGG_EDGE_MASK', '1') != '0' else None      # the solver always passes the edge mask
for by_var in (True, False):
    out = p.neural_aggregate_edges(aw, by_var, state, em, am, old); torch.cuda.synchronize()
    native.kernel_timing(True)
    for i in range(reps):
        out = p.neural_aggregate_edges(aw, by_var, state, em, am, old)
    torch.cuda.synchronize()
    tm = native.kernel_timing_read()
    native.kernel_timing(False)
    print('%s hidden %d by %s: %s  checksum %.9e' % (os.path.basename(os.environ.get('PDP_HIP_LIB', 'product')), H, 'variable' if by_var else 'clause',
          '  '.join('%s %.2f ms' % (k, v[0] / max(v[1], 1)) for k, v in tm.items() if v[1]), float(out.double().sum())), flush=True)
