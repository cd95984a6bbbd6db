"""Times the neural operator entry points alone (GRU cell, edge aggregator) on config 3's graph (5 000 x n=200 m=840, hidden 128).
Usage: python tools/neural_ops_time.py [reps [n [hidden]]] (m = 4.2 n); PDP_HIP_LIB selects another build of the library for same-box A/B runs."""
import sys, torch
sys.path.insert(0, '/root/repo/pdp-solver_amd')
from pdp.factorgraph import dataset
from pdp import native
dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tb = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(5000, nv, 3, m=int(round(4.2 * nv)), seed=0)), dev)
p = native.Problem(tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'])
E, H = p.E, (int(sys.argv[3]) if len(sys.argv) > 3 else 128)
g = torch.Generator(device='cpu'); g.manual_seed(1)
r = lambda *s: (torch.randn(*s, generator=g) * 0.2).to(dev)
gw = native.GruWeights(r(3 * H, H + 1), r(3 * H, H), r(3 * H), r(3 * H))
aw = native.AggregatorWeights(r(100, H + 1), r(100), r(50, 100), r(100, 51), r(100), r(H, 100), 1)
state = torch.randn(E, H, device=dev) * 0.5
h = torch.randn(E, H, device=dev) * 0.5
am = torch.ones(p.B, dtype=torch.uint8, device=dev)
def timed(name, fn, flop):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        out = fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2]
    print('%-22s %8.2f ms  %6.1f TFLOP/s  checksum %.6e' % (name, ms, flop / ms * 1e-9, float(out.double().sum())))
timed('gru', lambda: p.neural_gru(gw, state, h, am), 2.0 * E * 384 * 257)
timed('aggregate(by var)', lambda: p.neural_aggregate_edges(aw, True, state, None, am, h), 2.0 * E * (129 * 100 + 100 * 50 + 51 * 100 + 100 * 128))
timed('aggregate(by clause)', lambda: p.neural_aggregate_edges(aw, False, state, None, am, h), 2.0 * E * (129 * 100 + 100 * 50 + 51 * 100 + 100 * 128))
