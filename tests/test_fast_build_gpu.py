"""The opt-in fast build (libpdp_hip_fast.so: device math on v_exp_f32 / v_log_f32 / v_rcp_f32, include/pdp_math.h PDP_FAST_MATH).

Its results are not the oracle's bits, so it is gated by what the REFERENCE holds only: every golden trace's integer trajectory and the CLI
rows equal, floats within the tolerances those tests already state (tests/test_api_forward.py, tests/test_foreign_plugin.py,
tests/test_train_gpu.py run unchanged with PDP_BUILD=fast), the accuracy of the device functions against float64, and the solved counts
of the headline batch equal to the parity build's.  The parity build stays the default and keeps the whole bit-exact suite."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import REPO

pytestmark = pytest.mark.gpu


def test_reference_held_fixtures_pass_on_the_fast_build():
    "the three test files whose expectations all come from the reference, in a process that loads the fast library"
    env = dict(os.environ, PDP_BUILD='fast')
    files = ['tests/test_api_forward.py', 'tests/test_foreign_plugin.py', 'tests/test_train_gpu.py']
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider'] + files, cwd=REPO, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-6000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout.split('\n')[-2]
    # ... and the process really ran on the fast library
    probe = subprocess.run([sys.executable, '-c', "import sys; sys.path.insert(0, 'pdp-solver_amd'); from pdp import native; native.lib(); print(native.BUILD, native.LIB_PATH)"],
                           cwd=REPO, env=env, stdout=subprocess.PIPE, universal_newlines=True, timeout=300)
    assert probe.stdout.split()[0] == 'fast' and probe.stdout.strip().endswith('libpdp_hip_fast.so')


def _ulp_err(got, want64):
    want32 = want64.astype(np.float32)
    ulp = np.maximum(np.abs(np.spacing(want32)).astype(np.float64), 1.4012984643e-45)
    return np.abs(got.astype(np.float64) - want64) / ulp


def test_device_math_of_the_fast_build_against_float64():
    """the transcendental-unit forms: a few ulp on the ranges the path uses, denormals handled explicitly (the reference relies on
    log(1e-40) = -92.1034 and exp(-92.1034) = 1e-40), special values as the parity forms"""
    from pdp import native
    prev = native.use_build('fast')
    try:
        dev = torch.device('cuda:0')
        rng = np.random.RandomState(3)
        f = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(dev)
        g = lambda name, a: native.math_apply(name, f(a)).cpu().numpy()
        x = np.concatenate([rng.uniform(-104, 30, 200000), rng.uniform(-3, 3, 200000), [-92.1034, -87.5, -103.9, 0.0, 30.0, -1e-30]]).astype(np.float32)
        e = g('exp_fin', x)
        assert _ulp_err(e, np.exp(x.astype(np.float64))).max() <= 4.0
        assert abs(float(g('exp_fin', [-92.1034])[0]) - 1e-40) <= 2e-45
        se = g('safe_exp', np.concatenate([x, [31.0, 1e30, np.inf]]).astype(np.float32))
        assert _ulp_err(se[:-3], np.exp(np.minimum(x.astype(np.float64), 30.0))).max() <= 4.0 and np.all(se[-3:] == se[-3]) and abs(se[-1] / np.exp(30.0) - 1) < 3e-7
        y = np.concatenate([10.0 ** rng.uniform(-44.5, 0.3, 300000), rng.uniform(0.5, 1.0, 100000), [1e-40, 1.4e-45, 1.0, 0.99999994]]).astype(np.float32)
        lg = g('safe_log_fin', y)
        want = np.log(np.maximum(y.astype(np.float64), 1e-40))
        # log2 m + e is exact to an ulp of the SUM: near x = 1 that is an absolute, not a relative, statement
        assert np.all(np.abs(lg.astype(np.float64) - want) <= 3.0 * np.maximum(np.abs(np.spacing(want.astype(np.float32))), 1.2e-7))
        assert abs(float(g('safe_log_fin', [1e-40])[0]) + 92.1034) < 2e-5 and abs(float(g('safe_log_fin', [0.0])[0]) + 92.1034) < 2e-5
        z = np.concatenate([rng.uniform(-30, 30, 300000), [-100.0, 100.0, 0.0]]).astype(np.float32)
        z64 = z.astype(np.float64)
        ls = g('logsigmoid', z)
        assert np.all(np.abs(ls - (np.minimum(z64, 0) - np.log1p(np.exp(-np.abs(z64))))) <= 2e-7 * np.maximum(1.0, np.abs(z64)))
        assert np.all(np.abs(g('sigmoid', z) - 1.0 / (1.0 + np.exp(-z64))) <= 2.5e-7)
        assert np.all(np.abs(g('tanh', z) - np.tanh(z64)) <= 3e-7) and np.all(np.abs(g('tanh_abs', z) - np.tanh(z64)) <= 3e-7)
        nan = g('exp_fin', [np.nan])[0], g('safe_log_fin', [np.nan])[0], g('logsigmoid', [np.nan])[0], g('sigmoid', [np.nan])[0], g('tanh', [np.nan])[0]
        assert all(v != v for v in nan)
        assert g('exp', [np.inf])[0] == np.inf and g('exp', [-np.inf])[0] == 0.0 and g('log', [0.0])[0] == -np.inf and g('log', [np.inf])[0] == np.inf
        assert np.isnan(g('log', [-1.0])[0])
    finally:
        native.use_build(prev)


@pytest.mark.parametrize('dx', [128, 3, 2])
def test_gru_on_three_term_bf16_products_against_float64(dx):
    """The fast build runs the hidden-128 GRU cell on bf16 MFMAs with both operands split into high + low parts (hi hi + hi lo + lo hi, fp32
    accumulation: `k_gru_bf3`, csrc/pdp_neural.hip; reference `nn.GRUCell`, pdp_decimate.py:38-41,75,83).  Against a float64 evaluation of the
    same cell the error stays within 2e-5 of the largest |h'| -- the parity build's fp32 chain is at 2e-6 on the same input -- for the 129-wide
    input of np-nd-np and the 4- / 3-wide ones of p-nd-np; rows of inactive instances pass through unchanged, bit for bit."""
    from pdp import native
    from pdp.factorgraph import dataset
    dev = torch.device('cuda:0')
    b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(40, 200, 3, m=840, seed=3)), dev)
    torch.manual_seed(11)
    cell = torch.nn.GRUCell(dx + 1, 128).to(dev)
    prev = native.BUILD
    out = {}
    try:
        for build in ('parity', 'fast'):
            native.use_build(build)
            prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
            E = prob.E
            g = torch.Generator(device='cpu'); g.manual_seed(5)
            state = (torch.randn(E, dx, generator=g) * 0.7).to(dev); h = (torch.randn(E, 128, generator=g) * 0.5).to(dev)
            am = torch.ones(prob.B, dtype=torch.uint8, device=dev); am[::5] = 0
            w = native.GruWeights(cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
            out[build] = prob.neural_gru(w, state, h, am).cpu().numpy()
            name = native.kernel_name('gru')
            assert name.startswith('k_gru_bf3<') if build == 'fast' else name.startswith('k_gru_pipe<'), name
            torch.cuda.synchronize()
            del prob, w                                        # handles belong to the library that made them (native.use_build)
    finally:
        native.use_build(prev)
    # float64 reference of the cell on [state | sign] (the sign column is the loader's edge feature)
    ef = b['edge_feature'].cpu().numpy().reshape(-1).astype(np.float64)
    x = np.concatenate([state.cpu().numpy().astype(np.float64), ef[:, None]], axis=1)
    h64 = h.cpu().numpy().astype(np.float64)
    Wi, Wh = cell.weight_ih.detach().cpu().numpy().astype(np.float64), cell.weight_hh.detach().cpu().numpy().astype(np.float64)
    bi, bh = cell.bias_ih.detach().cpu().numpy().astype(np.float64), cell.bias_hh.detach().cpu().numpy().astype(np.float64)
    gi, gh = x @ Wi.T + bi, h64 @ Wh.T + bh
    sg = lambda v: 1.0 / (1.0 + np.exp(-v))
    r, z = sg(gi[:, :128] + gh[:, :128]), sg(gi[:, 128:256] + gh[:, 128:256])
    n = np.tanh(gi[:, 256:] + r * gh[:, 256:])
    want = (1.0 - z) * n + z * h64
    inst = b['batch_variable_map'].cpu().numpy()[b['graph_map'].cpu().numpy()[0]]
    live = am.cpu().numpy()[inst] == 1
    scale = np.abs(want).max()
    err_fast = np.abs(out['fast'][live] - want[live]).max() / scale
    err_parity = np.abs(out['parity'][live] - want[live]).max() / scale
    assert err_parity < 3e-6 and err_fast < 2e-5, (err_parity, err_fast)
    np.testing.assert_array_equal(out['fast'][~live], h.cpu().numpy()[~live])


@pytest.mark.parametrize('by_variable', [True, False])
def test_aggregator_on_three_term_bf16_products_against_float64(by_variable):
    """The fast build runs both halves of the hidden-128 MessageAggregator (util.py:51-77; inner widths 100 / 50) on split bf16 products
    (`k_agg_pre_bf3`, `k_agg_post_bf3`).  Against a float64 evaluation of the same call -- per-edge MLP, row sums without the edge's own term,
    second MLP on [sum | sign], blend with the previous state where the instance is inactive -- the error stays within 5e-5 of the largest |out|
    (the parity build: 5e-6); inactive rows pass through bit for bit."""
    from pdp import native
    from pdp.factorgraph import dataset
    dev = torch.device('cuda:0')
    b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(30, 200, 3, m=840, seed=4)), dev)
    torch.manual_seed(13)
    H = 128
    lin = lambda i, o, bias=True: torch.nn.Linear(i, o, bias=bias).to(dev)
    l1m, l2m, l1a, l2a = lin(H + 1, 100), lin(100, 50, False), lin(51, 100), lin(100, H, False)
    prev = native.BUILD
    out = {}
    try:
        for build in ('parity', 'fast'):
            native.use_build(build)
            prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
            E = prob.E
            g = torch.Generator(device='cpu'); g.manual_seed(6)
            state = (torch.randn(E, H, generator=g) * 0.7).to(dev); old = (torch.randn(E, H, generator=g) * 0.5).to(dev)
            am = torch.ones(prob.B, dtype=torch.uint8, device=dev); am[::4] = 0
            w = native.AggregatorWeights(l1m.weight, l1m.bias, l2m.weight, l1a.weight, l1a.bias, l2a.weight, 1)
            out[build] = prob.neural_aggregate_edges(w, by_variable, state, None, am, old).cpu().numpy()
            names = native.kernel_name('agg_pre'), native.kernel_name('agg_post')
            assert names == (('k_agg_pre_bf3', 'k_agg_post_bf3') if build == 'fast' else ('k_agg_pre_wave<65, 4, 50, 2, true>', 'k_agg_post_pf<26, 4, 50, 4>')), names
            torch.cuda.synchronize()
            del prob, w                                        # handles belong to the library that made them (native.use_build)
    finally:
        native.use_build(prev)
    gm = b['graph_map'].cpu().numpy()
    row = gm[0] if by_variable else gm[1]
    ef = b['edge_feature'].cpu().numpy().reshape(-1).astype(np.float64)
    f64 = lambda t_: t_.detach().cpu().numpy().astype(np.float64)
    ls = lambda v: np.minimum(v, 0.0) - np.log1p(np.exp(-np.abs(v)))
    x = np.concatenate([f64(state), ef[:, None]], axis=1)
    s_e = ls(ls(x @ f64(l1m.weight).T + f64(l1m.bias)) @ f64(l2m.weight).T)                        # [E, 50]
    A = np.zeros((int(row.max()) + 1, 50)); np.add.at(A, row, s_e)
    r = np.concatenate([A[row] - s_e, ef[:, None]], axis=1)
    want = ls(ls(r @ f64(l1a.weight).T + f64(l1a.bias)) @ f64(l2a.weight).T)
    inst = b['batch_variable_map'].cpu().numpy()[gm[0]]
    live = am.cpu().numpy()[inst] == 1
    scale = np.abs(want[live]).max()
    err_fast = np.abs(out['fast'][live] - want[live]).max() / scale
    err_parity = np.abs(out['parity'][live] - want[live]).max() / scale
    assert err_parity < 5e-6 and err_fast < 5e-5, (err_parity, err_fast)
    np.testing.assert_array_equal(out['fast'][~live], old.cpu().numpy()[~live])


def test_headline_family_solved_counts_equal_the_parity_build():
    """random 3-SAT n=200 m=840, 600 instances (a NaN-poisoned batch like the headline), T=100 + Walk-SAT: the two builds run the same
    number of sweeps, poison the batch in the same sweep, fix almost the same variables and solve the same number of instances"""
    from pdp import native
    from pdp.factorgraph import dataset
    dev = torch.device('cuda:0')
    B = 600
    b = dataset.to_torch(dataset.collate_segment(dataset.random_ksat_items(B, 200, 3, m=840, seed=0)), dev)
    out = {}
    prev = native.BUILD
    try:
        for build in ('parity', 'fast'):
            native.use_build(build)
            prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'], batch_size=B)
            prob.simplify()
            q = torch.full((prob.E, 3), 1.0, device=dev) / 3.0; fs = torch.zeros(prob.E, 2, device=dev); fs[:, 0] = 0.5
            am = torch.ones(B, dtype=torch.uint8, device=dev); dec = native.Decimator(prob)
            it, lds = prob.sp_solve(q, fs, am, dec, 100, 0.02, 100, inputs_disposable=True)
            replays = prob.last_solve_stats['replays']
            prob.random_fill(seed=5)
            o, ws = prob.local_search(prob.solution.clone(), 100, 0.5, seed=6)
            pred = prob.update_solution(o.reshape(-1).contiguous())
            solved, unsat = prob.cnf_eval(pred.reshape(-1).contiguous())
            out[build] = dict(iters=it, lds=lds, replays=replays, av=prob.active_variables.cpu().numpy().copy(), q=q.cpu().numpy(),
                              solved=int(solved.sum().item()), unsat=int(unsat.sum().item()))
            del prob, dec
    finally:
        native.use_build(prev)
    a, c = out['parity'], out['fast']
    assert (a['iters'], a['lds']) == (c['iters'], c['lds'])          # (how many instances a poison replays depends on the dispatch order, not on the build)
    assert np.array_equal(np.isnan(a['q']).any(axis=1), np.isnan(c['q']).any(axis=1))         # the same instances carry the NaN
    assert (a['av'] == c['av']).mean() >= 0.9995                                             # decimated variables: all but a handful
    assert a['solved'] == c['solved'] and abs(a['unsat'] - c['unsat']) <= max(3, a['unsat'] // 500)
