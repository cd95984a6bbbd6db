"""The Python boundary in the reference's own call shapes (reference: src/pdp/nn/util.py:51-77,121-176,257-286,
src/pdp/nn/solver.py:84-178).

tests/golden/foreign_plugin.py is a propagator / decimator / predictor triple written against the reference's API surface only.
tests/golden/generate_golden.py ran that same file inside the imported reference (CPU) -> foreign_plugin.npz; here it runs on the GPU on
this repository's ``pdp`` package through ``PropagatorDecimatorSolverBase._forward_core_stepwise``.  Bars: every integer output equal
(decimated variables, active flags, active mask, final assignment, clause counts, random-stream consumption); floats rtol 2e-4 / atol 2e-6
(torch's logsigmoid / exp / tanh against the library's, different summation trees in the layers)."""
import importlib.util
import logging
import os

import numpy as np
import pytest
import torch

from helpers import load_golden, REPO, random_batch

LOG = logging.getLogger('test')
RTOL, ATOL = 2e-4, 2e-6


def _plugin():
    spec = importlib.util.spec_from_file_location('foreign_plugin', os.path.join(REPO, 'tests', 'golden', 'foreign_plugin.py'))
    fp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fp)
    return fp


def test_plugin_file_imports_on_this_package_without_a_gpu():
    "the API names the plug-in needs exist (no compute): pdp.nn.util / pdp.nn.solver as the reference spells them"
    fp = _plugin()
    import pdp.nn.util as U
    import pdp.nn.solver as S
    assert fp.util is U and fp.solver is S
    for name in ('MessageAggregator', 'SatLossEvaluator', 'SatCNFEvaluator', 'PerceptronTanh', 'sparse_argmax', 'sparse_max', 'safe_exp',
                 'sparse_smooth_max'):
        assert hasattr(U, name), name
    for name in ('safe_log', 'compute_masks', 'compute_batch_mask'):
        assert callable(getattr(U.SatLossEvaluator, name)), name
    import inspect
    assert list(inspect.signature(U.sparse_argmax).parameters)[:3] == ['x', 'mask', 'device']
    assert list(inspect.signature(U.sparse_max).parameters)[:3] == ['x', 'mask', 'device']
    assert list(inspect.signature(U.sparse_smooth_max).parameters) == ['x', 'mask', 'device', 'alpha']
    assert list(inspect.signature(U.safe_exp).parameters) == ['x', 'device']
    assert list(inspect.signature(U.MessageAggregator.forward).parameters) == ['self', 'state', 'feature', 'mask', 'mask_transpose', 'edge_mask']
    for name in ('_graph_mask_tuple', '_batch_mask_tuple', '_pos_mask_tuple', '_neg_mask_tuple', '_signed_mask_tuple', '_vf_mask_tuple',
                 '_replication_mask_tuple', 'set_variables', 'simplify'):
        assert hasattr(S.SATProblem, name), name
    # the generator classes under the reference's import path (src/pdp/generator.py:98,163,270)
    from pdp.generator import UniformCNFGenerator, ModularCNFGenerator, VariableModularCNFGenerator, CNFGeneratorBase  # noqa: F401
    # the partial forms of the aggregator the reference's constructor accepts (util.py:24-42) build and register the same parameters
    a = U.MessageAggregator(torch.device('cpu'), 9, 8, 0, 6, 10, 1, include_self_message=False)
    assert sorted(k for k in a.state_dict() if not k.startswith('_module_list')) == ['_W1_a.bias', '_W1_a.weight', '_W2_a.weight']
    assert a._W1_a.weight.shape == (10, 10)
    b = U.MessageAggregator(torch.device('cpu'), 9, 8, 12, 6, 0, 1, include_self_message=True)
    assert sorted(k for k in b.state_dict() if not k.startswith('_module_list')) == ['_W1_m.bias', '_W1_m.weight', '_W2_m.weight']


def test_compute_masks_are_the_reference_matrices():
    "SatLossEvaluator.compute_masks / compute_batch_mask (util.py:125-176) as dense matrices, on the host"
    import pdp.nn.util as U
    gm = torch.tensor([[0, 1, 2, 1, 3], [0, 0, 1, 1, 2]], dtype=torch.int32)
    bvm = torch.tensor([0, 0, 0, 1], dtype=torch.int32); bfm = torch.tensor([0, 0, 1], dtype=torch.int32)
    ef = torch.tensor([[1.], [-1.], [1.], [1.], [-1.]])
    vm, fm = U.SatLossEvaluator.compute_masks(gm, bvm, bfm, ef, torch.device('cpu'))
    want_v = torch.zeros(5, 4); want_f = torch.zeros(3, 5)
    for e in range(5):
        want_v[e, gm[0, e]] = ef[e, 0]; want_f[gm[1, e], e] = 1
    assert torch.equal(vm.to_dense(), want_v) and torch.equal(fm.to_dense(), want_f)
    bv, bvt, bf, bft = U.SatLossEvaluator.compute_batch_mask(bvm, bfm, torch.device('cpu'))
    assert bv.shape == (4, 2) and bf.shape == (3, 2)
    assert torch.equal(bv.to_dense(), torch.tensor([[1., 0.], [1., 0.], [1., 0.], [0., 1.]])) and torch.equal(bvt.to_dense(), bv.to_dense().t())
    assert torch.equal(bf.to_dense(), torch.tensor([[1., 0.], [1., 0.], [0., 1.]])) and torch.equal(bft.to_dense(), bf.to_dense().t())
    x = torch.tensor([0.5, 1e-9]); eps = torch.tensor([1e-6])
    assert torch.equal(U.SatLossEvaluator.safe_log(x, eps), torch.max(x, eps).log())


# ---- GPU ---------------------------------------------------------------------------------------------------------------------------

def _dense_reference_max(x, rows, cols, shape):
    "util.sparse_max / sparse_argmax restated on the host with a dense matrix (util.py:257-275)"
    dense = torch.zeros(shape)
    dense[rows, cols] = x - x.min() + 1
    return torch.max(dense, 0)[0] + x.min() - 1, torch.argmax(dense, 0)


@pytest.mark.gpu
def test_sparse_forms_on_problem_masks_and_on_foreign_masks():
    """util.sparse_* with (a) the masks SATProblem built -- mapped to the resident kernels --, (b) the same incidence as masks built from the
    raw tensors -- generic index-list kernels --, (c) the dense restatement on the host: max / arg-max equal bit for bit, smooth max within
    tolerance of torch and (a) == (b) bit for bit (both add a variable's edges in ascending edge order)."""
    import pdp.nn.util as U
    from pdp.nn.solver import SATProblem
    from pdp.factorgraph import dataset
    dev = torch.device('cuda:0')
    tb = dataset.to_torch(random_batch(9, 24, mixed=True, seed=300), dev)
    sp = SATProblem((tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'], None, None), dev, 1)
    V, F, E, B = sp._variable_num, sp._function_num, sp._edge_num, sp._batch_size
    g = torch.Generator().manual_seed(5)
    own_bv, _, own_bf, _ = U.SatLossEvaluator.compute_batch_mask(tb['batch_variable_map'], tb['batch_function_map'], dev)
    bvm = tb['batch_variable_map'].long().cpu(); bfm = tb['batch_function_map'].long().cpu()
    for trial in range(4):
        xv = torch.randn(V, generator=g)
        if trial == 1:
            xv[::3] = xv[0]                      # ties: the first index wins
        if trial == 2:
            xv = xv * 1e6 + 3e7                  # the x - min + 1 rounding shows
        want_max, want_arg = _dense_reference_max(xv, torch.arange(V), bvm, (V, B))
        for mask in (sp._batch_mask_tuple[0], own_bv, sp):
            assert torch.equal(U.sparse_max(xv.to(dev), mask, dev).cpu(), want_max)
            assert torch.equal(U.sparse_argmax(xv.to(dev), mask, dev).cpu(), want_arg)
        xf = torch.randn(F, generator=g)
        want_max, want_arg = _dense_reference_max(xf, torch.arange(F), bfm, (F, B))
        for mask in (sp._batch_mask_tuple[2], own_bf):
            assert torch.equal(U.sparse_max(xf.to(dev), mask, dev).cpu(), want_max)
            assert torch.equal(U.sparse_argmax(xf.to(dev), mask, dev).cpu(), want_arg)
    # an empty column and a NaN
    rows = torch.tensor([0, 1, 2, 3]); cols = torch.tensor([0, 0, 2, 2])
    m = torch.sparse_coo_tensor(torch.stack([rows, cols]), torch.ones(4), (4, 3)).to(dev)
    x = torch.tensor([0.25, 4.0, -1.0, -1.0])
    wm, wa = _dense_reference_max(x, rows, cols, (4, 3))
    assert torch.equal(U.sparse_max(x.to(dev), m, dev).cpu(), wm) and torch.equal(U.sparse_argmax(x.to(dev), m, dev).cpu(), wa)
    x[1] = float('nan')
    got = U.sparse_max(x.to(dev), m, dev).cpu()
    assert torch.isnan(got).all()                                    # x.min() is NaN: every column is (util.py:275)
    assert U.sparse_argmax(x.to(dev), m, dev).cpu().tolist() == [0, 0, 2]      # the first NaN of a column, 0 for the empty one
    # smooth max: resident kernel == generic kernel; both within tolerance of the torch formula; alpha other than 30 on the generic kernel
    xe = (torch.rand(E, 1, generator=g) - 0.3)
    own_vm = torch.sparse_coo_tensor(torch.stack([tb['graph_map'][0].long(), torch.arange(E, device=dev)]), torch.ones(E, device=dev), (V, E))
    a = U.sparse_smooth_max(xe.to(dev), sp._graph_mask_tuple[0], dev)
    b = U.sparse_smooth_max(xe.to(dev), own_vm, dev)
    assert a.shape == (V, 1) and torch.equal(a, b)
    dense = own_vm.to_dense().cpu()
    for alpha, got in ((30, a.cpu()), (5, U.sparse_smooth_max(xe.to(dev), sp._graph_mask_tuple[0], dev, alpha=5).cpu())):
        coeff = torch.min(alpha * xe, torch.tensor([30.0])).exp()
        want = torch.mm(dense, xe * coeff) / torch.max(torch.mm(dense, coeff), torch.ones(1))
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(U.safe_exp(40 * xe.to(dev), dev).cpu().numpy(), torch.min(40 * xe, torch.tensor([30.0])).exp().numpy(), rtol=2e-6)
    # the replication masks (solver.py:84-99)
    sp3 = SATProblem((tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'], None, None), dev, 3)
    rm, rmt = sp3._replication_mask_tuple
    want = torch.zeros(3 * B, B); want[torch.arange(3 * B), torch.arange(B).repeat(3)] = 1
    assert torch.equal(rm.to_dense().cpu(), want) and torch.equal(rmt.to_dense().cpu(), want.t())
    assert sp._replication_mask_tuple is None
    energy = torch.randint(0, 4, (3 * B,), generator=g).float()
    wm, wa = _dense_reference_max(-energy, torch.arange(3 * B), torch.arange(B).repeat(3), (3 * B, B))
    assert torch.equal(U.sparse_argmax(-energy.to(dev), rm, dev).cpu(), wa)           # _deduplicate's call (solver.py:409)
    # the tensors behave as torch sparse matrices as well (the reference multiplies them directly)
    am = (torch.rand(B, 1, generator=g) > 0.4).float().to(dev)
    edge_active = torch.mm(sp._graph_mask_tuple[1], torch.mm(sp._batch_mask_tuple[0], am))
    assert torch.equal(edge_active[:, 0].cpu(), am.cpu()[bvm[tb['graph_map'][0].long().cpu()], 0])


@pytest.mark.gpu
@pytest.mark.parametrize('include_self', [False, True])
def test_message_aggregator_reference_call_equals_the_fused_kernels(include_self):
    """MessageAggregator.forward(state, feature, mask, mask_transpose, edge_mask) -- on SATProblem's masks and on foreign masks -- against the
    fused inference kernels the native plug-ins call (which are pinned to the oracle bit for bit) and against plain torch."""
    import pdp.nn.util as U
    import torch.nn.functional as Fn
    from pdp.nn.solver import SATProblem
    from pdp.factorgraph import dataset
    dev = torch.device('cuda:0')
    tb = dataset.to_torch(random_batch(7, 30, mixed=True, seed=41), dev)
    sp = SATProblem((tb['graph_map'], tb['batch_variable_map'], tb['batch_function_map'], tb['edge_feature'], None, None), dev, 1)
    E, V, F = sp._edge_num, sp._variable_num, sp._function_num
    H = 32
    torch.manual_seed(8)
    agg = U.MessageAggregator(dev, H + 1, H, 40, 20, 36, 0 if include_self else 1, include_self_message=include_self).to(dev)
    state = (torch.randn(E, H) * 0.5).to(dev)
    edge_mask = (torch.rand(E, 1) > 0.2).float().to(dev)
    sign = sp._edge_feature
    own_v = torch.sparse_coo_tensor(torch.stack([tb['graph_map'][0].long(), torch.arange(E, device=dev)]), torch.ones(E, device=dev), (V, E))
    own_f = torch.sparse_coo_tensor(torch.stack([tb['graph_map'][1].long(), torch.arange(E, device=dev)]), torch.ones(E, device=dev), (F, E))
    for by_variable, tagged, own in ((True, sp._graph_mask_tuple[:2], own_v), (False, sp._graph_mask_tuple[2:], own_f)):
        for em in (None, edge_mask):
            x = torch.cat((state, sign), 1)
            feature = None if include_self else sign
            with torch.no_grad():
                a = agg(x, feature, tagged[0], tagged[1], em)
                b = agg(x, feature, own, own.transpose(0, 1), em)
                # plain torch (util.py:51-77)
                s = Fn.logsigmoid(agg._W2_m(Fn.logsigmoid(agg._W1_m(x))))
                if em is not None:
                    s = s * em
                t = torch.mm(own.to_dense(), s)
                if not include_self:
                    t = torch.mm(own.to_dense().t(), t) - (s * em if em is not None else s)
                    t = torch.cat((t, sign), 1)
                want = Fn.logsigmoid(agg._W2_a(Fn.logsigmoid(agg._W1_a(t))))
            assert a.shape == want.shape == ((V if by_variable else F, H) if include_self else (E, H))
            assert torch.equal(a, b)                   # same ascending-edge sums on the resident rows and on the mask's index lists
            np.testing.assert_allclose(a.cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=2e-6)
            if not include_self:
                fused = sp._native.neural_aggregate_edges(agg.native_weights(), by_variable, state, None if em is None else em.reshape(-1).contiguous(),
                                                          None, torch.zeros(E, H, device=dev))
                np.testing.assert_allclose(a.cpu().numpy(), fused.cpu().numpy(), rtol=2e-5, atol=2e-6)
    # differentiable like the reference's module: gradients reach the parameters and the input through both mask kinds
    x = torch.cat((state, sign), 1).requires_grad_(True)
    feature = None if include_self else sign
    out_a = agg(x, feature, sp._graph_mask_tuple[0], sp._graph_mask_tuple[1], edge_mask)
    ga = torch.autograd.grad(out_a.sum(), [x, agg._W1_m.weight, agg._W2_a.weight])
    out_b = agg(x, feature, own_v, own_v.transpose(0, 1), edge_mask)
    gb = torch.autograd.grad(out_b.sum(), [x, agg._W1_m.weight, agg._W2_a.weight])
    for u, v in zip(ga, gb):
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert float(ga[0].abs().sum()) > 0


@pytest.mark.gpu
def test_foreign_plugin_triple_equals_its_run_inside_the_reference():
    fp = _plugin()
    d = load_golden('foreign_plugin')
    T, w, seed, R = [int(v) for v in d['meta']]
    dev = torch.device('cuda:0')
    model = fp.build_solver(dev, local_search_iterations=w)
    # the parameters the reference's constructors drew, under the reference's state-dict keys (strict: the key sets must coincide)
    ref_state = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith('w::')}
    own = model.state_dict()
    assert set(own) == set(ref_state)
    for k in own:                                           # the same seed draws the same parameters here
        np.testing.assert_array_equal(own[k].cpu().numpy(), ref_state[k].numpy(), err_msg=k)
    model.load_state_dict(ref_state, strict=True)
    gm = torch.from_numpy(d['graph_map']).to(dev); bvm = torch.from_numpy(d['batch_variable_map']).to(dev)
    bfm = torch.from_numpy(d['batch_function_map']).to(dev); ef = torch.from_numpy(d['edge_feature']).to(dev)
    out = fp.run(model, dev, gm, bvm, bfm, ef, iterations=T, batch_replication=R, seed=seed)
    assert model.last_run['path'] == 'stepwise' and model.last_run['iterations'] == d['trace_decided'].shape[0]
    trace = out['trace']
    assert len(trace) == d['trace_decided'].shape[0]
    # integer trajectory
    np.testing.assert_array_equal(np.stack([t['decided'].numpy() for t in trace]), d['trace_decided'])
    np.testing.assert_array_equal(np.stack([t['active_variables'].numpy() for t in trace]), d['trace_active_variables'])
    np.testing.assert_array_equal(np.stack([t['active_functions'].numpy() for t in trace]), d['trace_active_functions'])
    np.testing.assert_array_equal(np.stack([t['active_mask'].numpy() for t in trace]), d['trace_active_mask'])
    # the prediction keeps the predictor's beliefs for variables of instances the Walk-SAT pass did not touch: assignment bits equal, values close
    got_pred = out['prediction'].cpu().numpy()[:, 0]
    np.testing.assert_array_equal(got_pred > 0.5, d['final_prediction'] > 0.5)
    np.testing.assert_array_equal(np.isin(got_pred, (0.0, 1.0)), np.isin(d['final_prediction'], (0.0, 1.0)))
    np.testing.assert_allclose(got_pred, d['final_prediction'], rtol=RTOL, atol=ATOL)
    np.testing.assert_array_equal(out['solved'].cpu().numpy()[:, 0], d['final_solved'])
    np.testing.assert_array_equal(out['unsat'].cpu().numpy()[:, 0], d['final_unsat'])
    np.testing.assert_array_equal(out['counts'][0].cpu().numpy()[:, 0], d['counts_variables'])
    np.testing.assert_array_equal(out['counts'][1].cpu().numpy()[:, 0], d['counts_functions'])
    # floats
    np.testing.assert_allclose(np.stack([t['score'].numpy() for t in trace]), d['trace_score'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(np.stack([t['pressure'].numpy() for t in trace]), d['trace_pressure'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(np.array(out['check_log']), d['check_log'], rtol=1e-6)
    for i, x in enumerate(out['states'][0]):
        np.testing.assert_allclose(x.cpu().numpy(), d['final_prop_%d' % i], rtol=RTOL, atol=ATOL)
    for i, x in enumerate(out['states'][1]):
        np.testing.assert_allclose(x.cpu().numpy(), d['final_dec_%d' % i], rtol=RTOL, atol=ATOL)
    assert len(out['states'][1]) == sum(1 for k in d.files if k.startswith('final_dec_'))
    # the host generator stands where the reference's stands (initial states + the Walk-SAT draws)
    torch.manual_seed(seed)
    torch.rand(int(d['rand_sizes'].sum()))
    expected_next = torch.rand(3)
    out2 = fp.run(model, dev, gm, bvm, bfm, ef, iterations=T, batch_replication=R, seed=seed)
    np.testing.assert_array_equal(torch.rand(3).numpy(), expected_next.numpy())
    np.testing.assert_array_equal(out2['prediction'].cpu().numpy(), out['prediction'].cpu().numpy())
