"""Training path on the GPU: the native forward / adjoint pairs behind pdp/nn/train_ops.py against plain PyTorch
fp32 autograd of the same operators, the loss gradient against finite differences of the CPU oracle, and ``_train_batch`` / ``train()``
against what the reference computes for the same seeds (tests/golden/generate_golden.py::gen_train)."""
import json
import logging
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden, random_batch, REPO
from test_hip_ops import t, npy, make_pair

pytestmark = pytest.mark.gpu
LOG = logging.getLogger('test')
GOLD = os.path.join(REPO, 'tests', 'golden')
DEV = 'cuda:0'


def _leaf(*shape, scale=0.3, seed=0):
    g = torch.Generator(device='cpu'); g.manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).requires_grad_(True)


@pytest.mark.parametrize('R,K,N,act', [(1000, 33, 100, 'logsigmoid'), (77, 100, 50, 'logsigmoid'), (4097, 51, 128, 'relu'), (300, 50, 1, 'sigmoid'),
                                       (65, 129, 150, 'tanh'), (5000, 151, 100, 'none'),
                                       # the row-stripe kernel (>= 4 096 rows, act none / logsigmoid; forward and dX): whole and partial k slabs,
                                       # one to four column blocks, two and three column chunks, a ragged last stripe
                                       (8200, 129, 100, 'logsigmoid'), (4200, 100, 50, 'logsigmoid'), (4100, 51, 100, 'none'), (4099, 7, 33, 'none'),
                                       (6000, 129, 384, 'none'), (4500, 129, 129, 'none'), (4097, 128, 3, 'logsigmoid'), (4608, 32, 64, 'none')])
def test_linear_forward_and_adjoint_vs_torch(R, K, N, act):
    from pdp.nn import train_ops as T
    x, w, b = _leaf(R, K, seed=1), _leaf(N, K, seed=2), _leaf(N, seed=3)
    fn = {'logsigmoid': F.logsigmoid, 'relu': torch.relu, 'sigmoid': torch.sigmoid, 'tanh': torch.tanh, 'none': lambda z: z}[act]
    g = torch.randn(R, N, device=DEV)
    y = T.LinearAct.apply(x, w, b, act)
    y.backward(g)
    got = [y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()]
    for p_ in (x, w, b):
        p_.grad = None
    yr = fn(F.linear(x, w, b))
    yr.backward(g)
    for a, r, name in zip(got, (yr.detach(), x.grad, w.grad, b.grad), ('y', 'dx', 'dw', 'db')):
        torch.testing.assert_close(a, r, rtol=2e-4, atol=2e-4 * float(r.abs().max().clamp(min=1e-3)), msg=lambda m: name + ': ' + m)


def test_gru_forward_and_adjoint_vs_torch():
    from pdp.nn import train_ops as T
    for R, Kx, H in ((513, 33, 32), (2000, 129, 128), (100, 151, 150), (64, 4, 32)):
        cell = torch.nn.GRUCell(Kx, H).to(DEV)
        x, h = _leaf(R, Kx, seed=4), _leaf(R, H, seed=5)
        g = torch.randn(R, H, device=DEV)
        hn = T.GruCell.apply(x, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
        hn.backward(g)
        got = [hn.detach().clone(), x.grad.clone(), h.grad.clone()] + [p_.grad.clone() for p_ in cell.parameters()]
        x.grad = h.grad = None
        for p_ in cell.parameters():
            p_.grad = None
        hr = cell(x, h)
        hr.backward(g)
        ref = [hr.detach(), x.grad, h.grad] + [p_.grad for p_ in cell.parameters()]
        for a, r, name in zip(got, ref, ('h', 'dx', 'dh', 'dW_ih', 'dW_hh', 'db_ih', 'db_hh')):
            torch.testing.assert_close(a, r, rtol=3e-4, atol=3e-4 * float(r.abs().max().clamp(min=1e-3)), msg=lambda m: '%s (R=%d): %s' % (name, R, m))


@pytest.mark.parametrize('R,K,N,act', [(8200, 128, 100, 'logsigmoid'), (4100, 50, 100, 'logsigmoid'), (5000, 128, 384, 'none'), (4099, 7, 33, 'none'), (300, 128, 100, 'logsigmoid')])
def test_linear_on_a_separate_sign_column_vs_torch(R, K, N, act):
    """act([x | sign] W^T + b) without the concatenation (train_ops.linear_sign -> LinearActS: the K-wide block on the row-stripe GEMM, the sign
    column a rank-one term of its epilogue; dW's last column from the dZ pass) against torch on the concatenated operand; the last case is
    below the kernel's row threshold and takes the concatenating fallback."""
    from pdp.nn import train_ops as T
    x, w, b = _leaf(R, K, seed=11), _leaf(N, K + 1, seed=12), _leaf(N, seed=13)
    sign = torch.sign(torch.randn(R, 1, device=DEV))
    fn = {'logsigmoid': F.logsigmoid, 'none': lambda z: z}[act]
    g = torch.randn(R, N, device=DEV)
    y = T.linear_sign(x, sign, w, b, act)
    y.backward(g)
    got = [y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()]
    for p_ in (x, w, b):
        p_.grad = None
    yr = fn(F.linear(torch.cat((x, sign), 1), w, b))
    yr.backward(g)
    for a, r, name in zip(got, (yr.detach(), x.grad, w.grad, b.grad), ('y', 'dx', 'dw', 'db')):
        torch.testing.assert_close(a, r, rtol=2e-4, atol=2e-4 * float(r.abs().max().clamp(min=1e-3)), msg=lambda m: name + ': ' + m)
    # and without a bias
    x.grad = w.grad = None
    y = T.linear_sign(x, sign, w, None, act); y.backward(g)
    got = [y.detach().clone(), x.grad.clone(), w.grad.clone()]
    x.grad = w.grad = None
    yr = fn(F.linear(torch.cat((x, sign), 1), w)); yr.backward(g)
    for a, r, name in zip(got, (yr.detach(), x.grad, w.grad), ('y', 'dx', 'dw')):
        torch.testing.assert_close(a, r, rtol=2e-4, atol=2e-4 * float(r.abs().max().clamp(min=1e-3)), msg=lambda m: name + ' (no bias): ' + m)


def test_gru_cell_on_its_two_input_pieces_vs_torch():
    "train_ops.GruCellS (state [R,Ks], sign [R,1] held apart; forward on the inference kernel, adjoint without the sign column) against torch.nn.GRUCell"
    from pdp import native
    from pdp.nn import train_ops as T
    for R, Ks in ((64 * 41 + 7, 128), (128, 128), (33, 128), (64 * 9 + 5, 3), (64 * 3, 2)):      # Ks = 3 / 2: p-nd-np's survey columns
        cell = torch.nn.GRUCell(Ks + 1, 128).to(DEV)
        state, h = _leaf(R, Ks, seed=21), _leaf(R, 128, seed=22)
        sign = torch.sign(torch.randn(R, 1, device=DEV))
        packed = native.GruWeights(cell.weight_ih.data, cell.weight_hh.data, cell.bias_ih.data, cell.bias_hh.data)
        g = torch.randn(R, 128, device=DEV)
        hn = T.GruCellS.apply(state, sign, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, packed)
        hn.backward(g)
        got = [hn.detach().clone(), state.grad.clone(), h.grad.clone()] + [p_.grad.clone() for p_ in cell.parameters()]
        state.grad = h.grad = None
        for p_ in cell.parameters():
            p_.grad = None
        hr = cell(torch.cat((state, sign), 1), h)
        hr.backward(g)
        ref = [hr.detach(), state.grad, h.grad] + [p_.grad for p_ in cell.parameters()]
        for a, r, name in zip(got, ref, ('h', 'dstate', 'dh', 'dW_ih', 'dW_hh', 'db_ih', 'db_hh')):
            torch.testing.assert_close(a, r, rtol=3e-4, atol=3e-4 * float(r.abs().max().clamp(min=1e-3)), msg=lambda m: '%s (R=%d): %s' % (name, R, m))


def test_fused_gru_forward_equals_the_two_gemm_form():
    """The 129 -> 128 cell of the training path runs its full 64-row tiles in one launch of the pipelined inference kernel, which also writes
    the gates the adjoint reads (r | z | n | W_hn h + b_hn); rows behind the last full tile take the two-GEMM form.  Same results as that
    form on all rows, and the gradients through the saved gates equal torch's."""
    from pdp import native
    from pdp.nn import train_ops as T

    class Ctx(object):
        def save_for_backward(self, *a):
            self.saved = a

    for R in (64 * 37 + 13, 64 * 5, 40):
        cell = torch.nn.GRUCell(129, 128).to(DEV)
        state, h = _leaf(R, 128, seed=6), _leaf(R, 128, seed=7)
        sign = torch.sign(torch.randn(R, 1, device=DEV))
        packed = native.GruWeights(cell.weight_ih.data, cell.weight_hh.data, cell.bias_ih.data, cell.bias_hh.data)
        x = torch.cat((state, sign), 1)
        c1, c2 = Ctx(), Ctx()
        with torch.no_grad():
            h1 = T.GruCell.forward(c1, x, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, packed, state, sign)
            h2 = T.GruCell.forward(c2, x, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
        torch.testing.assert_close(h1, h2, rtol=0, atol=2e-6)
        torch.testing.assert_close(c1.saved[4], c2.saved[4], rtol=0, atol=4e-6)
        g = torch.randn(R, 128, device=DEV)
        hn = T.GruCell.apply(x, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, packed, state, sign)
        hn.backward(g)
        got = [hn.detach().clone(), state.grad.clone(), h.grad.clone()] + [p_.grad.clone() for p_ in cell.parameters()]
        state.grad = h.grad = None
        for p_ in cell.parameters():
            p_.grad = None
        hr = cell(torch.cat((state, sign), 1), h)
        hr.backward(g)
        ref = [hr.detach(), state.grad, h.grad] + [p_.grad for p_ in cell.parameters()]
        for a, r, name in zip(got, ref, ('h', 'dstate', 'dh', 'dW_ih', 'dW_hh', 'db_ih', 'db_hh')):
            torch.testing.assert_close(a, r, rtol=3e-4, atol=3e-4 * float(r.abs().max().clamp(min=1e-3)), msg=lambda m: '%s (R=%d): %s' % (name, R, m))


def _rel_to_max(a, r):
    return float((a.double() - r).abs().max() / r.abs().max().clamp(min=1e-30))


@pytest.mark.parametrize('R,K,N,act', [(1000, 33, 100, 'logsigmoid'), (4097, 51, 128, 'relu'), (8200, 129, 100, 'logsigmoid'), (6000, 129, 384, 'none'),
                                       (100000, 129, 100, 'logsigmoid')])
def test_linear_and_its_adjoint_against_float64(R, K, N, act):
    """The tolerance of the fp32-vs-fp32 comparisons above (rtol 2e-4) says little about near-cancelling sums.  Against the same operator in
    float64 every output -- y, dX, dW (a sum over all R rows), db -- is within 4e-6 of the tensor's largest magnitude, with and without the
    separate sign column (tools/train_accuracy.py prints the figures next to torch's own fp32 error, which is 2-9 x larger on dW)."""
    from pdp.nn import train_ops as T
    fn = {'logsigmoid': F.logsigmoid, 'relu': torch.relu, 'none': lambda z: z}[act]
    g = torch.randn(R, N, device=DEV)
    x, w, b = _leaf(R, K, seed=1), _leaf(N, K, seed=2), _leaf(N, seed=3)
    y = T.LinearAct.apply(x, w, b, act); y.backward(g)
    xd, wd, bd = [t_.detach().double().requires_grad_(True) for t_ in (x, w, b)]
    yr = fn(F.linear(xd, wd, bd)); yr.backward(g.double())
    for a, r, name in zip((y.detach(), x.grad, w.grad, b.grad), (yr.detach(), xd.grad, wd.grad, bd.grad), ('y', 'dx', 'dw', 'db')):
        assert _rel_to_max(a, r) < 4e-6, (name, _rel_to_max(a, r))
    if act != 'relu':
        x, w, b = _leaf(R, K, seed=4), _leaf(N, K + 1, seed=5), _leaf(N, seed=6)
        sign = torch.sign(torch.randn(R, 1, device=DEV))
        y = T.linear_sign(x, sign, w, b, act); y.backward(g)
        xd, wd, bd = [t_.detach().double().requires_grad_(True) for t_ in (x, w, b)]
        yr = fn(F.linear(torch.cat((xd, sign.double()), 1), wd, bd)); yr.backward(g.double())
        for a, r, name in zip((y.detach(), x.grad, w.grad, b.grad), (yr.detach(), xd.grad, wd.grad, bd.grad), ('y', 'dx', 'dw', 'db')):
            assert _rel_to_max(a, r) < 4e-6, (name + ' (sign column apart)', _rel_to_max(a, r))


@pytest.mark.parametrize('R,Ks', [(64 * 41 + 7, 128), (64 * 9 + 5, 3), (50000, 128)])
def test_gru_cell_and_its_adjoint_against_float64(R, Ks):
    "the cell on its two input pieces (forward: the pipelined inference kernel) against torch.nn.GRUCell in float64: everything within 4e-6 of the tensor's largest magnitude"
    import copy
    from pdp import native
    from pdp.nn import train_ops as T
    cell = torch.nn.GRUCell(Ks + 1, 128).to(DEV)
    state, h = _leaf(R, Ks, seed=21), _leaf(R, 128, seed=22)
    sign = torch.sign(torch.randn(R, 1, device=DEV))
    packed = native.GruWeights(cell.weight_ih.data, cell.weight_hh.data, cell.bias_ih.data, cell.bias_hh.data)
    g = torch.randn(R, 128, device=DEV)
    hn = T.GruCellS.apply(state, sign, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, packed); hn.backward(g)
    got = [hn.detach(), state.grad, h.grad] + [p_.grad for p_ in cell.parameters()]
    cd = copy.deepcopy(cell).double()
    for p_ in cd.parameters():
        p_.grad = None
    sd, hd = state.detach().double().requires_grad_(True), h.detach().double().requires_grad_(True)
    hr = cd(torch.cat((sd, sign.double()), 1), hd); hr.backward(g.double())
    ref = [hr.detach(), sd.grad, hd.grad] + [p_.grad for p_ in cd.parameters()]
    for a, r, name in zip(got, ref, ('h', 'dstate', 'dh', 'dW_ih', 'dW_hh', 'db_ih', 'db_hh')):
        assert _rel_to_max(a, r) < 4e-6, (name, _rel_to_max(a, r))


@pytest.mark.parametrize('by_variable,include_self', [(True, False), (False, False), (True, True)])
def test_row_aggregate_forward_and_adjoint_vs_torch(oracle, by_variable, include_self):
    from pdp.nn import train_ops as T
    b = random_batch(batch=9, n=25, mixed=True, seed=77)
    hp, op = make_pair(oracle, b)
    A = 50
    s = _leaf(hp.E, A, seed=6)
    rows = torch.from_numpy(b['graph_map'][0 if by_variable else 1].astype(np.int64)).to(DEV)
    nrows = hp.V if by_variable else hp.F
    out = T.RowAggregate.apply(s, hp, by_variable, include_self)
    g = torch.randn_like(out)
    out.backward(g)
    got, gs = out.detach().clone(), s.grad.clone()
    s.grad = None
    agg = torch.zeros(nrows, A, device=DEV).index_add(0, rows, s)
    ref = agg if include_self else agg[rows] - s
    ref.backward(g)
    torch.testing.assert_close(got, ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gs, s.grad, rtol=1e-5, atol=1e-5)


def test_sat_loss_gradient_vs_finite_differences_of_the_oracle(oracle):
    """d loss / d prediction from pdp_sat_loss_grad against central differences of the ORACLE's loss (double-checked with torch autograd of the
    same formula): the loss forward is bit-exact against the oracle (tests/test_hip_ops.py), so this ties the adjoint to the checker."""
    from pdp.nn import train_ops as T
    b = random_batch(batch=7, n=30, mixed=True, seed=5)
    hp, op = make_pair(oracle, b)
    rng = np.random.RandomState(1)
    pred = rng.uniform(0.05, 0.95, size=op.V).astype(np.float32)
    coeff, eps, sharp = 2.0, 1e-8, 5
    x = t(pred).requires_grad_(True)
    loss = T.SatLoss.apply(x, hp, coeff, eps, sharp)
    assert abs(float(loss) - op.sat_loss(pred, coeff, eps, sharp)) <= 1e-6 * abs(float(loss))
    loss.backward()
    grad = npy(x.grad)
    h = 2e-3
    for v in rng.choice(op.V, size=24, replace=False):
        pp, pm = pred.copy(), pred.copy(); pp[v] += h; pm[v] -= h
        fd = (op.sat_loss(pp, coeff, eps, sharp) - op.sat_loss(pm, coeff, eps, sharp)) / (2 * h)
        assert abs(fd - grad[v]) <= 0.05 * abs(grad[v]) + 2e-4, (v, fd, grad[v])
    # the same formula in torch (util.py:178-197), autograd on the GPU
    gm = torch.from_numpy(b['graph_map'].astype(np.int64)).to(DEV); sgn = t(b['edge_feature']).reshape(-1)
    xr = t(pred).requires_grad_(True)
    ev = sgn * xr[gm[0]] + (1 - sgn) / 2
    w = (coeff * ev).exp()
    F_ = op.F
    nom = torch.zeros(F_, device=DEV).index_add(0, gm[1], w * ev); den = torch.zeros(F_, device=DEV).index_add(0, gm[1], w)
    cv = 1 + (den / torch.clamp(nom, min=eps) - 1).pow(sharp)
    torch.clamp(cv, min=eps).log().mean().backward()
    np.testing.assert_allclose(grad, npy(xr.grad), rtol=2e-4, atol=2e-7)


def _train_cfg(**kw):
    c = dict(model_type='np-nd-np', model_name='t-train', verbose=False, dropout=0.0, error_dim=3, exploration=0.1, hidden_dim=32, local_search_iteration=0,
             epsilon=0.5, tolerance=0.02, t_max=100, edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1, mem_hidden_dim=100, agg_hidden_dim=100,
             mem_agg_hidden_dim=50, classifier_dim=50, loss_sharpness=5, randomized=True, train_inner_recurrence_num=1, train_outer_recurrence_num=3,
             clip_norm=0.65, batch_size=6, epoch_num=2, repetition_num=1, train_batch_limit=4000000, test_batch_limit=40000000, max_cache_size=100000,
             test_recurrence_num=5, rng='torch', dropout_rng='torch', random_seed=0)
    c['lambda'] = 0.9
    c.update(kw)
    return c


def _load(m, d, prefix='w__', alias_file='state_dict_alias_map_train.json'):
    alias = json.load(open(os.path.join(GOLD, alias_file)))
    sd = {}
    for key, canon in alias.items():
        k = prefix + canon.replace('.', '__')
        if key == '_global_step':
            sd[key] = torch.zeros(1)
        elif k in d.files:
            sd[key] = torch.from_numpy(d[k])
    m.load_state_dict(sd, strict=True)


GRAD_RTOL = 5e-4          # per element; plus GRAD_ATOL x the tensor's largest |gradient| (the sums behind a weight gradient run over all edges in
GRAD_ATOL = 5e-4          # another order than MKL's sgemm / torch's sparse mm: elements that nearly cancel carry the tensor's absolute error)


@pytest.mark.parametrize('model_type,golden,alias_file,min_grads', [('np-nd-np', 'train_batch', 'state_dict_alias_map_train.json', 20),
                                                                    ('p-nd-np', 'train_batch_p_nd_np', 'state_dict_alias_map_train_pndnp.json', 19)])
def test_train_batch_equals_reference(model_type, golden, alias_file, min_grads):
    """One ``_train_batch`` (base.py:149-182), hidden 32, three outer recurrences with lambda 0.9, random initial state from the torch CPU
    stream: the per-recurrence losses and the first prediction, the gradient of EVERY parameter after loss.backward(), and the parameters
    after the clipped Adam step, against the reference's values for the same seeds -- for the fully neural solver and for p-nd-np (SP
    propagator with learned adaptors + GRU decimator + neural predictor: config/Train/p-prodec2-nsp-cnf-3-10-pytorch.yaml's model type;
    reference + the App. B-5 width shim)."""
    import torch.optim as optim
    from pdp.trainer import SatFactorGraphTrainer
    d = load_golden(golden)
    gm, bvm, bfm, ef = [torch.from_numpy(d[k]).to(DEV) for k in ('graph_map', 'batch_variable_map', 'batch_function_map', 'edge_feature')]
    label = torch.from_numpy(d['label']).to(DEV)
    cfg = _train_cfg(model_type=model_type)
    tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=LOG)
    m = tr._model_list[0]
    _load(m, d, alias_file=alias_file)
    m._global_step.data = torch.tensor([3.0], device=m._global_step.device)
    torch.manual_seed(31)
    state = m.get_init_state(gm, bvm, bfm, ef, None, cfg['randomized'])
    loss = torch.zeros(1, device=DEV)
    steps = []
    for k in range(3):
        prediction, state = m(init_state=state, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                              is_training=True, iteration_num=1)
        if k == 0:
            np.testing.assert_allclose(npy(prediction[0])[:, 0], d['first_prediction'], rtol=3e-4, atol=3e-5)
        lt = tr._compute_loss(model=m, loss=tr._loss, prediction=prediction, label=label, graph_map=gm, batch_variable_map=bvm,
                              batch_function_map=bfm, edge_feature=ef, meta_data=None)
        steps.append(float(lt))
        loss = loss + lt * (0.9 ** (3 - k - 1))
    np.testing.assert_allclose(steps, d['step_losses'], rtol=2e-5)
    np.testing.assert_allclose(float(loss), float(d['loss'][0]), rtol=2e-5)
    loss.backward()
    checked = 0
    for name, prm in m.named_parameters(remove_duplicate=False):
        key = 'g__' + name.replace('.', '__')
        if key in d.files:
            ref = d[key]
            np.testing.assert_allclose(npy(prm.grad), ref, rtol=GRAD_RTOL, atol=GRAD_ATOL * float(np.abs(ref).max()) + 1e-9, err_msg=name)
            checked += 1
    assert checked == sum(1 for k in d.files if k.startswith('g__')) and checked >= min_grads
    # the whole step through _train_batch on a fresh model
    tr2 = SatFactorGraphTrainer(cfg, use_cuda=True, logger=LOG)
    m2 = tr2._model_list[0]
    _load(m2, d, alias_file=alias_file)
    m2._global_step.data = torch.tensor([3.0], device=m2._global_step.device)
    opt = optim.Adam(tr2.get_parameter_list(), lr=1e-3, weight_decay=1e-10)
    total = np.zeros(1, dtype=np.float32)
    torch.manual_seed(31)
    tr2._train_batch(total, opt, gm, bvm, bfm, ef, None, label)
    np.testing.assert_allclose(total, d['train_batch_total_loss'], rtol=2e-5)
    sd = m2.state_dict()
    alias = json.load(open(os.path.join(GOLD, alias_file)))
    n_el = n_far = 0
    for key, canon in alias.items():
        k = 'after__w__' + canon.replace('.', '__')
        if key == canon and k in d.files:
            got, ref, before = npy(sd[key]), d[k], d['w__' + canon.replace('.', '__')]
            assert np.abs(got - ref).max() <= 2.1e-3                           # an Adam step moves every weight by at most lr
            far = np.abs(got - ref) > 2e-6                                     # (a sign flip of a vanishing gradient is a full 2 lr apart)
            n_el += got.size; n_far += int(far.sum())
            if d['g__' + canon.replace('.', '__')].any() if ('g__' + canon.replace('.', '__')) in d.files else True:
                assert np.abs(ref - before).max() > 5e-4, canon                # the step moved the weights
    assert n_el > 15000 and n_far <= 0.002 * n_el, (n_far, n_el)


def _det_weights(name, shape):
    "tests/golden/generate_golden.py::det_weights: parameter values from the canonical parameter name alone"
    import zlib
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    scale = 1.0 / np.sqrt(shape[-1]) if len(shape) >= 2 else 0.1
    return (rs.uniform(-1.0, 1.0, size=shape) * scale).astype(np.float32)


def _sample_index(name, numel, count=384):
    import zlib
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ 0x5bd1e995) & 0x7fffffff)
    return np.sort(rs.choice(numel, size=min(count, numel), replace=False)).astype(np.int64)


@pytest.mark.parametrize('model_type,alias_file,min_grads', [('np-nd-np', 'state_dict_alias_map_train.json', 29), ('p-nd-np', 'state_dict_alias_map_train_pndnp.json', 19)])
def test_train_batch_hidden128_equals_reference(model_type, alias_file, min_grads):
    """The statements of ``_train_batch`` (base.py:149-182) at hidden 128 -- the width of the 128-wide fast kernels' shapes and of the
    shipped training configs' order of magnitude -- on 8 instances, three outer recurrences: per-recurrence losses, first prediction, and
    of every parameter gradient 384 sampled entries, the largest magnitude and the L2 norm, against the reference's values
    (generate_golden.py train_h128; the fixture holds no weights: both sides build them from the parameter names)."""
    from pdp.trainer import SatFactorGraphTrainer
    d = load_golden('train_h128_' + model_type.replace('-', '_'))
    gm, bvm, bfm, ef = [torch.from_numpy(d[k]).to(DEV) for k in ('graph_map', 'batch_variable_map', 'batch_function_map', 'edge_feature')]
    label = torch.from_numpy(d['label']).to(DEV)
    cfg = _train_cfg(model_type=model_type, hidden_dim=128)
    tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=LOG)
    m = tr._model_list[0]
    alias = json.load(open(os.path.join(GOLD, alias_file)))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = {}
    for key, canon in alias.items():
        sd[key] = torch.zeros(1) if key == '_global_step' else torch.from_numpy(_det_weights(canon, shapes[key]))
    m.load_state_dict(sd, strict=True)
    m._global_step.data = torch.tensor([3.0], device=m._global_step.device)
    torch.manual_seed(37)
    state = m.get_init_state(gm, bvm, bfm, ef, None, cfg['randomized'])
    loss = torch.zeros(1, device=DEV)
    steps = []
    for k in range(3):
        prediction, state = m(init_state=state, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                              is_training=True, iteration_num=1)
        if k == 0:
            np.testing.assert_allclose(npy(prediction[0])[:, 0], d['first_prediction'], rtol=3e-4, atol=3e-5)
        lt = tr._compute_loss(model=m, loss=tr._loss, prediction=prediction, label=label, graph_map=gm, batch_variable_map=bvm,
                              batch_function_map=bfm, edge_feature=ef, meta_data=None)
        steps.append(float(lt))
        loss = loss + lt * (0.9 ** (3 - k - 1))
    np.testing.assert_allclose(steps, d['step_losses'], rtol=2e-5)
    np.testing.assert_allclose(float(loss), float(d['loss'][0]), rtol=2e-5)
    loss.backward()
    checked = 0
    for name, prm in m.named_parameters(remove_duplicate=False):
        key = name.replace('.', '__')
        if 'gs__' + key in d.files:
            g = npy(prm.grad).reshape(-1)
            gmax, gnorm = d['gn__' + key]
            np.testing.assert_allclose(g[_sample_index(name, g.size)], d['gs__' + key], rtol=GRAD_RTOL, atol=GRAD_ATOL * float(gmax) + 1e-9, err_msg=name)
            np.testing.assert_allclose(np.sqrt((g.astype(np.float64) ** 2).sum()), gnorm, rtol=2e-4, err_msg=name + ' (norm)')
            np.testing.assert_allclose(np.abs(g).max(), gmax, rtol=1e-3, err_msg=name + ' (max)')
            checked += 1
    assert checked == sum(1 for k in d.files if k.startswith('gs__')) and checked >= min_grads


def test_sp_adapted_sweep_forward_and_adjoint_vs_torch(oracle):
    """train_ops.SpAdaptedSweep (pdp_sp_propagate_adapted / pdp_train_sp_adapted_backward) against torch autograd of the reference's
    formulation (pdp_propagate.py:163-221 written with index_add for the sparse products), with and without an edge mask, pi 0 and 0.1."""
    from pdp.nn import train_ops as T
    b = random_batch(batch=11, n=30, mixed=True, seed=91)
    hp, op = make_pair(oracle, b)
    E = hp.E
    gmap = torch.from_numpy(b['graph_map'].astype(np.int64)).to(DEV)
    s = t(b['edge_feature']).reshape(-1)
    g = torch.Generator(device='cpu'); g.manual_seed(5)
    for pi, with_mask in ((0.0, False), (0.1, True)):
        z = (torch.randn(E, generator=g) * 2).to(DEV).requires_grad_(True)
        u0 = (torch.randn(E, generator=g) * 2).to(DEV).requires_grad_(True)
        force = torch.sign(torch.randn(E, generator=g)).to(DEV)
        em = ((torch.rand(E, generator=g) > 0.2).float().to(DEV)) if with_mask else None
        gq = torch.randn(E, 3, generator=g).to(DEV); ge = torch.randn(E, generator=g).to(DEV)

        def ours():
            xlog = F.logsigmoid(z)
            fs2 = torch.stack((torch.sigmoid(u0), force), 1)
            return T.SpAdaptedSweep.apply(xlog, fs2, hp, em, pi)

        def ref():
            eps, mx = 1e-40, 30.0
            x = F.logsigmoid(z)
            if em is not None:
                x = x * em
            S = torch.zeros(hp.F, device=DEV).index_add(0, gmap[1], x)
            eta = torch.clamp(S[gmap[1]] - x, max=mx).exp()
            y = torch.clamp(1 - torch.sigmoid(u0), min=eps).log()
            if em is not None:
                y = y * em
            P = torch.zeros(hp.V, device=DEV).index_add(0, gmap[0], y * (s > 0).float())[gmap[0]]
            N = torch.zeros(hp.V, device=DEV).index_add(0, gmap[0], y * (s < 0).float())[gmap[0]]
            lg = lambda c: torch.clamp(1.0 - pi * c.float(), min=eps).log()
            same = 0.5 * (1 + s) * P + 0.5 * (1 - s) * N - y + lg(force == s)
            opp = 0.5 * (1 - s) * P + 0.5 * (1 + s) * N + lg(force == -s)
            dc = torch.clamp(same + opp, max=mx).exp()
            A, B = torch.clamp(same, max=mx).exp(), torch.clamp(opp, max=mx).exp()
            qu, qs = A * (1 - B), B * (1 - A)
            tot = qu + qs + dc
            return torch.stack((qu, qs, dc), 1) / tot.unsqueeze(1), torch.stack((eta, force), 1)

        outs = []
        for fn in (ours, ref):
            z.grad = u0.grad = None
            q, fs = fn()
            ((q * gq).sum() + (fs[:, 0] * ge).sum()).backward()
            outs.append((q.detach().clone(), fs.detach().clone(), z.grad.clone(), u0.grad.clone()))
        for a, r, name in zip(outs[0], outs[1], ('q', 'fs', 'dz', 'du0')):
            torch.testing.assert_close(a, r, rtol=2e-4, atol=2e-5 * float(r.abs().max().clamp(min=1e-3)), msg=lambda m_: '%s (pi %g): %s' % (name, pi, m_))


def test_np_d_np_training_is_refused_like_the_reference_fails():
    """The reference cannot train np-d-np: its prediction is sat_problem._solution (IdentityPredictor), and loss.backward() raises (the fixture
    records the reference's own exception for _train_batch's statements).  This build refuses that model type up front, with the reason."""
    from pdp.trainer import SatFactorGraphTrainer
    ref = json.load(open(os.path.join(GOLD, 'train_np_d_np_reference.json')))
    assert ref['raised'] == 'RuntimeError' and not all(ref['prediction_requires_grad_per_step'])
    tr = SatFactorGraphTrainer(_train_cfg(model_type='np-d-np', tolerance=0.02, t_max=10), use_cuda=True, logger=LOG)
    with pytest.raises(Exception, match='np-d-np'):
        tr.train([os.path.join(GOLD, 'train_small.json')], [os.path.join(GOLD, 'train_small.json')], None)


def test_train_run_equals_reference():
    """``train()`` (base.py:311-404) for two epochs on a 12-instance file with dropout 0.2: shuffled loader (the sampler's draws), random
    initial states, dropout masks -- all from the torch CPU stream in the reference's order --, validation pass per epoch.  Losses and
    validation errors per epoch equal the reference's, the global generator ends at the same position, the trained weights agree."""
    import torch.optim as optim
    from pdp.trainer import SatFactorGraphTrainer
    d = load_golden('train_run')
    cfg = _train_cfg(dropout=0.2, max_cache_size=1)
    tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=LOG)
    m = tr._model_list[0]
    _load(m, d)
    opt = optim.Adam(tr.get_parameter_list(), lr=1e-3, weight_decay=1e-10)
    torch.manual_seed(41); np.random.seed(41)
    path = os.path.join(GOLD, 'train_small.json')
    _, errors, losses = tr.train([path], [path], opt, last_export_path_base=None, best_export_path_base=None, metric_index=0)
    np.testing.assert_array_equal(torch.rand(4).numpy(), d['next_rand'])        # same consumption of the random stream
    np.testing.assert_allclose(losses, d['losses'], rtol=2e-4)
    np.testing.assert_allclose(errors[:2], d['errors'][:2], rtol=0, atol=1e-6)
    assert np.array_equal(np.isinf(errors[2]), np.isinf(d['errors'][2]))
    np.testing.assert_array_equal(npy(m._global_step), d['global_step'])
    sd = m.state_dict()
    alias = json.load(open(os.path.join(GOLD, 'state_dict_alias_map_train.json')))
    n_el = n_far = 0
    for key, canon in alias.items():
        k = 'after__w__' + canon.replace('.', '__')
        if key == canon and k in d.files:
            got, ref = npy(sd[key]), d[k]
            assert np.abs(got - ref).max() <= 4.1e-3                           # four Adam steps of lr 1e-3
            n_el += got.size; n_far += int((np.abs(got - ref) > 1e-5).sum())
    assert n_far <= 0.01 * n_el, (n_far, n_el)


def test_train_script_with_generator(tmp_path):
    """satyr-train-test.py without -t on a Train-style YAML (the reference's keys), ``-g``: one epoch of 16 generated instances, validation
    and test on a labelled file; checkpoints (best / last), the losses / errors arrays, and a trained model whose loss went down."""
    import importlib.util
    import yaml
    cfg = dict(model_name='t-np', model_type='np-nd-np', version='0.1', has_meta_data=False, train_path=[os.path.join(GOLD, 'train_small.json')],
               validation_path=[os.path.join(GOLD, 'train_small.json')], test_path=[os.path.join(GOLD, 'train_small.json')], model_path=str(tmp_path),
               repetition_num=1, train_epoch_size=16, epoch_num=3, label_dim=1, edge_feature_dim=1, meta_feature_dim=0, error_dim=3, metric_index=0,
               prediction_dim=1, hidden_dim=32, mem_hidden_dim=100, agg_hidden_dim=100, mem_agg_hidden_dim=50, classifier_dim=50, batch_size=8,
               learning_rate=0.002, exploration=0.1, verbose=False, randomized=True, train_inner_recurrence_num=1, train_outer_recurrence_num=4,
               test_recurrence_num=10, max_cache_size=100000, dropout=0.2, clip_norm=0.65, weight_decay=1e-10, loss_sharpness=5,
               train_batch_limit=4000000, test_batch_limit=40000000, generator='uniform', min_n=6, max_n=14, min_alpha=2, max_alpha=4, min_k=2, max_k=4,
               local_search_iteration=20, epsilon=0.5, rng='torch', init_rng='torch', dropout_rng='torch')   # seeded CPU streams: a reproducible run
    cfg['lambda'] = 1
    path = tmp_path / 'train.yaml'
    path.write_text(yaml.safe_dump(cfg))
    spec = importlib.util.spec_from_file_location('satyr_train_test', os.path.join(REPO, 'pdp-solver_amd', 'satyr-train-test.py'))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    np.random.seed(3); torch.manual_seed(3)
    res = mod.run(3, str(path), True, None, False, False, True, 1)
    base = os.path.join(os.path.relpath(str(tmp_path)), 't-np', '0.1')
    for sub in ('best', 'last'):
        assert os.path.exists(os.path.join(base, sub, 't-np'))
    losses = np.load(os.path.join(base, 'best', 'losses.npy'))
    assert losses.shape == (1, 3, 1) and np.all(np.isfinite(losses)) and losses[0, 2, 0] < losses[0, 0, 0]
    assert len(res) == 1 and np.asarray(res[0][1]).shape == (3, 1)


@pytest.mark.parametrize('model_type,weights', [('np-nd-np', 'demo-np-nd-np-h128.pt'), ('p-nd-np', 'demo-p-nd-np-h128.pt')])
def test_trained_weights_solve_held_out_instances(model_type, weights):
    """models/*.pt were trained on this GPU by tools/train_demo.py (the reference's energy loss on generated 3-SAT: 1 500 steps for np-nd-np,
    2 500 for p-nd-np).  Loaded strictly into a fresh solver and run through the INFERENCE kernels (fused fp32 MFMA aggregators / GRU cells,
    the adaptor form of the SP sweep; T = 30, no Walk-SAT, deterministic initial state) they solve most of a seeded held-out set of the
    training distribution; random weights solve ~1 %."""
    from pdp.trainer import SatFactorGraphTrainer
    from pdp.factorgraph import dataset
    cfg = _train_cfg(model_type=model_type, hidden_dim=128, dropout=0.0, test_recurrence_num=30, rng='philox')
    rng = np.random.RandomState(77)
    items = []
    for k in range(600):
        n = int(rng.randint(10, 41)); m = max(1, int(rng.uniform(2.0, 4.0) * n))
        items += dataset.random_ksat_items(1, n, 3, m=m, seed=55_000_000 + k)
    b = dataset.to_torch(dataset.collate_segment(items), torch.device(DEV))
    gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
    fractions = []
    for trained in (False, True):
        torch.manual_seed(5)
        tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=LOG)
        m_ = tr._model_list[0]
        if trained:
            m_.load_state_dict(torch.load(os.path.join(REPO, 'models', weights), map_location=DEV), strict=True)
        with torch.no_grad():
            st = m_.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1)
            pred, _ = m_(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                         is_training=False, iteration_num=30, check_termination=tr._check_recurrence_termination, batch_replication=1)
            solved, _ = tr._cnf_evaluator(pred[0], gm, bvm, bfm, ef, None, sat_problem=m_._last_problem)
        assert m_.last_run['train_path'] is False
        fractions.append(float(solved.mean().item()))
    assert fractions[0] < 0.1 and fractions[1] > 0.75, fractions


@pytest.mark.parametrize('yaml_name', ['PDP-np-nd-np-demo-h128.yaml', 'PDP-p-nd-np-demo-h128.yaml'])
def test_cli_predicts_with_the_trained_weights(tmp_path, monkeypatch, yaml_name):
    """train -> save -> satyr.py: the Predict YAMLs that point at models/ run the CLI on a directory of generated DIMACS files (the training
    distribution) and most rows come back solved (T = 30, no Walk-SAT)."""
    import satyr
    from pdp import generator
    ddir = tmp_path / 'cnf'
    ddir.mkdir()
    rng = np.random.RandomState(123)
    for k in range(60):
        n = int(rng.randint(10, 41)); m_ = max(1, int(rng.uniform(2.0, 4.0) * n))
        clauses = generator.uniform_ksat(n, m_, 3, np.random.RandomState(7_000 + k))
        (ddir / ('g%02d_1.cnf' % k)).write_text('p cnf %d %d\n' % (n, len(clauses)) + ''.join(' '.join(str(x) for x in c) + ' 0\n' for c in clauses))
    out = tmp_path / 'out.jsonl'
    monkeypatch.chdir(REPO)
    satyr.main([os.path.join(REPO, 'config', 'Predict', yaml_name), str(ddir), '30', '-d', '-z', '100', '-s', '3', '-w', '0', '--rng', 'philox', '-o', str(out)])
    rows = [json.loads(l) for l in out.read_text().split('\n') if l.strip()]
    assert len(rows) == 60
    assert sum(r['solved'] for r in rows) >= 42, sum(r['solved'] for r in rows)
