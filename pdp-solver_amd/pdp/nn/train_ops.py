"""Differentiable building blocks of the neural PDP solver for the training path (reference: what torch autograd derives for
src/pdp/nn/util.py:51-77, src/pdp/nn/pdp_decimate.py:70-83, src/pdp/trainer.py:28-29 and src/pdp/nn/util.py:178-197 when
``FactorGraphTrainerBase._train_batch`` calls ``loss.backward()``, src/pdp/factorgraph/base.py:149-182).

Each ``torch.autograd.Function`` below pairs a forward entry point of libpdp_hip.so with its adjoint (csrc/pdp_train.hip).  PyTorch keeps
the graph, accumulates the parameter gradients and runs the optimizer the caller hands to ``train()``; the arithmetic is native.
"""

import ctypes as C
import os

import torch

from pdp import native

ACT = {'none': 0, 'logsigmoid': 1, 'relu': 2, 'sigmoid': 3, 'tanh': 4}


def _f(t):
    return t.contiguous() if t is not None else None


class LinearAct(torch.autograd.Function):
    "y = act(x W^T + b): nn.Linear followed by an activation, on the fp32 matrix cores"

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x, weight = _f(x), _f(weight)
        R, K = x.shape
        N = weight.shape[0]
        y = torch.empty(R, N, dtype=torch.float32, device=x.device)
        native.check(native.lib().pdp_train_linear(native.ptr(x, torch.float32), C.c_int64(R), C.c_int(K), C.c_int64(K), native.ptr(weight, torch.float32),
                                                   native.ptr(_f(bias), torch.float32), C.c_int(N), C.c_int(ACT[act]), native.ptr(y), native._stream()))
        ctx.save_for_backward(x, weight, y)
        ctx.act, ctx.has_bias = act, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = _f(dy)
        R, K = x.shape
        N = weight.shape[0]
        dz = torch.empty_like(dy)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(weight)
        db = torch.empty(N, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        native.check(native.lib().pdp_train_linear_backward(native.ptr(dy, torch.float32), native.ptr(y), native.ptr(x), C.c_int64(R), C.c_int(K), C.c_int64(K),
                                                            native.ptr(weight), C.c_int(N), C.c_int(ACT[ctx.act]), native.ptr(dz), native.ptr(dx), C.c_int64(K),
                                                            native.ptr(dw), native.ptr(db), native._stream()))
        return dx, dw, db, None


class LinearActS(torch.autograd.Function):
    """y = act([x | xs] W^T + b) with the operand's last column ``xs`` [R] held apart (the edge sign in front of the training path's layers):
    no [R, K + 1] concatenation is made -- the K-wide block runs on the row-stripe GEMM, the column is a rank-one term of its epilogue
    (include/pdp_hip.h: pdp_train_linear_s).  ``xs`` gets no gradient.  Use ``linear_sign`` below, which falls back to the concatenation
    for shapes the kernel does not take."""

    @staticmethod
    def forward(ctx, x, xs, weight, bias, act):
        x, xs, weight = _f(x), _f(xs.reshape(-1)), _f(weight)
        R, K = x.shape
        N = weight.shape[0]
        y = torch.empty(R, N, dtype=torch.float32, device=x.device)
        native.check(native.lib().pdp_train_linear_s(native.ptr(x, torch.float32), native.ptr(xs, torch.float32), C.c_int64(R), C.c_int(K), C.c_int64(K),
                                                     native.ptr(weight, torch.float32), native.ptr(_f(bias), torch.float32), C.c_int(N), C.c_int(ACT[act]),
                                                     native.ptr(y), native._stream()))
        ctx.save_for_backward(x, xs, weight, y)
        ctx.act, ctx.has_bias = act, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, xs, weight, y = ctx.saved_tensors
        dy = _f(dy)
        R, K = x.shape
        N = weight.shape[0]
        dz = torch.empty_like(dy)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(weight)
        db = torch.empty(N, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        native.check(native.lib().pdp_train_linear_s_backward(native.ptr(dy, torch.float32), native.ptr(y), native.ptr(x), native.ptr(xs), C.c_int64(R), C.c_int(K),
                                                              C.c_int64(K), native.ptr(weight), C.c_int(N), C.c_int(ACT[ctx.act]), native.ptr(dz), native.ptr(dx),
                                                              C.c_int64(K), native.ptr(dw), native.ptr(db), native._stream()))
        return dx, None, dw, db, None


def linear_sign(x, sign, weight, bias, act):
    """act([x | sign] W^T + b) for ``sign`` [R, 1]: without the concatenation where the row-stripe kernel takes the shape, with it otherwise
    (``PDP_TRAIN_CAT=1``: always with it -- the form of rounds 1-3, for A/B measurements)."""
    R, K = x.shape
    N = weight.shape[0]
    if (sign.dim() == 2 and sign.size(1) == 1 and weight.shape[1] == K + 1 and os.environ.get('PDP_TRAIN_CAT', '0') != '1'
            and native.lib().pdp_train_linear_s_supported(C.c_int64(R), C.c_int(K), C.c_int(N), C.c_int(ACT[act]))):
        return LinearActS.apply(x, sign, weight, bias, act)
    return LinearAct.apply(torch.cat((x, sign), 1), weight, bias, act)


class RowAggregate(torch.autograd.Function):
    """Deep-set aggregation over the rows of the factor graph (util.py:60-69): the ordered sum of the edge values of every variable
    (by_variable) or clause; with include_self False every edge gets its row's sum minus its own value ([E, A]), else the rows ([rows, A])."""

    @staticmethod
    def forward(ctx, s, problem, by_variable, include_self):
        s = _f(s)
        A = s.shape[1]
        L = native.lib()
        rows = torch.empty(problem.V if by_variable else problem.F, A, dtype=torch.float32, device=s.device)
        native.check(L.pdp_train_row_sum(problem._h, C.c_int(1 if by_variable else 0), native.ptr(s, torch.float32), C.c_int(A), native.ptr(rows), native._stream()))
        ctx.problem, ctx.by_variable, ctx.include_self = problem, by_variable, include_self
        if include_self:
            return rows
        out = torch.empty_like(s)
        native.check(L.pdp_train_row_spread(problem._h, C.c_int(1 if by_variable else 0), native.ptr(rows), native.ptr(s), C.c_int(A), native.ptr(out), native._stream()))
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _f(dout)
        p, bv = ctx.problem, C.c_int(1 if ctx.by_variable else 0)
        A = dout.shape[1]
        L = native.lib()
        ds = torch.empty(p.E, A, dtype=torch.float32, device=dout.device)
        if ctx.include_self:
            native.check(L.pdp_train_row_spread(p._h, bv, native.ptr(dout, torch.float32), None, C.c_int(A), native.ptr(ds), native._stream()))
        else:
            rows = torch.empty(p.V if ctx.by_variable else p.F, A, dtype=torch.float32, device=dout.device)
            native.check(L.pdp_train_row_sum(p._h, bv, native.ptr(dout, torch.float32), C.c_int(A), native.ptr(rows), native._stream()))
            native.check(L.pdp_train_row_spread(p._h, bv, native.ptr(rows), native.ptr(dout), C.c_int(A), native.ptr(ds), native._stream()))
        return ds, None, None, None


class MaskMatmul(torch.autograd.Function):
    """torch.mm(mask, x) for an arbitrary sparse mask (util.py:60,63 with a mask that is not a SATProblem's own): entries in row-sorted order,
    added per row in ascending entry order (pdp_csr_matmul).  The adjoint is the same product with the transposed mask."""

    @staticmethod
    def forward(ctx, x, mask):
        from pdp.nn import util
        ctx.mask = mask
        return util._csr(mask).matmul(_f(x))

    @staticmethod
    def backward(ctx, dout):
        from pdp.nn import util
        mt = getattr(ctx.mask, '_pdp_transposed', None)
        if mt is None:
            mt = ctx.mask.transpose(0, 1)
            ctx.mask._pdp_transposed = mt
        return util._csr(mt).matmul(_f(dout)), None


class GruCell(torch.autograd.Function):
    "torch.nn.GRUCell (gate order r, z, n) on the fp32 matrix cores"

    @staticmethod
    def forward(ctx, x, h, w_ih, w_hh, b_ih, b_hh, packed=None, state=None, sign=None):
        """``packed``: the cell's weights as ``native.GruWeights`` (the inference layout).  With it the 129 -> 128 cell runs its full 64-row
        tiles in one launch of the pipelined inference kernel, which also writes the gates the adjoint needs; the rows behind the last
        full tile take the two-GEMM form.  ``state`` [R, 128] / ``sign`` [R]: the two pieces x = [state | sign] was concatenated from
        (the kernel reads them where they lie; their gradient travels through x)."""
        x, h, w_ih, w_hh, b_ih, b_hh = [_f(t) for t in (x, h, w_ih, w_hh, b_ih, b_hh)]
        R, Kx = x.shape
        H = h.shape[1]
        hnew = torch.empty_like(h)
        saved = torch.empty(R, 4 * H, dtype=torch.float32, device=x.device)
        full = 0
        if packed is not None and H == 128 and Kx == 129 and os.environ.get('PDP_TRAIN_GRU', 'fused') == 'fused':
            full = (R // 64) * 64
            if full:
                # the kernel wants the state rows contiguous [full, 128] and the sign column contiguous [full]
                state = _f(state.detach()) if state is not None else x[:full, :128].contiguous()
                sign = _f(sign.detach().reshape(-1)) if sign is not None else x[:full, 128].contiguous()
                native.check(native.lib().pdp_train_gru_fused(C.byref(packed.desc), native.ptr(state, torch.float32), native.ptr(sign, torch.float32),
                                                              native.ptr(h, torch.float32), C.c_int64(full), native.ptr(hnew), native.ptr(saved), native._stream()))
        if full < R:
            xt, ht = x[full:].contiguous(), h[full:].contiguous()
            hn_t = torch.empty_like(ht)
            sv_t = torch.empty(R - full, 4 * H, dtype=torch.float32, device=x.device)
            scratch = torch.empty(R - full, 6 * H, dtype=torch.float32, device=x.device)
            native.check(native.lib().pdp_train_gru(native.ptr(xt, torch.float32), native.ptr(ht, torch.float32), native.ptr(w_ih), native.ptr(w_hh), native.ptr(b_ih),
                                                    native.ptr(b_hh), C.c_int64(R - full), C.c_int(Kx), C.c_int(H), native.ptr(hn_t), native.ptr(sv_t), native.ptr(scratch),
                                                    native._stream()))
            if full:
                hnew[full:] = hn_t; saved[full:] = sv_t
            else:
                hnew, saved = hn_t, sv_t
        ctx.save_for_backward(x, h, w_ih, w_hh, saved)
        return hnew

    @staticmethod
    def backward(ctx, dhnew):
        x, h, w_ih, w_hh, saved = ctx.saved_tensors
        dhnew = _f(dhnew)
        R, Kx = x.shape
        H = h.shape[1]
        dx, dh = torch.empty_like(x), torch.empty_like(h)
        dw_ih, dw_hh = torch.empty_like(w_ih), torch.empty_like(w_hh)
        db_ih = torch.empty(3 * H, dtype=torch.float32, device=x.device); db_hh = torch.empty_like(db_ih)
        scratch = torch.empty(R, 7 * H, dtype=torch.float32, device=x.device)
        native.check(native.lib().pdp_train_gru_backward(native.ptr(dhnew, torch.float32), native.ptr(saved), native.ptr(x), native.ptr(h), native.ptr(w_ih),
                                                         native.ptr(w_hh), C.c_int64(R), C.c_int(Kx), C.c_int(H), native.ptr(dx), native.ptr(dh), native.ptr(dw_ih),
                                                         native.ptr(dw_hh), native.ptr(db_ih), native.ptr(db_hh), native.ptr(scratch), native._stream()))
        return dx, dh, dw_ih, dw_hh, db_ih, db_hh, None, None, None


class GruCellS(torch.autograd.Function):
    """A hidden-128 cell of the neural decimator on its two input pieces ``state`` [R, Ks] (Ks = 128: np-nd-np; 3 / 2: p-nd-np's surveys) and
    ``sign`` [R, 1] (torch.nn.GRUCell on their concatenation): the forward is one launch of the pipelined inference kernel on the full 64-row tiles (the ragged tail takes the
    two-GEMM form on a concatenated copy of its <= 63 rows), the adjoint multiplies by W_ih[:, :128] only (the sign has no gradient) and gets
    the last column of dW_ih from the pointwise pass.  ``packed``: the weights as native.GruWeights."""

    @staticmethod
    def forward(ctx, state, sign, h, w_ih, w_hh, b_ih, b_hh, packed):
        state, sign, h, w_ih, w_hh, b_ih, b_hh = [_f(t) for t in (state, sign.reshape(-1), h, w_ih, w_hh, b_ih, b_hh)]
        R, Ks = state.shape
        H = h.shape[1]
        hnew = torch.empty_like(h)
        saved = torch.empty(R, 4 * H, dtype=torch.float32, device=state.device)
        full = (R // 64) * 64
        if full:
            native.check(native.lib().pdp_train_gru_fused(C.byref(packed.desc), native.ptr(state, torch.float32), native.ptr(sign, torch.float32),
                                                          native.ptr(h, torch.float32), C.c_int64(full), native.ptr(hnew), native.ptr(saved), native._stream()))
        if full < R:
            xt = torch.cat((state[full:], sign[full:].unsqueeze(1)), 1).contiguous()
            ht = h[full:].contiguous()
            hn_t = torch.empty_like(ht)
            sv_t = torch.empty(R - full, 4 * H, dtype=torch.float32, device=state.device)
            scratch = torch.empty(R - full, 6 * H, dtype=torch.float32, device=state.device)
            native.check(native.lib().pdp_train_gru(native.ptr(xt, torch.float32), native.ptr(ht, torch.float32), native.ptr(w_ih), native.ptr(w_hh), native.ptr(b_ih),
                                                    native.ptr(b_hh), C.c_int64(R - full), C.c_int(Ks + 1), C.c_int(H), native.ptr(hn_t), native.ptr(sv_t),
                                                    native.ptr(scratch), native._stream()))
            hnew[full:] = hn_t; saved[full:] = sv_t
        ctx.save_for_backward(state, sign, h, w_ih, w_hh, saved)
        return hnew

    @staticmethod
    def backward(ctx, dhnew):
        state, sign, h, w_ih, w_hh, saved = ctx.saved_tensors
        dhnew = _f(dhnew)
        R, Ks = state.shape
        H = h.shape[1]
        dstate, dh = torch.empty_like(state), torch.empty_like(h)
        dw_ih, dw_hh = torch.empty_like(w_ih), torch.empty_like(w_hh)
        db_ih = torch.empty(3 * H, dtype=torch.float32, device=state.device); db_hh = torch.empty_like(db_ih)
        scratch = torch.empty(R, 6 * H, dtype=torch.float32, device=state.device)
        native.check(native.lib().pdp_train_gru_backward_s(native.ptr(dhnew, torch.float32), native.ptr(saved), native.ptr(state), native.ptr(sign), native.ptr(h),
                                                           native.ptr(w_ih), native.ptr(w_hh), C.c_int64(R), C.c_int(Ks), C.c_int(H), native.ptr(dstate), native.ptr(dh),
                                                           native.ptr(dw_ih), native.ptr(dw_hh), native.ptr(db_ih), native.ptr(db_hh), native.ptr(scratch),
                                                           native._stream()))
        return dstate, None, dh, dw_ih, dw_hh, db_ih, db_hh, None


class SpAdaptedSweep(torch.autograd.Function):
    """The SP sweep of the adaptor form of SurveyPropagator (model type p-nd-np; pdp_propagate.py:163-221 as the training path runs it: no
    active mask) on the log-domain clause message ``xlog`` [E] and the variable-side input ``fs2`` [E, 2] = [eta, force]: returns the new
    surveys [E, 3] and the new function state [E, 2].  The force column passes through without a gradient (torch.sign)."""

    @staticmethod
    def forward(ctx, xlog, fs2, problem, edge_mask, pi):
        ctx.xshape = xlog.shape
        xlog, fs2 = _f(xlog.reshape(-1)), _f(fs2)
        em = None if edge_mask is None else _f(edge_mask.reshape(-1))
        E = xlog.numel()
        old_q = torch.zeros(E, 3, dtype=torch.float32, device=xlog.device)         # (what an inactive instance would keep: the blend weight is 0 here)
        old_fs = torch.zeros(E, 2, dtype=torch.float32, device=xlog.device)
        q, fs = problem.sp_propagate_adapted(xlog, fs2, em, None, old_q, old_fs, pi)
        ctx.save_for_backward(xlog, fs2, em if em is not None else torch.empty(0, device=xlog.device))
        ctx.problem, ctx.pi, ctx.has_em = problem, pi, em is not None
        return q, fs

    @staticmethod
    def backward(ctx, gq, gfs):
        xlog, fs2, em = ctx.saved_tensors
        gq = _f(gq)
        geta = _f(gfs[:, 0])
        dxlog = torch.empty_like(xlog)
        deta = torch.empty_like(xlog)
        native.check(native.lib().pdp_train_sp_adapted_backward(ctx.problem._h, native.ptr(xlog, torch.float32), native.ptr(fs2, torch.float32),
                                                                native.ptr(em, torch.float32) if ctx.has_em else None, C.c_float(ctx.pi),
                                                                native.ptr(gq, torch.float32), native.ptr(geta, torch.float32), native.ptr(dxlog),
                                                                native.ptr(deta), native._stream()))
        dfs2 = torch.stack((deta, torch.zeros_like(deta)), dim=1)
        return dxlog.reshape(ctx.xshape), dfs2, None, None, None


class SatLoss(torch.autograd.Function):
    "energy loss of a prediction (SatLossEvaluator.forward, util.py:178-197) and its gradient with respect to the prediction"

    @staticmethod
    def forward(ctx, pred, problem, coeff, eps, sharpness):
        p = _f(pred.reshape(-1))
        out = problem.sat_loss(p, coeff, eps, sharpness)
        ctx.save_for_backward(p)
        ctx.problem, ctx.args, ctx.shape = problem, (coeff, eps, sharpness), pred.shape
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        coeff, eps, sharpness = ctx.args
        d = torch.empty_like(p)
        native.check(native.lib().pdp_sat_loss_grad(ctx.problem._h, native.ptr(p, torch.float32), C.c_float(coeff), C.c_float(eps), C.c_int(int(sharpness)),
                                                    C.c_float(1.0), native.ptr(d), native._stream()))
        return (d * g).reshape(ctx.shape), None, None, None, None


def dropout(x, p, rng):
    """F.dropout(x, p, training=True) (pdp_propagate.py:80,91).  Default: torch's device generator.  rng 'torch' (config key dropout_rng, the
    golden tests): the mask comes from the global CPU generator exactly as the reference's --cpu_mode run draws it (the draw depends on
    the shape and p only) -- a host-side [E, H] mask per call, so not for real training runs."""
    if p <= 0:
        return x
    if rng == 'torch':
        mask = torch.nn.functional.dropout(torch.ones(x.shape, dtype=torch.float32), p=p, training=True).to(x.device)
        return x * mask
    return torch.nn.functional.dropout(x, p=p, training=True)
