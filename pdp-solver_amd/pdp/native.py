"""ctypes binding of libpdp_hip.so (the C ABI declared in include/pdp_hip.h).

PyTorch is used for device memory and streams only: every call hands raw device pointers
(``tensor.data_ptr()``) and the current HIP stream to the library.  There is NO CPU fallback: if the
library is missing or no GPU is visible the calls raise (loudly), they never reroute.
"""

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# two builds of the same sources (csrc/Makefile): 'parity' -- the default, IEEE-only device math, the oracle's bits -- and the opt-in 'fast'
# (device math on the transcendental unit, include/pdp_math.h PDP_FAST_MATH; gated by the reference-held fixtures only).  PDP_BUILD=fast
# selects it for a process, use_build() inside one; PDP_HIP_LIB names any other file (A/B builds).
BUILDS = {'parity': 'libpdp_hip.so', 'fast': 'libpdp_hip_fast.so'}
BUILD = os.environ.get('PDP_BUILD', 'parity')
if BUILD not in BUILDS:
    raise ImportError("PDP_BUILD must be one of %s, got %r" % (sorted(BUILDS), BUILD))
LIB_PATH = os.environ.get('PDP_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'csrc', BUILDS[BUILD])

PDP_OK = 0
PDP_ERR_SPECULATION = 5
RNG_STREAM, RNG_PHILOX = 0, 1
MODEL_SP, MODEL_WALKSAT, MODEL_REINFORCE = 0, 1, 2

EXPORTED_SYMBOLS = [
    'pdp_abi_version', 'pdp_last_error', 'pdp_device_count', 'pdp_problem_create', 'pdp_problem_destroy',
    'pdp_problem_dims', 'pdp_problem_set_rng_base', 'pdp_problem_set_exchange', 'pdp_problem_export_graph', 'pdp_problem_bind_state', 'pdp_simplify', 'pdp_set_variables',
    'pdp_refresh_edge_mask', 'pdp_smooth_max', 'pdp_instance_max', 'pdp_instance_argmax', 'pdp_sp_propagate', 'pdp_sp_adaptors', 'pdp_sp_propagate_adapted',
    'pdp_survey_score', 'pdp_cnf_eval', 'pdp_sat_loss', 'pdp_update_solution', 'pdp_check_termination', 'pdp_loop_begin', 'pdp_loop_step', 'pdp_loop_read', 'pdp_decimator_create',
    'pdp_decimator_destroy', 'pdp_decimator_reset', 'pdp_sequential_decimate', 'pdp_sequential_decimate_gate',
    'pdp_sequential_decimate_apply', 'pdp_reinforce_decimate', 'pdp_reinforce_predict', 'pdp_energy',
    'pdp_energy_diff', 'pdp_random_fill', 'pdp_local_search', 'pdp_deduplicate', 'pdp_sp_solve', 'pdp_math_apply',
    'pdp_neural_aggregate_edges', 'pdp_neural_gru', 'pdp_neural_predict', 'pdp_dimacs_open', 'pdp_dimacs_read', 'pdp_dimacs_close', 'pdp_dimacs_open_many',
    'pdp_kernel_timing', 'pdp_kernel_timing_read', 'pdp_kernel_name',
    'pdp_train_linear', 'pdp_train_linear_backward', 'pdp_train_linear_s_supported', 'pdp_train_linear_s', 'pdp_train_linear_s_backward', 'pdp_train_row_sum', 'pdp_train_row_spread', 'pdp_train_gru', 'pdp_train_gru_fused', 'pdp_train_gru_backward', 'pdp_train_gru_backward_s',
    'pdp_sat_loss_grad', 'pdp_train_sp_adapted_backward',
    'pdp_coo_max', 'pdp_coo_argmax', 'pdp_coo_row_ptr', 'pdp_csr_matmul', 'pdp_csr_smooth_max',
]


class NativeError(RuntimeError):
    pass


class SpeculationFailed(NativeError):
    """The persistent solver met one of the reference's cross-instance couplings; rerun step-wise."""


class CoupledForwardFailed(NativeError):
    """A coupled forward spread over several processes (--split-forward) met a coupling only the single-process loops reproduce (the
    batch-global minimum of a sweep was not 0 in any part).  Every part raises it (the outcome is agreed on across the parts); the predict
    driver then solves that segment whole on the rank of its first part."""


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError("libpdp_hip.so is not built (%s); run `make -C pdp-solver_amd/csrc` or "
                              "__graft_entry__.build()" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.pdp_last_error.restype = C.c_char_p
        for name in EXPORTED_SYMBOLS:
            if name != 'pdp_last_error':
                getattr(_lib, name).restype = C.c_int
    return _lib


def use_build(name):
    """Switch this process to the other build of the library.  Handles (Problem, Decimator, weight descriptors) belong to the library that
    made them: drop every one of them first.  Returns the previous build's name."""
    global _lib, LIB_PATH, BUILD
    if name not in BUILDS:
        raise NativeError("unknown build %r (have %s)" % (name, sorted(BUILDS)))
    previous = BUILD
    if name != BUILD:
        BUILD = name
        LIB_PATH = os.path.join(os.path.dirname(_HERE), 'csrc', BUILDS[name])
        _lib = None
    return previous


def check(status):
    if status == PDP_OK:
        return
    msg = lib().pdp_last_error().decode('utf-8', 'replace')
    if status == PDP_ERR_SPECULATION:
        raise SpeculationFailed(msg)
    raise NativeError("libpdp_hip error %d: %s" % (status, msg))


def require_gpu():
    if not torch.cuda.is_available():
        raise NativeError("no HIP device visible: the PDP hot path has no CPU fallback")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_DT = {torch.float32: 'f32', torch.int32: 'i32', torch.uint8: 'u8', torch.int64: 'i64'}


def ptr(t, dtype=None, numel=None, name='tensor'):
    """Device pointer of a dense tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise NativeError("%s must live on the GPU" % name)
    if not t.is_contiguous():
        raise NativeError("%s must be contiguous" % name)
    if dtype is not None and t.dtype != dtype:
        raise NativeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if numel is not None and t.numel() != numel:
        raise NativeError("%s must have %d elements, got %d" % (name, numel, t.numel()))
    return C.c_void_p(t.data_ptr())


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32), C.c_int)


class SolveArgs(C.Structure):
    _fields_ = [('model', C.c_int32), ('iterations', C.c_int32), ('tolerance', C.c_float), ('t_max', C.c_float),
                ('pi', C.c_float), ('decimation_probability', C.c_float), ('seed', C.c_uint64),
                ('coins', C.c_void_p), ('q', C.c_void_p), ('fs', C.c_void_p), ('active_mask', C.c_void_p),
                ('decimator', C.c_void_p), ('check_termination', C.c_int32), ('iterations_run_host', C.c_int32),
                ('used_lds_host', C.c_int32), ('kernel_launches_host', C.c_int32), ('replay_launches_host', C.c_int32),
                ('time_kernels', C.c_int32), ('solve_kernel_ms_host', C.c_float), ('replay_kernel_ms_host', C.c_float),
                ('replicas_identical', C.c_int32), ('isolate_instances', C.c_int32), ('hbm_instances_host', C.c_int32),
                ('inputs_disposable', C.c_int32)]


class AggDesc(C.Structure):
    _fields_ = [('Wt1m', C.c_void_p), ('b1m', C.c_void_p), ('Wt2m', C.c_void_p), ('Wt1a', C.c_void_p), ('b1a', C.c_void_p),
                ('Wt2a', C.c_void_p), ('din', C.c_int32), ('m1', C.c_int32), ('a', C.c_int32), ('g', C.c_int32),
                ('out', C.c_int32), ('fd', C.c_int32)]


class GruDesc(C.Structure):
    _fields_ = [('Wt_ih', C.c_void_p), ('Wt_hh', C.c_void_p), ('b_ih', C.c_void_p), ('b_hh', C.c_void_p), ('dx', C.c_int32), ('H', C.c_int32)]


class HeadDesc(C.Structure):
    _fields_ = [('Wt1', C.c_void_p), ('b1', C.c_void_p), ('w2', C.c_void_p), ('H', C.c_int32), ('C', C.c_int32), ('out_act', C.c_int32)]


def _pad_linear(weight, bias):
    "nn.Linear weight [N,K] (+bias) -> (Wt [Kp,Np] zero padded, bias [Np]) in the layout include/pdp_hip.h documents"
    N, K = weight.shape
    Kp, Np = K + (K & 1), (N + 31) // 32 * 32
    Wt = torch.zeros(Kp, Np, dtype=torch.float32, device=weight.device)
    Wt[:K, :N] = weight.detach().t()
    bp = torch.zeros(Np, dtype=torch.float32, device=weight.device)
    if bias is not None:
        bp[:N] = bias.detach()
    return Wt.contiguous(), bp


class AggregatorWeights(object):
    "device-side, padded copy of a MessageAggregator's four layers (keeps the tensors alive for the descriptor)"

    def __init__(self, W1m, b1m, W2m, W1a, b1a, W2a, feature_dim):
        self.t = []
        Wt1m, bb1m = _pad_linear(W1m, b1m); Wt2m, _ = _pad_linear(W2m, None)
        Wt1a, bb1a = _pad_linear(W1a, b1a); Wt2a, _ = _pad_linear(W2a, None)
        self.t = [Wt1m, bb1m, Wt2m, Wt1a, bb1a, Wt2a]
        d = AggDesc()
        d.Wt1m, d.b1m, d.Wt2m, d.Wt1a, d.b1a, d.Wt2a = [x.data_ptr() for x in self.t]
        d.din = W1m.shape[1]; d.m1 = W1m.shape[0]; d.a = W2m.shape[0]; d.g = W1a.shape[0]; d.out = W2a.shape[0]; d.fd = feature_dim
        assert W1a.shape[1] == d.a + feature_dim
        self.desc = d


class GruWeights(object):
    def __init__(self, W_ih, W_hh, b_ih, b_hh):
        H = W_hh.shape[1]; dx1 = W_ih.shape[1]
        Hp = (H + 31) // 32 * 32
        Kpx, Kph = dx1 + (dx1 & 1), H + (H & 1)
        dev = W_ih.device
        Wt_ih = torch.zeros(Kpx, 3 * Hp, dtype=torch.float32, device=dev); Wt_hh = torch.zeros(Kph, 3 * Hp, dtype=torch.float32, device=dev)
        bi = torch.zeros(3 * Hp, dtype=torch.float32, device=dev); bh = torch.zeros(3 * Hp, dtype=torch.float32, device=dev)
        for g in range(3):
            Wt_ih[:dx1, g * Hp:g * Hp + H] = W_ih.detach()[g * H:(g + 1) * H].t()
            Wt_hh[:H, g * Hp:g * Hp + H] = W_hh.detach()[g * H:(g + 1) * H].t()
            bi[g * Hp:g * Hp + H] = b_ih.detach()[g * H:(g + 1) * H]; bh[g * Hp:g * Hp + H] = b_hh.detach()[g * H:(g + 1) * H]
        self.t = [Wt_ih.contiguous(), Wt_hh.contiguous(), bi, bh]
        d = GruDesc()
        d.Wt_ih, d.Wt_hh, d.b_ih, d.b_hh = [x.data_ptr() for x in self.t]
        d.dx = dx1 - 1; d.H = H
        self.desc = d


class HeadWeights(object):
    def __init__(self, W1, b1, W2, out_act):
        Wt1, bb1 = _pad_linear(W1, b1)
        w2 = W2.detach().reshape(-1).to(torch.float32).contiguous()
        self.t = [Wt1, bb1, w2]
        d = HeadDesc()
        d.Wt1, d.b1, d.w2 = [x.data_ptr() for x in self.t]
        d.H = W1.shape[1]; d.C = W1.shape[0]; d.out_act = {'sigmoid': 3, 'tanh': 4}[out_act]
        self.desc = d


class Decimator(object):
    """Native SequentialDecimator / ReinforceDecimator state (previous survey + counters)."""

    def __init__(self, problem):
        self.problem = problem
        h = C.c_void_p()
        self._made_by = lib()                         # a handle is destroyed by the library that made it, whatever build the process uses by then
        check(self._made_by.pdp_decimator_create(C.byref(h), problem._h))
        self._h = h

    def reset(self):
        check(lib().pdp_decimator_reset(self._h, _stream()))

    def __del__(self):
        try:
            if self._h:
                self._made_by.pdp_decimator_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Problem(object):
    """HBM-resident batch of CNF instances (native SATProblem)."""

    def __init__(self, graph_map, batch_variable_map, batch_function_map, edge_feature, batch_size=None, replication=1):
        require_gpu()
        L = lib()
        self.device = graph_map.device
        gm = graph_map.to(torch.int32).contiguous()
        bvm = batch_variable_map.to(torch.int32).contiguous()
        bfm = batch_function_map.to(torch.int32).contiguous()
        ef = edge_feature.to(torch.float32).reshape(-1).contiguous()
        E, V, F = gm.size(1), bvm.numel(), bfm.numel()
        if batch_size is None:
            batch_size = int(bvm.max().item()) + 1
        h = C.c_void_p()
        check(L.pdp_problem_create(C.byref(h), C.c_int(E), C.c_int(V), C.c_int(F), C.c_int(batch_size), C.c_int(replication),
                                   ptr(gm, torch.int32), ptr(bvm, torch.int32), ptr(bfm, torch.int32), ptr(ef, torch.float32),
                                   _stream()))
        self._h = h
        self._made_by = L                             # destroyed by the library that made it (see Decimator)
        dims = (C.c_int32 * 8)()
        check(L.pdp_problem_dims(self._h, dims))
        self.E, self.V, self.F, self.B, self.R, self.max_n, self.max_m, self.max_e = [int(x) for x in dims]
        dev = self.device
        self.active_variables = torch.empty(self.V, 1, dtype=torch.float32, device=dev)
        self.active_functions = torch.empty(self.F, 1, dtype=torch.float32, device=dev)
        self.solution = torch.empty(self.V, dtype=torch.float32, device=dev)
        self.is_sat = torch.empty(self.B, dtype=torch.float32, device=dev)
        self.edge_mask = torch.empty(self.E, 1, dtype=torch.float32, device=dev)
        check(L.pdp_problem_bind_state(self._h, ptr(self.active_variables), ptr(self.active_functions), ptr(self.solution),
                                       ptr(self.is_sat), ptr(self.edge_mask), _stream()))

    def __del__(self):
        try:
            if self._h:
                self._made_by.pdp_problem_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- graph export (replicated arrays for the python-visible attributes) --------------------------------
    def export_graph(self):
        dev = self.device
        gm = torch.empty(2, self.E, dtype=torch.int32, device=dev)
        bvm = torch.empty(self.V, dtype=torch.int32, device=dev)
        bfm = torch.empty(self.F, dtype=torch.int32, device=dev)
        ef = torch.empty(self.E, 1, dtype=torch.float32, device=dev)
        check(lib().pdp_problem_export_graph(self._h, ptr(gm), ptr(bvm), ptr(bfm), ptr(ef), _stream()))
        return gm, bvm, bfm, ef

    # -- K7 / K8 ---------------------------------------------------------------------------------------------
    def simplify(self):
        check(lib().pdp_simplify(self._h, _stream()))

    def set_variables(self, assignment):
        check(lib().pdp_set_variables(self._h, ptr(assignment, torch.float32, self.V, 'assignment'), _stream()))

    def refresh_edge_mask(self, want_flag=True):
        flag = C.c_int32(0)
        check(lib().pdp_refresh_edge_mask(self._h, C.byref(flag) if want_flag else None, _stream()))
        return bool(flag.value) if want_flag else None

    # -- K4 / K5 ---------------------------------------------------------------------------------------------
    def smooth_max(self, x):
        out = torch.empty(self.V, 1, dtype=torch.float32, device=self.device)
        check(lib().pdp_smooth_max(self._h, ptr(x, torch.float32, self.E, 'x'), ptr(out), _stream()))
        return out

    def instance_max(self, x):
        out = torch.empty(self.B, dtype=torch.float32, device=self.device)
        check(lib().pdp_instance_max(self._h, ptr(x, torch.float32, self.V, 'x'), ptr(out), _stream()))
        return out

    def instance_argmax(self, x):
        out = torch.empty(self.B, dtype=torch.int64, device=self.device)
        check(lib().pdp_instance_argmax(self._h, ptr(x, torch.float32, self.V, 'x'), ptr(out), _stream()))
        return out

    # -- K1-K3, K6 -------------------------------------------------------------------------------------------
    def sp_propagate(self, dec_q, dec_fs, edge_mask, active_mask, init_q, init_fs, pi=0.0):
        out_q = torch.empty(self.E, 3, dtype=torch.float32, device=self.device)
        out_fs = torch.empty(self.E, 2, dtype=torch.float32, device=self.device)
        check(lib().pdp_sp_propagate(self._h, ptr(dec_q, torch.float32, 3 * self.E, 'decimator_state[0]'),
                                     ptr(dec_fs, torch.float32, 2 * self.E, 'decimator_state[1]'),
                                     ptr(edge_mask, torch.float32, self.E, 'edge_mask'),
                                     ptr(active_mask, torch.uint8, self.B, 'active_mask'),
                                     ptr(init_q, torch.float32, 3 * self.E, 'init_state[0]'),
                                     ptr(init_fs, torch.float32, 2 * self.E, 'init_state[1]'),
                                     C.c_float(pi), ptr(out_q), ptr(out_fs), _stream()))
        return out_q, out_fs

    def sp_adaptors(self, dec_v, dec_f, w_f, W_v):
        "adaptor form of the propagator inputs (p-nd-np): [E,H] decimator states -> xlog [E], fs2 [E,2]"
        H = dec_v.shape[1]
        xlog = torch.empty(self.E, dtype=torch.float32, device=self.device)
        fs2 = torch.empty(self.E, 2, dtype=torch.float32, device=self.device)
        check(lib().pdp_sp_adaptors(self._h, C.c_int(H), ptr(dec_v, torch.float32, self.E * H, 'decimator_state[0]'),
                                    ptr(dec_f, torch.float32, self.E * H, 'decimator_state[1]'), ptr(w_f, torch.float32, H, 'w_f'),
                                    ptr(W_v, torch.float32, 2 * H, 'W_v'), ptr(xlog), ptr(fs2), _stream()))
        return xlog, fs2

    def sp_propagate_adapted(self, xlog, dec_fs, edge_mask, active_mask, init_q, init_fs, pi=0.0, out=None):
        out_q = torch.empty(self.E, 3, dtype=torch.float32, device=self.device) if out is None else out[0]
        out_fs = torch.empty(self.E, 2, dtype=torch.float32, device=self.device) if out is None else out[1]
        check(lib().pdp_sp_propagate_adapted(self._h, ptr(xlog, torch.float32, self.E, 'xlog'), ptr(dec_fs, torch.float32, 2 * self.E, 'fs2'),
                                             ptr(edge_mask, torch.float32, self.E, 'edge_mask'), ptr(active_mask, torch.uint8, self.B, 'active_mask'),
                                             ptr(init_q, torch.float32, 3 * self.E, 'init_state[0]'), ptr(init_fs, torch.float32, 2 * self.E, 'init_state[1]'),
                                             C.c_float(pi), ptr(out_q, torch.float32, 3 * self.E, 'out[0]'), ptr(out_fs, torch.float32, 2 * self.E, 'out[1]'), _stream()))
        return out_q, out_fs

    def survey_score(self, fs, pi=0.0):
        out = torch.empty(self.V, 1, dtype=torch.float32, device=self.device)
        check(lib().pdp_survey_score(self._h, ptr(fs, torch.float32, 2 * self.E, 'message_state[1]'), C.c_float(pi), ptr(out), _stream()))
        return out

    # -- K9 / K13 --------------------------------------------------------------------------------------------
    def cnf_eval(self, pred):
        solved = torch.empty(self.B, 1, dtype=torch.float32, device=self.device)
        unsat = torch.empty(self.B, 1, dtype=torch.float32, device=self.device)
        check(lib().pdp_cnf_eval(self._h, ptr(pred, torch.float32, self.V, 'prediction'), ptr(solved), ptr(unsat), _stream()))
        return solved, unsat

    def sat_loss(self, pred, coeff, eps, sharpness):
        "energy loss of a prediction (SatLossEvaluator.forward): device float [1]"
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        check(lib().pdp_sat_loss(self._h, ptr(pred, torch.float32, self.V, 'prediction'), C.c_float(coeff), C.c_float(eps),
                                 C.c_int(int(sharpness)), ptr(out), _stream()))
        return out

    def update_solution(self, pred):
        out = torch.empty(self.V, 1, dtype=torch.float32, device=self.device)
        check(lib().pdp_update_solution(self._h, ptr(pred, torch.float32, self.V, 'prediction'), ptr(out), _stream()))
        return out

    def check_termination(self, active_mask, pred):
        check(lib().pdp_check_termination(self._h, ptr(active_mask, torch.uint8, self.B, 'active_mask'),
                                          ptr(pred, torch.float32, self.V, 'prediction'), _stream()))

    # -- device-driven loop (pdp_loop_*): the sweeps of a forward enqueued without a host read per sweep ------------
    def loop_begin(self):
        check(lib().pdp_loop_begin(self._h, _stream()))

    def loop_step(self, active_mask):
        check(lib().pdp_loop_step(self._h, ptr(active_mask, torch.uint8, self.B, 'active_mask'), _stream()))

    def loop_read(self, end=False):
        "(stopped, executed sweeps); waits for the stream"
        stopped, iters = C.c_int32(0), C.c_int32(0)
        check(lib().pdp_loop_read(self._h, C.byref(stopped), C.byref(iters), C.c_int(1 if end else 0), _stream()))
        return bool(stopped.value), int(iters.value)

    # -- decimators --------------------------------------------------------------------------------------------
    def sequential_decimate(self, dec, fs, active_mask, tolerance, t_max, pi=0.0):
        check(lib().pdp_sequential_decimate(self._h, dec._h, ptr(fs, torch.float32, 2 * self.E, 'message_state[1]'),
                                            ptr(active_mask, torch.uint8, self.B, 'active_mask'), C.c_float(tolerance),
                                            C.c_float(t_max), C.c_float(pi), _stream()))

    def sequential_decimate_gate(self, dec, fs, active_mask, tolerance, t_max):
        flag = C.c_int32(0)
        check(lib().pdp_sequential_decimate_gate(self._h, dec._h, ptr(fs, torch.float32, 2 * self.E), ptr(active_mask, torch.uint8, self.B),
                                                 C.c_float(tolerance), C.c_float(t_max), C.byref(flag), _stream()))
        return bool(flag.value)

    def sequential_decimate_apply(self, dec, fs, score, active_mask):
        check(lib().pdp_sequential_decimate_apply(self._h, dec._h, ptr(fs, torch.float32, 2 * self.E),
                                                  ptr(score, torch.float32, self.V), ptr(active_mask, torch.uint8, self.B), _stream()))

    def reinforce_decimate(self, dec, fs, active_mask, coin, decimation_probability, pi):
        check(lib().pdp_reinforce_decimate(self._h, dec._h, ptr(fs, torch.float32, 2 * self.E), ptr(active_mask, torch.uint8, self.B),
                                           C.c_float(coin), C.c_float(decimation_probability), C.c_float(pi), _stream()))

    def reinforce_predict(self, fs):
        out = torch.empty(self.V, 1, dtype=torch.float32, device=self.device)
        check(lib().pdp_reinforce_predict(self._h, ptr(fs, torch.float32, 2 * self.E), ptr(out), _stream()))
        return out

    # -- K14 ---------------------------------------------------------------------------------------------------
    def energy(self, assignment):
        en = torch.empty(self.B, 1, dtype=torch.float32, device=self.device)
        uf = torch.empty(self.F, 1, dtype=torch.float32, device=self.device)
        check(lib().pdp_energy(self._h, ptr(assignment, torch.float32, self.V), ptr(en), ptr(uf), _stream()))
        return en, uf

    def energy_diff(self, assignment):
        d = torch.empty(self.V, 1, dtype=torch.float32, device=self.device)
        check(lib().pdp_energy_diff(self._h, ptr(assignment, torch.float32, self.V), ptr(d), _stream()))
        return d

    def set_rng_base(self, first_variable, first_instance):
        """this batch is a contiguous part of a larger forward: the Philox draws are counted from there (include/pdp_hip.h)"""
        check(lib().pdp_problem_set_rng_base(self._h, C.c_uint32(int(first_variable)), C.c_uint32(int(first_instance))))

    def set_exchange(self, fn):
        """``fn(mins, maxs, ors)``: three numpy uint32 arrays to be replaced, in place, by their element-wise minimum / maximum / bit-wise OR
        over all parts of a coupled forward solved by several processes (include/pdp_hip.h: pdp_problem_set_exchange); None removes it."""
        if fn is None:
            self._exchange_cb = None
            check(lib().pdp_problem_set_exchange(self._h, None, None))
            return
        import numpy as np

        def trampoline(user, mins, n_mins, maxs, n_maxs, ors, n_ors):
            try:
                view = lambda ptr, n: np.ctypeslib.as_array(ptr, shape=(n,)) if n > 0 else np.zeros(0, np.uint32)
                fn(view(mins, n_mins), view(maxs, n_maxs), view(ors, n_ors))
                return 0
            except Exception:                      # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1

        self._exchange_cb = EXCHANGE_FN(trampoline)          # (kept alive as long as the problem uses it)
        check(lib().pdp_problem_set_exchange(self._h, self._exchange_cb, None))

    def random_fill(self, values=None, seed=0):
        mode = RNG_STREAM if values is not None else RNG_PHILOX
        check(lib().pdp_random_fill(self._h, C.c_int(mode), ptr(values, torch.float32), C.c_uint64(seed), _stream()))

    def local_search(self, pred, iterations, epsilon, var_rand=None, coin_rand=None, seed=0):
        out = torch.empty(self.V, 1, dtype=torch.float32, device=self.device)
        steps = C.c_int32(0)
        mode = RNG_STREAM if var_rand is not None else RNG_PHILOX
        check(lib().pdp_local_search(self._h, ptr(pred, torch.float32, self.V, 'prediction'), C.c_int(iterations), C.c_float(epsilon),
                                     C.c_int(mode), ptr(var_rand, torch.float32), ptr(coin_rand, torch.float32),
                                     C.c_uint64(seed), ptr(out), C.byref(steps), _stream()))
        return out, int(steps.value)

    def deduplicate(self, pred):
        out = torch.empty(self.V // self.R, 1, dtype=torch.float32, device=self.device)
        chosen = torch.empty(self.B // self.R, dtype=torch.int32, device=self.device)
        check(lib().pdp_deduplicate(self._h, ptr(pred, torch.float32, self.V), ptr(out), ptr(chosen), _stream()))
        return out, chosen

    # -- neural plug-ins -------------------------------------------------------------------------------------------------
    def neural_aggregate_edges(self, agg_w, by_variable, state, edge_mask, active_mask, old, out=None):
        if out is None:
            out = torch.empty(self.E, agg_w.desc.out, dtype=torch.float32, device=self.device)
        check(lib().pdp_neural_aggregate_edges(self._h, C.byref(agg_w.desc), C.c_int(1 if by_variable else 0),
                                               ptr(state, torch.float32, self.E * (agg_w.desc.din - 1), 'state'),
                                               ptr(edge_mask, torch.float32, self.E, 'edge_mask'), ptr(active_mask, torch.uint8, self.B, 'active_mask'),
                                               ptr(old, torch.float32, self.E * agg_w.desc.out, 'init_state'),
                                               ptr(out, torch.float32, self.E * agg_w.desc.out, 'out'), _stream()))
        return out

    def neural_gru(self, gru_w, state, h, active_mask, out=None):
        if out is None:
            out = torch.empty(self.E, gru_w.desc.H, dtype=torch.float32, device=self.device)
        check(lib().pdp_neural_gru(self._h, C.byref(gru_w.desc), ptr(state, torch.float32, self.E * gru_w.desc.dx, 'message_state'),
                                   ptr(h, torch.float32, self.E * gru_w.desc.H, 'init_state'), ptr(active_mask, torch.uint8, self.B, 'active_mask'),
                                   ptr(out, torch.float32, self.E * gru_w.desc.H, 'out'), _stream()))
        return out

    def neural_predict(self, agg_w, head_w, state, edge_mask):
        out = torch.empty(self.V, 1, dtype=torch.float32, device=self.device)
        check(lib().pdp_neural_predict(self._h, C.byref(agg_w.desc), C.byref(head_w.desc), ptr(state, torch.float32, self.E * (agg_w.desc.din - 1), 'state'),
                                       ptr(edge_mask, torch.float32, self.E, 'edge_mask'), ptr(out), _stream()))
        return out

    # -- persistent solve -----------------------------------------------------------------------------------------
    def sp_solve(self, q, fs, active_mask, dec, iterations, tolerance, t_max, pi=0.0, model=MODEL_SP,
                 decimation_probability=0.5, seed=0, coins=None, check_termination=True, time_kernels=False, replicas_identical=False, isolate_instances=False,
                 inputs_disposable=False):
        a = SolveArgs()
        a.inputs_disposable = 1 if inputs_disposable else 0
        a.isolate_instances = 1 if isolate_instances else 0
        a.time_kernels = 1 if time_kernels else 0
        a.replicas_identical = 1 if replicas_identical else 0
        a.model = model; a.iterations = iterations; a.tolerance = tolerance; a.t_max = t_max; a.pi = pi
        a.decimation_probability = decimation_probability; a.seed = seed
        a.coins = ptr(coins, torch.float32).value if coins is not None else None
        a.q = ptr(q, torch.float32, 3 * self.E).value; a.fs = ptr(fs, torch.float32, 2 * self.E).value
        a.active_mask = ptr(active_mask, torch.uint8, self.B).value
        a.decimator = dec._h.value
        a.check_termination = 1 if check_termination else 0
        check(lib().pdp_sp_solve(self._h, C.byref(a), _stream()))
        self.last_solve_launches = int(a.kernel_launches_host)
        self.last_solve_stats = dict(launches=int(a.kernel_launches_host), replays=int(a.replay_launches_host),
                                     solve_kernel_ms=float(a.solve_kernel_ms_host), replay_kernel_ms=float(a.replay_kernel_ms_host),
                                     hbm_instances=int(a.hbm_instances_host))
        return int(a.iterations_run_host), bool(a.used_lds_host)


# -- the reference's L0 primitives on a sparse COO mask (include/pdp_hip.h: pdp_coo_*, pdp_csr_*) -------------------------------------------
def coo_reduce(kind, rows, cols, x, n_rows, n_cols):
    "util.sparse_max ('max') / sparse_argmax ('argmax') of x [nnz] paired with the entries (rows[i], cols[i]) of a [n_rows, n_cols] mask"
    require_gpu()
    nnz = int(rows.numel())
    scratch = torch.empty(n_cols + 2, dtype=torch.int64, device=x.device)
    out = torch.empty(n_cols, dtype=torch.float32 if kind == 'max' else torch.int64, device=x.device)
    fn = lib().pdp_coo_max if kind == 'max' else lib().pdp_coo_argmax
    check(fn(ptr(rows, torch.int64, nnz, 'mask rows'), ptr(cols, torch.int64, nnz, 'mask columns'), C.c_int64(nnz), ptr(x, torch.float32, nnz, 'x'),
             C.c_int64(n_rows), C.c_int64(n_cols), ptr(scratch), ptr(out), _stream()))
    return out


class CsrMask(object):
    "row offsets / columns / values of the row-sorted entries of a sparse mask [n_rows, n_cols] (built once per mask, cached on the tensor)"

    def __init__(self, mask):
        require_gpu()
        m = mask if mask.is_coalesced() else mask.coalesce()
        idx = m.indices()
        self.n_rows, self.n_cols = int(mask.size(0)), int(mask.size(1))
        self.rows = idx[0].contiguous()
        self.cols = idx[1].contiguous()
        self.vals = m.values().to(torch.float32).contiguous()
        self.nnz = int(self.cols.numel())
        self.row_ptr = torch.empty(self.n_rows + 1, dtype=torch.int64, device=mask.device)
        check(lib().pdp_coo_row_ptr(ptr(self.rows, torch.int64), C.c_int64(self.nnz), C.c_int64(self.n_rows), ptr(self.row_ptr), _stream()))

    def matmul(self, X, sub=None):
        "mask @ X (- sub): X [n_cols, d] dense"
        d = int(X.size(1))
        if X.size(0) != self.n_cols:
            raise NativeError("mask [%d, %d] times a matrix with %d rows" % (self.n_rows, self.n_cols, X.size(0)))
        out = torch.empty(self.n_rows, d, dtype=torch.float32, device=X.device)
        check(lib().pdp_csr_matmul(ptr(self.row_ptr), ptr(self.cols), ptr(self.vals), C.c_int64(self.n_rows), ptr(X, torch.float32, self.n_cols * d, 'X'),
                                   C.c_int(d), C.c_int64(d), ptr(sub, torch.float32, self.n_rows * d, 'sub') if sub is not None else None, ptr(out), _stream()))
        return out

    def smooth_max(self, x, alpha):
        out = torch.empty(self.n_rows, 1, dtype=torch.float32, device=x.device)
        check(lib().pdp_csr_smooth_max(ptr(self.row_ptr), ptr(self.cols), ptr(self.vals), C.c_int64(self.n_rows), ptr(x, torch.float32, self.n_cols, 'x'),
                                       C.c_float(alpha), ptr(out), _stream()))
        return out


def dimacs_parse(path):
    """One DIMACS file -> (var_num, clause_num, signed_vars int32 [E], clause_ids int32 [E]) in the compact conventions of the
    reference's converter (include/pdp_hip.h, pdp_dimacs_open).  Host code only: works without a GPU."""
    import numpy as np
    h = C.c_void_p()
    nv, nc, ne = C.c_int32(), C.c_int32(), C.c_int64()
    check(lib().pdp_dimacs_open(os.fsencode(path), C.byref(h), C.byref(nv), C.byref(nc), C.byref(ne)))
    try:
        sv = np.empty(ne.value, np.int32); ci = np.empty(ne.value, np.int32)
        check(lib().pdp_dimacs_read(h, sv.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p)))
    finally:
        lib().pdp_dimacs_close(h)
    return int(nv.value), int(nc.value), sv, ci


def dimacs_parse_many(paths, threads=8):
    """Several DIMACS files at once (parsed by host threads inside the library): list of dimacs_parse tuples in input order."""
    import numpy as np
    n = len(paths)
    if n == 0:
        return []
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    handles = (C.c_void_p * n)()
    nv = (C.c_int32 * n)(); nc = (C.c_int32 * n)(); ne = (C.c_int64 * n)()
    check(lib().pdp_dimacs_open_many(arr, C.c_int32(n), C.c_int32(threads), handles, nv, nc, ne))
    out = []
    try:
        for i in range(n):
            sv = np.empty(ne[i], np.int32); ci = np.empty(ne[i], np.int32)
            check(lib().pdp_dimacs_read(C.c_void_p(handles[i]), sv.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p)))
            out.append((int(nv[i]), int(nc[i]), sv, ci))
    finally:
        for i in range(n):
            if handles[i]:
                lib().pdp_dimacs_close(C.c_void_p(handles[i]))
    return out


TIMING_KEYS = ('agg_pre', 'row_sum', 'agg_post', 'gru', 'predict_head', 'walksat', 'sp_adaptors', 'sp_sweep')       # include/pdp_hip.h: PDP_TK_*


def kernel_timing(enable):
    "bracket the library's neural / Walk-SAT kernels with HIP events on their launch stream (measurement only)"
    global _TIMING_ON
    check(lib().pdp_kernel_timing(C.c_int(1 if enable else 0)))
    _TIMING_ON = bool(enable)


_TIMING_ON = False


def kernel_timing_enabled():
    "event pairs around every launch cannot be read back from a replayed graph: the solver's graph loop stands back while this is on"
    return _TIMING_ON


def kernel_timing_read():
    "{key: (summed device ms, launches)} since the last read; synchronises on the recorded events"
    ms = (C.c_float * len(TIMING_KEYS))(); n = (C.c_int32 * len(TIMING_KEYS))()
    check(lib().pdp_kernel_timing_read(ms, n))
    return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(TIMING_KEYS)}


NAME_KEYS = TIMING_KEYS + ('sp_solve', 'sp_replay')                                         # include/pdp_hip.h: PDP_KN_*


def kernel_name(key):
    "name, with template arguments, of the kernel the library launched last for this key ('' before the first launch)"
    buf = C.create_string_buffer(160)
    check(lib().pdp_kernel_name(C.c_int(NAME_KEYS.index(key)), buf, C.c_int(160)))
    return buf.value.decode('ascii')


def math_apply(fn, x):
    names = {'exp': 0, 'log': 1, 'logsigmoid': 2, 'sigmoid': 3, 'tanh': 4, 'safe_exp': 5, 'safe_log': 6, 'philox': 7, 'rcp': 8, 'safe_exp_fast': 9, 'safe_log_fin': 10, 'safe_log_fin_scorer': 11, 'exp_fin': 12, 'tanh_abs': 13, 'rcp_ge1': 14}
    require_gpu()
    y = torch.empty_like(x)
    check(lib().pdp_math_apply(C.c_int(names[fn]), ptr(x, torch.float32), ptr(y), C.c_int64(x.numel()), _stream()))
    return y
