// pdp_common.hpp -- handle layout, error plumbing and workgroup primitives shared by the kernels.
//
// Design: one workgroup per CNF instance.  Nothing in the PDP path couples two
// instances except a handful of batch-global reductions in the reference;
// those are separate tiny kernels here.  Instance topology is stored with instance-LOCAL ids so a
// workgroup can copy it into LDS verbatim (u16 when the instance is small enough).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pdp_hip.h"
#include "../../include/pdp_math.h"

#define PDP_WAVE 64
#define PDP_NT 256            // threads per workgroup of the step-wise kernels

// ---- error plumbing -------------------------------------------------------------------------
void pdp_set_error(const char *fmt, ...);

// ---- device memory: a small cache in front of hipMalloc / hipFree (pdp_problem.hip) --------------------------------------
// A batch costs ~60 allocations and as many frees; hipFree of a large block synchronises the device (7.5 ms per destroyed
// config-2 problem).  Freed blocks are kept (up to PDP_POOL_LIMIT_BYTES) and handed out again for requests of a similar size.
int pdp_dev_alloc(void **out, size_t bytes);
void pdp_dev_free(void *ptr);

// ---- optional kernel timing (pdp_kernel_timing, pdp_problem.hip): events on the launch stream around a scope ------------------------
void pdp_timing_mark(int key, hipStream_t st, bool begin);
extern int g_pdp_timing_on;
// name of the kernel last launched for a key (pdp_kernel_name): bench.py labels its per-kernel lines with what the library really ran
extern const char *g_pdp_kernel_name[];
inline void pdp_note_kernel(int key, const char *name) { g_pdp_kernel_name[key] = name; }
struct pdp_timed_scope {
    int key; hipStream_t st;
    pdp_timed_scope(int k, hipStream_t s) : key(k), st(s) { if (g_pdp_timing_on) pdp_timing_mark(key, st, true); }
    ~pdp_timed_scope() { if (g_pdp_timing_on) pdp_timing_mark(key, st, false); }
};

#define PDP_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            pdp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PDP_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

#define PDP_LAUNCH_CHECK()                                                                       \
    do {                                                                                         \
        hipError_t _e = hipGetLastError();                                                       \
        if (_e != hipSuccess) {                                                                  \
            pdp_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return PDP_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

#define PDP_REQUIRE(cond, msg)                                                                   \
    do { if (!(cond)) { pdp_set_error("%s (%s:%d)", msg, __FILE__, __LINE__); return PDP_ERR_INVALID; } } while (0)

// ---- device flag slots (uint32 words in pdp_problem::flags) -----------------------------------
enum {
    FL_LAYOUT_BAD = 0,
    FL_GMIN0 = 1,        // encoded global min #0
    FL_GMIN1 = 2,
    FL_GMIN2 = 3,
    FL_NAN0 = 4,         // NaN seen in the vector reduced into GMIN0
    FL_NAN1 = 5,
    FL_NAN2 = 6,
    FL_ANY_ACTIVE_VAR = 7,
    FL_ANY_CONV = 8,
    FL_ANY_POS = 9,
    FL_N_SEL = 10,
    FL_ACTIVE_EDGES = 11,
    FL_ANY_UNSAT = 12,
    FL_SPEC_VIOLATION = 13,
    FL_ITERS_RUN = 14,
    FL_WS_STEPS = 15,
    FL_PERM_ZERO = 16,   // first iteration from which an exited instance guarantees exact zeros
    FL_TEAM_TIMEOUT = 17,// sticky: a team barrier ran out of patience (the team's workgroups were not resident together); results are void
    FL_LOOP_STOP = 18,   // device-driven step-wise loop (pdp_loop_*): every instance has left the loop -- the state-writing kernels of later sweeps return at once
    FL_LOOP_ITERS = 19,  // ... sweeps executed before that
    FL_COUNT = 32
};

// ---- handle -------------------------------------------------------------------------------------
struct pdp_problem {
    int E, V, F, B, R;          // sizes of the (replicated) batch
    int E0, V0, F0, B0;         // sizes of one replica
    int max_n, max_m, max_e;    // largest instance
    int fn_edges_identity;      // edges are clause-major sorted (f_edges[k] == k)
    // topology
    int32_t *graph_map;         // [2,E] global ids (replicated)
    int32_t *var_inst, *fn_inst;// [V], [F]
    float *edge_sign;           // [E]
    int8_t *e_sgn;              // [E]
    int32_t *e_var, *e_fn;      // [E] instance-local ids
    int32_t *inst_v0, *inst_f0, *inst_e0;   // [B+1]
    int32_t *v_ptr, *v_edges;   // [V+B] local CSR offsets (instance b at v0+b, n+1 entries), [E] local edge ids
    int32_t *f_ptr, *f_edges;   // [F+B], [E]
    int32_t *nv_ptr, *nv_edges; // [V+1], [E] by-variable CSR with GLOBAL edge ids (neural row sums)
    int32_t *nf_ptr, *nf_edges; // [F+1], [E] by-clause
    // bound state (caller owned)
    float *av, *af, *sol, *is_sat, *emask;
    int has_edge_mask;          // refresh_edge_mask was called (sat_problem._edge_mask is not None)
    // workspace
    float *ws_e[4];             // [E] each
    float *ws_v[6];             // [V] each
    float *ws_f[2];             // [F] each
    float *ws_b[4];             // [B] each
    int64_t *ws_bi[2];          // [B] each
    int32_t *ws_vi[3];          // [V] each
    uint8_t *ws_fu[2];          // [F] each
    uint32_t *flags;            // [FL_COUNT] device
    uint32_t *flags_host;       // pinned host mirror
    void *cub_tmp; size_t cub_tmp_bytes;
    // persistent-solver scratch, allocated on first use and kept (hipMalloc/hipFree of ~0.5 GB cost milliseconds)
    char *solve_blob; size_t solve_blob_bytes;
    uint32_t *solve_host; size_t solve_host_words;   // pinned
    float *solve_extra_v;
    float *solve_rec;           // [E][4] per-edge records of the HBM-resident solver
    // LDS-resident solver: private instance records (pdp_solve.hip, BlobLayout) and device-side control blocks
    int64_t *res_stat_off;                           // [2B] static | dynamic record offsets
    char *res_stat; size_t res_stat_bytes; int res_static_built;
    char *res_dyn[2]; size_t res_dyn_bytes;
    float *res_prev_slots;
    char *res_ctl; size_t res_ctl_bytes;
    hipEvent_t *res_events; int res_events_n;       // 2 per chunk + 1, created on demand (pdp_solve_args.time_kernels)
    // k_simplify_lds: the instances' topology in the slot form it works on (16-bit words), kept from the second simplify() of a problem on
    uint16_t *simp_topo; int simp_calls;            // slot words | clause of slot | edge -> slot | row pointers, every instance's part 4-byte aligned
    // per-instance routing of the persistent solver: instances whose image fits the LDS / the others (HBM-resident kernel, same chunk loop)
    int32_t *res_fit_list, *res_big_list;            // device, [res_nfit] / [res_nbig] instance ids in ascending order
    uint8_t *res_is_big;                             // device, [B]
    int res_nfit, res_nbig, res_fit_n, res_fit_m, res_fit_e;   // counts; largest fitting instance
    char *res_big_snap; size_t res_big_snap_bytes;   // chunk-entry state of the big instances (the NaN-poison replay restarts them from it)
    // persistent Walk-SAT, per-instance routing (pdp_walksat.hip::ws_prepare)
    int ws_route_ready, ws_nfit, ws_nbig, ws_fit_n, ws_fit_m, ws_fit_e;
    int32_t *ws_fit_list, *ws_big_list; int64_t *ws_big_off; int64_t ws_big_E, ws_big_V, ws_big_F;
    hipStream_t ws_side_stream; hipEvent_t ws_side_ev[2];
    uint32_t *team_ws;          // barrier counters and reduction mailboxes of the workgroup teams (k_sp_solve<NT, true>)
    hipStream_t res_side_stream; hipEvent_t res_side_ev[2];   // the big instances' launches overlap the LDS-resident kernel on a stream of their own
    float *nws[4]; size_t nws_floats[4];             // neural workspaces (grow on demand)
    uint32_t rng_var_base, rng_inst_base;            // pdp_problem_set_rng_base: position of this batch inside the forward it is a part of
    // pdp_problem_set_exchange: the batch is a part of a COUPLED forward solved by several processes -- the reductions the reference takes over
    // the whole batch are completed across the parts through this host callback (element-wise min / max / or over all parts, in place)
    int (*exchange)(void *user, uint32_t *mins, int n_mins, uint32_t *maxs, int n_maxs, uint32_t *ors, int n_ors);
    void *exchange_user;
    uint32_t *exchange_host; size_t exchange_host_words;   // pinned staging block of the callback's arrays
};

struct pdp_decimator {
    pdp_problem *p;
    float *prev;                // [E] SequentialDecimator._previous_function_state
    float *counters;            // [B] SequentialDecimator._counters
    int has_prev;
};

// ---- by-value view handed to kernels -----------------------------------------------------------
struct PView {
    const int32_t *inst_v0, *inst_f0, *inst_e0;
    const int32_t *e_var, *e_fn, *v_ptr, *v_edges, *f_ptr, *f_edges;
    const int8_t *e_sgn;
    const int32_t *var_inst, *fn_inst;
    float *av, *af, *sol, *is_sat, *emask;
    uint32_t *flags;
    int B, R, B0, V, F, E;
    uint32_t rng_v0, rng_b0;    // added to the variable / instance index of every Philox counter (0 unless the batch is a part of a forward)
};

static inline PView make_view(const pdp_problem *p)
{
    PView v;
    v.inst_v0 = p->inst_v0; v.inst_f0 = p->inst_f0; v.inst_e0 = p->inst_e0;
    v.e_var = p->e_var; v.e_fn = p->e_fn; v.v_ptr = p->v_ptr; v.v_edges = p->v_edges;
    v.f_ptr = p->f_ptr; v.f_edges = p->f_edges; v.e_sgn = p->e_sgn;
    v.var_inst = p->var_inst; v.fn_inst = p->fn_inst;
    v.av = p->av; v.af = p->af; v.sol = p->sol; v.is_sat = p->is_sat; v.emask = p->emask;
    v.flags = p->flags; v.B = p->B; v.R = p->R; v.B0 = p->B0; v.V = p->V; v.F = p->F; v.E = p->E;
    v.rng_v0 = p->rng_var_base; v.rng_b0 = p->rng_inst_base;
    return v;
}

// One instance as seen by its workgroup (pointers already offset to the instance's slice).
struct Inst {
    int b, v0, f0, e0, n, m, e;
    const int32_t *e_var, *e_fn, *v_ptr, *v_edges, *f_ptr, *f_edges;
    const int8_t *sgn;
    float *av, *af, *sol, *emask;
};

__device__ __forceinline__ Inst load_inst(const PView &pv, int b)
{
    Inst I;
    I.b = b;
    I.v0 = pv.inst_v0[b]; I.f0 = pv.inst_f0[b]; I.e0 = pv.inst_e0[b];
    I.n = pv.inst_v0[b + 1] - I.v0; I.m = pv.inst_f0[b + 1] - I.f0; I.e = pv.inst_e0[b + 1] - I.e0;
    I.e_var = pv.e_var + I.e0; I.e_fn = pv.e_fn + I.e0; I.sgn = pv.e_sgn + I.e0;
    I.v_ptr = pv.v_ptr + I.v0 + b; I.v_edges = pv.v_edges + I.e0;
    I.f_ptr = pv.f_ptr + I.f0 + b; I.f_edges = pv.f_edges + I.e0;
    I.av = pv.av + I.v0; I.af = pv.af + I.f0; I.sol = pv.sol + I.v0;
    I.emask = pv.emask ? pv.emask + I.e0 : nullptr;
    return I;
}

// ---- ordered encoding of floats for atomicMin ------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t pdp_enc_ordered(float f)
{
    const uint32_t u = pdp_f2bits(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float pdp_dec_ordered(uint32_t k)
{
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return pdp_bits2f(u);
}
#define PDP_ENC_PLUS_INF 0xff800000u   /* enc(+inf) */

// ---- workgroup primitives ----------------------------------------------------------------------------
// All take a small LDS scratch array (>= 2 * nwaves entries of the payload type) and must be called by
// every thread of the workgroup.  They end with a barrier, so the scratch can be reused immediately.

// Wave-level reduction with DPP moves (a __shfl_down is a ds_bpermute: an LDS round trip per step): Hillis-Steele steps inside the
// 16-lane rows, then the lane-15 / lane-31 broadcasts.  The total ends up in LANE 63.  T is a 32-bit type.
template <int CTRL, int ROW_MASK, typename T, typename Op>
__device__ __forceinline__ T wave_dpp_step(T v, Op op, T identity)
{
    static_assert(sizeof(T) == 4, "32-bit types only");
    int vi, idi;
    __builtin_memcpy(&vi, &v, 4); __builtin_memcpy(&idi, &identity, 4);
    const int oi = __builtin_amdgcn_update_dpp(idi, vi, CTRL, ROW_MASK, 0xf, false);      // lanes without a source see the identity
    T o;
    __builtin_memcpy(&o, &oi, 4);
    return op(v, o);
}
template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce(T v, Op op, T identity)
{
    v = wave_dpp_step<0x111, 0xf>(v, op, identity);      // row_shr:1
    v = wave_dpp_step<0x112, 0xf>(v, op, identity);      // row_shr:2
    v = wave_dpp_step<0x114, 0xf>(v, op, identity);      // row_shr:4
    v = wave_dpp_step<0x118, 0xf>(v, op, identity);      // row_shr:8
    v = wave_dpp_step<0x142, 0xa>(v, op, identity);      // row_bcast:15
    v = wave_dpp_step<0x143, 0xc>(v, op, identity);      // row_bcast:31
    return v;
}

// (nt: the workgroup size, for callers that must not fetch blockDim -- out-of-line device functions read it from the dispatch packet in memory)
template <typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, Op op, T identity, T *scratch, int nt = (int)blockDim.x)
{
    const int lane = threadIdx.x & (PDP_WAVE - 1), wid = threadIdx.x / PDP_WAVE;
    const int nw = (nt + PDP_WAVE - 1) / PDP_WAVE;
    v = wave_reduce(v, op, identity);
    if (lane == PDP_WAVE - 1) scratch[wid] = v;
    __syncthreads();
    T r = identity;
    for (int i = 0; i < nw; ++i) r = op(r, scratch[i]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ bool pdp_finite(float x) { return (x - x) == 0.0f; }      // neither NaN nor +-inf
struct OpMaxNan { __device__ float operator()(float a, float b) const { return pdp_max(a, b); } };
struct OpMinNan { __device__ float operator()(float a, float b) const { return pdp_min(a, b); } };
struct OpAddI { __device__ int operator()(int a, int b) const { return a + b; } };
struct OpOrI { __device__ int operator()(int a, int b) const { return a | b; } };
struct OpMinI { __device__ int operator()(int a, int b) const { return a < b ? a : b; } };
struct OpMaxI { __device__ int operator()(int a, int b) const { return a > b ? a : b; } };

// arg-max with torch.argmax semantics: larger value wins, NaN is maximal, first index wins ties.
struct ArgPair { float v; int i; };
__device__ __forceinline__ bool arg_better(float av, int ai, float bv, int bi)
{
    if (ai < 0) return false;
    if (bi < 0) return true;
    const bool an = av != av, bn = bv != bv;
    if (an || bn) { if (an && bn) return ai < bi; return an; }
    if (av > bv) return true;
    if (av < bv) return false;
    return ai < bi;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void argmax_dpp_step(float &v, int &i)
{
    const int ov = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    const int oi = __builtin_amdgcn_update_dpp(-1, i, CTRL, ROW_MASK, 0xf, false);          // lanes without a source see index -1: never better
    if (arg_better(__int_as_float(ov), oi, v, i)) { v = __int_as_float(ov); i = oi; }
}
__device__ __forceinline__ ArgPair block_argmax(float v, int i, float *sv, int *si)
{
    const int lane = threadIdx.x & (PDP_WAVE - 1), wid = threadIdx.x / PDP_WAVE;
    const int nw = (blockDim.x + PDP_WAVE - 1) / PDP_WAVE;
    argmax_dpp_step<0x111, 0xf>(v, i); argmax_dpp_step<0x112, 0xf>(v, i); argmax_dpp_step<0x114, 0xf>(v, i);
    argmax_dpp_step<0x118, 0xf>(v, i); argmax_dpp_step<0x142, 0xa>(v, i); argmax_dpp_step<0x143, 0xc>(v, i);     // total in lane 63
    if (lane == PDP_WAVE - 1) { sv[wid] = v; si[wid] = i; }
    __syncthreads();
    ArgPair r; r.v = sv[0]; r.i = si[0];
    for (int k = 1; k < nw; ++k) if (arg_better(sv[k], si[k], r.v, r.i)) { r.v = sv[k]; r.i = si[k]; }
    __syncthreads();
    return r;
}

// arg-max of (value, index) pairs as the maximum of 64-bit keys: order-preserving bits of the (non-NaN) value on top, inverted
// index below, so the larger value wins and the first index wins ties (util.sparse_argmax); 0 = "no candidate".
__device__ __forceinline__ unsigned long long argkey(float t, int v)
{
    const uint32_t b = __float_as_uint(t);
    const uint32_t ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)ord << 32) | (uint32_t)(0xffffffffu - (uint32_t)v);
}
__device__ __forceinline__ int argkey_index(unsigned long long k) { return k ? (int)(0xffffffffu - (uint32_t)k) : -1; }

// wave-level maximum with DPP moves (a __shfl_down is an LDS round trip of ~100 cycles, a DPP move a few): Hillis-Steele steps
// inside the 16-lane rows, then lane 15 / lane 31 broadcasts; the result is in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long k)
{
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k >> 32), CTRL, ROW_MASK, 0xf, false);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k, CTRL, ROW_MASK, 0xf, false);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;      // lanes without a source see 0
    return o > k ? o : k;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
    k = dpp_max_u64<0x111, 0xf>(k);      // row_shr:1
    k = dpp_max_u64<0x112, 0xf>(k);      // row_shr:2
    k = dpp_max_u64<0x114, 0xf>(k);      // row_shr:4
    k = dpp_max_u64<0x118, 0xf>(k);      // row_shr:8   -> lane 15 of every row holds its row's maximum
    k = dpp_max_u64<0x142, 0xa>(k);      // row_bcast:15 into rows 1 and 3
    k = dpp_max_u64<0x143, 0xc>(k);      // row_bcast:31 into rows 2 and 3 -> lane 63
    return k;
}


// block-wide arg-max through keys (values must not be NaN): every thread gets the winning key
__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long k, unsigned long long *scratch /*[>= waves]*/)
{
    const int lane = threadIdx.x & (PDP_WAVE - 1), wid = threadIdx.x / PDP_WAVE;
    const int nw = (blockDim.x + PDP_WAVE - 1) / PDP_WAVE;
    k = wave_max_u64(k);
    if (lane == PDP_WAVE - 1) scratch[wid] = k;
    __syncthreads();
    unsigned long long r = scratch[0];
    for (int i = 1; i < nw; ++i) r = scratch[i] > r ? scratch[i] : r;
    __syncthreads();
    return r;
}

// number of LDS scratch floats the primitives need
#define PDP_RED_SCRATCH 64

// ---- execution team of a per-instance routine ---------------------------------------------------------------------
// The per-instance device routines (pdp_device.hpp) are written against these six calls.  For every instance view but
// TeamView the team IS the workgroup and they are the plain block primitives; a TeamView (pdp_solve.hip) spreads one big
// instance over several workgroups and overloads them with device-scope versions.
template <class I> __device__ __forceinline__ int team_tid(const I &) { return threadIdx.x; }
template <class I> __device__ __forceinline__ int team_nt(const I &) { return blockDim.x; }
template <class I> __device__ __forceinline__ void team_sync(const I &) { __syncthreads(); }
template <class I> __device__ __forceinline__ int team_any(const I &, int x) { return __syncthreads_or(x); }
template <class I, typename T, typename Op>
__device__ __forceinline__ T team_reduce(const I &, T v, Op op, T identity, T *scratch) { return block_reduce(v, op, identity, scratch); }
template <class I>
__device__ __forceinline__ ArgPair team_argmax(const I &, float v, int i, float *sv, int *si) { return block_argmax(v, i, sv, si); }

// One big instance spread over `size` workgroups (the team): thread ids run over the whole team, barriers and reductions
// are device-scope.  Everything a team barrier orders must live in HBM.  Teamed<V> adds the team's state to an instance view V; the
// overloads below are more specialised than the generic team_* calls above, so the per-instance routines pick them up.
template <class Base>
struct Teamed : Base {
    int rank, size;             // this workgroup's place in the team
    int same_xcd;               // 1: every workgroup of the team reported the same XCC id (checked at kernel start); 0: agent-scope fences; 2: mailboxes only
    uint32_t *bar;              // arrival counter, zeroed before the launch; it only grows: barrier k is complete at k * size arrivals
    uint32_t *box;              // [2][size][PDP_BOX_WORDS] reduction mailboxes, alternating with the parity of the barrier they ride on
    mutable uint32_t epoch;     // team barriers passed (identical on every thread of the team)
    // A team's workgroups wait for each other, so all of them must be resident at once.  The launches are sized for that (pdp_team_plan,
    // lockstep_possible), but nothing GUARANTEES it -- another process on the GPU, a changed occupancy -- and a barrier that can never
    // complete would hang the process.  The wait is therefore bounded: after spin_limit polls (seconds; a legitimate wait is micro- to
    // milliseconds) the barrier gives up, marks the team's workspace so that its other workgroups give up at their next poll, and the team
    // is `failed` from then on: barriers return at once, reductions return their identity (every fix-point loop ends), the kernel runs
    // to its end and reports the failure through its violation / FL_TEAM_TIMEOUT word; the host restores the state and fails over.
    mutable uint32_t failed;
    uint32_t spin_limit;
};
#define PDP_SPIN_LIMIT_DEFAULT (1u << 24)                // polls of ~0.5 us each
#define PDP_TEAM_MAX 256                                 // team size of one instance (mailbox reads: thread r folds ranks r, r + blockDim, ...)
#define PDP_LOCK_MAX 1024                                // workgroups of a lock-step launch (one instance each)
#define PDP_BOX_WORDS 8                                   // words of one rank's mailbox
#define PDP_TEAM_WORDS (32 + 2 * PDP_TEAM_MAX * PDP_BOX_WORDS)      // words of team workspace per instance: the counter on a 128 B line of its own, then the mailboxes

template <class B> __device__ __forceinline__ int team_tid(const Teamed<B> &t) { return t.rank * (int)blockDim.x + (int)threadIdx.x; }
template <class B> __device__ __forceinline__ int team_nt(const Teamed<B> &t) { return t.size * (int)blockDim.x; }
// Team barrier.  Workgroups on DIFFERENT XCDs only see each other's stores through agent-scope release / acquire fences, which
// write back and invalidate the XCD's whole L2 -- tens of microseconds next to a kernel that streams instance records.  The
// launch numbers the workgroups so that a team lands on ONE XCD (k_sp_solve), every team verifies that at its first barrier,
// and then the shared L2 is the point of coherence: a store is visible once it left the write-through vector cache
// (s_waitcnt vmcnt(0)), and a reader only has to drop its CU's vector cache (buffer_inv).
template <class B> __device__ __forceinline__ void team_sync(const Teamed<B> &t)
{
    if (t.size == 1 || t.failed) { __syncthreads(); return; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // this wave's stores have left the CU's write-through vector cache
    __syncthreads();
    t.epoch += 1;
    int gave_up = 0;
    if (threadIdx.x < PDP_WAVE) {
        // One wave speaks for the workgroup.  Team on several XCDs: the release writes this XCD's L2 back (every wave's stores are in it
        // by now), the acquire drops the vector cache and the stale L2 lines.  Team on one XCD: the L2 is the point of coherence, only
        // the CU's vector cache has to go (sixteen waves doing that cost ~7 us per barrier, one wave well under 1).
        if (!t.same_xcd) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(t.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t target = t.epoch * (uint32_t)t.size;
            uint32_t polls = 0;
            while (__hip_atomic_load(t.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                // every 1024th poll: out of patience, or has another workgroup of the team given up already (word 1 of the counter's line)?
                if ((++polls & 0x3ffu) == 0u && (polls >= t.spin_limit || __hip_atomic_load(t.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    __hip_atomic_store(t.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gave_up = 1;
                    break;
                }
            }
        }
        if (t.same_xcd == 1) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        else if (!t.same_xcd) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // (same_xcd == 2: the workgroups exchange mailbox words only -- agent-scope atomics -- and no cache has to be touched)
    }
    if (__syncthreads_or(gave_up)) t.failed = 1u;                       // workgroup-uniform
}
// A mailbox written for the barrier of epoch e is read right after that barrier; the next writer of the same half has passed
// barrier e + 1, which every workgroup only reaches after its reads.  Thread r of every workgroup fetches rank r's mailbox (all
// fetches in flight together), a block reduction folds them.
template <class B> __device__ __forceinline__ uint32_t *team_box(const Teamed<B> &t) { return t.box + (size_t)(t.epoch & 1u) * t.size * PDP_BOX_WORDS; }
__device__ __forceinline__ void box_put(uint32_t *w, uint32_t v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t box_get(const uint32_t *w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class B, typename T, typename Op>
__device__ __forceinline__ T team_reduce(const Teamed<B> &t, T v, Op op, T identity, T *scratch)
{
    static_assert(sizeof(T) == 4, "mailboxes hold 32-bit values");
    v = block_reduce(v, op, identity, scratch);
    if (t.size == 1) return v;
    uint32_t *box = team_box(t);
    if (threadIdx.x == 0) box_put(&box[PDP_BOX_WORDS * t.rank], __builtin_bit_cast(uint32_t, v));
    team_sync(t);
    if (t.failed) return identity;
    T theirs = identity;          // thread r folds the mailboxes of ranks r, r + blockDim, ...
    for (int r = (int)threadIdx.x; r < t.size; r += (int)blockDim.x) theirs = op(theirs, __builtin_bit_cast(T, box_get(&box[PDP_BOX_WORDS * r])));
    return block_reduce(theirs, op, identity, scratch);
}
template <class B> __device__ __forceinline__ int team_any(const Teamed<B> &t, int x)      // like __syncthreads_or: is x non-zero anywhere (NOT the bitwise or)
{
    x = __syncthreads_or(x);
    if (t.size == 1) return x;
    uint32_t *box = team_box(t);
    if (threadIdx.x == 0) box_put(&box[PDP_BOX_WORDS * t.rank], (uint32_t)x);
    team_sync(t);
    if (t.failed) return 0;                                            // "nothing changed anywhere": every fix-point loop of a failed team ends
    int theirs = 0;
    for (int r = (int)threadIdx.x; r < t.size; r += (int)blockDim.x) theirs |= (int)(box_get(&box[PDP_BOX_WORDS * r]) != 0u);
    return __syncthreads_or(theirs);
}
template <class B> __device__ __forceinline__ ArgPair team_argmax(const Teamed<B> &t, float v, int i, float *sv, int *si)
{
    ArgPair r = block_argmax(v, i, sv, si);
    if (t.size == 1) return r;
    uint32_t *box = team_box(t);
    if (threadIdx.x == 0) { box_put(&box[PDP_BOX_WORDS * t.rank], __float_as_uint(r.v)); box_put(&box[PDP_BOX_WORDS * t.rank + 1], (uint32_t)r.i); }
    team_sync(t);
    if (t.failed) { ArgPair none; none.v = 0.0f; none.i = -1; return none; }
    float ov = 0.0f; int oi = -1;
    for (int r = (int)threadIdx.x; r < t.size; r += (int)blockDim.x) {
        const float rv = __uint_as_float(box_get(&box[PDP_BOX_WORDS * r])); const int ri = (int)box_get(&box[PDP_BOX_WORDS * r + 1]);
        if (arg_better(rv, ri, ov, oi)) { ov = rv; oi = ri; }
    }
    return block_argmax(ov, oi, sv, si);
}
// Places this workgroup in its team (launch numbering: slot-minor over `slots`) and makes the first barrier: with full agent-scope
// fences, to find out whether the whole team sits on one XCD (HW_REG_XCC_ID = 20, bits 3:0).  Returns the slot of the launch, or -1 for a
// padding workgroup (one-XCD teams pad the slot count to the XCD count so that a team's workgroups share an XCD).
struct TeamLaunch { int size, count, slots, no_xcd; uint32_t *ws; uint32_t spin_limit; };
uint32_t pdp_spin_limit();                     // PDP_TEAM_SPIN_LIMIT (polls) or the default: how long a team barrier waits before it gives up
struct pdp_problem;
int pdp_simplify_lds(pdp_problem *p, hipStream_t st);      // pdp_solve.hip: simplify() with the instances in LDS; 0 if the batch does not qualify
int pdp_device_cus();                         // CUs of the current device (workgroups that are certainly resident together)
int pdp_exchange_call(pdp_problem *p, const uint32_t *mins, int n_mins, const uint32_t *maxs, int n_maxs, const uint32_t *ors, int n_ors, uint32_t **out);   // pdp_problem.hip
int pdp_edge_rows(const pdp_problem *p);     // workgroups per instance of the flat per-edge kernels (gridDim.y): 1 unless an instance is big
int pdp_team_plan(pdp_problem *p, int count, bool wide, int threads, TeamLaunch *out, hipStream_t st);
template <class B>
__device__ __forceinline__ int team_begin(Teamed<B> &t, const TeamLaunch &tl, int *redi)
{
    const int slot = (int)blockIdx.x % tl.slots;
    if (slot >= tl.count) return -1;
    t.rank = (int)blockIdx.x / tl.slots; t.size = tl.size; t.epoch = 0; t.failed = 0u;
    t.spin_limit = tl.spin_limit ? tl.spin_limit : PDP_SPIN_LIMIT_DEFAULT;
    t.bar = tl.ws + (size_t)slot * PDP_TEAM_WORDS; t.box = t.bar + 32;
    t.same_xcd = 0;
    const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf);
    const int seen = team_reduce(t, 1 << xcc, OpOrI(), 0, redi);
    t.same_xcd = ((seen & (seen - 1)) == 0 && !tl.no_xcd) ? 1 : 0;
    return slot;
}


