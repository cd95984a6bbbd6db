// pdp_walksat.hip -- Walk-SAT post-processing, random fill and replica de-duplication.
// replaces: PropagatorDecimatorSolverBase._local_search / _compute_energy / _compute_energy_diff /
// _deduplicate (reference: src/pdp/nn/solver.py:401-496) and IdentityPredictor's random fill
// (src/pdp/nn/pdp_predict.py:118-128).  One workgroup per instance; all quantities except the
// uniform draws are small integers.
#include "pdp_device.hpp"
#include <type_traits>
#include <vector>

#include <hipcub/hipcub.hpp>
#include <stdlib.h>
#include <string.h>

#define ST(s) ((hipStream_t)(s))
#define DECL_RED __shared__ float redf[PDP_RED_SCRATCH]; __shared__ int redi[PDP_RED_SCRATCH];

static inline int grid_for(int64_t n, int nt = 256) { int64_t g = (n + nt - 1) / nt; if (g < 1) g = 1; if (g > 4096) g = 4096; return (int)g; }

// ---- energy ---------------------------------------------------------------------------------------
// per clause: agg = sum s * (a * active), deg = sum active; unsat = (agg == -deg) * active_fn
template <class I>
__device__ __forceinline__ int d_clause_energy(const I &in, const float *a /*[n]*/, float *agg_out, float *deg_out, uint8_t *unsat_out)
{
    int cnt = 0;
    for (int c = threadIdx.x; c < in.m; c += blockDim.x) {
        float agg = 0.0f, deg = 0.0f;
        for (int k = in.f_ptr[c]; k < in.f_ptr[c + 1]; ++k) {
            const int e = in.f_edges[k];
            const int v = in.e_var[e];
            agg = agg + (0.0f + (float)in.sgn[e] * (a[v] * in.av[v]));
            deg = deg + (0.0f + in.av[v]);
        }
        const float u = ((agg == -deg) ? 1.0f : 0.0f) * in.af[c];
        if (agg_out) { agg_out[c] = agg; deg_out[c] = deg; }
        if (unsat_out) unsat_out[c] = (u == 1.0f) ? 1 : 0;
        cnt += (u == 1.0f) ? 1 : 0;
    }
    return cnt;
}

__global__ void __launch_bounds__(PDP_NT) k_energy(PView pv, const float *assignment, float *energy, float *unsat_fn, uint8_t *unsat_u8,
                                                   float *agg_ws, float *deg_ws)
{
    DECL_RED
    (void)redf;
    const Inst I = load_inst(pv, blockIdx.x);
    int cnt = d_clause_energy(I, assignment + I.v0, agg_ws + I.f0, deg_ws + I.f0, unsat_u8 + I.f0);
    cnt = block_reduce(cnt, OpAddI(), 0, redi);
    if (unsat_fn) for (int c = threadIdx.x; c < I.m; c += blockDim.x) unsat_fn[I.f0 + c] = (float)unsat_u8[I.f0 + c];
    if (threadIdx.x == 0) energy[I.b] = (float)cnt;
}

extern "C" int pdp_energy(pdp_problem *p, const float *assignment, float *energy, float *unsat_functions, void *stream)
{
    PDP_REQUIRE(p && p->av && assignment && energy, "NULL argument / state not bound");
    hipLaunchKernelGGL(k_energy, dim3(p->B), dim3(PDP_NT), 0, ST(stream), make_view(p), assignment, energy, unsat_functions, p->ws_fu[0],
                       p->ws_f[0], p->ws_f[1]);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// delta[v] = sum over edges of critical * dist (solver.py:469-484); needs agg/deg per clause
template <class I>
__device__ __forceinline__ float d_var_delta(const I &in, int v, const float *a, const float *agg, const float *deg, const float *emask)
{
    float delta = 0.0f;
    const float dist_v = a[v] * in.av[v];
    for (int k = in.v_ptr[v]; k < in.v_ptr[v + 1]; ++k) {
        const int e = in.v_edges[k];
        const int c = in.e_fn[e];
        const float dist = 0.0f + (float)in.sgn[e] * dist_v;
        const float others = (0.0f + agg[c]) - dist;
        const float fd = 0.0f + deg[c];
        const float critical = ((others == (1.0f - fd)) ? 1.0f : 0.0f) * emask[e];
        delta = delta + critical * dist;
    }
    return delta;
}

__global__ void __launch_bounds__(PDP_NT) k_energy_diff(PView pv, const float *assignment, float *delta, float *agg_ws, float *deg_ws)
{
    const Inst I = load_inst(pv, blockIdx.x);
    d_clause_energy(I, assignment + I.v0, agg_ws + I.f0, deg_ws + I.f0, (uint8_t *)nullptr);
    __syncthreads();
    for (int v = threadIdx.x; v < I.n; v += blockDim.x)
        delta[I.v0 + v] = d_var_delta(I, v, assignment + I.v0, agg_ws + I.f0, deg_ws + I.f0, I.emask);
}

extern "C" int pdp_energy_diff(pdp_problem *p, const float *assignment, float *delta, void *stream)
{
    PDP_REQUIRE(p && p->av && assignment && delta, "NULL argument / state not bound");
    hipLaunchKernelGGL(k_energy_diff, dim3(p->B), dim3(PDP_NT), 0, ST(stream), make_view(p), assignment, delta, p->ws_f[0], p->ws_f[1]);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- random fill ---------------------------------------------------------------------------------------
__global__ void k_active_flags(int V, const float *av, int32_t *flag)
{
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) flag[v] = (av[v] > 0.0f) ? 1 : 0;
}
__global__ void k_fill_stream(int V, const float *av, const int32_t *rank, const float *values, float *sol)
{
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x)
        if (av[v] > 0.0f) sol[v] = values[rank[v]];
}
__global__ void k_fill_philox(int V, const float *av, uint64_t seed, uint32_t base, float *sol)
{
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x)
        if (av[v] > 0.0f) sol[v] = pdp_philox_uniform(seed, PDP_RNG_STREAM_FILL, 0u, base + (uint32_t)v);
}

extern "C" int pdp_random_fill(pdp_problem *p, int rng_mode, const float *values, uint64_t seed, void *stream)
{
    PDP_REQUIRE(p && p->av, "NULL argument / state not bound");
    hipStream_t st = ST(stream);
    if (rng_mode == PDP_RNG_PHILOX) {
        hipLaunchKernelGGL(k_fill_philox, dim3(grid_for(p->V)), dim3(256), 0, st, p->V, p->av, seed, p->rng_var_base, p->sol);
    } else {
        PDP_REQUIRE(values, "stream mode needs the drawn values");
        hipLaunchKernelGGL(k_active_flags, dim3(grid_for(p->V)), dim3(256), 0, st, p->V, p->av, p->ws_vi[0]);
        size_t need = 0;
        PDP_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, p->ws_vi[0], p->ws_vi[1], p->V, st));
        if (need > p->cub_tmp_bytes) {
            if (p->cub_tmp) pdp_dev_free(p->cub_tmp);
            { int st_ = pdp_dev_alloc(&p->cub_tmp, need); if (st_ != PDP_OK) return st_; }
            p->cub_tmp_bytes = need;
        }
        PDP_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(p->cub_tmp, need, p->ws_vi[0], p->ws_vi[1], p->V, st));
        hipLaunchKernelGGL(k_fill_stream, dim3(grid_for(p->V)), dim3(256), 0, st, p->V, p->av, p->ws_vi[1], values, p->sol);
    }
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- Walk-SAT, strict step-wise form ---------------------------------------------------------------------
__global__ void k_ws_init(int V, const float *av, const float *pred, float *a)
{
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
        const float bit = (pred[v] > 0.5f) ? 1.0f : 0.0f;
        a[v] = av[v] * (2.0f * bit - 1.0f);
    }
}

// (a) energies; unsat flag per (replica) instance
__global__ void __launch_bounds__(PDP_NT) k_ws_energy(PView pv, const float *a, float *unsat_b, uint8_t *unsat_u8, float *agg_ws, float *deg_ws)
{
    DECL_RED
    (void)redf;
    const Inst I = load_inst(pv, blockIdx.x);
    int cnt = d_clause_energy(I, a + I.v0, agg_ws + I.f0, deg_ws + I.f0, unsat_u8 + I.f0);
    cnt = block_reduce(cnt, OpAddI(), 0, redi);
    if (threadIdx.x == 0) {
        unsat_b[I.b] = (cnt > 0) ? 1.0f : 0.0f;
        if (pv.R == 1 && cnt > 0) atomicOr(&pv.flags[FL_ANY_UNSAT], 1u);
    }
}
// (a') replication: an original instance counts as unsat while ALL its replicas are unsat (solver.py:446-449)
__global__ void k_ws_compact(int B0, int R, const float *unsat_b, uint32_t *flags)
{
    const int b0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (b0 >= B0) return;
    float s = 0.0f;
    for (int r = 0; r < R; ++r) s = s + (1.0f - unsat_b[b0 + r * B0]);
    if (!(s > 0.0f)) atomicOr(&flags[FL_ANY_UNSAT], 1u);
}

// (b) delta energy + random candidates; batch-global min of the candidate vector
__global__ void __launch_bounds__(PDP_NT) k_ws_delta(PView pv, const float *a, const uint8_t *unsat_u8, const float *agg_ws, const float *deg_ws,
                                                     int rng_mode, const float *var_rand, uint64_t seed, int step, float *negdelta, float *uv)
{
    DECL_RED
    const Inst I = load_inst(pv, blockIdx.x);
    float m = PDP_INF; bool nn = false;
    for (int v = threadIdx.x; v < I.n; v += blockDim.x) {
        negdelta[I.v0 + v] = -d_var_delta(I, v, a + I.v0, agg_ws + I.f0, deg_ws + I.f0, I.emask);
        float acc = 0.0f;
        for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) acc = acc + (float)unsat_u8[I.f0 + I.e_fn[I.v_edges[k]]];
        acc = acc * I.av[v];
        const float u = (rng_mode == PDP_RNG_STREAM) ? var_rand[I.v0 + v]
                                                     : pdp_philox_uniform(seed, PDP_RNG_STREAM_WSVAR, (uint32_t)step, pv.rng_v0 + (uint32_t)(I.v0 + v));
        const float r = ((acc > 0.0f) ? 1.0f : 0.0f) * u;
        uv[I.v0 + v] = r;
        if (r != r) nn = true; else if (r < m) m = r;
    }
    publish_min(m, nn, pv.flags, FL_GMIN0, FL_NAN0, redf, redi);
}

// (c) choose and flip (solver.py:454-465)
__global__ void __launch_bounds__(PDP_NT) k_ws_flip(PView pv, float *a, const float *unsat_b, const float *negdelta, const float *uv,
                                                    int rng_mode, const float *coin_rand, uint64_t seed, int step, float epsilon)
{
    DECL_RED
    const Inst I = load_inst(pv, blockIdx.x);
    // -delta holds small integers: the (x - min + 1) shift is exact for any min, so a local shift of 0 gives
    // the reference's arg-max; the random candidates need the batch-global min (rounding of x - min + 1)
    const int greedy = d_instance_argmax(I, negdelta + I.v0, 0.0f, redf, redi);
    const float gmin = pdp_dec_ordered(pv.flags[FL_GMIN0]);
    const int randi = d_instance_argmax(I, uv + I.v0, pv.flags[FL_NAN0] ? PDP_NAN : gmin, redf, redi);
    if (threadIdx.x == 0 && unsat_b[I.b] > 0.0f && I.n > 0) {
        const float u = (rng_mode == PDP_RNG_STREAM) ? coin_rand[I.b]
                                                     : pdp_philox_uniform(seed, PDP_RNG_STREAM_WSCOIN, (uint32_t)step, pv.rng_b0 + (uint32_t)I.b);
        const int ind = (u > epsilon) ? greedy : randi;
        a[I.v0 + ind] = -a[I.v0 + ind];
    }
}

__global__ void k_ws_finish(int V, const float *a, float *out)
{
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) out[v] = (a[v] + 1.0f) / 2.0f;
}

__global__ void k_ws_reset(uint32_t *flags)
{
    if (threadIdx.x == 0) { flags[FL_ANY_UNSAT] = 0u; flags[FL_GMIN0] = PDP_ENC_PLUS_INF; flags[FL_NAN0] = 0u; }
}

__global__ void __launch_bounds__(PDP_NT) k_edge_mask2(PView pv)
{
    const Inst I = load_inst(pv, blockIdx.x);
    for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < I.e; e += gridDim.y * blockDim.x) {
        const float a = 0.0f + I.av[I.e_var[e]];
        const float b = 0.0f + I.af[I.e_fn[e]];
        I.emask[e] = a * b;
    }
}

// ---- Walk-SAT, persistent form: all steps of one instance in one workgroup, state in LDS -----------------------------
// Same arithmetic as the step-wise kernels above.  The only batch-global quantity of a step is the min of the random
// candidate vector (util.sparse_argmax's x - x.min() + 1); the kernel assumes it is 0 (true whenever ANY variable of the
// batch is inactive, outside every unsat clause, or belongs to a finished instance) and records per step whether an
// exact zero existed; pdp_local_search falls back to the strict loop if the record has a hole.
struct WsParams {
    const float *pred; float *out;
    int steps_cap; float epsilon; int rng_mode; const float *var_rand, *coin_rand; uint64_t seed;
    int32_t *first_sat;        // [B] step at which the instance had no unsat clause (steps_cap if never)
    uint32_t *spec_used, *spec_zero;   // bit maps [(steps + 31) / 32]: any unsat instance evaluated the arg-max / an exact zero existed
    const int32_t *inst_list;  // the instances of this launch (routing / replay subset) or NULL: instance = blockIdx.x
    const int32_t *cap_b;      // per-instance step cap (replication replay) or NULL
    // HBM-resident form (instances past the LDS limit): per launch slot the offsets of the instance's pieces in the workspace arrays
    const int64_t *big_off;    // [slots][3]: edge / variable / clause offset
    uint32_t *ws_e;            // [3][sum e]: pvv | pcc | cl
    int32_t *ws_v;             // [3][sum n]: a (float bits) | delta | nuns
    float *ws_f;               // [2][sum m]: aggc | degc
    uint8_t *ws_u;             // [sum m]: unsat
    int64_t ws_E, ws_V, ws_F;  // the sums (strides of the stacked arrays)
    int *ws_cnt;               // [slots] team form: unsat clauses of the instance
};

static size_t ws_lds_bytes(int n, int m, int e)
{
    auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    return 3 * a16((size_t)e * 2) + a16((size_t)(n + 1) * 2) + a16((size_t)(m + 1) * 2) + 4 * a16((size_t)n * 4) + 3 * a16((size_t)m * 4) + a16((size_t)m);
}

template <class T> __device__ __forceinline__ T *ws_carve(unsigned char *&p, size_t count)
{
    T *r = reinterpret_cast<T *>(p);
    p += (count * sizeof(T) + 15) & ~(size_t)15;
    return r;
}

// Incremental form.  The reference re-evaluates every clause and every variable in each step (solver.py:469-496); all of those
// quantities are small integers, so they can be carried from step to step exactly: flipping variable f changes
//   aggc[c]  (signed literal sum)            only for the clauses c of f,
//   unsat[c], the per-instance unsat count   only for those clauses,
//   delta[u] (energy change if u flips)      only for the variables u of those clauses: minus the old, plus the new contribution
//                                            of clause c (integer LDS atomics),
//   nuns[u]  (# unsat clauses containing u)  only where unsat[c] changed.
// A step is then: one scan over the variables feeding two 64-bit LDS max-atomics (value | inverted index: larger value wins,
// first index wins ties -- util.sparse_argmax), the flip, and one pass over the ~deg(f) clauses of f.
#ifdef PDP_PHASE_PROF
__device__ unsigned long long g_ws_cycles[8];
extern "C" int pdp_debug_ws_cycles(unsigned long long *out_host, int reset)
{
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_ws_cycles), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ws_cycles), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#define WS_PROF_DECL unsigned long long _t0 = __builtin_readcyclecounter(), _t1; uint32_t _acc[8] = {0};
#define WS_PROF_MARK(i) do { _t1 = __builtin_readcyclecounter(); _acc[i] += (uint32_t)(_t1 - _t0); _t0 = _t1; } while (0)
#define WS_PROF_FLUSH() do { if (threadIdx.x == 0) { _Pragma("unroll") for (int _i = 0; _i < 8; ++_i) if (_acc[_i]) atomicAdd(&g_ws_cycles[_i], (unsigned long long)_acc[_i]); } } while (0)
#else
#define WS_PROF_DECL
#define WS_PROF_MARK(i)
#define WS_PROF_FLUSH()
#endif

// W = uint16_t: the LDS-resident form (packed words and offsets in LDS).  W = uint32_t: the HBM-resident form for instances past the LDS
// limit -- the same statements on words twice as wide, in a per-call workspace; offsets, active flags and clause weights are read where
// the problem keeps them.  Routed per instance by pdp_local_search.
template <class W, int NT>
__global__ void __launch_bounds__(NT) k_walksat(PView pv, WsParams wp)
{
    constexpr bool HBM = sizeof(W) == 4;
    constexpr W SB = (W)((W)1 << (8 * sizeof(W) - 1)), MB = (W)((W)1 << (8 * sizeof(W) - 2)), VM = (W)(MB - 1);    // sign / mask bit, id mask
    constexpr int NWV = NT / 64;
    typedef typename std::conditional<HBM, int32_t, uint16_t>::type PT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int redi[PDP_RED_SCRATCH];
    __shared__ int s_cnt;
    __shared__ float s_coin;
    __shared__ unsigned long long s_keys[2 * NWV];   // per-wave maxima of the two arg-max keys
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wid = tid >> 6, nw = nt >> 6;
    const Inst G = load_inst(pv, __builtin_amdgcn_readfirstlane(wp.inst_list ? wp.inst_list[blockIdx.x] : (int)blockIdx.x));
    const int n = G.n, m = G.m, ne = G.e;
    // by-variable slots: pvv = variable | sign bit, pcc = clause | edge-mask bit (top);  by-clause edges: cl = variable | mask bit | sign bit
    W *pvv, *pcc, *cl; const PT *v_ptr, *f_ptr; const float *av, *af; float *a, *aggc, *degc; int *delta, *nuns; uint8_t *unsat;
    if constexpr (HBM) {
        const int64_t *off = wp.big_off + 3 * (size_t)blockIdx.x;
        pvv = wp.ws_e + off[0]; pcc = pvv + wp.ws_E; cl = pcc + wp.ws_E;
        a = reinterpret_cast<float *>(wp.ws_v + off[1]); delta = wp.ws_v + wp.ws_V + off[1]; nuns = delta + wp.ws_V;
        aggc = wp.ws_f + off[2]; degc = aggc + wp.ws_F; unsat = wp.ws_u + off[2];
        v_ptr = G.v_ptr; f_ptr = G.f_ptr; av = G.av; af = G.af;
    } else {
        unsigned char *cp = smem;
        pvv = ws_carve<W>(cp, ne); pcc = ws_carve<W>(cp, ne); cl = ws_carve<W>(cp, ne);
        PT *vp = ws_carve<PT>(cp, n + 1), *fp = ws_carve<PT>(cp, m + 1);
        float *avl = ws_carve<float>(cp, n); a = ws_carve<float>(cp, n);
        delta = ws_carve<int>(cp, n); nuns = ws_carve<int>(cp, n);
        float *afl = ws_carve<float>(cp, m); aggc = ws_carve<float>(cp, m); degc = ws_carve<float>(cp, m);
        unsat = ws_carve<uint8_t>(cp, m);
        for (int v = tid; v <= n; v += nt) vp[v] = (PT)G.v_ptr[v];
        for (int c = tid; c <= m; c += nt) fp[c] = (PT)G.f_ptr[c];
        for (int v = tid; v < n; v += nt) avl[v] = G.av[v];
        for (int c = tid; c < m; c += nt) afl[c] = G.af[c];
        v_ptr = vp; f_ptr = fp; av = avl; af = afl;
    }
    for (int p = tid; p < ne; p += nt) {
        const int e = G.v_edges[p];
        const bool em = G.emask[e] == 1.0f, neg = G.sgn[e] < 0;
        pvv[p] = (W)((W)G.e_var[e] | (neg ? SB : (W)0));
        pcc[p] = (W)((W)G.e_fn[e] | (em ? SB : (W)0));
        cl[e] = (W)((W)G.e_var[e] | (em ? MB : (W)0) | (neg ? SB : (W)0));      // edges are clause-major: edge id == position in the clause list
    }
    for (int v = tid; v < n; v += nt) {
        const float bit = (wp.pred[G.v0 + v] > 0.5f) ? 1.0f : 0.0f;
        a[v] = G.av[v] * (2.0f * bit - 1.0f);
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    // ---- full evaluation once (the reference's per-step formulas) --------------------------------------------------------
    int dup_any = 0;                                  // some variable occurs twice in one clause (not in loader output; handled anyway)
    {
        int cnt = 0;
        for (int c = tid; c < m; c += nt) {
            float deg = 0.0f, agg = 0.0f;
            for (int k = f_ptr[c]; k < f_ptr[c + 1]; ++k) {
                const W w = cl[k];
                const int v = (int)(w & VM);
                deg = deg + (0.0f + av[v]);
                agg = agg + (0.0f + ((w & SB) ? -1.0f : 1.0f) * (a[v] * av[v]));
                for (int k2 = f_ptr[c]; k2 < k; ++k2) if ((int)(cl[k2] & VM) == v) dup_any = 1;
            }
            degc[c] = deg; aggc[c] = agg;
            const float u = ((agg == -deg) ? 1.0f : 0.0f) * af[c];
            unsat[c] = (u == 1.0f) ? 1 : 0;
            cnt += unsat[c];
        }
        cnt = block_reduce(cnt, OpAddI(), 0, redi);
        dup_any = __syncthreads_or(dup_any);
        if (tid == 0) s_cnt = cnt;
        for (int v = tid; v < n; v += nt) {
            const float dist_v = a[v] * av[v];
            float d = 0.0f, acc = 0.0f;
            for (int p = v_ptr[v]; p < v_ptr[v + 1]; ++p) {
                const W cw = pcc[p];
                const int c = (int)(cw & (W)~SB);
                const float dist = 0.0f + ((pvv[p] & SB) ? -1.0f : 1.0f) * dist_v;
                const float others = (0.0f + aggc[c]) - dist;
                const float critical = ((others == (1.0f - (0.0f + degc[c]))) ? 1.0f : 0.0f) * ((cw & SB) ? 1.0f : 0.0f);
                d = d + critical * dist;
                acc = acc + (float)unsat[c];
            }
            delta[v] = (int)d; nuns[v] = (int)acc;           // exact: sums of at most deg(v) terms in {-1, 0, 1}
        }
        __syncthreads();
    }
    const int cap = wp.cap_b ? wp.cap_b[G.b] : wp.steps_cap;
    auto var_rand = [&](int step, int v) -> float {
        return (wp.rng_mode == PDP_RNG_STREAM) ? wp.var_rand[(size_t)step * pv.V + G.v0 + v]
                                               : pdp_philox_uniform(wp.seed, PDP_RNG_STREAM_WSVAR, (uint32_t)step, pv.rng_v0 + (uint32_t)(G.v0 + v));
    };
    auto coin_rand = [&](int step) -> float {
        return (wp.rng_mode == PDP_RNG_STREAM) ? wp.coin_rand[(size_t)step * pv.B + G.b]
                                               : pdp_philox_uniform(wp.seed, PDP_RNG_STREAM_WSCOIN, (uint32_t)step, pv.rng_b0 + (uint32_t)G.b);
    };
    int first_sat = cap;
    uint32_t used32 = 0, zero32 = 0;                            // thread 0: speculation bits of the current block of 32 steps
    // the random numbers of a step are fetched one step ahead (first variable of every thread, and the coin by the last thread)
    float u_pref = (tid < n && cap > 0) ? var_rand(0, tid) : 0.0f;
    float coin_pref = (tid == nt - 1 && cap > 0) ? coin_rand(0) : 0.0f;
    int it = 0;
    WS_PROF_DECL
    WS_PROF_MARK(0);
    int pending = -1;                                           // variable picked in the previous step whose a[] entry is not flipped yet
    for (; it < cap; ++it) {
        if (s_cnt == 0) { first_sat = it; break; }              // uniform: s_cnt is only written before a barrier
        if (tid == 0 && pending >= 0) a[pending] = -a[pending];
        pending = -1;
        // ---- the two arg-maxes of the step (solver.py:452-458): one scan, DPP maxima per wave, every lane joins the waves ------
        unsigned long long kg = 0ull, kr = 0ull; int has_zero = 0;
        if (tid == nt - 1) s_coin = coin_pref;
        for (int v = tid; v < n; v += nt) {
            const float acc = (float)nuns[v] * av[v];
            const float u = (v == tid) ? u_pref : var_rand(it, v);
            const float r = ((acc > 0.0f) ? 1.0f : 0.0f) * u;
            if (r == 0.0f) has_zero = 1;
            const float tg = ((-(float)delta[v]) - 0.0f) + 1.0f, tr = (r - 0.0f) + 1.0f;      // util.sparse_argmax's x - min + 1 with min == 0
            const unsigned long long k0 = argkey(tg, v), k1 = argkey(tr, v);
            kg = k0 > kg ? k0 : kg; kr = k1 > kr ? k1 : kr;
        }
        WS_PROF_MARK(1);                                        // scan
        kg = wave_max_u64(kg);
        kr = wave_max_u64(kr);
        has_zero = __builtin_amdgcn_ballot_w64(has_zero != 0) != 0 ? 1 : 0;
        if (lane == 63) { s_keys[wid] = kg; s_keys[NWV + wid] = kr; redi[wid] = has_zero; }
        __syncthreads();
        WS_PROF_MARK(2);                                        // wave maxima + barrier
        unsigned long long bg = s_keys[0], br = s_keys[NWV]; has_zero = redi[0];
        for (int k = 1; k < nw; ++k) { bg = s_keys[k] > bg ? s_keys[k] : bg; br = s_keys[NWV + k] > br ? s_keys[NWV + k] : br; has_zero |= redi[k]; }
        const int f = (s_coin > wp.epsilon) ? argkey_index(bg) : argkey_index(br);        // identical on every lane
        if (tid == 0) {
            // one global atomic per step and workgroup on the same word serialises the whole batch: collect 32 steps per flush
            used32 |= 1u << (it & 31);
            if (has_zero) zero32 |= 1u << (it & 31);
            if ((it & 31) == 31) {
                atomicOr(&wp.spec_used[it >> 5], used32);
                if (zero32) atomicOr(&wp.spec_zero[it >> 5], zero32);
                used32 = 0; zero32 = 0;
            }
        }
        if (it + 1 < cap) {                                     // next step's random numbers: in flight during the update pass
            if (tid < n) u_pref = var_rand(it + 1, tid);
            if (tid == nt - 1) coin_pref = coin_rand(it + 1);
        }
        WS_PROF_MARK(3);                                        // join + prefetch
        // ---- flip f and carry the change through its clauses ---------------------------------------------------------------------
        if (f >= 0) {
            const int pa = v_ptr[f], deg_f = v_ptr[f + 1] - pa;
            const float a_new = -a[f] * av[f];                // 0 for an inactive variable: nothing changes then
            for (int j = tid; j < deg_f; j += nt) {
                const int p = pa + j;
                const int c = (int)(pcc[p] & (W)~SB);
                float sum_s = (pvv[p] & SB) ? -1.0f : 1.0f; bool first = true;
                if (dup_any) {                                   // f may occur more than once in a clause: handle the clause once
                    sum_s = 0.0f;
                    for (int p2 = pa; p2 < pa + deg_f; ++p2)
                        if ((int)(pcc[p2] & (W)~SB) == c) { if (p2 < p) first = false; sum_s += (pvv[p2] & SB) ? -1.0f : 1.0f; }
                }
                if (!first || a_new == 0.0f) continue;
                const float old_agg = aggc[c], new_agg = old_agg + 2.0f * sum_s * a_new;
                const float target = 1.0f - (0.0f + degc[c]);
                const int u_new = (((new_agg == -degc[c]) ? 1.0f : 0.0f) * af[c] == 1.0f) ? 1 : 0;
                const int du = u_new - (int)unsat[c];
                auto touch = [&](W w) {                   // one literal of clause c: unsat count and contribution change
                    const int u = (int)(w & VM);
                    if (du) atomicAdd(&nuns[u], du);
                    if (!(w & MB)) return;                   // masked edge: contributes 0 before and after
                    const float sg = (w & SB) ? -1.0f : 1.0f;
                    const float dist_new = sg * ((u == f) ? a_new : a[u] * av[u]);      // a[f] itself is written at the end of the pass
                    const float dist_old = (u == f) ? -dist_new : dist_new;
                    const float c_old = ((old_agg - dist_old) == target) ? dist_old : 0.0f;
                    const float c_new = ((new_agg - dist_new) == target) ? dist_new : 0.0f;
                    const int dd = (int)(c_new - c_old);
                    if (dd) atomicAdd(&delta[u], dd);
                };
                const int k0 = f_ptr[c], klen = f_ptr[c + 1] - k0;
                if (klen == 3) { const W w0 = cl[k0], w1 = cl[k0 + 1], w2 = cl[k0 + 2]; touch(w0); touch(w1); touch(w2); }
                else for (int k = k0; k < k0 + klen; ++k) touch(cl[k]);
                aggc[c] = new_agg;
                if (du) { unsat[c] = (uint8_t)u_new; atomicAdd(&s_cnt, du); }
            }
        }
        pending = f;                                            // a[f] itself is flipped after the barrier: nobody reads `a` before the next one
        __syncthreads();
        WS_PROF_MARK(4);                                        // update pass + barrier
    }
    WS_PROF_FLUSH();
    if (tid == 0 && pending >= 0) a[pending] = -a[pending];
    __syncthreads();
    for (int v = tid; v < n; v += nt) wp.out[G.v0 + v] = (a[v] + 1.0f) / 2.0f;
    if (tid == 0) {
        wp.first_sat[G.b] = first_sat;
        if (used32) {                                            // the steps of the last, partial block end at it - 1
            const int w = (it - 1) >> 5;
            atomicOr(&wp.spec_used[w], used32);
            if (zero32) atomicOr(&wp.spec_zero[w], zero32);
        }
    }
}

// ---- Walk-SAT of a BIG instance as a team of workgroups -------------------------------------------------------------------------------
// The HBM-resident form above gives a big instance one workgroup, and a step scans all its variables: 0.68 ms per step at n = 300 000.
// Here the instance's team (pdp_common.hpp: Teamed<>, chip-wide when nothing else runs) shares the scan; a step is the scan, one team
// exchange of the two arg-max keys (mailboxes, the block maxima of every rank), the flip's O(degree) update by rank 0, one team barrier.
// Same statements and the same integer state as k_walksat, in the same per-call workspace; the unsat count is a word of that workspace.
struct WsNone {};
template <int NT>
__global__ void __launch_bounds__(NT) k_walksat_team(PView pv, WsParams wp, TeamLaunch tl)
{
    typedef uint32_t W;
    constexpr W SB = 0x80000000u, MB = 0x40000000u, VM = MB - 1u;
    constexpr int NWV = NT / 64;
    DECL_RED
    (void)redf;
    __shared__ unsigned long long s_keys[2 * NWV];
    Teamed<WsNone> TT;
    const int slot = team_begin(TT, tl, redi);
    if (slot < 0) return;
    const Inst G = load_inst(pv, __builtin_amdgcn_readfirstlane(wp.inst_list[slot]));
    const int n = G.n, m = G.m, ne = G.e;
    const int tid = team_tid(TT), nt = team_nt(TT);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t *off = wp.big_off + 3 * (size_t)slot;
    W *pvv = wp.ws_e + off[0], *pcc = pvv + wp.ws_E, *cl = pcc + wp.ws_E;
    float *a = reinterpret_cast<float *>(wp.ws_v + off[1]);
    int *delta = wp.ws_v + wp.ws_V + off[1], *nuns = delta + wp.ws_V;
    float *aggc = wp.ws_f + off[2], *degc = aggc + wp.ws_F;
    uint8_t *unsat = wp.ws_u + off[2];
    int *gcnt = wp.ws_cnt + slot;                     // unsat clauses of the instance
    const int32_t *v_ptr = G.v_ptr, *f_ptr = G.f_ptr;
    const float *av = G.av, *af = G.af;
    for (int p = tid; p < ne; p += nt) {
        const int e = G.v_edges[p];
        const bool em = G.emask[e] == 1.0f, neg = G.sgn[e] < 0;
        pvv[p] = (W)G.e_var[e] | (neg ? SB : 0u);
        pcc[p] = (W)G.e_fn[e] | (em ? SB : 0u);
        cl[e] = (W)G.e_var[e] | (em ? MB : 0u) | (neg ? SB : 0u);
    }
    for (int v = tid; v < n; v += nt) {
        const float bit = (wp.pred[G.v0 + v] > 0.5f) ? 1.0f : 0.0f;
        a[v] = G.av[v] * (2.0f * bit - 1.0f);
    }
    team_sync(TT);
    // ---- full evaluation once (the reference's per-step formulas) ------------------------------------------------------------------
    int dup_any = 0;
    {
        int cnt = 0;
        for (int c = tid; c < m; c += nt) {
            float deg = 0.0f, agg = 0.0f;
            for (int k = f_ptr[c]; k < f_ptr[c + 1]; ++k) {
                const W w = cl[k];
                const int v = (int)(w & VM);
                deg = deg + (0.0f + av[v]);
                agg = agg + (0.0f + ((w & SB) ? -1.0f : 1.0f) * (a[v] * av[v]));
                for (int k2 = f_ptr[c]; k2 < k; ++k2) if ((int)(cl[k2] & VM) == v) dup_any = 1;
            }
            degc[c] = deg; aggc[c] = agg;
            const float u = ((agg == -deg) ? 1.0f : 0.0f) * af[c];
            unsat[c] = (u == 1.0f) ? 1 : 0;
            cnt += unsat[c];
        }
        cnt = team_reduce(TT, cnt, OpAddI(), 0, redi);           // (its barrier publishes aggc / degc / unsat)
        dup_any = team_any(TT, dup_any);
        if (tid == 0) __hip_atomic_store(gcnt, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int v = tid; v < n; v += nt) {
            const float dist_v = a[v] * av[v];
            float d = 0.0f, acc = 0.0f;
            for (int p = v_ptr[v]; p < v_ptr[v + 1]; ++p) {
                const W cw = pcc[p];
                const int c = (int)(cw & ~SB);
                const float dist = 0.0f + ((pvv[p] & SB) ? -1.0f : 1.0f) * dist_v;
                const float others = (0.0f + aggc[c]) - dist;
                const float critical = ((others == (1.0f - (0.0f + degc[c]))) ? 1.0f : 0.0f) * ((cw & SB) ? 1.0f : 0.0f);
                d = d + critical * dist;
                acc = acc + (float)unsat[c];
            }
            delta[v] = (int)d; nuns[v] = (int)acc;
        }
        team_sync(TT);
    }
    const int cap = wp.steps_cap;
    auto var_rand = [&](int step, int v) -> float {
        return (wp.rng_mode == PDP_RNG_STREAM) ? wp.var_rand[(size_t)step * pv.V + G.v0 + v]
                                               : pdp_philox_uniform(wp.seed, PDP_RNG_STREAM_WSVAR, (uint32_t)step, pv.rng_v0 + (uint32_t)(G.v0 + v));
    };
    auto coin_rand = [&](int step) -> float {
        return (wp.rng_mode == PDP_RNG_STREAM) ? wp.coin_rand[(size_t)step * pv.B + G.b]
                                               : pdp_philox_uniform(wp.seed, PDP_RNG_STREAM_WSCOIN, (uint32_t)step, pv.rng_b0 + (uint32_t)G.b);
    };
    // block-wide maxima of the two keys and the OR of the flag: every thread gets them
    auto block_keys = [&](unsigned long long &kg, unsigned long long &kr, int &hz) {
        kg = wave_max_u64(kg); kr = wave_max_u64(kr);
        hz = __builtin_amdgcn_ballot_w64(hz != 0) != 0 ? 1 : 0;
        if (lane == 63) { s_keys[wid] = kg; s_keys[NWV + wid] = kr; redi[wid] = hz; }
        __syncthreads();
        kg = s_keys[0]; kr = s_keys[NWV]; hz = redi[0];
        for (int k = 1; k < NWV; ++k) { kg = s_keys[k] > kg ? s_keys[k] : kg; kr = s_keys[NWV + k] > kr ? s_keys[NWV + k] : kr; hz |= redi[k]; }
        __syncthreads();
    };
    int first_sat = cap;
    uint32_t used32 = 0, zero32 = 0;
    int it = 0, pending = -1;
    for (; it < cap; ++it) {
        if (__hip_atomic_load(gcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) { first_sat = it; break; }      // uniform: only written before a team barrier
        if (tid == 0 && pending >= 0) a[pending] = -a[pending];
        pending = -1;
        unsigned long long kg = 0ull, kr = 0ull; int has_zero = 0;
        for (int v = tid; v < n; v += nt) {
            const float acc = (float)nuns[v] * av[v];
            const float u = var_rand(it, v);
            const float r = ((acc > 0.0f) ? 1.0f : 0.0f) * u;
            if (r == 0.0f) has_zero = 1;
            const float tg = ((-(float)delta[v]) - 0.0f) + 1.0f, tr = (r - 0.0f) + 1.0f;
            const unsigned long long k0 = argkey(tg, v), k1 = argkey(tr, v);
            kg = k0 > kg ? k0 : kg; kr = k1 > kr ? k1 : kr;
        }
        block_keys(kg, kr, has_zero);
        if (TT.size > 1) {
            uint32_t *box = team_box(TT);
            if (threadIdx.x == 0) {
                uint32_t *mine = &box[PDP_BOX_WORDS * TT.rank];
                box_put(&mine[0], (uint32_t)kg); box_put(&mine[1], (uint32_t)(kg >> 32)); box_put(&mine[2], (uint32_t)kr); box_put(&mine[3], (uint32_t)(kr >> 32));
                box_put(&mine[4], (uint32_t)has_zero);
            }
            team_sync(TT);                                         // (also publishes rank 0's flip of the previous step's variable)
            kg = 0ull; kr = 0ull; has_zero = 0;
            if ((int)threadIdx.x < TT.size) {
                const uint32_t *w = &box[PDP_BOX_WORDS * threadIdx.x];
                kg = (unsigned long long)box_get(&w[0]) | ((unsigned long long)box_get(&w[1]) << 32);
                kr = (unsigned long long)box_get(&w[2]) | ((unsigned long long)box_get(&w[3]) << 32);
                has_zero = (int)box_get(&w[4]);
            }
            block_keys(kg, kr, has_zero);
        }
        const int f = (coin_rand(it) > wp.epsilon) ? argkey_index(kg) : argkey_index(kr);          // identical on every thread of the team
        if (tid == 0) {
            used32 |= 1u << (it & 31);
            if (has_zero) zero32 |= 1u << (it & 31);
            if ((it & 31) == 31) {
                atomicOr(&wp.spec_used[it >> 5], used32);
                if (zero32) atomicOr(&wp.spec_zero[it >> 5], zero32);
                used32 = 0; zero32 = 0;
            }
        }
        // ---- flip f and carry the change through its clauses: O(degree), rank 0 ----------------------------------------------------------
        if (f >= 0 && TT.rank == 0) {
            const int pa = v_ptr[f], deg_f = v_ptr[f + 1] - pa;
            const float a_new = -a[f] * av[f];
            for (int j = threadIdx.x; j < deg_f; j += blockDim.x) {
                const int p = pa + j;
                const int c = (int)(pcc[p] & ~SB);
                float sum_s = (pvv[p] & SB) ? -1.0f : 1.0f; bool first = true;
                if (dup_any) {
                    sum_s = 0.0f;
                    for (int p2 = pa; p2 < pa + deg_f; ++p2)
                        if ((int)(pcc[p2] & ~SB) == c) { if (p2 < p) first = false; sum_s += (pvv[p2] & SB) ? -1.0f : 1.0f; }
                }
                if (!first || a_new == 0.0f) continue;
                const float old_agg = aggc[c], new_agg = old_agg + 2.0f * sum_s * a_new;
                const float target = 1.0f - (0.0f + degc[c]);
                const int u_new = (((new_agg == -degc[c]) ? 1.0f : 0.0f) * af[c] == 1.0f) ? 1 : 0;
                const int du = u_new - (int)unsat[c];
                for (int k = f_ptr[c]; k < f_ptr[c + 1]; ++k) {
                    const W w = cl[k];
                    const int u = (int)(w & VM);
                    if (du) atomicAdd(&nuns[u], du);
                    if (!(w & MB)) continue;
                    const float sg = (w & SB) ? -1.0f : 1.0f;
                    const float dist_new = sg * ((u == f) ? a_new : a[u] * av[u]);
                    const float dist_old = (u == f) ? -dist_new : dist_new;
                    const float c_old = ((old_agg - dist_old) == target) ? dist_old : 0.0f;
                    const float c_new = ((new_agg - dist_new) == target) ? dist_new : 0.0f;
                    const int dd = (int)(c_new - c_old);
                    if (dd) atomicAdd(&delta[u], dd);
                }
                aggc[c] = new_agg;
                if (du) { unsat[c] = (uint8_t)u_new; atomicAdd(gcnt, du); }
            }
        }
        pending = f;
        team_sync(TT);
    }
    if (tid == 0 && pending >= 0) a[pending] = -a[pending];
    team_sync(TT);
    for (int v = tid; v < n; v += nt) wp.out[G.v0 + v] = (a[v] + 1.0f) / 2.0f;
    if (tid == 0) {
        wp.first_sat[G.b] = first_sat;
        if (used32) {
            const int w = (it - 1) >> 5;
            atomicOr(&wp.spec_used[w], used32);
            if (zero32) atomicOr(&wp.spec_zero[w], zero32);
        }
    }
    if (TT.failed && threadIdx.x == 0) atomicOr(&pv.flags[FL_TEAM_TIMEOUT], 1u);      // a team barrier gave up (pdp_common.hpp): the search is void
}

__global__ void k_ws_group_stop(int B0, int R, int cap, const int32_t *first_sat, uint32_t *stop /*max over originals of min over replicas*/)
{
    const int b0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (b0 >= B0) return;
    int mn = cap;
    for (int r = 0; r < R; ++r) { const int f = first_sat[b0 + r * B0]; mn = f < mn ? f : mn; }
    atomicMax(stop, (uint32_t)mn);
}
// fit_list / nfit: the instances of the LDS-resident form (NULL: all B of them); only those are listed
__global__ void k_ws_replay_list(int B, int stop, const int32_t *first_sat, int32_t *cap_b, int32_t *list, uint32_t *count, const int32_t *fit_list, int nfit)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    cap_b[i] = stop;
    if (i >= nfit) return;
    const int b = fit_list ? fit_list[i] : i;
    if (first_sat[b] > stop) list[atomicAdd(count, 1u)] = b;     // ran past the global stop in pass 1: redo with the cap
}

// Per-instance routing of the persistent Walk-SAT: which instances fit the LDS-resident form, which take the HBM-resident one
// (computed once per problem from the instance sizes).
static bool ws_fits_lds(int n, int m, int e) { return ws_lds_bytes(n, m, e) <= 64 * 1024 && n < 16384 && m < 16384 && e < 65535; }
static int ws_prepare(pdp_problem *p)
{
    if (p->ws_route_ready) return PDP_OK;
    const size_t B = p->B;
    std::vector<int32_t> v0(B + 1), f0(B + 1), e0(B + 1);
    PDP_HIP_CHECK(hipMemcpy(v0.data(), p->inst_v0, (B + 1) * 4, hipMemcpyDeviceToHost));
    PDP_HIP_CHECK(hipMemcpy(f0.data(), p->inst_f0, (B + 1) * 4, hipMemcpyDeviceToHost));
    PDP_HIP_CHECK(hipMemcpy(e0.data(), p->inst_e0, (B + 1) * 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> fit, big;
    std::vector<int64_t> off;
    int64_t se = 0, sv = 0, sf = 0;
    p->ws_fit_n = p->ws_fit_m = p->ws_fit_e = 0;
    for (size_t b = 0; b < B; ++b) {
        const int n = v0[b + 1] - v0[b], m = f0[b + 1] - f0[b], e = e0[b + 1] - e0[b];
        if (ws_fits_lds(n, m, e)) {
            fit.push_back((int32_t)b);
            if (n > p->ws_fit_n) p->ws_fit_n = n;
            if (m > p->ws_fit_m) p->ws_fit_m = m;
            if (e > p->ws_fit_e) p->ws_fit_e = e;
        } else {
            big.push_back((int32_t)b);
            off.push_back(se); off.push_back(sv); off.push_back(sf);
            se += (e + 3) & ~3; sv += (n + 3) & ~3; sf += (m + 3) & ~3;
        }
    }
    p->ws_nfit = (int)fit.size(); p->ws_nbig = (int)big.size();
    p->ws_big_E = se; p->ws_big_V = sv; p->ws_big_F = sf;
    { int st_ = pdp_dev_alloc((void **)&p->ws_fit_list, (fit.size() + 1) * 4); if (st_ != PDP_OK) return st_; }
    { int st_ = pdp_dev_alloc((void **)&p->ws_big_list, (big.size() + 1) * 4); if (st_ != PDP_OK) return st_; }
    { int st_ = pdp_dev_alloc((void **)&p->ws_big_off, (off.size() + 1) * 8); if (st_ != PDP_OK) return st_; }
    if (!fit.empty()) PDP_HIP_CHECK(hipMemcpy(p->ws_fit_list, fit.data(), fit.size() * 4, hipMemcpyHostToDevice));
    if (!big.empty()) PDP_HIP_CHECK(hipMemcpy(p->ws_big_list, big.data(), big.size() * 4, hipMemcpyHostToDevice));
    if (!off.empty()) PDP_HIP_CHECK(hipMemcpy(p->ws_big_off, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    p->ws_route_ready = 1;
    return PDP_OK;
}

// Everything a persistent search holds until it returns -- device scratch, the host copy of the bit maps, work on the side stream that
// still reads the caller's arrays -- is released by this guard on EVERY path out, error returns included: the side stream is joined first.
namespace {
struct WsCallGuard {
    hipStream_t side = nullptr; bool side_busy = false;
    void *dev[2] = {nullptr, nullptr}; void *host = nullptr;
    ~WsCallGuard()
    {
        if (side_busy && side) (void)hipStreamSynchronize(side);
        for (void *d : dev) if (d) pdp_dev_free(d);
        if (host) free(host);
    }
};
}

// returns PDP_OK and *done = 1 if the persistent search produced the reference result, *done = 0 if the caller must run the strict loop
static int local_search_persistent(pdp_problem *p, const float *pred, int iterations, float epsilon, int rng_mode, const float *var_rand,
                                   const float *coin_rand, uint64_t seed, float *out, int32_t *steps_host, hipStream_t st, int *done)
{
    *done = 0;
    if (iterations <= 0) return PDP_OK;
    if (p->exchange) {
        // A coupled forward over several processes: whether THIS part can take the persistent search is a local fact (its edge order, a
        // big instance with the routing switched off), and a part that left for the strict loop would skip the collectives the others
        // issue.  The parts agree first; one that cannot -> every part reports PDP_ERR_SPECULATION with its inputs untouched, and the host
        // solves the segment whole on one process (pdp.native.CoupledForwardFailed).
        int can = p->fn_edges_identity ? 1 : 0;
        if (can) { int st_ = ws_prepare(p); if (st_ != PDP_OK) return st_; if (p->ws_nbig && getenv("PDP_WALKSAT_NO_ROUTING")) can = 0; }
        const uint32_t ors[1] = {can ? 0u : 1u};
        uint32_t *m = nullptr;
        { const int st_ = pdp_exchange_call(p, nullptr, 0, nullptr, 0, ors, 1, &m); if (st_ != PDP_OK) return st_; }
        if (m[0]) {
            pdp_set_error("Walk-SAT of a coupled multi-process forward: a part cannot take the persistent search (edge order / routing switch); "
                          "solve the segment on one process");
            return PDP_ERR_SPECULATION;
        }
    }
    if (!p->fn_edges_identity) return PDP_OK;
    { int st_ = ws_prepare(p); if (st_ != PDP_OK) return st_; }
    const int nfit = p->ws_nfit, nbig = p->ws_nbig;
    if (nbig && getenv("PDP_WALKSAT_NO_ROUTING")) return PDP_OK;
    const size_t lds = ws_lds_bytes(p->ws_fit_n, p->ws_fit_m, p->ws_fit_e);
    const size_t bw = ((size_t)iterations + 31) / 32;               // words per bit map
    const size_t words = 2 * bw + 4;
    WsCallGuard guard;
    uint32_t *spec = nullptr;
    { int st_ = pdp_dev_alloc((void **)&spec, words * 4 + (size_t)p->B * 4 * 3); if (st_ != PDP_OK) return st_; }
    guard.dev[0] = spec;
    int32_t *first_sat = (int32_t *)(spec + words), *cap_b = first_sat + p->B, *list = cap_b + p->B;
    PDP_HIP_CHECK(hipMemsetAsync(spec, 0, words * 4, st));
    hipLaunchKernelGGL(k_edge_mask2, dim3(p->B, pdp_edge_rows(p)), dim3(PDP_NT), 0, st, make_view(p));       // solver.py:439-440
    p->has_edge_mask = 1;
    WsParams wp;
    memset(&wp, 0, sizeof(wp));
    wp.pred = pred; wp.out = out; wp.steps_cap = iterations; wp.epsilon = epsilon; wp.rng_mode = rng_mode; wp.var_rand = var_rand;
    wp.coin_rand = coin_rand; wp.seed = seed; wp.first_sat = first_sat; wp.spec_used = spec; wp.spec_zero = spec + bw;
    wp.inst_list = nullptr; wp.cap_b = nullptr;
    int ws_nt = 256;
    if (const char *env = getenv("PDP_WALKSAT_THREADS")) { const int v = atoi(env); if (v == 64 || v == 128 || v == 256) ws_nt = v; }
    char *big_ws = nullptr;
    WsParams wb = wp;
    // the instances past the LDS limit: one 1024-thread workgroup each on the HBM-resident form -- few of them: a team of workgroups each, chip-wide
    // when no LDS-resident launch runs next to it -- on stream `on` (pass 1: a stream of its own next to the LDS-resident launch; they share
    // nothing but the speculation bit maps).  The kernels build their workspace from the problem and `pred` at entry, so a second launch with
    // a smaller step cap (the replica truncation below) simply runs the same search again and stops earlier.
    auto launch_big = [&](int steps_cap, bool alone, hipStream_t on) -> int {
        wb.steps_cap = steps_cap;
        TeamLaunch tl; tl.size = 1;
        if (!getenv("PDP_WALKSAT_NO_TEAM")) { const int st_ = pdp_team_plan(p, nbig, alone, 256, &tl, on); if (st_ != PDP_OK) return st_; }
        if (tl.size > 1) hipLaunchKernelGGL((k_walksat_team<256>), dim3(tl.size * tl.slots), dim3(256), 0, on, make_view(p), wb, tl);
        else hipLaunchKernelGGL((k_walksat<uint32_t, 1024>), dim3(nbig), dim3(1024), 0, on, make_view(p), wb);
        return PDP_OK;
    };
    if (nbig) {
        const size_t bytes = (size_t)p->ws_big_E * 12 + (size_t)p->ws_big_V * 12 + (size_t)p->ws_big_F * 9 + 64 + (size_t)nbig * 4 + 16;
        { int st_ = pdp_dev_alloc((void **)&big_ws, bytes); if (st_ != PDP_OK) return st_; }
        guard.dev[1] = big_ws;
        wb.inst_list = p->ws_big_list; wb.big_off = p->ws_big_off;
        wb.ws_E = p->ws_big_E; wb.ws_V = p->ws_big_V; wb.ws_F = p->ws_big_F;
        wb.ws_e = (uint32_t *)big_ws; wb.ws_v = (int32_t *)(wb.ws_e + 3 * wb.ws_E); wb.ws_f = (float *)(wb.ws_v + 3 * wb.ws_V); wb.ws_u = (uint8_t *)(wb.ws_f + 2 * wb.ws_F);
        wb.ws_cnt = (int *)(big_ws + ((bytes - (size_t)nbig * 4 - 16) & ~(size_t)15));
        if (!p->ws_side_stream) {
            PDP_HIP_CHECK(hipStreamCreateWithFlags(&p->ws_side_stream, hipStreamNonBlocking));
            for (int i = 0; i < 2; ++i) PDP_HIP_CHECK(hipEventCreateWithFlags(&p->ws_side_ev[i], hipEventDisableTiming));
        }
        PDP_HIP_CHECK(hipEventRecord(p->ws_side_ev[0], st)); PDP_HIP_CHECK(hipStreamWaitEvent(p->ws_side_stream, p->ws_side_ev[0], 0));
        guard.side = p->ws_side_stream; guard.side_busy = true;
        { const int st_ = launch_big(iterations, nfit == 0, p->ws_side_stream); if (st_ != PDP_OK) return st_; }
        PDP_HIP_CHECK(hipEventRecord(p->ws_side_ev[1], p->ws_side_stream));
    }
    if (nfit) {
        PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_walksat<uint16_t, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wp.inst_list = nbig ? p->ws_fit_list : nullptr;
        pdp_timed_scope timed(PDP_TK_WALKSAT, st);
        pdp_note_kernel(PDP_TK_WALKSAT, "k_walksat<unsigned short, 256>");
        hipLaunchKernelGGL((k_walksat<uint16_t, 256>), dim3(nfit), dim3(ws_nt), lds, st, make_view(p), wp);
    }
    if (nbig) PDP_HIP_CHECK(hipStreamWaitEvent(st, p->ws_side_ev[1], 0));
    PDP_LAUNCH_CHECK();
    uint32_t *ctl = spec + 2 * bw;                       // [0] global stop step, [1] replay count
    hipLaunchKernelGGL(k_ws_group_stop, dim3((p->B0 + 255) / 256), dim3(256), 0, st, p->B0, p->R, iterations, first_sat, ctl);
    uint32_t *host = (uint32_t *)malloc(words * 4);
    PDP_REQUIRE(host, "out of host memory");
    guard.host = host;
    PDP_HIP_CHECK(hipMemcpyAsync(host, spec, words * 4, hipMemcpyDeviceToHost, st));
    PDP_HIP_CHECK(hipStreamSynchronize(st));             // (the main stream waited for the side stream's event: both are drained here)
    guard.side_busy = false;
    const int stop = (int)host[2 * bw];
    int status = PDP_OK;
    if (p->R > 1 && stop < iterations) {
        // replicas that were still searching at the global stop step must be truncated there (solver.py:446-449)
        hipLaunchKernelGGL(k_ws_replay_list, dim3((p->B + 255) / 256), dim3(256), 0, st, p->B, stop, first_sat, cap_b, list, ctl + 1,
                           (const int32_t *)(nbig ? p->ws_fit_list : nullptr), nfit);
        uint32_t cnt = 0;
        PDP_HIP_CHECK(hipMemcpyAsync(&cnt, ctl + 1, 4, hipMemcpyDeviceToHost, st));
        PDP_HIP_CHECK(hipStreamSynchronize(st));
        // the big instances (not in the list: they cannot run LDS-resident) run again, all of them, with the global stop as their cap -- a search
        // that had ended before the stop ends at the same step again (same random numbers, keyed by step), the others are cut there
        if (nbig) { const int st_ = launch_big(stop, cnt == 0, st); if (st_ != PDP_OK) return st_; }
        if (cnt) {
            wp.inst_list = list; wp.cap_b = cap_b;
            hipLaunchKernelGGL((k_walksat<uint16_t, 256>), dim3(cnt), dim3(ws_nt), lds, st, make_view(p), wp);
        }
        if (cnt || nbig) PDP_HIP_CHECK(hipStreamSynchronize(st));
    }
    // speculation record: every step < stop in which some instance searched needs an exact zero somewhere.  An instance that
    // finished at step f supplies zeros for every later step (its candidates are all 0), so only steps before the first finish count.
    bool ok = true;
    int first_finish = iterations;
    {
        int32_t *fs_host = (int32_t *)malloc(sizeof(int32_t) * (size_t)p->B);
        if (hipMemcpy(fs_host, first_sat, sizeof(int32_t) * (size_t)p->B, hipMemcpyDeviceToHost) != hipSuccess) status = PDP_ERR_HIP;
        for (int b = 0; b < p->B; ++b) first_finish = fs_host[b] < first_finish ? fs_host[b] : first_finish;
        free(fs_host);
    }
    int stop_all = stop;
    if (p->exchange && status == PDP_OK) {
        // a coupled forward over several processes: the record of the batch-global minimum is completed across the parts (an exact zero in
        // ANY part serves every part; the first finished instance of the whole batch; the step at which the whole batch was solved)
        const uint32_t mins[1] = {(uint32_t)first_finish}, maxs[1] = {(uint32_t)stop};
        uint32_t *m = nullptr;
        { const int st_ = pdp_exchange_call(p, mins, 1, maxs, 1, host, (int)(2 * bw), &m); if (st_ != PDP_OK) return st_; }
        first_finish = (int)m[0]; stop_all = (int)m[1];
        for (size_t i = 0; i < 2 * (size_t)bw; ++i) host[i] = m[2 + i];
    }
    for (int t = 0; t < stop_all && t < first_finish && ok; ++t)
        if (((host[t >> 5] >> (t & 31)) & 1u) && !((host[bw + (t >> 5)] >> (t & 31)) & 1u)) ok = false;
    if (status == PDP_OK && nbig) {
        // the big instances ran as teams of workgroups: did one of their barriers give up (workgroups not resident together)?
        uint32_t timed_out = 0;
        if (hipMemcpy(&timed_out, p->flags + FL_TEAM_TIMEOUT, 4, hipMemcpyDeviceToHost) != hipSuccess) status = PDP_ERR_HIP;
        else if (timed_out) {
            pdp_set_error("Walk-SAT: a team of workgroups that shares one big instance was not resident together: its barrier gave up instead of "
                          "hanging; the search result is void (is another process using this GPU?)");
            status = PDP_ERR_HIP;
        }
    }
    if (status != PDP_OK) return status;
    if (!ok && p->exchange) {
        // (every part sees the same completed record, so every part takes this exit: the host solves the segment whole on one process)
        pdp_set_error("Walk-SAT of a coupled multi-process forward: a step's batch-global minimum was not 0 in any part; the strict loop that "
                      "computes it is single-process -- the segment is solved on one process");
        return PDP_ERR_SPECULATION;
    }
    if (!ok) return PDP_OK;          // caller runs the strict loop on the untouched inputs
    if (steps_host) *steps_host = stop;
    *done = 1;
    return PDP_OK;
}

extern "C" int pdp_local_search(pdp_problem *p, const float *pred, int iterations, float epsilon, int rng_mode,
                                const float *var_rand, const float *coin_rand, uint64_t seed, float *out,
                                int32_t *steps_host, void *stream)
{
    PDP_REQUIRE(p && p->av && pred && out, "NULL argument / state not bound");
    PDP_REQUIRE(rng_mode == PDP_RNG_PHILOX || iterations == 0 || (var_rand && coin_rand), "stream mode needs the drawn values");
    hipStream_t st = ST(stream);
    PDP_REQUIRE(!p->exchange || (!getenv("PDP_WALKSAT_STRICT") && p->R == 1), "coupled multi-process forward: the persistent Walk-SAT only");
    if (!getenv("PDP_WALKSAT_STRICT")) {
        int done = 0;
        const int rc = local_search_persistent(p, pred, iterations, epsilon, rng_mode, var_rand, coin_rand, seed, out, steps_host, st, &done);
        if (rc != PDP_OK) return rc;
        if (done) return PDP_OK;
    }
    // iterations <= 0 (satyr -w 0): the persistent search declines and the loop below is init / edge mask / copy-out -- no batch-global
    // reduction, so every part of a coupled forward runs it on its own.  With steps to do, the persistent search of a coupled forward
    // either finished or reported above; the strict loop has no exchange.
    if (p->exchange && iterations > 0) {
        pdp_set_error("coupled multi-process forward: the strict Walk-SAT loop is single-process");
        return PDP_ERR_SPECULATION;
    }
    const PView pv = make_view(p);
    float *a = p->ws_v[0], *negdelta = p->ws_v[1], *uv = p->ws_v[2], *unsat_b = p->ws_b[0];
    hipLaunchKernelGGL(k_ws_init, dim3(grid_for(p->V)), dim3(256), 0, st, p->V, p->av, pred, a);
    hipLaunchKernelGGL(k_edge_mask2, dim3(p->B, pdp_edge_rows(p)), dim3(PDP_NT), 0, st, pv);       // solver.py:439-440
    p->has_edge_mask = 1;
    int it = 0;
    for (; it < iterations; ++it) {
        hipLaunchKernelGGL(k_ws_reset, dim3(1), dim3(64), 0, st, p->flags);
        hipLaunchKernelGGL(k_ws_energy, dim3(p->B), dim3(PDP_NT), 0, st, pv, a, unsat_b, p->ws_fu[0], p->ws_f[0], p->ws_f[1]);
        if (p->R > 1) hipLaunchKernelGGL(k_ws_compact, dim3((p->B0 + 255) / 256), dim3(256), 0, st, p->B0, p->R, unsat_b, p->flags);
        PDP_HIP_CHECK(hipMemcpyAsync(p->flags_host, p->flags + FL_ANY_UNSAT, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PDP_HIP_CHECK(hipStreamSynchronize(st));
        if (p->flags_host[0] == 0u) break;
        hipLaunchKernelGGL(k_ws_delta, dim3(p->B), dim3(PDP_NT), 0, st, pv, a, p->ws_fu[0], p->ws_f[0], p->ws_f[1], rng_mode,
                           var_rand ? var_rand + (size_t)it * p->V : nullptr, seed, it, negdelta, uv);
        hipLaunchKernelGGL(k_ws_flip, dim3(p->B), dim3(PDP_NT), 0, st, pv, a, unsat_b, negdelta, uv, rng_mode,
                           coin_rand ? coin_rand + (size_t)it * p->B : nullptr, seed, it, epsilon);
    }
    hipLaunchKernelGGL(k_ws_finish, dim3(grid_for(p->V)), dim3(256), 0, st, p->V, a, out);
    PDP_LAUNCH_CHECK();
    PDP_HIP_CHECK(hipStreamSynchronize(st));
    if (steps_host) *steps_host = it;
    return PDP_OK;
}

// ---- de-duplication -----------------------------------------------------------------------------------------
__global__ void k_dedup_assign(int V, const float *pred, float *a)
{
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) a[v] = 2.0f * pred[v] - 1.0f;
}
__global__ void k_dedup_choose(int B0, int R, const float *energy, int32_t *chosen)
{
    const int b0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (b0 >= B0) return;
    // arg-max of -energy over the replicas, first (lowest replica) index wins ties
    int best = 0; float bv = -energy[b0];
    for (int r = 1; r < R; ++r) { const float v = -energy[b0 + r * B0]; if (v > bv) { bv = v; best = r; } }
    chosen[b0] = best;
}
__global__ void k_dedup_select(int V0, const int32_t *var_inst, const int32_t *chosen, const float *pred, float *out)
{
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V0; v += (int64_t)gridDim.x * blockDim.x) {
        const int b0 = var_inst[v];              // replica 0 carries the original instance ids
        out[v] = 0.0f + pred[v + (int64_t)chosen[b0] * V0];
    }
}

extern "C" int pdp_deduplicate(pdp_problem *p, const float *pred, float *out, int32_t *chosen, void *stream)
{
    PDP_REQUIRE(p && p->av && pred && out, "NULL argument / state not bound");
    hipStream_t st = ST(stream);
    int32_t *ch = chosen ? chosen : p->ws_vi[0];
    hipLaunchKernelGGL(k_dedup_assign, dim3(grid_for(p->V)), dim3(256), 0, st, p->V, pred, p->ws_v[0]);
    hipLaunchKernelGGL(k_energy, dim3(p->B), dim3(PDP_NT), 0, st, make_view(p), p->ws_v[0], p->ws_b[0], (float *)nullptr, p->ws_fu[0],
                       p->ws_f[0], p->ws_f[1]);
    hipLaunchKernelGGL(k_dedup_choose, dim3((p->B0 + 255) / 256), dim3(256), 0, st, p->B0, p->R, p->ws_b[0], ch);
    hipLaunchKernelGGL(k_dedup_select, dim3(grid_for(p->V0)), dim3(256), 0, st, p->V0, p->var_inst, ch, pred, out);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}
