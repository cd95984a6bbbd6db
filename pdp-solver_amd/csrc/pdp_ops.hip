// pdp_ops.hip -- step-wise entry points: one launch (or a short guarded chain of launches) per
// reference operator group, one workgroup per instance.  These back the plug-in API
// (propagator / decimator / predictor classes) and are the strict, batch-coupled fallback of the
// persistent solver in pdp_solve.hip.  Batch-global quantities of the reference (x.min() inside
// sparse_max/argmax, `.sum() > 0` guards) live in device flag words so that no
// entry point needs a host round trip unless its signature returns something to the host.
#include "pdp_device.hpp"
#include <type_traits>

#define ST(s) ((hipStream_t)(s))

// ---- flag helpers -------------------------------------------------------------------------------------
__global__ void k_reset_flags(uint32_t *flags)
{
    const int i = threadIdx.x;
    if (i >= FL_COUNT) return;
    if (i == FL_GMIN0 || i == FL_GMIN1 || i == FL_GMIN2) flags[i] = PDP_ENC_PLUS_INF;
    else if (i == FL_LAYOUT_BAD || i == FL_SPEC_VIOLATION || i == FL_TEAM_TIMEOUT || i == FL_LOOP_STOP || i == FL_LOOP_ITERS) { /* sticky / owned by pdp_loop_* */ }
    else flags[i] = 0u;
}
static inline void reset_flags(pdp_problem *p, hipStream_t st) { hipLaunchKernelGGL(k_reset_flags, dim3(1), dim3(64), 0, st, p->flags); }

#define DECL_RED __shared__ float redf[PDP_RED_SCRATCH]; __shared__ int redi[PDP_RED_SCRATCH];

// ---- K7 ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ SimplifyScratch make_scratch(const Inst &I, float *assign, int32_t *deg, int32_t *sdeg,
                                                        int32_t *flagv_words, uint8_t *ff, uint8_t *ff2, int *red)
{
    SimplifyScratch s;
    s.assign = assign + I.v0; s.deg = deg + I.v0; s.sdeg = sdeg + I.v0;
    s.flag_v = reinterpret_cast<uint8_t *>(flagv_words) + I.v0;
    s.flag_f = ff + I.f0; s.flag_f2 = ff2 + I.f0; s.red = red;
    return s;
}

// TEAM: a big instance spread over a team of workgroups (pdp_common.hpp: Teamed<>, pdp_team_plan); the routines are the same
template <bool TEAM>
__global__ void __launch_bounds__(PDP_NT) k_simplify(PView pv, float *assign_ws, int32_t *deg, int32_t *sdeg, int32_t *fv,
                                                     uint8_t *ff, uint8_t *ff2, TeamLaunch tl)
{
    DECL_RED
    (void)redf;
    std::conditional_t<TEAM, Teamed<Inst>, Inst> I;
    int slot = (int)blockIdx.x;
    if constexpr (TEAM) { slot = team_begin(I, tl, redi); if (slot < 0) return; }
    static_cast<Inst &>(I) = load_inst(pv, slot);
    const SimplifyScratch s = make_scratch(I, assign_ws, deg, sdeg, fv, ff, ff2, redi);
    d_simplify(I, s, pv.is_sat + I.b);
    if constexpr (TEAM) { if (I.failed && threadIdx.x == 0) atomicOr(&pv.flags[FL_TEAM_TIMEOUT], 1u); }
}

// set_variables: assignment lives in caller memory [V]; guard_slot < 0 -> always run
template <bool TEAM>
__global__ void __launch_bounds__(PDP_NT) k_set_variables(PView pv, float *assignment, float *assign_ws, int32_t *deg, int32_t *sdeg,
                                                          int32_t *fv, uint8_t *ff, uint8_t *ff2, int guard_slot, TeamLaunch tl)
{
    DECL_RED
    (void)redf;
    if (guard_slot >= 0 && pv.flags[guard_slot] == 0u) return;
    std::conditional_t<TEAM, Teamed<Inst>, Inst> I;
    int slot = (int)blockIdx.x;
    if constexpr (TEAM) { slot = team_begin(I, tl, redi); if (slot < 0) return; }
    static_cast<Inst &>(I) = load_inst(pv, slot);
    // the caller's assignment is only masked in place (solver.py:210); simplify works on private scratch
    const SimplifyScratch s0 = make_scratch(I, assignment, deg, sdeg, fv, ff, ff2, redi);
    d_set_variable_core(I, s0);
    const SimplifyScratch s = make_scratch(I, assign_ws, deg, sdeg, fv, ff, ff2, redi);
    d_simplify(I, s, pv.is_sat + I.b);
    if constexpr (TEAM) { if (I.failed && threadIdx.x == 0) atomicOr(&pv.flags[FL_TEAM_TIMEOUT], 1u); }
}

// few big instances: teams (chip-wide when the instances are huge: nothing else runs next to these launches)
static int simplify_plan(pdp_problem *p, TeamLaunch *tl, hipStream_t st)
{
    tl->size = 1; tl->count = p->B; tl->slots = p->B; tl->no_xcd = 0; tl->ws = nullptr;
    if (p->max_e < 16384 || p->B > 128) return PDP_OK;
    return pdp_team_plan(p, p->B, true, PDP_NT, tl, st);
}

extern "C" int pdp_simplify(pdp_problem *p, void *stream)
{
    PDP_REQUIRE(p && p->av, "problem state is not bound");
    if (pdp_simplify_lds(p, ST(stream))) { PDP_LAUNCH_CHECK(); return PDP_OK; }       // every instance fits the LDS: the same routines, LDS-resident
    TeamLaunch tl;
    { const int st_ = simplify_plan(p, &tl, ST(stream)); if (st_ != PDP_OK) return st_; }
    if (tl.size > 1)
        hipLaunchKernelGGL((k_simplify<true>), dim3(tl.size * tl.slots), dim3(PDP_NT), 0, ST(stream), make_view(p), p->ws_v[5], p->ws_vi[0], p->ws_vi[1],
                           p->ws_vi[2], p->ws_fu[0], p->ws_fu[1], tl);
    else
        hipLaunchKernelGGL((k_simplify<false>), dim3(p->B), dim3(PDP_NT), 0, ST(stream), make_view(p), p->ws_v[5], p->ws_vi[0], p->ws_vi[1],
                           p->ws_vi[2], p->ws_fu[0], p->ws_fu[1], tl);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

extern "C" int pdp_set_variables(pdp_problem *p, float *assignment, void *stream)
{
    PDP_REQUIRE(p && p->av && assignment, "NULL argument / state not bound");
    TeamLaunch tl;
    { const int st_ = simplify_plan(p, &tl, ST(stream)); if (st_ != PDP_OK) return st_; }
    if (tl.size > 1)
        hipLaunchKernelGGL((k_set_variables<true>), dim3(tl.size * tl.slots), dim3(PDP_NT), 0, ST(stream), make_view(p), assignment, p->ws_v[5], p->ws_vi[0],
                           p->ws_vi[1], p->ws_vi[2], p->ws_fu[0], p->ws_fu[1], -1, tl);
    else
        hipLaunchKernelGGL((k_set_variables<false>), dim3(p->B), dim3(PDP_NT), 0, ST(stream), make_view(p), assignment, p->ws_v[5], p->ws_vi[0],
                           p->ws_vi[1], p->ws_vi[2], p->ws_fu[0], p->ws_fu[1], -1, tl);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- K8 ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PDP_NT) k_edge_mask(PView pv)
{
    DECL_RED
    (void)redf;
    const Inst I = load_inst(pv, blockIdx.x);
    int cnt = 0;
    for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < I.e; e += gridDim.y * blockDim.x) {        // (gridDim.y > 1: big instances)
        const float a = 0.0f + I.av[I.e_var[e]];
        const float b = 0.0f + I.af[I.e_fn[e]];
        const float m = a * b;
        I.emask[e] = m;
        cnt += (m == 1.0f) ? 1 : 0;
    }
    cnt = block_reduce(cnt, OpAddI(), 0, redi);
    if (threadIdx.x == 0 && cnt) atomicAdd(&pv.flags[FL_ACTIVE_EDGES], (uint32_t)cnt);
}

static int read_flags(pdp_problem *p, hipStream_t st)
{
    PDP_HIP_CHECK(hipMemcpyAsync(p->flags_host, p->flags, FL_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PDP_HIP_CHECK(hipStreamSynchronize(st));
    if (p->flags_host[FL_TEAM_TIMEOUT]) {
        pdp_set_error("a team of workgroups that shares one big instance was not resident together: its barrier gave up instead of hanging; "
                      "the problem's state is void (is another process using this GPU?)");
        return PDP_ERR_HIP;
    }
    return PDP_OK;
}

extern "C" int pdp_refresh_edge_mask(pdp_problem *p, int32_t *all_active_host, void *stream)
{
    PDP_REQUIRE(p && p->emask, "problem state is not bound");
    reset_flags(p, ST(stream));
    hipLaunchKernelGGL(k_edge_mask, dim3(p->B, pdp_edge_rows(p)), dim3(PDP_NT), 0, ST(stream), make_view(p));
    PDP_LAUNCH_CHECK();
    p->has_edge_mask = 1;
    if (all_active_host) {
        int s = read_flags(p, ST(stream));
        if (s != PDP_OK) return s;
        *all_active_host = (p->flags_host[FL_ACTIVE_EDGES] == (uint32_t)p->E) ? 1 : 0;
    }
    return PDP_OK;
}

// ---- K4 / K5 ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PDP_NT) k_smooth_max(PView pv, const float *x, float *out)
{
    const Inst I = load_inst(pv, blockIdx.x);
    for (int v = threadIdx.x; v < I.n; v += blockDim.x) out[I.v0 + v] = d_smooth_max_var(I, v, x + I.e0);
}

extern "C" int pdp_smooth_max(pdp_problem *p, const float *x, float *out, void *stream)
{
    PDP_REQUIRE(p && x && out, "NULL argument");
    hipLaunchKernelGGL(k_smooth_max, dim3(p->B), dim3(PDP_NT), 0, ST(stream), make_view(p), x, out);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

__global__ void k_global_min(const float *x, int64_t n, uint32_t *flags, int slot_min, int slot_nan)
{
    DECL_RED
    float m = PDP_INF; bool nn = false;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (v != v) nn = true; else if (v < m) m = v;
    }
    publish_min(m, nn, flags, slot_min, slot_nan, redf, redi);
}

__global__ void __launch_bounds__(PDP_NT) k_instance_max(PView pv, const float *x, float *out, int slot_min, int slot_nan)
{
    DECL_RED
    (void)redi;
    const Inst I = load_inst(pv, blockIdx.x);
    const float gmin = read_gmin(pv.flags, slot_min, slot_nan);
    const float r = d_instance_max(I, x + I.v0, gmin, I.n < pv.V, redf);
    if (threadIdx.x == 0) out[I.b] = r;
}

__global__ void __launch_bounds__(PDP_NT) k_instance_argmax(PView pv, const float *x, int64_t *out, int slot_min, int slot_nan)
{
    DECL_RED
    const Inst I = load_inst(pv, blockIdx.x);
    const float gmin = read_gmin(pv.flags, slot_min, slot_nan);
    const int li = d_instance_argmax(I, x + I.v0, gmin, redf, redi);
    if (threadIdx.x == 0) out[I.b] = (li < 0) ? 0 : (int64_t)(I.v0 + li);
}

static inline int grid_for(int64_t n, int nt = 256) { int64_t g = (n + nt - 1) / nt; if (g < 1) g = 1; if (g > 4096) g = 4096; return (int)g; }

extern "C" int pdp_instance_max(pdp_problem *p, const float *x, float *out, void *stream)
{
    PDP_REQUIRE(p && x && out, "NULL argument");
    reset_flags(p, ST(stream));
    hipLaunchKernelGGL(k_global_min, dim3(grid_for(p->V)), dim3(256), 0, ST(stream), x, (int64_t)p->V, p->flags, FL_GMIN0, FL_NAN0);
    hipLaunchKernelGGL(k_instance_max, dim3(p->B), dim3(PDP_NT), 0, ST(stream), make_view(p), x, out, FL_GMIN0, FL_NAN0);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

extern "C" int pdp_instance_argmax(pdp_problem *p, const float *x, int64_t *out, void *stream)
{
    PDP_REQUIRE(p && x && out, "NULL argument");
    reset_flags(p, ST(stream));
    hipLaunchKernelGGL(k_global_min, dim3(grid_for(p->V)), dim3(256), 0, ST(stream), x, (int64_t)p->V, p->flags, FL_GMIN0, FL_NAN0);
    hipLaunchKernelGGL(k_instance_argmax, dim3(p->B), dim3(PDP_NT), 0, ST(stream), make_view(p), x, out, FL_GMIN0, FL_NAN0);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- K1-K3 ------------------------------------------------------------------------------------------------------
// LOGX: dq is [E] and already holds the log-domain clause message (the adaptor form of the propagator, model type p-nd-np:
// logsigmoid of a learned projection, pdp_propagate.py:166-167) instead of [E,3] surveys whose first column goes through safe_log
// PHASE -1: the whole sweep in one launch, one workgroup per instance (its barriers separate the three phases).  PHASE 0 / 1 / 2: one
// phase per launch with gridDim.y workgroups per instance -- what a batch takes that cannot fill the chip with a workgroup per instance
// (the reference's default memory limit cuts configs[4] with -b 4 into segments of <= 31 instances x 4 replicas, up to 47 520 edges each:
// 124 workgroups of 256 threads ran the sweep in 0.35 ms, as long as 20 M edges take in one segment).  Same statements per edge / row.
template <bool LOGX, int PHASE>
__global__ void __launch_bounds__(PDP_NT) k_sp_propagate(PView pv, const float *dq, const float *dfs, const float *emask,
                                                         const uint8_t *amask, const float *iq, const float *ifs, float pi,
                                                         float *oq, float *ofs, float *xs, float *ys, float *Sw, float *Pw, float *Nw,
                                                         const uint32_t *stop /* FL_LOOP_STOP: a device-driven loop has ended, write nothing */)
{
    if (amask && __builtin_amdgcn_readfirstlane((int)__builtin_nontemporal_load(stop)) != 0) return;
    const Inst I = load_inst(pv, blockIdx.x);
    const int nt = (int)blockDim.x * (PHASE < 0 ? 1 : (int)gridDim.y), tid = (int)threadIdx.x + (PHASE < 0 ? 0 : (int)blockIdx.y * (int)blockDim.x);
    dq += (LOGX ? 1 : 3) * (size_t)I.e0; dfs += 2 * (size_t)I.e0; iq += 3 * (size_t)I.e0; ifs += 2 * (size_t)I.e0;
    oq += 3 * (size_t)I.e0; ofs += 2 * (size_t)I.e0;
    xs += I.e0; ys += I.e0; Sw += I.f0; Pw += I.v0; Nw += I.v0;
    const float *em = emask ? emask + I.e0 : nullptr;
    const float mask = amask ? (0.0f + (0.0f + (float)amask[I.b])) : 1.0f;
    const float L0 = pdp_safe_log(1.0f - pi * 0.0f, PDP_SP_EPS), L1 = pdp_safe_log(1.0f - pi * 1.0f, PDP_SP_EPS);
    if (PHASE < 0 || PHASE == 0)
    for (int e = tid; e < I.e; e += nt) {
        float x = LOGX ? dq[e] : pdp_safe_log(dq[3 * e], PDP_SP_EPS);
        float y = pdp_safe_log(1.0f - dfs[2 * e], PDP_SP_EPS);
        if (em) { x = x * em[e]; y = y * em[e]; }
        xs[e] = x; ys[e] = y;
    }
    if (PHASE < 0) __syncthreads();
    if (PHASE < 0 || PHASE == 1) {
        for (int c = tid; c < I.m; c += nt) {
            float acc = 0.0f;
            for (int k = I.f_ptr[c]; k < I.f_ptr[c + 1]; ++k) acc = acc + xs[I.f_edges[k]];
            Sw[c] = acc;
        }
        for (int v = tid; v < I.n; v += nt) {
            float P = 0.0f, N = 0.0f;
            for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
                const int e = I.v_edges[k];
                const float y = ys[e];
                const int sg = I.sgn[e];
                P = P + ((sg == 1) ? 1.0f : 0.0f) * y;
                N = N + ((sg == -1) ? 1.0f : 0.0f) * y;
            }
            Pw[v] = P; Nw[v] = N;
        }
    }
    if (PHASE < 0) __syncthreads();
    if (PHASE < 0 || PHASE == 2)
    for (int e = tid; e < I.e; e += nt) {
        const int v = I.e_var[e], c = I.e_fn[e];
        const float s = (float)I.sgn[e];
        const float agg = (0.0f + Sw[c]) - xs[e];
        const float eta = mask * pdp_safe_exp(agg) + (1.0f - mask) * ifs[2 * e];
        const float force = dfs[2 * e + 1];
        const SpOut o = d_sp_edge(s, Pw[v], Nw[v], ys[e], force, L0, L1);
        oq[3 * e + 0] = mask * o.qu + (1.0f - mask) * iq[3 * e + 0];
        oq[3 * e + 1] = mask * o.qs + (1.0f - mask) * iq[3 * e + 1];
        oq[3 * e + 2] = mask * o.dc + (1.0f - mask) * iq[3 * e + 2];
        ofs[2 * e + 0] = eta;
        ofs[2 * e + 1] = force;
    }
}

// workgroups per instance of the phase-split sweep: 1 (the fused launch) while a workgroup per instance fills the chip, otherwise enough
// to have ~4 workgroups per CU, but no more than the largest instance has edges for (64 per thread at least)
static int sp_sweep_rows(const pdp_problem *p)
{
    if (getenv("PDP_SP_SWEEP_FUSED")) return 1;
    const long cus = pdp_device_cus();
    if ((long)p->B >= 2 * cus || p->B <= 0) return 1;
    long rows = (4 * cus + p->B - 1) / p->B;
    const long by_size = ((long)p->max_e + 4L * PDP_NT - 1) / (4L * PDP_NT);
    if (rows > by_size) rows = by_size;
    if (rows > 64) rows = 64;
    return rows < 2 ? 1 : (int)rows;
}
template <bool LOGX>
static void sp_sweep_launch(pdp_problem *p, const float *dq, const float *dfs, const float *edge_mask, const uint8_t *active_mask, const float *init_q,
                            const float *init_fs, float pi, float *out_q, float *out_fs, hipStream_t st)
{
    const int rows = sp_sweep_rows(p);
    const PView pv = make_view(p);
    if (rows == 1) {
        pdp_note_kernel(PDP_TK_SP_SWEEP, LOGX ? "k_sp_propagate<true>" : "k_sp_propagate<false>");
        hipLaunchKernelGGL((k_sp_propagate<LOGX, -1>), dim3(p->B), dim3(PDP_NT), 0, st, pv, dq, dfs, edge_mask, active_mask, init_q, init_fs, pi, out_q, out_fs,
                           p->ws_e[0], p->ws_e[1], p->ws_f[0], p->ws_v[0], p->ws_v[1], p->flags + FL_LOOP_STOP);
        return;
    }
    pdp_note_kernel(PDP_TK_SP_SWEEP, LOGX ? "k_sp_propagate<true> in three phases" : "k_sp_propagate<false> in three phases");
    hipLaunchKernelGGL((k_sp_propagate<LOGX, 0>), dim3(p->B, rows), dim3(PDP_NT), 0, st, pv, dq, dfs, edge_mask, active_mask, init_q, init_fs, pi, out_q, out_fs,
                       p->ws_e[0], p->ws_e[1], p->ws_f[0], p->ws_v[0], p->ws_v[1], p->flags + FL_LOOP_STOP);
    hipLaunchKernelGGL((k_sp_propagate<LOGX, 1>), dim3(p->B, rows), dim3(PDP_NT), 0, st, pv, dq, dfs, edge_mask, active_mask, init_q, init_fs, pi, out_q, out_fs,
                       p->ws_e[0], p->ws_e[1], p->ws_f[0], p->ws_v[0], p->ws_v[1], p->flags + FL_LOOP_STOP);
    hipLaunchKernelGGL((k_sp_propagate<LOGX, 2>), dim3(p->B, rows), dim3(PDP_NT), 0, st, pv, dq, dfs, edge_mask, active_mask, init_q, init_fs, pi, out_q, out_fs,
                       p->ws_e[0], p->ws_e[1], p->ws_f[0], p->ws_v[0], p->ws_v[1], p->flags + FL_LOOP_STOP);
}

extern "C" int pdp_sp_propagate(pdp_problem *p, const float *dec_q, const float *dec_fs, const float *edge_mask,
                                const uint8_t *active_mask, const float *init_q, const float *init_fs, float pi,
                                float *out_q, float *out_fs, void *stream)
{
    PDP_REQUIRE(p && dec_q && dec_fs && init_q && init_fs && out_q && out_fs, "NULL argument");
    PDP_REQUIRE(out_q != dec_q && out_q != init_q && out_fs != dec_fs && out_fs != init_fs, "outputs must not alias inputs");
    pdp_timed_scope timed(PDP_TK_SP_SWEEP, ST(stream));
    sp_sweep_launch<false>(p, dec_q, dec_fs, edge_mask, active_mask, init_q, init_fs, pi, out_q, out_fs, ST(stream));
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- adaptor form of the propagator (model type p-nd-np) -----------------------------------------------------------------------
// replaces: the include_adaptors=True branches of SurveyPropagator.forward (pdp_propagate.py:166-167, 179-182): per edge
//   xlog = logsigmoid(w_f . dec_v[e]),  eta = sigmoid(W_v[0] . dec_f[e]),  force = sign(W_v[1] . dec_f[e])
// with every dot product the k-ascending fmaf chain from 0 that the oracle computes.  A wave takes 64 edges: their rows go through LDS
// (coalesced 128-byte row segments in, one lane per edge out), 32 columns at a time.
__global__ void __launch_bounds__(64) k_sp_adaptors(int E, int H, const float *__restrict__ dv, const float *__restrict__ df,
                                                    const float *__restrict__ wf, const float *__restrict__ Wv,
                                                    float *__restrict__ xlog, float *__restrict__ fs2)
{
    __shared__ float T[64 * 33];                           // 64 edges x 32 columns (8.4 KB: many single-wave workgroups per CU)
    const int l = threadIdx.x, half = l >> 5, c = l & 31;
    for (int64_t e0 = (int64_t)blockIdx.x * 64; e0 < E; e0 += (int64_t)gridDim.x * 64) {
        const int64_t e = e0 + l;
        float a = 0.0f, b0 = 0.0f, b1 = 0.0f;
        for (int pass = 0; pass < 2; ++pass) {
            const float *src = pass == 0 ? dv : df;
            for (int c0 = 0; c0 < H; c0 += 32) {
                const int nc = H - c0 < 32 ? H - c0 : 32;
                float v[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) {             // rows 2j and 2j + 1: two 128-byte row segments per load instruction
                    const int64_t er = e0 + 2 * j + half;
                    v[j] = (er < E && c < nc) ? src[er * H + c0 + c] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 32; ++j) T[(2 * j + half) * 33 + c] = v[j];
                if (pass == 0) { for (int k = 0; k < nc; ++k) a = fmaf(T[l * 33 + k], wf[c0 + k], a); }
                else { for (int k = 0; k < nc; ++k) { const float x = T[l * 33 + k]; b0 = fmaf(x, Wv[c0 + k], b0); b1 = fmaf(x, Wv[H + c0 + k], b1); } }
            }
        }
        if (e < E) {
            xlog[e] = pdp_logsigmoidf(a);
            fs2[2 * e + 0] = pdp_sigmoidf(b0);
            fs2[2 * e + 1] = (b1 != b1) ? b1 : pdp_sign(b1);                 // torch.sign(NaN) = NaN
        }
    }
}

extern "C" int pdp_sp_adaptors(pdp_problem *p, int H, const float *dec_v, const float *dec_f, const float *w_f, const float *W_v,
                               float *xlog, float *fs2, void *stream)
{
    PDP_REQUIRE(p && dec_v && dec_f && w_f && W_v && xlog && fs2 && H > 0, "NULL argument");
    const int64_t groups = ((int64_t)p->E + 63) / 64;
    pdp_timed_scope timed(PDP_TK_SP_ADAPTORS, ST(stream));
    pdp_note_kernel(PDP_TK_SP_ADAPTORS, "k_sp_adaptors");
    hipLaunchKernelGGL(k_sp_adaptors, dim3((unsigned)(groups < 16384 ? (groups < 1 ? 1 : groups) : 16384)), dim3(64), 0, ST(stream), p->E, H, dec_v, dec_f, w_f, W_v, xlog, fs2);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

extern "C" int pdp_sp_propagate_adapted(pdp_problem *p, const float *xlog, const float *dec_fs, const float *edge_mask,
                                        const uint8_t *active_mask, const float *init_q, const float *init_fs, float pi,
                                        float *out_q, float *out_fs, void *stream)
{
    PDP_REQUIRE(p && xlog && dec_fs && init_q && init_fs && out_q && out_fs, "NULL argument");
    PDP_REQUIRE(out_q != init_q && out_fs != dec_fs && out_fs != init_fs, "outputs must not alias inputs");
    pdp_timed_scope timed(PDP_TK_SP_SWEEP, ST(stream));
    sp_sweep_launch<true>(p, xlog, dec_fs, edge_mask, active_mask, init_q, init_fs, pi, out_q, out_fs, ST(stream));
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- K6 ----------------------------------------------------------------------------------------------------------
// score [V]; optionally coeff = |score| * active * conv and the batch-global min / any-positive / NaN flags
__device__ void d_survey_score(const Inst &I, const float *fs /*inst slice [e,2]*/, float pi, float *fm /*[e] scratch*/, float *score /*[n]*/)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int e = tid; e < I.e; e += nt)
        fm[e] = pdp_safe_log(1.0f - fs[2 * e], PDP_SCORER_EPS) * (0.0f + I.af[I.e_fn[e]]);
    __syncthreads();
    for (int v = tid; v < I.n; v += nt) {
        float ext = 0.0f, pos = 0.0f, neg = 0.0f, all = 0.0f;
        for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
            const int e = I.v_edges[k];
            const int sg = I.sgn[e];
            const float f = fm[e];
            ext = ext + fs[2 * e + 1];
            pos = pos + ((sg == 1) ? 1.0f : 0.0f) * f;
            neg = neg + ((sg == -1) ? 1.0f : 0.0f) * f;
            all = all + f;
        }
        score[v] = d_score_from_sums(pos, neg, all, ext, pi);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(PDP_NT) k_survey_score(PView pv, const float *fs, float pi, float *score, float *fm)
{
    const Inst I = load_inst(pv, blockIdx.x);
    d_survey_score(I, fs + 2 * (size_t)I.e0, pi, fm + I.e0, score + I.v0);
}

extern "C" int pdp_survey_score(pdp_problem *p, const float *fs, float pi, float *score, void *stream)
{
    PDP_REQUIRE(p && fs && score, "NULL argument");
    hipLaunchKernelGGL(k_survey_score, dim3(p->B), dim3(PDP_NT), 0, ST(stream), make_view(p), fs, pi, score, p->ws_e[0]);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- K9 / K13 -------------------------------------------------------------------------------------------------------
template <bool TEAM>
__global__ void __launch_bounds__(PDP_NT) k_cnf_eval(PView pv, const float *pred, float *solved, float *unsat, TeamLaunch tl)
{
    DECL_RED
    (void)redf;
    std::conditional_t<TEAM, Teamed<Inst>, Inst> I;
    int slot = (int)blockIdx.x;
    if constexpr (TEAM) { slot = team_begin(I, tl, redi); if (slot < 0) return; }
    static_cast<Inst &>(I) = load_inst(pv, slot);
    const int nsat = d_cnf_sat_count(I, pred + I.v0, redi);
    if (team_tid(I) == 0) {
        const float max_sat = (float)I.m, bv = (float)nsat;
        solved[I.b] = (max_sat == bv) ? 1.0f : 0.0f;
        unsat[I.b] = max_sat - bv;
    }
    if constexpr (TEAM) { if (I.failed && threadIdx.x == 0) atomicOr(&pv.flags[FL_TEAM_TIMEOUT], 1u); }
}

static int launch_cnf_eval(pdp_problem *p, const float *pred, float *solved, float *unsat, hipStream_t st)
{
    TeamLaunch tl;
    { const int st_ = simplify_plan(p, &tl, st); if (st_ != PDP_OK) return st_; }          // few big instances: teams
    if (tl.size > 1) hipLaunchKernelGGL((k_cnf_eval<true>), dim3(tl.size * tl.slots), dim3(PDP_NT), 0, st, make_view(p), pred, solved, unsat, tl);
    else hipLaunchKernelGGL((k_cnf_eval<false>), dim3(p->B), dim3(PDP_NT), 0, st, make_view(p), pred, solved, unsat, tl);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

extern "C" int pdp_cnf_eval(pdp_problem *p, const float *pred, float *solved, float *unsat, void *stream)
{
    PDP_REQUIRE(p && pred && solved && unsat, "NULL argument");
    return launch_cnf_eval(p, pred, solved, unsat, ST(stream));
}

// ---- energy loss of a prediction (test mode: SatLossEvaluator.forward, util.py:178-197) ----------------------------------------
// per edge   ev = s x_v + (1 - s) / 2,  w = exp(coeff ev);  per clause (edges in ascending id)  nom = sum w ev,  den = sum w,
// cv = 1 + (den / max(nom, eps) - 1)^sharpness,  term = log(max(cv, eps));  loss = mean over all clauses of the batch.
// The mean is taken in a fixed order (clauses of an instance in ascending id by one thread, then the instances in ascending id) so that
// the CPU oracle reproduces it bit for bit; the reference's torch.mean uses another order (compared with a tolerance).
__global__ void __launch_bounds__(PDP_NT) k_sat_loss(PView pv, const float *pred, float coeff, float eps, int sharpness, float *term, float *inst_sum)
{
    const Inst I = load_inst(pv, blockIdx.x);
    pred += I.v0; term += I.f0;
    for (int c = threadIdx.x; c < I.m; c += blockDim.x) {
        float nom = 0.0f, den = 0.0f;
        for (int k = I.f_ptr[c]; k < I.f_ptr[c + 1]; ++k) {
            const int e = I.f_edges[k];
            const float s = (float)I.sgn[e];
            const float ev = s * pred[I.e_var[e]] + (1.0f - s) / 2.0f;
            const float w = pdp_expf(coeff * ev);
            nom = nom + w * ev; den = den + w;
        }
        const float d = den / pdp_max_c(nom, eps) - 1.0f;
        float pw = d;
        for (int j = 1; j < sharpness; ++j) pw = pw * d;
        term[c] = pdp_safe_log(1.0f + pw, eps);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = 0.0f;
        for (int c = 0; c < I.m; ++c) acc = acc + term[c];
        inst_sum[I.b] = acc;
    }
}

__global__ void k_sat_loss_finish(int B, int F, const float *inst_sum, float *loss)
{
    float acc = 0.0f;
    for (int b = 0; b < B; ++b) acc = acc + inst_sum[b];
    loss[0] = acc / (float)F;
}

extern "C" int pdp_sat_loss(pdp_problem *p, const float *pred, float coeff, float eps, int sharpness, float *loss, void *stream)
{
    PDP_REQUIRE(p && pred && loss, "NULL argument");
    PDP_REQUIRE(sharpness >= 1, "loss_sharpness must be a positive integer");
    hipLaunchKernelGGL(k_sat_loss, dim3(p->B), dim3(PDP_NT), 0, ST(stream), make_view(p), pred, coeff, eps, sharpness, p->ws_f[0], p->ws_b[0]);
    hipLaunchKernelGGL(k_sat_loss_finish, dim3(1), dim3(1), 0, ST(stream), p->B, p->F, p->ws_b[0], loss);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

__global__ void k_update_solution(int V, const float *av, float *sol, const float *pred, float *out)
{
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
        const float a = av[v];
        const float r = a * pred[v] + (1.0f - a) * sol[v];
        out[v] = r;
        if (a == 1.0f) sol[v] = r;
    }
}

extern "C" int pdp_update_solution(pdp_problem *p, const float *pred, float *out, void *stream)
{
    PDP_REQUIRE(p && p->av && pred && out, "NULL argument / state not bound");
    hipLaunchKernelGGL(k_update_solution, dim3(grid_for(p->V)), dim3(256), 0, ST(stream), p->V, p->av, p->sol, pred, out);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

__global__ void k_termination(int B0, int R, const float *solved, uint8_t *amask)
{
    const int b0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (b0 >= B0) return;
    if (R > 1) {
        float real = 0.0f;
        for (int r = 0; r < R; ++r) real = real + ((solved[b0 + r * B0] > 0.5f) ? 1.0f : 0.0f);
        for (int r = 0; r < R; ++r) { const int b = b0 + r * B0; if (amask[b]) amask[b] = (real == 0.0f) ? 1 : 0; }
    } else {
        if (amask[b0]) amask[b0] = (solved[b0] <= 0.5f) ? 1 : 0;
    }
}

extern "C" int pdp_check_termination(pdp_problem *p, uint8_t *active_mask, const float *pred, void *stream)
{
    PDP_REQUIRE(p && active_mask && pred, "NULL argument");
    { const int st_ = launch_cnf_eval(p, pred, p->ws_b[0], p->ws_b[1], ST(stream)); if (st_ != PDP_OK) return st_; }
    hipLaunchKernelGGL(k_termination, dim3((p->B0 + 255) / 256), dim3(256), 0, ST(stream), p->B0, p->R, p->ws_b[0], active_mask);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- device-driven step-wise loop (the plug-in loop of solver.py:355-386 without its per-sweep host read) -----------------------------------
// The host enqueues (or replays, as a captured HIP graph) one sweep after the other; pdp_loop_step closes a sweep on the device: it counts
// the sweep and raises FL_LOOP_STOP once `int(active_mask.sum()) <= 0` (solver.py:383-384).  Behind the stop the kernels that write the
// message states return at once (pdp_neural.hip: loop_stopped; k_sp_propagate), so whatever the host still has in flight changes nothing:
// the states, the solution and the mask are the ones of the sweep that ended the loop, and FL_LOOP_ITERS is the reference's iteration count.
__global__ void __launch_bounds__(256) k_loop_step(int B, const uint8_t *amask, uint32_t *words /* [stop, iterations] */)
{
    if (words[0]) return;
    int any = 0;
    if (amask) { for (int b = threadIdx.x; b < B; b += blockDim.x) any |= amask[b] ? 1 : 0; }
    else any = 1;
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) { words[1] += 1u; if (!any) words[0] = 1u; }
}
extern "C" int pdp_loop_begin(pdp_problem *p, void *stream)
{
    PDP_REQUIRE(p, "NULL argument");
    PDP_HIP_CHECK(hipMemsetAsync(p->flags + FL_LOOP_STOP, 0, 2 * sizeof(uint32_t), ST(stream)));
    return PDP_OK;
}
extern "C" int pdp_loop_step(pdp_problem *p, const uint8_t *active_mask, void *stream)
{
    PDP_REQUIRE(p, "NULL argument");
    static_assert(FL_LOOP_ITERS == FL_LOOP_STOP + 1, "the two loop words are adjacent");
    hipLaunchKernelGGL(k_loop_step, dim3(1), dim3(256), 0, ST(stream), p->B, active_mask, p->flags + FL_LOOP_STOP);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}
extern "C" int pdp_loop_read(pdp_problem *p, int32_t *stopped_host, int32_t *iterations_host, int end, void *stream)
{
    PDP_REQUIRE(p && stopped_host && iterations_host, "NULL argument");
    PDP_HIP_CHECK(hipMemcpyAsync(p->flags_host + FL_LOOP_STOP, p->flags + FL_LOOP_STOP, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ST(stream)));
    if (end) PDP_HIP_CHECK(hipMemsetAsync(p->flags + FL_LOOP_STOP, 0, 2 * sizeof(uint32_t), ST(stream)));    // the loop is over: later calls write again
    PDP_HIP_CHECK(hipStreamSynchronize(ST(stream)));
    *stopped_host = (int32_t)p->flags_host[FL_LOOP_STOP]; *iterations_host = (int32_t)p->flags_host[FL_LOOP_ITERS];
    return PDP_OK;
}

// ---- decimators ---------------------------------------------------------------------------------------------------------
extern "C" int pdp_decimator_create(pdp_decimator **out, pdp_problem *p)
{
    PDP_REQUIRE(out && p, "NULL argument");
    pdp_decimator *d = new pdp_decimator();
    d->p = p; d->has_prev = 0; d->prev = nullptr; d->counters = nullptr;
    { int st_ = pdp_dev_alloc((void **)&d->prev, sizeof(float) * (size_t)p->E); if (st_ != PDP_OK) return st_; }
    { int st_ = pdp_dev_alloc((void **)&d->counters, sizeof(float) * (size_t)p->B); if (st_ != PDP_OK) return st_; }
    PDP_HIP_CHECK(hipMemset(d->counters, 0, sizeof(float) * (size_t)p->B));
    PDP_HIP_CHECK(hipMemset(d->prev, 0, sizeof(float) * (size_t)p->E));
    *out = d;
    return PDP_OK;
}
extern "C" int pdp_decimator_destroy(pdp_decimator *d)
{
    if (!d) return PDP_OK;
    if (d->prev) pdp_dev_free(d->prev);
    if (d->counters) pdp_dev_free(d->counters);
    delete d;
    return PDP_OK;
}
extern "C" int pdp_decimator_reset(pdp_decimator *d, void *stream)
{
    PDP_REQUIRE(d, "NULL argument");
    d->has_prev = 0;
    PDP_HIP_CHECK(hipMemsetAsync(d->counters, 0, sizeof(float) * (size_t)d->p->B, ST(stream)));
    return PDP_OK;
}

// (1) survey gate input: xv = smooth_max(eta) * active; batch-global min -> GMIN0; any active variable -> flag
__global__ void __launch_bounds__(PDP_NT) k_dec_survey(PView pv, const float *fs, float *eta_ws, float *xv)
{
    DECL_RED
    const Inst I = load_inst(pv, blockIdx.x);
    const int tid = threadIdx.x, nt = blockDim.x;
    fs += 2 * (size_t)I.e0; eta_ws += I.e0; xv += I.v0;
    for (int e = tid; e < I.e; e += nt) eta_ws[e] = fs[2 * e];
    __syncthreads();
    float m = PDP_INF; bool nn = false; int anyact = 0;
    for (int v = tid; v < I.n; v += nt) {
        const float r = d_smooth_max_var(I, v, eta_ws) * I.av[v];
        xv[v] = r;
        if (r != r) nn = true; else if (r < m) m = r;
        anyact |= (I.av[v] > 0.0f) ? 1 : 0;
    }
    publish_min(m, nn, pv.flags, FL_GMIN0, FL_NAN0, redf, redi);
    anyact = __syncthreads_or(anyact);
    if (tid == 0 && anyact) atomicOr(&pv.flags[FL_ANY_ACTIVE_VAR], 1u);
}

// (2) active_mask[survey <= 1e-10] = 0
__global__ void __launch_bounds__(PDP_NT) k_dec_gate(PView pv, const float *xv, uint8_t *amask)
{
    DECL_RED
    (void)redi;
    const Inst I = load_inst(pv, blockIdx.x);
    const float gmin = read_gmin(pv.flags, FL_GMIN0, FL_NAN0);
    const float mx = d_instance_max(I, xv + I.v0, gmin, I.n < pv.V, redf);
    if (threadIdx.x == 0 && mx <= 1e-10f) amask[I.b] = 0;
}

// (3) xv = smooth_max(|prev - eta| * edge_mask) * active; global min -> GMIN1
__global__ void __launch_bounds__(PDP_NT) k_dec_diff(PView pv, const float *fs, const float *prev, int use_emask, float *dws, float *xv)
{
    DECL_RED
    if (pv.flags[FL_ANY_ACTIVE_VAR] == 0u) return;
    const Inst I = load_inst(pv, blockIdx.x);
    const int tid = threadIdx.x, nt = blockDim.x;
    fs += 2 * (size_t)I.e0; prev += I.e0; dws += I.e0; xv += I.v0;
    for (int e = tid; e < I.e; e += nt) {
        float d = pdp_abs(prev[e] - fs[2 * e]);
        if (use_emask) d = d * I.emask[e];
        dws[e] = d;
    }
    __syncthreads();
    float m = PDP_INF; bool nn = false;
    for (int v = tid; v < I.n; v += nt) {
        const float r = d_smooth_max_var(I, v, dws) * I.av[v];
        xv[v] = r;
        if (r != r) nn = true; else if (r < m) m = r;
    }
    publish_min(m, nn, pv.flags, FL_GMIN1, FL_NAN1, redf, redi);
}

// (4) convergence bookkeeping (pdp_decimate.py:143-150)
__global__ void __launch_bounds__(PDP_NT) k_dec_conv(PView pv, const float *xv, float *counters, float tol, float t_max, float *conv_b)
{
    DECL_RED
    (void)redi;
    if (pv.flags[FL_ANY_ACTIVE_VAR] == 0u) return;
    const Inst I = load_inst(pv, blockIdx.x);
    const float gmin = read_gmin(pv.flags, FL_GMIN1, FL_NAN1);
    const float sd = d_instance_max(I, xv + I.v0, gmin, I.n < pv.V, redf);
    if (threadIdx.x == 0) {
        float cnt = counters[I.b];
        if (sd < tol) cnt = 0.0f;
        float conv = (sd < tol) ? 1.0f : 0.0f;
        if (cnt >= t_max) conv = 1.0f;
        if (cnt >= t_max) cnt = 0.0f;
        counters[I.b] = cnt;
        conv_b[I.b] = conv;
        if (conv > 0.0f && I.n > 0) atomicOr(&pv.flags[FL_ANY_CONV], 1u);
    }
}

// (5) score + coeff; global min of coeff -> GMIN2; any positive / NaN flags
__global__ void __launch_bounds__(PDP_NT) k_dec_score(PView pv, const float *fs, float pi, const float *ext_score, const float *conv_b,
                                                      float *fm, float *score, float *coeff)
{
    DECL_RED
    if (pv.flags[FL_ANY_ACTIVE_VAR] == 0u || pv.flags[FL_ANY_CONV] == 0u) return;
    const Inst I = load_inst(pv, blockIdx.x);
    const int tid = threadIdx.x, nt = blockDim.x;
    score += I.v0; coeff += I.v0;
    if (ext_score) { for (int v = tid; v < I.n; v += nt) score[v] = ext_score[I.v0 + v]; __syncthreads(); }
    else d_survey_score(I, fs + 2 * (size_t)I.e0, pi, fm + I.e0, score);
    const float conv = 0.0f + conv_b[I.b];
    float m = PDP_INF; bool nn = false; int anypos = 0;
    for (int v = tid; v < I.n; v += nt) {
        const float c = (pdp_abs(score[v]) * I.av[v]) * conv;
        coeff[v] = c;
        if (c != c) nn = true; else { if (c < m) m = c; if (c > 0.0f) anypos = 1; }
    }
    publish_min(m, nn, pv.flags, FL_GMIN2, FL_NAN2, redf, redi);
    anypos = __syncthreads_or(anypos);
    if (tid == 0 && anypos) atomicOr(&pv.flags[FL_ANY_POS], 1u);
}

// (6) arg-max per instance -> assignment (pdp_decimate.py:158-169)
__global__ void __launch_bounds__(PDP_NT) k_dec_pick(PView pv, const float *score, const float *coeff, const uint8_t *amask, float *assignment)
{
    DECL_RED
    // `coeff.sum() > 0` is False when the sum is NaN
    if (pv.flags[FL_ANY_ACTIVE_VAR] == 0u || pv.flags[FL_ANY_CONV] == 0u || pv.flags[FL_ANY_POS] == 0u || pv.flags[FL_NAN2] != 0u) return;
    const Inst I = load_inst(pv, blockIdx.x);
    const int tid = threadIdx.x, nt = blockDim.x;
    score += I.v0; coeff += I.v0; assignment += I.v0;
    const float gmin = read_gmin(pv.flags, FL_GMIN2, FL_NAN2);
    int anypos = 0;
    for (int v = tid; v < I.n; v += nt) { assignment[v] = 0.0f; anypos |= (coeff[v] != 0.0f) ? 1 : 0; }
    anypos = __syncthreads_or(anypos);                       // norm != 0  <=>  some coeff != 0 (coeff >= 0, no NaN here)
    const int li = d_instance_argmax(I, coeff, gmin, redf, redi);
    const int sel = anypos && (amask ? (amask[I.b] != 0) : 1);
    if (tid == 0 && sel && li >= 0) {
        assignment[li] = pdp_sign(score[li]);
        atomicAdd(&pv.flags[FL_N_SEL], 1u);
    }
}

// (8) counters += 1 (only when block (3)-(7) executed); prev = eta
__global__ void __launch_bounds__(PDP_NT) k_dec_finish(PView pv, const float *fs, float *prev, float *counters, int had_prev)
{
    const Inst I = load_inst(pv, blockIdx.x);
    fs += 2 * (size_t)I.e0; prev += I.e0;
    for (int e = threadIdx.x; e < I.e; e += blockDim.x) prev[e] = fs[2 * e];
    if (threadIdx.x == 0 && had_prev && pv.flags[FL_ANY_ACTIVE_VAR] != 0u) counters[I.b] = counters[I.b] + 1.0f;
}

static int decimate_gate_chain(pdp_problem *p, pdp_decimator *d, const float *fs, uint8_t *active_mask, float tol, float t_max, hipStream_t st)
{
    const PView pv = make_view(p);
    reset_flags(p, st);
    // the survey kernel also publishes FL_ANY_ACTIVE_VAR, so it runs even without an active mask
    hipLaunchKernelGGL(k_dec_survey, dim3(p->B), dim3(PDP_NT), 0, st, pv, fs, p->ws_e[0], p->ws_v[0]);
    if (active_mask) hipLaunchKernelGGL(k_dec_gate, dim3(p->B), dim3(PDP_NT), 0, st, pv, p->ws_v[0], active_mask);
    if (d->has_prev) {
        hipLaunchKernelGGL(k_dec_diff, dim3(p->B), dim3(PDP_NT), 0, st, pv, fs, d->prev, p->has_edge_mask, p->ws_e[1], p->ws_v[1]);
        hipLaunchKernelGGL(k_dec_conv, dim3(p->B), dim3(PDP_NT), 0, st, pv, p->ws_v[1], d->counters, tol, t_max, p->ws_b[2]);
    }
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

static int decimate_apply_chain(pdp_problem *p, pdp_decimator *d, const float *fs, const float *ext_score, const uint8_t *active_mask,
                                float pi, hipStream_t st)
{
    const PView pv = make_view(p);
    if (d->has_prev) {
        hipLaunchKernelGGL(k_dec_score, dim3(p->B), dim3(PDP_NT), 0, st, pv, fs, pi, ext_score, p->ws_b[2], p->ws_e[2], p->ws_v[2], p->ws_v[3]);
        hipLaunchKernelGGL(k_dec_pick, dim3(p->B), dim3(PDP_NT), 0, st, pv, p->ws_v[2], p->ws_v[3], active_mask, p->ws_v[4]);
        TeamLaunch tl;
        { const int st_ = simplify_plan(p, &tl, st); if (st_ != PDP_OK) return st_; }
        if (tl.size > 1)
            hipLaunchKernelGGL((k_set_variables<true>), dim3(tl.size * tl.slots), dim3(PDP_NT), 0, st, pv, p->ws_v[4], p->ws_v[5], p->ws_vi[0], p->ws_vi[1], p->ws_vi[2],
                               p->ws_fu[0], p->ws_fu[1], (int)FL_N_SEL, tl);
        else
            hipLaunchKernelGGL((k_set_variables<false>), dim3(p->B), dim3(PDP_NT), 0, st, pv, p->ws_v[4], p->ws_v[5], p->ws_vi[0], p->ws_vi[1], p->ws_vi[2],
                               p->ws_fu[0], p->ws_fu[1], (int)FL_N_SEL, tl);
    }
    hipLaunchKernelGGL(k_dec_finish, dim3(p->B), dim3(PDP_NT), 0, st, pv, fs, d->prev, d->counters, d->has_prev);
    PDP_LAUNCH_CHECK();
    d->has_prev = 1;
    return PDP_OK;
}

extern "C" int pdp_sequential_decimate(pdp_problem *p, pdp_decimator *d, const float *fs, uint8_t *active_mask,
                                       float tolerance, float t_max, float pi, void *stream)
{
    PDP_REQUIRE(p && d && fs && p->av, "NULL argument / state not bound");
    int s = decimate_gate_chain(p, d, fs, active_mask, tolerance, t_max, ST(stream));
    if (s != PDP_OK) return s;
    return decimate_apply_chain(p, d, fs, nullptr, active_mask, pi, ST(stream));
}

extern "C" int pdp_sequential_decimate_gate(pdp_problem *p, pdp_decimator *d, const float *fs, uint8_t *active_mask,
                                            float tolerance, float t_max, int32_t *any_converged_host, void *stream)
{
    PDP_REQUIRE(p && d && fs && p->av, "NULL argument / state not bound");
    int s = decimate_gate_chain(p, d, fs, active_mask, tolerance, t_max, ST(stream));
    if (s != PDP_OK) return s;
    if (any_converged_host) {
        s = read_flags(p, ST(stream));
        if (s != PDP_OK) return s;
        *any_converged_host = (d->has_prev && p->flags_host[FL_ANY_ACTIVE_VAR] && p->flags_host[FL_ANY_CONV]) ? 1 : 0;
    }
    return PDP_OK;
}

extern "C" int pdp_sequential_decimate_apply(pdp_problem *p, pdp_decimator *d, const float *fs, const float *score,
                                             const uint8_t *active_mask, void *stream)
{
    PDP_REQUIRE(p && d && fs && p->av, "NULL argument / state not bound");
    return decimate_apply_chain(p, d, fs, score, active_mask, 0.0f, ST(stream));
}

// Reinforce decimator (pdp_decimate.py:202-234)
__global__ void __launch_bounds__(PDP_NT) k_reinforce_gate(PView pv, const float *xv, uint8_t *amask)
{
    DECL_RED
    (void)redi;
    if (pv.flags[FL_ANY_ACTIVE_VAR] == 0u) return;
    const Inst I = load_inst(pv, blockIdx.x);
    const float gmin = read_gmin(pv.flags, FL_GMIN1, FL_NAN1);
    const float sd = d_instance_max(I, xv + I.v0, gmin, I.n < pv.V, redf);
    if (threadIdx.x == 0 && sd <= 0.01f) amask[I.b] = 0;
}

__global__ void __launch_bounds__(PDP_NT) k_reinforce_force(PView pv, float *fs, const uint8_t *amask, float pi, float *fm, float *score)
{
    const Inst I = load_inst(pv, blockIdx.x);
    fs += 2 * (size_t)I.e0; score += I.v0;
    d_survey_score(I, fs, pi, fm + I.e0, score);
    const float mask = amask ? (0.0f + (0.0f + (float)amask[I.b])) : 1.0f;
    for (int e = threadIdx.x; e < I.e; e += blockDim.x) {
        const float sc = 0.0f + pdp_sign(score[I.e_var[e]]);       // torch.sign(NaN) is 0
        fs[2 * e + 1] = mask * sc + (1.0f - mask) * fs[2 * e + 1];
    }
}

__global__ void __launch_bounds__(PDP_NT) k_any_active(PView pv)
{
    const Inst I = load_inst(pv, blockIdx.x);
    int anyact = 0;
    for (int v = threadIdx.x; v < I.n; v += blockDim.x) anyact |= (I.av[v] > 0.0f) ? 1 : 0;
    anyact = __syncthreads_or(anyact);
    if (threadIdx.x == 0 && anyact) atomicOr(&pv.flags[FL_ANY_ACTIVE_VAR], 1u);
}

__global__ void __launch_bounds__(PDP_NT) k_copy_eta(PView pv, const float *fs, float *prev)
{
    const Inst I = load_inst(pv, blockIdx.x);
    fs += 2 * (size_t)I.e0; prev += I.e0;
    for (int e = threadIdx.x; e < I.e; e += blockDim.x) prev[e] = fs[2 * e];
}

extern "C" int pdp_reinforce_decimate(pdp_problem *p, pdp_decimator *d, float *fs, uint8_t *active_mask, float coin,
                                      float decimation_probability, float pi, void *stream)
{
    PDP_REQUIRE(p && d && fs && p->av, "NULL argument / state not bound");
    hipStream_t st = ST(stream);
    const PView pv = make_view(p);
    if (active_mask && d->has_prev) {
        reset_flags(p, st);
        hipLaunchKernelGGL(k_any_active, dim3(p->B), dim3(PDP_NT), 0, st, pv);
        hipLaunchKernelGGL(k_dec_diff, dim3(p->B), dim3(PDP_NT), 0, st, pv, fs, d->prev, p->has_edge_mask, p->ws_e[1], p->ws_v[1]);
        hipLaunchKernelGGL(k_reinforce_gate, dim3(p->B), dim3(PDP_NT), 0, st, pv, p->ws_v[1], active_mask);
    }
    hipLaunchKernelGGL(k_copy_eta, dim3(p->B), dim3(PDP_NT), 0, st, pv, fs, d->prev);
    d->has_prev = 1;
    if (coin < decimation_probability)
        hipLaunchKernelGGL(k_reinforce_force, dim3(p->B), dim3(PDP_NT), 0, st, pv, fs, active_mask, pi, p->ws_e[2], p->ws_v[2]);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

__global__ void __launch_bounds__(PDP_NT) k_reinforce_predict(PView pv, const float *fs, float *pred)
{
    const Inst I = load_inst(pv, blockIdx.x);
    fs += 2 * (size_t)I.e0;
    for (int v = threadIdx.x; v < I.n; v += blockDim.x) {
        float acc = 0.0f;
        for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) acc = acc + fs[2 * I.v_edges[k] + 1];
        pred[I.v0 + v] = (acc > 0.0f) ? 1.0f : 0.0f;
    }
}

extern "C" int pdp_reinforce_predict(pdp_problem *p, const float *fs, float *pred, void *stream)
{
    PDP_REQUIRE(p && fs && pred, "NULL argument");
    hipLaunchKernelGGL(k_reinforce_predict, dim3(p->B), dim3(PDP_NT), 0, ST(stream), make_view(p), fs, pred);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- math probe -------------------------------------------------------------------------------------------------------------
__global__ void k_math_apply(int fn, const float *x, float *y, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float r;
        switch (fn) {
        case 0: r = pdp_expf(x[i]); break;
        case 1: r = pdp_logf(x[i]); break;
        case 2: r = pdp_logsigmoidf(x[i]); break;
        case 3: r = pdp_sigmoidf(x[i]); break;
        case 4: r = pdp_tanhf(x[i]); break;
        case 13: r = pdp_tanhf_abs(x[i]); break;
        case 14: r = pdp_rcp_ge1(x[i]); break;
        case 5: r = pdp_safe_exp(x[i]); break;
        case 6: r = pdp_safe_log(x[i], PDP_SP_EPS); break;
        case 7: r = pdp_philox_uniform(0x1234abcdULL, 2u, 7u, (uint32_t)i); break;
        case 8: r = 1.0f / x[i]; break;
        case 9: r = pdp_safe_exp_fast(x[i]); break;
        case 10: r = pdp_safe_log_fin(x[i], PDP_SP_EPS); break;
        case 11: r = pdp_safe_log_fin(x[i], PDP_SCORER_EPS); break;
        case 12: r = pdp_expf_fin_le30(x[i]); break;
        default: r = x[i];
        }
        y[i] = r;
    }
}

extern "C" int pdp_math_apply(int fn, const float *x, float *y, int64_t n, void *stream)
{
    PDP_REQUIRE(x && y && n >= 0, "bad argument");
    hipLaunchKernelGGL(k_math_apply, dim3(grid_for(n)), dim3(256), 0, ST(stream), fn, x, y, n);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}
