// pdp_solve.hip -- persistent solver: the whole PDP iteration loop in ONE launch.
// replaces: PropagatorDecimatorSolverBase._forward_core (reference: src/pdp/nn/solver.py:355-386) for the
// classical triple SurveyPropagator + SequentialDecimator(SurveyScorer) + IdentityPredictor together with the
// per-iteration termination check (src/pdp/trainer.py:150-162).
//
// One workgroup owns one CNF instance for all T iterations.  When the instance fits (the common case: the
// BASELINE configs have <= ~5k edges per instance) its topology and its whole message state live in LDS and HBM
// is touched twice: load at entry, store at exit.  Larger instances run the same code on HBM-resident arrays.
//
// Cross-instance couplings of the reference cannot be honoured inside independent
// workgroups, so the kernel SPECULATES that they are inert -- batch-global min of each arg-max/max operand is 0
// and no NaN appears -- and records per (iteration, call site) whether some instance really had an exact zero.
// The host verifies the record; on a miss the call returns PDP_ERR_SPECULATION and the caller reruns the batch
// through the strict step-wise entry points (pdp_ops.hip), so results always equal the reference semantics.
#include "pdp_device.hpp"
#include <stdlib.h>

#define ST(s) ((hipStream_t)(s))

// instance view used by the solver: same field names as Inst, index type templated (u16 in LDS, i32 in HBM)
template <class IT>
struct SView {
    int b, n, m, e;
    const IT *e_var, *e_fn, *v_ptr, *v_edges, *f_ptr, *f_edges;
    const int8_t *sgn;
    float *av, *af, *sol;
    float *emask;           // [e]
    float *qu; int qstride; // q[:,0]
    float *eta; int estride;// fs[:,0]
    const float *force; int fstride;   // fs[:,1]
    float *s0, *s1, *s2, *s3;           // [e] scratch
    float *S;               // [m]
    float *P, *N, *xv1, *xv2, *score, *coeff, *assign;   // [n]
    int32_t *deg, *sdeg;    // [n]
    uint8_t *flag_v, *flag_f, *flag_f2;
};

struct SolveParams {
    int T;
    float tol, t_max, pi;
    float *q, *fs;              // [E,3], [E,2]
    uint8_t *amask;             // [B]
    float *prev, *counters;     // decimator state
    int has_prev;
    int check_termination;
    int has_edge_mask;          // problem->has_edge_mask at entry
    uint32_t *spec_zero;        // [T] bits: site0 (survey max), site1 (diff max), site2 (coeff argmax): some instance had an exact 0
    uint32_t *spec_used;        // [T] bits: some instance evaluated the site
    // HBM-mode scratch
    float *ws_e[4]; float *ws_f; float *ws_v[7]; int32_t *ws_vi[3]; uint8_t *ws_fu[2];
};

__device__ __forceinline__ size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

template <class T>
__device__ __forceinline__ T *carve(unsigned char *&p, size_t count)
{
    T *r = reinterpret_cast<T *>(p);
    p += align16(count * sizeof(T));
    return r;
}

// LDS bytes needed for an instance of (n, m, e): must mirror the carve sequence in the kernel
static size_t lds_bytes_for(int n, int m, int e)
{
    auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t s = 0;
    s += 4 * a16((size_t)e * 2);          // e_var, e_fn, v_edges, f_edges (u16)
    s += a16((size_t)(n + 1) * 2) + a16((size_t)(m + 1) * 2);
    s += a16((size_t)e);                  // sgn
    s += 8 * a16((size_t)e * 4);          // emask, qu, eta, force, s0..s3
    s += a16((size_t)m * 4) * 2;          // af, S
    s += a16((size_t)n * 4) * 9;          // av, sol, P, N, xv1, xv2, score, coeff, assign
    s += a16((size_t)n * 4) * 2;          // deg, sdeg
    s += a16((size_t)n) + 2 * a16((size_t)m);
    return s;
}

template <class IT, bool LDS>
__global__ void __launch_bounds__(LDS ? 512 : 256) k_sp_solve(PView pv, SolveParams sp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float redf[PDP_RED_SCRATCH];
    __shared__ int redi[PDP_RED_SCRATCH];
    __shared__ int sh_flag[4];

    const int tid = threadIdx.x, nt = blockDim.x;
    const Inst G = load_inst(pv, blockIdx.x);
    const int n = G.n, m = G.m, ne = G.e;
    float *gq = sp.q + 3 * (size_t)G.e0;
    float *gfs = sp.fs + 2 * (size_t)G.e0;

    SView<IT> I;
    I.b = G.b; I.n = n; I.m = m; I.e = ne;
    if constexpr (LDS) {
        unsigned char *p = smem;
        IT *e_var = carve<IT>(p, ne), *e_fn = carve<IT>(p, ne), *v_edges = carve<IT>(p, ne), *f_edges = carve<IT>(p, ne);
        IT *v_ptr = carve<IT>(p, n + 1), *f_ptr = carve<IT>(p, m + 1);
        int8_t *sgn = carve<int8_t>(p, ne);
        I.emask = carve<float>(p, ne); I.qu = carve<float>(p, ne); I.eta = carve<float>(p, ne);
        float *force = carve<float>(p, ne);
        I.s0 = carve<float>(p, ne); I.s1 = carve<float>(p, ne); I.s2 = carve<float>(p, ne); I.s3 = carve<float>(p, ne);
        I.af = carve<float>(p, m); I.S = carve<float>(p, m);
        I.av = carve<float>(p, n); I.sol = carve<float>(p, n); I.P = carve<float>(p, n); I.N = carve<float>(p, n);
        I.xv1 = carve<float>(p, n); I.xv2 = carve<float>(p, n); I.score = carve<float>(p, n); I.coeff = carve<float>(p, n);
        I.assign = carve<float>(p, n);
        I.deg = carve<int32_t>(p, n); I.sdeg = carve<int32_t>(p, n);
        I.flag_v = carve<uint8_t>(p, n); I.flag_f = carve<uint8_t>(p, m); I.flag_f2 = carve<uint8_t>(p, m);
        for (int e = tid; e < ne; e += nt) {
            e_var[e] = (IT)G.e_var[e]; e_fn[e] = (IT)G.e_fn[e]; v_edges[e] = (IT)G.v_edges[e]; f_edges[e] = (IT)G.f_edges[e];
            sgn[e] = G.sgn[e];
            I.emask[e] = G.emask[e];
            I.qu[e] = gq[3 * e]; I.eta[e] = gfs[2 * e]; force[e] = gfs[2 * e + 1];
        }
        for (int v = tid; v <= n; v += nt) v_ptr[v] = (IT)G.v_ptr[v];
        for (int c = tid; c <= m; c += nt) f_ptr[c] = (IT)G.f_ptr[c];
        for (int v = tid; v < n; v += nt) { I.av[v] = G.av[v]; I.sol[v] = G.sol[v]; }
        for (int c = tid; c < m; c += nt) I.af[c] = G.af[c];
        I.e_var = e_var; I.e_fn = e_fn; I.v_edges = v_edges; I.f_edges = f_edges; I.v_ptr = v_ptr; I.f_ptr = f_ptr; I.sgn = sgn;
        I.force = force; I.qstride = 1; I.estride = 1; I.fstride = 1;
    } else {
        I.e_var = G.e_var; I.e_fn = G.e_fn; I.v_edges = G.v_edges; I.f_edges = G.f_edges; I.v_ptr = G.v_ptr; I.f_ptr = G.f_ptr;
        I.sgn = G.sgn; I.av = G.av; I.af = G.af; I.sol = G.sol; I.emask = G.emask;
        I.qu = gq; I.qstride = 3; I.eta = gfs; I.estride = 2; I.force = gfs + 1; I.fstride = 2;
        I.s0 = sp.ws_e[0] + G.e0; I.s1 = sp.ws_e[1] + G.e0; I.s2 = sp.ws_e[2] + G.e0; I.s3 = sp.ws_e[3] + G.e0;
        I.S = sp.ws_f + G.f0;
        I.P = sp.ws_v[0] + G.v0; I.N = sp.ws_v[1] + G.v0; I.xv1 = sp.ws_v[2] + G.v0; I.xv2 = sp.ws_v[3] + G.v0;
        I.score = sp.ws_v[4] + G.v0; I.coeff = sp.ws_v[5] + G.v0; I.assign = sp.ws_v[6] + G.v0;
        I.deg = sp.ws_vi[0] + G.v0; I.sdeg = sp.ws_vi[1] + G.v0;
        I.flag_v = reinterpret_cast<uint8_t *>(sp.ws_vi[2]) + G.v0;
        I.flag_f = sp.ws_fu[0] + G.f0; I.flag_f2 = sp.ws_fu[1] + G.f0;
    }
    __syncthreads();

    SimplifyScratch ss;
    ss.assign = I.assign; ss.deg = I.deg; ss.sdeg = I.sdeg; ss.flag_v = I.flag_v; ss.flag_f = I.flag_f; ss.flag_f2 = I.flag_f2; ss.red = redi;

    int active = sp.amask[G.b] ? 1 : 0;
    int has_prev = sp.has_prev;
    int prev_from_global = sp.has_prev;       // first iteration compares with the decimator's stored survey
    int use_em = sp.has_edge_mask;            // sat_problem._edge_mask is not None
    float cnt = sp.counters[G.b];
    int iters = 0, did_prop = 0;
    int nsat = -1;                            // cached CNF result (solution only changes on decimation)
    int violation = 0;
    const bool other_rows = n < pv.V;

    for (int t = 0; t < sp.T; ++t) {
        if (!active) break;
        iters = t + 1;
        // ---- P1: per-edge logs (pdp_propagate.py:166-169,185-188)
        for (int e = tid; e < ne; e += nt) {
            float x = pdp_safe_log(I.qu[e * I.qstride], PDP_SP_EPS);
            float y = pdp_safe_log(1.0f - I.eta[e * I.estride], PDP_SP_EPS);
            if (use_em) { const float em = I.emask[e]; x = x * em; y = y * em; }
            I.s0[e] = x; I.s1[e] = y;
        }
        __syncthreads();
        // ---- P2: per-clause and per-variable sums (ascending edge id)
        for (int c = tid; c < m; c += nt) {
            float acc = 0.0f;
            for (int k = I.f_ptr[c]; k < I.f_ptr[c + 1]; ++k) acc = acc + I.s0[I.f_edges[k]];
            I.S[c] = acc;
        }
        for (int v = tid; v < n; v += nt) {
            float P = 0.0f, N = 0.0f;
            for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
                const int e = I.v_edges[k];
                const float y = I.s1[e];
                const int sg = I.sgn[e];
                P = P + ((sg == 1) ? 1.0f : 0.0f) * y;
                N = N + ((sg == -1) ? 1.0f : 0.0f) * y;
            }
            I.P[v] = P; I.N[v] = N;
        }
        __syncthreads();
        // ---- P3: new surveys + new q_u, and the decimator's per-edge terms (pdp_decimate.py:128-141)
        int nan_seen = 0;
        for (int e = tid; e < ne; e += nt) {
            const int v = I.e_var[e], c = I.e_fn[e];
            const float s = (float)I.sgn[e];
            const float eta_old = I.eta[e * I.estride];
            const float agg = (0.0f + I.S[c]) - I.s0[e];
            const float eta_new = 1.0f * pdp_safe_exp(agg) + (1.0f - 1.0f) * eta_old;
            const float force = I.force[e * I.fstride];
            const SpOut o = d_sp_edge(s, I.P[v], I.N[v], I.s1[e], force, sp.pi);
            const float qu_old = I.qu[e * I.qstride];
            const float qu_new = 1.0f * o.qu + (1.0f - 1.0f) * qu_old;
            if (qu_new != qu_new || eta_new != eta_new || o.qs != o.qs || o.dc != o.dc) nan_seen = 1;
            float d = 0.0f;
            if (has_prev) {
                const float pe = prev_from_global ? sp.prev[G.e0 + e] : eta_old;
                d = pdp_abs(pe - eta_new);
                if (use_em) d = d * I.emask[e];
            }
            I.qu[e * I.qstride] = qu_new;
            I.eta[e * I.estride] = eta_new;
            I.s0[e] = pdp_safe_exp(30.0f * eta_new);      // smooth-max weight of the survey
            I.s2[e] = d;
            I.s3[e] = pdp_safe_exp(30.0f * d);            // smooth-max weight of the difference
        }
        did_prop = 1;
        __syncthreads();
        // ---- P4: per-variable smooth maxima (util.py:282-286) times the active flag
        int z1 = 0, z2 = 0;
        for (int v = tid; v < n; v += nt) {
            float num1 = 0.0f, den1 = 0.0f, num2 = 0.0f, den2 = 0.0f;
            for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
                const int e = I.v_edges[k];
                const float c1 = I.s0[e], et = I.eta[e * I.estride];
                num1 = num1 + et * c1; den1 = den1 + c1;
                if (has_prev) { const float c2 = I.s3[e], d = I.s2[e]; num2 = num2 + d * c2; den2 = den2 + c2; }
            }
            const float a = I.av[v];
            const float r1 = (num1 / pdp_max(den1, 1.0f)) * a;
            I.xv1[v] = r1;
            if (r1 == 0.0f) z1 = 1;
            if (r1 != r1) nan_seen = 1;
            if (has_prev) {
                const float r2 = (num2 / pdp_max(den2, 1.0f)) * a;
                I.xv2[v] = r2;
                if (r2 == 0.0f) z2 = 1;
                if (r2 != r2) nan_seen = 1;
            }
        }
        __syncthreads();
        // ---- P5: per-instance maxima with the reference's (x - min + 1) rounding, speculating min == 0
        const float g = d_instance_max(I, I.xv1, 0.0f, other_rows, redf);
        float dmax = 0.0f;
        if (has_prev) dmax = d_instance_max(I, I.xv2, 0.0f, other_rows, redf);
        z1 = __syncthreads_or(z1); z2 = __syncthreads_or(z2); nan_seen = __syncthreads_or(nan_seen);
        if (nan_seen) violation = 1;
        int conv = 0;
        if (g <= 1e-10f) active = 0;                          // trivial surveys: leave the instance to Walk-SAT
        if (has_prev) {
            if (dmax < sp.tol) cnt = 0.0f;
            conv = (dmax < sp.tol) ? 1 : 0;
            if (cnt >= sp.t_max) { conv = 1; cnt = 0.0f; }
        }
        uint32_t used = 1u | (has_prev ? 2u : 0u);
        uint32_t zero = (z1 ? 1u : 0u) | ((has_prev && z2) ? 2u : 0u);
        // coeff = |score| * active * converged: every variable of a non-converged instance is an exact 0 of site 2
        if (has_prev && !conv && n > 0) zero |= 4u;
        // ---- P6: decimation (pdp_decimate.py:152-171)
        int decimated = 0;
        if (has_prev && conv) {
            // scorer (pdp_predict.py:155-192)
            for (int e = tid; e < ne; e += nt)
                I.s3[e] = pdp_safe_log(1.0f - I.eta[e * I.estride], PDP_SCORER_EPS) * (0.0f + I.af[I.e_fn[e]]);
            __syncthreads();
            int z3 = 0, anynz = 0, cn = 0;
            for (int v = tid; v < n; v += nt) {
                float ext = 0.0f, pos = 0.0f, neg = 0.0f, all = 0.0f;
                for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
                    const int e = I.v_edges[k];
                    const int sg = I.sgn[e];
                    const float f = I.s3[e];
                    ext = ext + I.force[e * I.fstride];
                    pos = pos + ((sg == 1) ? 1.0f : 0.0f) * f;
                    neg = neg + ((sg == -1) ? 1.0f : 0.0f) * f;
                    all = all + f;
                }
                const float sc = d_score_from_sums(pos, neg, all, ext, sp.pi);
                const float co = (pdp_abs(sc) * I.av[v]) * 1.0f;
                I.score[v] = sc; I.coeff[v] = co;
                if (co == 0.0f) z3 = 1;
                if (co != 0.0f) anynz = 1;
                if (co != co) cn = 1;
            }
            __syncthreads();
            z3 = __syncthreads_or(z3); anynz = __syncthreads_or(anynz); cn = __syncthreads_or(cn);
            if (cn) violation = 1;
            used |= 4u; if (z3) zero |= 4u;
            const int li = d_instance_argmax(I, I.coeff, 0.0f, redf, redi);
            if (active && anynz && !cn && li >= 0) {
                for (int v = tid; v < n; v += nt) I.assign[v] = 0.0f;
                __syncthreads();
                if (tid == 0) I.assign[li] = pdp_sign(I.score[li]);
                __syncthreads();
                d_set_variable_core(I, ss);
                d_simplify(I, ss, pv.is_sat + G.b);
                decimated = 1;
            }
        }
        if (has_prev) cnt = cnt + 1.0f;
        if (tid == 0) { atomicOr(&sp.spec_used[t], used); if (zero) atomicOr(&sp.spec_zero[t], zero); }
        // ---- P7: edge mask refresh (solver.py:370-371); values only change after a decimation
        if (decimated || !use_em) {
            for (int e = tid; e < ne; e += nt) {
                const float a = 0.0f + I.av[I.e_var[e]];
                const float b = 0.0f + I.af[I.e_fn[e]];
                I.emask[e] = a * b;
            }
            use_em = 1;
            __syncthreads();
        }
        // ---- P8: prediction = solution; termination check (trainer.py:150-162, util.py:226-236)
        if (sp.check_termination) {
            if (decimated || nsat < 0) nsat = d_cnf_sat_count(I, I.sol, redi);
            if (active && nsat == m) active = 0;
        }
        has_prev = 1; prev_from_global = 0;
    }

    // ---- write back -------------------------------------------------------------------------------------------
    if (did_prop) {
        // q_s / q_dc of the last sweep are recomputed from the per-variable sums that are still resident
        for (int e = tid; e < ne; e += nt) {
            const SpOut o = d_sp_edge((float)I.sgn[e], I.P[I.e_var[e]], I.N[I.e_var[e]], I.s1[e], I.force[e * I.fstride], sp.pi);
            const float qs_old = gq[3 * e + 1], qd_old = gq[3 * e + 2];
            gq[3 * e + 1] = 1.0f * o.qs + (1.0f - 1.0f) * qs_old;
            gq[3 * e + 2] = 1.0f * o.dc + (1.0f - 1.0f) * qd_old;
            if constexpr (LDS) { gq[3 * e] = I.qu[e]; gfs[2 * e] = I.eta[e]; }
            sp.prev[G.e0 + e] = I.eta[e * I.estride];
        }
    }
    if constexpr (LDS) {
        for (int e = tid; e < ne; e += nt) G.emask[e] = I.emask[e];
        for (int v = tid; v < n; v += nt) { G.av[v] = I.av[v]; G.sol[v] = I.sol[v]; }
        for (int c = tid; c < m; c += nt) G.af[c] = I.af[c];
    }
    // an instance with a de-activated variable contributes an exact 0 to every batch-global min from now on
    int any_inactive = 0;
    for (int v = tid; v < n; v += nt) any_inactive |= (I.av[v] == 0.0f) ? 1 : 0;
    any_inactive = __syncthreads_or(any_inactive);
    if (tid == 0) {
        if (any_inactive) atomicMin(&pv.flags[FL_PERM_ZERO], (uint32_t)iters);
        sp.amask[G.b] = (uint8_t)active;
        sp.counters[G.b] = cnt;
        atomicMax(&pv.flags[FL_ITERS_RUN], (uint32_t)iters);
        if (violation) atomicOr(&pv.flags[FL_SPEC_VIOLATION], 1u);
    }
    (void)sh_flag;
}

extern "C" int pdp_sp_solve(pdp_problem *p, pdp_solve_args *a, void *stream)
{
    PDP_REQUIRE(p && a && p->av, "NULL argument / state not bound");
    PDP_REQUIRE(a->model == PDP_MODEL_SP, "persistent solve: only the SP triple is implemented (use the step-wise path)");
    PDP_REQUIRE(a->q && a->fs && a->active_mask && a->decimator, "NULL state array");
    PDP_REQUIRE(p->R == 1, "persistent solve needs replication == 1 (replicas couple through the termination check)");
    hipStream_t st = ST(stream);
    const int T = a->iterations;
    a->iterations_run_host = 0; a->used_lds_host = 0;
    if (T <= 0) return PDP_OK;

    uint32_t *spec = nullptr;
    PDP_HIP_CHECK(hipMalloc((void **)&spec, sizeof(uint32_t) * 2 * (size_t)T));
    PDP_HIP_CHECK(hipMemsetAsync(spec, 0, sizeof(uint32_t) * 2 * (size_t)T, st));
    PDP_HIP_CHECK(hipMemsetAsync(p->flags + FL_SPEC_VIOLATION, 0, sizeof(uint32_t) * 2, st));   // violation + iters_run
    PDP_HIP_CHECK(hipMemsetAsync(p->flags + FL_PERM_ZERO, 0xff, sizeof(uint32_t), st));

    SolveParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.T = T; sp.tol = a->tolerance; sp.t_max = a->t_max; sp.pi = a->pi;
    sp.q = a->q; sp.fs = a->fs; sp.amask = a->active_mask;
    sp.prev = a->decimator->prev; sp.counters = a->decimator->counters; sp.has_prev = a->decimator->has_prev;
    sp.check_termination = a->check_termination; sp.has_edge_mask = p->has_edge_mask;
    sp.spec_used = spec; sp.spec_zero = spec + T;

    const size_t lds = lds_bytes_for(p->max_n, p->max_m, p->max_e);
    const bool fits = lds <= 160 * 1024 - 2048 && p->max_e < 65535 && p->max_n < 65535 && p->max_m < 65535;
    float *extra_v = nullptr;
    if (fits) {
        PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_sp_solve<uint16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_sp_solve<uint16_t, true>), dim3(p->B), dim3(512), lds, st, make_view(p), sp);
        a->used_lds_host = 1;
    } else {
        for (int i = 0; i < 4; ++i) sp.ws_e[i] = p->ws_e[i];
        sp.ws_f = p->ws_f[0];
        for (int i = 0; i < 6; ++i) sp.ws_v[i] = p->ws_v[i];
        PDP_HIP_CHECK(hipMalloc((void **)&extra_v, sizeof(float) * (size_t)p->V));
        sp.ws_v[6] = extra_v;
        for (int i = 0; i < 3; ++i) sp.ws_vi[i] = p->ws_vi[i];
        sp.ws_fu[0] = p->ws_fu[0]; sp.ws_fu[1] = p->ws_fu[1];
        hipLaunchKernelGGL((k_sp_solve<int32_t, false>), dim3(p->B), dim3(256), 0, st, make_view(p), sp);
    }
    PDP_LAUNCH_CHECK();
    // verify the speculation record
    uint32_t *host = (uint32_t *)malloc(sizeof(uint32_t) * 2 * (size_t)T);
    PDP_HIP_CHECK(hipMemcpyAsync(host, spec, sizeof(uint32_t) * 2 * (size_t)T, hipMemcpyDeviceToHost, st));
    PDP_HIP_CHECK(hipMemcpyAsync(p->flags_host, p->flags, FL_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PDP_HIP_CHECK(hipStreamSynchronize(st));
    bool ok = p->flags_host[FL_SPEC_VIOLATION] == 0u;
    const uint32_t perm_from = p->flags_host[FL_PERM_ZERO];
    for (int t = 0; t < T && ok; ++t) if ((uint32_t)t < perm_from && (host[t] & ~host[T + t]) != 0u) ok = false;
    if (getenv("PDP_DEBUG")) {
        fprintf(stderr, "[pdp_sp_solve] violation=%u iters=%u perm_from=%u lds=%zu\n", p->flags_host[FL_SPEC_VIOLATION],
                p->flags_host[FL_ITERS_RUN], perm_from, lds);
        for (int t = 0; t < T; ++t) if (t < 3 || (host[t] & ~host[T + t])) fprintf(stderr, "  t=%d used=%u zero=%u\n", t, host[t], host[T + t]);
    }
    free(host);
    (void)hipFree(spec);
    if (extra_v) (void)hipFree(extra_v);
    a->iterations_run_host = (int32_t)p->flags_host[FL_ITERS_RUN];
    a->decimator->has_prev = 1;
    p->has_edge_mask = 1;
    if (!ok) {
        pdp_set_error("persistent solve: a cross-instance coupling of the reference became active (batch-global min != 0 or NaN); "
                      "state is not reference-exact, rerun the batch step-wise");
        return PDP_ERR_SPECULATION;
    }
    return PDP_OK;
}
