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
    int poison_from;            // first iteration at which the batch is NaN-poisoned (INT_MAX: never); pass 2 only
    uint32_t *nan_iter;         // device word: min over instances of the first iteration whose surveys contain a NaN
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
        // pass 1 only: once some instance is known to poison the batch before t, this pass will be replayed anyway
        if (sp.poison_from == 0x7fffffff) {
            if (tid == 0) sh_flag[0] = (__hip_atomic_load(sp.nan_iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)t) ? 1 : 0;
            __syncthreads();
            const int abort_now = sh_flag[0];
            __syncthreads();
            if (abort_now) break;
        }
        const bool poisoned = t >= sp.poison_from;
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
            // only a NaN SURVEY poisons the batch-global reductions of this iteration; a NaN in q (0/0) reaches the
            // surveys one iteration later (x = log(max(NaN, eps)) = NaN)
            if (eta_new != eta_new) nan_seen = 1;
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
        // A NaN survey (0/0 in pdp_propagate.py:215-216) makes every batch-global min/max of the reference NaN from
        // this iteration on.  Pass 1 records the first such iteration, pass 2 replays with it.
        if (nan_seen && !poisoned) {
            if (tid == 0) atomicMin(sp.nan_iter, (uint32_t)t);
            if (sp.poison_from != 0x7fffffff) violation = 1;      // pass 2 must not find an earlier NaN
        }
        int conv = 0;
        if (!poisoned) {
            if (g <= 1e-10f) active = 0;                      // trivial surveys: leave the instance to Walk-SAT
            if (has_prev) {
                if (dmax < sp.tol) cnt = 0.0f;
                conv = (dmax < sp.tol) ? 1 : 0;
                if (cnt >= sp.t_max) { conv = 1; cnt = 0.0f; }
            }
        } else if (has_prev) {
            // all maxima are NaN: comparisons are False, only the counter overflow still "converges" an instance
            if (cnt >= sp.t_max) { conv = 1; cnt = 0.0f; }
        }
        uint32_t used = 1u | (has_prev ? 2u : 0u);
        uint32_t zero = (z1 ? 1u : 0u) | ((has_prev && z2) ? 2u : 0u);
        // coeff = |score| * active * converged: every variable of a non-converged instance is an exact 0 of site 2
        if (has_prev && !conv && n > 0) zero |= 4u;
        // ---- P6: decimation (pdp_decimate.py:152-171)
        int decimated = 0;
        if (has_prev && conv && !poisoned && !nan_seen) {
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
            if (cn) violation = 1;                                // cannot happen without a NaN survey
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
        if (tid == 0 && !poisoned) { atomicOr(&sp.spec_used[t], used); if (zero) atomicOr(&sp.spec_zero[t], zero); }
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

// ---- host side ---------------------------------------------------------------------------------------------------
struct SolveSnapshot {
    float *q, *fs, *av, *af, *sol, *sat, *emask, *prev, *cnt; uint8_t *amask;
};

static int snapshot_copy(pdp_problem *p, pdp_solve_args *a, SolveSnapshot &s, bool save, hipStream_t st)
{
    const size_t E = p->E, V = p->V, F = p->F, B = p->B;
    struct { void *live; void *snap; size_t bytes; } items[] = {
        {a->q, s.q, 3 * E * 4}, {a->fs, s.fs, 2 * E * 4}, {p->av, s.av, V * 4}, {p->af, s.af, F * 4}, {p->sol, s.sol, V * 4},
        {p->is_sat, s.sat, B * 4}, {p->emask, s.emask, E * 4}, {a->decimator->prev, s.prev, E * 4},
        {a->decimator->counters, s.cnt, B * 4}, {a->active_mask, s.amask, B}};
    for (auto &it : items) {
        if (save) PDP_HIP_CHECK(hipMemcpyAsync(it.snap, it.live, it.bytes, hipMemcpyDeviceToDevice, st));
        else PDP_HIP_CHECK(hipMemcpyAsync(it.live, it.snap, it.bytes, hipMemcpyDeviceToDevice, st));
    }
    return PDP_OK;
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
    const size_t E = p->E, V = p->V, F = p->F, B = p->B;

    // one allocation: speculation record [2T] + nan word + snapshot of everything the loop mutates
    const size_t words = 2 * (size_t)T + 4;
    const size_t snap_floats = 3 * E + 2 * E + V + F + V + B + E + E + B;
    char *blob = nullptr;
    PDP_HIP_CHECK(hipMalloc((void **)&blob, words * 4 + snap_floats * 4 + B + 64));
    uint32_t *spec = (uint32_t *)blob;
    uint32_t *nan_iter = spec + 2 * (size_t)T;
    float *f = (float *)(blob + words * 4);
    SolveSnapshot snap;
    snap.q = f; f += 3 * E; snap.fs = f; f += 2 * E; snap.av = f; f += V; snap.af = f; f += F; snap.sol = f; f += V;
    snap.sat = f; f += B; snap.emask = f; f += E; snap.prev = f; f += E; snap.cnt = f; f += B; snap.amask = (uint8_t *)f;
    const int had_prev = a->decimator->has_prev, had_emask = p->has_edge_mask;
    int status = snapshot_copy(p, a, snap, true, st);
    if (status != PDP_OK) { (void)hipFree(blob); return status; }

    SolveParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.T = T; sp.tol = a->tolerance; sp.t_max = a->t_max; sp.pi = a->pi;
    sp.q = a->q; sp.fs = a->fs; sp.amask = a->active_mask;
    sp.prev = a->decimator->prev; sp.counters = a->decimator->counters; sp.has_prev = had_prev;
    sp.check_termination = a->check_termination; sp.has_edge_mask = had_emask;
    sp.spec_used = spec; sp.spec_zero = spec + T; sp.nan_iter = nan_iter;

    const size_t lds = lds_bytes_for(p->max_n, p->max_m, p->max_e);
    const bool fits = lds <= 160 * 1024 - 2048 && p->max_e < 65535 && p->max_n < 65535 && p->max_m < 65535;
    float *extra_v = nullptr;
    if (fits) {
        PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_sp_solve<uint16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        a->used_lds_host = 1;
    } else {
        for (int i = 0; i < 4; ++i) sp.ws_e[i] = p->ws_e[i];
        sp.ws_f = p->ws_f[0];
        for (int i = 0; i < 6; ++i) sp.ws_v[i] = p->ws_v[i];
        PDP_HIP_CHECK(hipMalloc((void **)&extra_v, sizeof(float) * V));
        sp.ws_v[6] = extra_v;
        for (int i = 0; i < 3; ++i) sp.ws_vi[i] = p->ws_vi[i];
        sp.ws_fu[0] = p->ws_fu[0]; sp.ws_fu[1] = p->ws_fu[1];
    }
    uint32_t *host = (uint32_t *)malloc(words * 4);
    bool ok = true;
    int poison_from = 0x7fffffff;
    int passes = 0;
    for (int pass = 0; pass < 2; ++pass) {
        passes = pass + 1;
        sp.poison_from = poison_from;
        PDP_HIP_CHECK(hipMemsetAsync(spec, 0, sizeof(uint32_t) * 2 * (size_t)T, st));
        PDP_HIP_CHECK(hipMemsetAsync(nan_iter, 0xff, sizeof(uint32_t), st));
        PDP_HIP_CHECK(hipMemsetAsync(p->flags + FL_SPEC_VIOLATION, 0, sizeof(uint32_t) * 2, st));   // violation + iters_run
        PDP_HIP_CHECK(hipMemsetAsync(p->flags + FL_PERM_ZERO, 0xff, sizeof(uint32_t), st));
        if (fits) hipLaunchKernelGGL((k_sp_solve<uint16_t, true>), dim3(p->B), dim3(512), lds, st, make_view(p), sp);
        else hipLaunchKernelGGL((k_sp_solve<int32_t, false>), dim3(p->B), dim3(256), 0, st, make_view(p), sp);
        PDP_LAUNCH_CHECK();
        PDP_HIP_CHECK(hipMemcpyAsync(host, spec, words * 4, hipMemcpyDeviceToHost, st));
        PDP_HIP_CHECK(hipMemcpyAsync(p->flags_host, p->flags, FL_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PDP_HIP_CHECK(hipStreamSynchronize(st));
        const uint32_t t_nan = host[2 * (size_t)T];
        if (pass == 0 && t_nan < (uint32_t)T) {
            // some instance poisons the batch from iteration t_nan on: restore and replay with the poison applied
            poison_from = (int)t_nan;
            status = snapshot_copy(p, a, snap, false, st);
            if (status != PDP_OK) break;
            continue;
        }
        ok = p->flags_host[FL_SPEC_VIOLATION] == 0u;
        const uint32_t perm_from = p->flags_host[FL_PERM_ZERO];
        for (int t = 0; t < T && t < poison_from && ok; ++t)
            if ((uint32_t)t < perm_from && (host[t] & ~host[T + t]) != 0u) ok = false;
        if (getenv("PDP_DEBUG")) {
            fprintf(stderr, "[pdp_sp_solve] pass=%d violation=%u iters=%u perm_from=%u poison_from=%d lds=%zu ok=%d\n", pass,
                    p->flags_host[FL_SPEC_VIOLATION], p->flags_host[FL_ITERS_RUN], perm_from, poison_from, lds, (int)ok);
            for (int t = 0; t < T; ++t) if (t < 2 || (host[t] & ~host[T + t])) fprintf(stderr, "  t=%d used=%u zero=%u\n", t, host[t], host[T + t]);
        }
        break;
    }
    a->iterations_run_host = (int32_t)p->flags_host[FL_ITERS_RUN];
    if (status == PDP_OK && !ok) {
        // leave the caller's state exactly as it was so that it can rerun the batch step-wise
        status = snapshot_copy(p, a, snap, false, st);
        if (status == PDP_OK) status = hipStreamSynchronize(st) == hipSuccess ? PDP_OK : PDP_ERR_HIP;
    }
    free(host);
    (void)hipFree(blob);
    if (extra_v) (void)hipFree(extra_v);
    if (status != PDP_OK) return status;
    (void)passes;
    if (!ok) {
        pdp_set_error("persistent solve: a cross-instance coupling of the reference became active (batch-global min != 0); "
                      "state restored, rerun the batch step-wise");
        return PDP_ERR_SPECULATION;
    }
    a->decimator->has_prev = 1;
    p->has_edge_mask = 1;
    return PDP_OK;
}
