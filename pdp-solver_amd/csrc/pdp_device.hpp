// pdp_device.hpp -- per-instance device routines, executed cooperatively by one workgroup.
// They are templated on the instance view `I` (field names of struct Inst) so the same code runs on an
// HBM-resident view (int32 ids, step-wise kernels), on an LDS-resident view (u16 ids, persistent solver) and on a
// TeamView (one big instance spread over several workgroups).  Thread ids, barriers and reductions go through the
// team_* calls of pdp_common.hpp; every routine must be called by all threads of the team and leaves it synchronised.
//
// Arithmetic mirrors the reference statement by statement; all quantities here are small integers
// held in fp32 by the reference, so int32 arithmetic is exact and order-independent.
#pragma once

#include "pdp_common.hpp"

// ---- batch-global min plumbing (reference couplings) -------------------------------
__device__ __forceinline__ float read_gmin(const uint32_t *flags, int slot_min, int slot_nan)
{
    if (flags[slot_nan]) return PDP_NAN;
    return pdp_dec_ordered(flags[slot_min]);
}

// block-level contribution to a batch-global min (NaN tracked separately)
__device__ __forceinline__ void publish_min(float local_min, bool local_nan, uint32_t *flags, int slot_min, int slot_nan,
                                            float *redf, int *redi)
{
    struct OpMinPlain { __device__ float operator()(float a, float b) const { return a < b ? a : b; } };
    const float m = block_reduce(local_min, OpMinPlain(), PDP_INF, redf);
    const int nn = block_reduce(local_nan ? 1 : 0, OpOrI(), 0, redi);
    if (threadIdx.x == 0) {
        if (nn) atomicOr(&flags[slot_nan], 1u);
        atomicMin(&flags[slot_min], pdp_enc_ordered(m));
    }
}


// scratch the simplification routines need (instance-sized, any address space)
struct SimplifyScratch {
    float *assign;       // [n]
    int32_t *deg;        // [n]
    int32_t *sdeg;       // [n]
    uint8_t *flag_v;     // [n]
    uint8_t *flag_f;     // [m]
    uint8_t *flag_f2;    // [m]
    int *red;            // [PDP_RED_SCRATCH] LDS
};

// reference: SATProblem._set_variable_core (solver.py:205-226)
template <class I>
__device__ void d_set_variable_core(const I &in, const SimplifyScratch &s)
{
    const int tid = team_tid(in), nt = team_nt(in);
    for (int v = tid; v < in.n; v += nt) s.assign[v] = s.assign[v] * in.av[v];
    team_sync(in);
    for (int c = tid; c < in.m; c += nt) {
        float input_num = 0.0f, function_eval = 0.0f;
        for (int k = in.f_ptr[c]; k < in.f_ptr[c + 1]; ++k) {
            const int e = in.f_edges[k];
            const float a = s.assign[in.e_var[e]];
            input_num = input_num + pdp_abs(a);
            function_eval = function_eval + (float)in.sgn[e] * a;
        }
        const float deact = ((function_eval > -input_num) ? 1.0f : 0.0f) * in.af[c];
        s.flag_f2[c] = (deact == 1.0f) ? 1 : 0;
    }
    team_sync(in);
    for (int v = tid; v < in.n; v += nt) {
        const float a = s.assign[v];
        if (pdp_abs(a) == 1.0f) { in.av[v] = 0.0f; in.sol[v] = (a + 1.0f) / 2.0f; }
    }
    for (int c = tid; c < in.m; c += nt) if (s.flag_f2[c]) in.af[c] = 0.0f;
    team_sync(in);
}

// reference: SATProblem._propagate_single_clauses (solver.py:228-273)
template <class I>
__device__ void d_unit_propagate(const I &in, const SimplifyScratch &s, float *is_sat_b)
{
    const int tid = team_tid(in), nt = team_nt(in);
    for (;;) {
        int any = 0;
        for (int c = tid; c < in.m; c += nt) {
            float deg = 0.0f;
            for (int k = in.f_ptr[c]; k < in.f_ptr[c + 1]; ++k) deg = deg + in.av[in.e_var[in.f_edges[k]]];
            const float single = ((deg == 1.0f) ? 1.0f : 0.0f) * in.af[c];
            s.flag_f[c] = (single == 1.0f) ? 1 : 0;
            any |= s.flag_f[c];
        }
        any = team_any(in, any);
        if (!any) break;
        int nconf = 0;
        for (int v = tid; v < in.n; v += nt) {
            int inum = 0, ev = 0;
            for (int k = in.v_ptr[v]; k < in.v_ptr[v + 1]; ++k) {
                const int e = in.v_edges[k];
                const int sg = s.flag_f[in.e_fn[e]];
                inum += sg;
                ev += (int)in.sgn[e] * sg;
            }
            s.deg[v] = inum; s.sdeg[v] = ev;
            const int aev = ev < 0 ? -ev : ev;
            if (aev != inum && in.av[v] == 1.0f) nconf++;
        }
        nconf = team_reduce(in, nconf, OpAddI(), 0, s.red);
        if (nconf >= 1) {
            if (tid == 0) *is_sat_b = 0.0f;
            // the reference compares (count * active) with == 1 (solver.py:257,261)
            if (nconf == 1) {
                for (int c = tid; c < in.m; c += nt) if (in.af[c] == 1.0f) in.af[c] = 0.0f;
                for (int v = tid; v < in.n; v += nt) if (in.av[v] == 1.0f) in.av[v] = 0.0f;
            }
            team_sync(in);
        }
        for (int v = tid; v < in.n; v += nt) {
            const int inum = s.deg[v], ev = s.sdeg[v];
            const int aev = ev < 0 ? -ev : ev;
            const float assigned = ((aev == inum) ? 1.0f : 0.0f) * in.av[v];
            s.assign[v] = (float)((ev > 0) - (ev < 0)) * assigned;
        }
        for (int c = tid; c < in.m; c += nt) if (s.flag_f[c]) in.af[c] = 0.0f;
        team_sync(in);
        d_set_variable_core(in, s);
    }
}

// reference: SATProblem._peel (solver.py:180-203)
template <class I>
__device__ void d_peel(const I &in, const SimplifyScratch &s)
{
    const int tid = team_tid(in), nt = team_nt(in);
    for (int v = tid; v < in.n; v += nt) {
        int d = 0, sd = 0;
        for (int k = in.v_ptr[v]; k < in.v_ptr[v + 1]; ++k) {
            const int e = in.v_edges[k];
            const int a = (in.af[in.e_fn[e]] == 1.0f) ? 1 : 0;
            d += a; sd += (int)in.sgn[e] * a;
        }
        s.deg[v] = d; s.sdeg[v] = sd;
    }
    team_sync(in);
    for (;;) {
        int any = 0;
        for (int v = tid; v < in.n; v += nt) {
            const int sd = s.sdeg[v];
            const int single = (s.deg[v] == (sd < 0 ? -sd : sd)) && (in.av[v] == 1.0f);
            s.flag_v[v] = single ? 1 : 0;
            any |= single;
        }
        any = team_any(in, any);
        if (!any) break;
        for (int c = tid; c < in.m; c += nt) {
            int acc = 0;
            for (int k = in.f_ptr[c]; k < in.f_ptr[c + 1]; ++k) acc += s.flag_v[in.e_var[in.f_edges[k]]];
            s.flag_f[c] = (acc > 0 && in.af[c] == 1.0f) ? 1 : 0;
        }
        team_sync(in);
        for (int v = tid; v < in.n; v += nt) {
            int dd = 0, sd = 0;
            for (int k = in.v_ptr[v]; k < in.v_ptr[v + 1]; ++k) {
                const int e = in.v_edges[k];
                const int f = s.flag_f[in.e_fn[e]];
                dd += f; sd += (int)in.sgn[e] * f;
            }
            const int a = (in.av[v] == 1.0f) ? 1 : 0;     // degree_delta * active_variables
            dd *= a; sd *= a;
            if (s.flag_v[v]) {
                const int sg = s.sdeg[v];
                in.sol[v] = ((float)((sg > 0) - (sg < 0)) + 1.0f) / 2.0f;
            }
            s.deg[v] -= dd; s.sdeg[v] -= sd;
        }
        team_sync(in);
        for (int v = tid; v < in.n; v += nt) if (s.flag_v[v]) in.av[v] = 0.0f;
        for (int c = tid; c < in.m; c += nt) if (s.flag_f[c]) in.af[c] = 0.0f;
        team_sync(in);
    }
}

// reference: SATProblem.simplify (solver.py:281-285)
template <class I>
__device__ void d_simplify(const I &in, const SimplifyScratch &s, float *is_sat_b)
{
    d_unit_propagate(in, s, is_sat_b);
    d_peel(in, s);
}

// reference: SatCNFEvaluator.forward (util.py:226-236) restricted to one instance.
// Returns the number of satisfied clauses (identical on all threads).
template <class I>
__device__ int d_cnf_sat_count(const I &in, const float *pred /*[n]*/, int *red)
{
    int cnt = 0;
    for (int c = team_tid(in); c < in.m; c += team_nt(in)) {
        float clause = 0.0f;
        for (int k = in.f_ptr[c]; k < in.f_ptr[c + 1]; ++k) {
            const int e = in.f_edges[k];
            const float sg = (float)in.sgn[e];
            float ev = 0.0f + sg * pred[in.e_var[e]];
            ev = ev + (1.0f - sg) / 2.0f;
            clause = clause + ((ev > 0.5f) ? 1.0f : 0.0f);
        }
        cnt += (clause > 0.0f) ? 1 : 0;
    }
    return team_reduce(in, cnt, OpAddI(), 0, red);
}

// per-instance part of util.sparse_max (util.py:267-275): max over the dense column of
// (x - gmin) + 1 (plus the zero rows of the other instances), then + gmin - 1.
template <class I>
__device__ float d_instance_max(const I &in, const float *x /*[n]*/, float gmin, bool other_rows, float *red)
{
    float t = -PDP_INF;
    for (int v = team_tid(in); v < in.n; v += team_nt(in)) t = pdp_max(t, (x[v] - gmin) + 1.0f);
    t = team_reduce(in, t, OpMaxNan(), -PDP_INF, red);
    if (other_rows) t = pdp_max(t, 0.0f);
    return (t + gmin) - 1.0f;
}

// per-instance part of util.sparse_argmax (util.py:257-265); returns the LOCAL index (or -1 if n == 0)
template <class I>
__device__ int d_instance_argmax(const I &in, const float *x /*[n]*/, float gmin, float *redf, int *redi)
{
    float bv = 0.0f; int bi = -1;
    for (int v = team_tid(in); v < in.n; v += team_nt(in)) {
        const float t = (x[v] - gmin) + 1.0f;
        if (arg_better(t, v, bv, bi)) { bv = t; bi = v; }
    }
    return team_argmax(in, bv, bi, redf, redi).i;
}

// per-variable smooth max (util.sparse_smooth_max util.py:282-286) of an edge vector
template <class I>
__device__ __forceinline__ float d_smooth_max_var(const I &in, int v, const float *x /*[e]*/)
{
    float num = 0.0f, den = 0.0f;
    for (int k = in.v_ptr[v]; k < in.v_ptr[v + 1]; ++k) {
        const float xe = x[in.v_edges[k]];
        const float coeff = pdp_safe_exp(30.0f * xe);
        num = num + xe * coeff;
        den = den + coeff;
    }
    return num / pdp_max(den, 1.0f);
}

// SurveyScorer tail (pdp_predict.py:174-192) from the three per-variable sums
__device__ __forceinline__ float d_score_from_sums(float pos, float neg, float all, float ext_sum, float pi)
{
    const float ef = pdp_sign(ext_sum);
    float ps = pos + pdp_safe_log(1.0f - pi * ((ef == 1.0f) ? 1.0f : 0.0f), PDP_SCORER_EPS);
    float ng = neg + pdp_safe_log(1.0f - pi * ((ef == -1.0f) ? 1.0f : 0.0f), PDP_SCORER_EPS);
    float pns = ps + ng;
    float dc = all + pdp_safe_log(1.0f - pi, PDP_SCORER_EPS);
    const float bias = (2.0f * pns + dc) / 4.0f;
    ps = ps - bias; ng = ng - bias; pns = pns - bias;
    dc = pdp_safe_exp(dc - bias);
    const float q0 = pdp_safe_exp(ps) - pdp_safe_exp(pns);
    const float q1 = pdp_safe_exp(ng) - pdp_safe_exp(pns);
    const float total = pdp_safe_log((q0 + q1) + dc, PDP_SCORER_EPS);
    return pdp_safe_exp(pdp_safe_log(q1, PDP_SCORER_EPS) - total) - pdp_safe_exp(pdp_safe_log(q0, PDP_SCORER_EPS) - total);
}

// SurveyPropagator per-edge update (pdp_propagate.py:195-218) from the per-variable sums
struct SpOut { float qu, qs, dc; };
// L0 / L1: the two values log(max(1 - pi * [force == +-s], eps)) can take (pdp_propagate.py:197,201), computed once by the caller
__device__ __forceinline__ SpOut d_sp_edge(float s, float P, float N, float y, float force, float L0, float L1)
{
    const float pos = 0.0f + P, neg = 0.0f + N;
    float same = (0.5f * (1.0f + s)) * pos + (0.5f * (1.0f - s)) * neg;
    same = same - y;
    same = same + ((force == s) ? L1 : L0);
    float opp = (0.5f * (1.0f - s)) * pos + (0.5f * (1.0f + s)) * neg;
    opp = opp + ((force == -s) ? L1 : L0);
    float dc = same + opp;
    dc = pdp_safe_exp(dc);
    const float A = pdp_safe_exp(same), Bv = pdp_safe_exp(opp);
    const float qu = A * (1.0f - Bv), qs = Bv * (1.0f - A);
    const float total = (qu + qs) + dc;
    SpOut o; o.qu = qu / total; o.qs = qs / total; o.dc = dc / total;
    return o;
}
