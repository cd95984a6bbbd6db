/*
 * pdp_math.h -- exact, platform-independent fp32 math shared by the HIP kernels and the CPU oracle.
 *
 * Why this exists: the PDP hot path takes DISCRETE decisions (decimation arg-max, convergence
 * thresholds, clause satisfaction) on top of fp32 messages that go through exp/log
 * (reference: src/pdp/nn/pdp_propagate.py:133-137, src/pdp/nn/pdp_predict.py:149-153,
 * src/pdp/nn/util.py:277-286).  A vendor libm on the host (glibc / Sleef inside torch) and OCML on
 * gfx950 differ in the last bit, which makes end-to-end integer parity unprovable.  Every function
 * below is therefore written ONLY in terms of IEEE-754 basic operations (+ - * / fma, compares,
 * integer bit moves), which round identically on x86-64 (SSE/FMA) and on gfx950 (denormals
 * enabled, correctly rounded division = hipcc defaults).  Both sides must be compiled with
 * -ffp-contract=off so that no additional fusion happens.
 *
 * Accuracy: pdp_expf / pdp_logf are Cephes-style minimax kernels, < 1.5 ulp over the ranges the
 * path uses, denormal inputs and outputs handled exactly (the reference relies on
 * log(1e-40) = -92.1034 and exp(-92.1034) = 1e-40).
 *
 * NaN semantics follow torch: max/min propagate NaN (torch.max(x, eps)), sign(NaN) = 0.
 */
#ifndef PDP_MATH_H
#define PDP_MATH_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define PDP_HD __host__ __device__ __forceinline__
#else
#define PDP_HD static inline
#endif

typedef union { uint32_t u; float f; } pdp_f32_bits;

PDP_HD float pdp_bits2f(uint32_t u) { pdp_f32_bits c; c.u = u; return c.f; }
PDP_HD uint32_t pdp_f2bits(float f) { pdp_f32_bits c; c.f = f; return c.u; }

#define PDP_INF  (pdp_bits2f(0x7f800000u))
#define PDP_NAN  (pdp_bits2f(0x7fc00000u))

/* torch.max / torch.min (elementwise, NaN-propagating) */
PDP_HD float pdp_max(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
PDP_HD float pdp_min(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
/* torch.sign: sign(NaN) == 0 (also for the Reinforce force of a NaN score, pdp_decimate.py:230: pinned by trace_reinforce_nan_leak) */
PDP_HD float pdp_sign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
PDP_HD float pdp_abs(float x) { return pdp_bits2f(pdp_f2bits(x) & 0x7fffffffu); }

/* 2^n for -126 <= n <= 127 */
PDP_HD float pdp_pow2i(int n) { return pdp_bits2f((uint32_t)(n + 127) << 23); }

/* ---- opt-in fast build (-DPDP_FAST_MATH, device code only: libpdp_hip_fast.so) ---------------------------------------------------
 * Every function of this header keeps its name, domain and special-value behaviour; on the device its value comes from the
 * transcendental unit (v_exp_f32 = 2^x, v_log_f32 = log2 x, v_rcp_f32; about 1 ulp each) instead of the IEEE-only polynomials below.
 * The results are NOT the oracle's bits any more -- about 1-2 ulp from them -- so this build is gated by the reference-held fixtures
 * only (integer trajectories equal, floats within the tolerances written in the tests: tests/test_fast_build_gpu.py), never by the
 * bit-exact suite; the parity build stays the default.  What the hardware units do not do by themselves is done here explicitly:
 *   exp: v_exp_f32 flushes denormal results, and 2^(x log2 e) loses |x| 2^-24 in the product -- the argument is split into an integer n and
 *        a fraction f carried in two floats (the product's exact residual + the low part of log2 e), the unit sees only f in [-0.5, 0.5],
 *        and v_ldexp_f32 applies 2^n with one correct rounding, also into the denormal range (the reference relies on exp(-92.1) = 1e-40);
 *   log: v_log_f32 flushes denormal arguments -- v_frexp_mant / v_frexp_exp split x = m 2^e exactly (denormals included, log(1e-40) = -92.1)
 *        and the unit sees only m in [0.5, 1). */
#if defined(PDP_FAST_MATH) && (defined(__HIP_DEVICE_COMPILE__))
#define PDP_FAST_DEVICE 1
PDP_HD float pdp_fmaxf_dev(float a, float b) { return __builtin_fmaxf(a, b); }     /* v_max_f32 / v_min_f32: the non-NaN operand wins */
PDP_HD float pdp_fminf_dev(float a, float b) { return __builtin_fminf(a, b); }
/* e^x for a finite x (any magnitude) or NaN */
PDP_HD float pdp_hw_expf(float x)
{
    const float t = x * 1.44269502162933349609375f;                     /* float(log2 e) */
    const float n = __builtin_rintf(t);                                   /* v_rndne_f32 */
    float lo = fmaf(x, 1.44269502162933349609375f, -t);                  /* what the product dropped, exactly */
    lo = fmaf(x, 1.92596299112661746e-8f, lo);                           /* + x * (log2 e - float(log2 e)) */
    const float f = (t - n) + lo;
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);          /* NaN: every step carries it, (int)NaN = 0 */
}
/* log(x) for x > 0 (denormals included); 0 -> -inf, negative -> NaN, NaN -> NaN; +inf is the caller's business */
PDP_HD float pdp_hw_logf(float x)
{
    const float m = __builtin_amdgcn_frexp_mantf(x);
    const int e = __builtin_amdgcn_frexp_expf(x);
    return ((float)e + __builtin_amdgcn_logf(m)) * 0.693147180559945309f;
}
#endif

/* e^x.  Result is correctly scaled into the denormal range (single final rounding).
 * Written branch-free (selects only): on gfx950 early returns would become divergent exec-mask branches. */
PDP_HD float pdp_expf(float x)
{
#ifdef PDP_FAST_DEVICE
    { const float r = pdp_hw_expf(pdp_fminf_dev(pdp_fmaxf_dev(x, -150.0f), 89.0f)); return (x != x) ? x : r; }
#endif
    const int is_nan = (x != x);
    float xc = is_nan ? 0.0f : x;
    xc = (xc > 89.0f) ? 89.0f : xc;
    xc = (xc < -104.5f) ? -104.5f : xc;
    const float t = xc * 1.44269504088896341f;
    const float nf = (t + 12582912.0f) - 12582912.0f;     /* round-to-nearest-even integer */
    float r = fmaf(nf, -0.693359375f, xc);                 /* x - n*ln2 (hi, exact product) */
    r = fmaf(nf, 2.12194440e-4f, r);                       /*           (lo)                */
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, z, r);
    p = p + 1.0f;
    const int n = (int)nf;                                 /* |n| <= 151 after the clamp */
    const int n1 = n >> 1;                                 /* any split with both halves in the normal range works: */
    const int n2 = n - n1;                                 /* the first product is exact, the second rounds once     */
    float res = (p * pdp_pow2i(n1)) * pdp_pow2i(n2);
    res = (x > 88.7228394f) ? PDP_INF : res;
    res = (x < -104.0f) ? 0.0f : res;
    return is_nan ? x : res;
}

/* natural log, x > 0 expected (denormals fine); x == 0 -> -inf, x < 0 -> NaN.  Branch-free like pdp_expf. */
PDP_HD float pdp_logf(float x)
{
#ifdef PDP_FAST_DEVICE
    { const float r = pdp_hw_logf(x); return (pdp_f2bits(x) == 0x7f800000u) ? x : r; }
#endif
    const int is_nan = (x != x);
    const uint32_t u0 = pdp_f2bits(x);
    const int is_inf = (u0 == 0x7f800000u);
    const int not_pos = !(x > 0.0f);                       /* x <= 0 or NaN */
    const float xs = (not_pos || is_inf) ? 1.0f : x;      /* keep the main path on sane input */
    const int den = pdp_f2bits(xs) < 0x00800000u;
    const float xn = xs * (den ? 8388608.0f : 1.0f);
    const uint32_t u = pdp_f2bits(xn);
    int e = (den ? -23 : 0) + (int)(u >> 23) - 126;        /* x = m * 2^e, m in [0.5, 1) */
    float m = pdp_bits2f((u & 0x007fffffu) | 0x3f000000u);
    const int lt = m < 0.707106781186547524f;
    e = e - lt;
    m = (lt ? (m + m) : m) - 1.0f;
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = fmaf(y, m, -1.1514610310e-1f);
    y = fmaf(y, m, 1.1676998740e-1f);
    y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f);
    y = fmaf(y, m, -1.6668057665e-1f);
    y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f);
    y = fmaf(y, m, 3.3333331174e-1f);
    y = (y * m) * z;
    const float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(fe, 0.693359375f, r);
    r = is_inf ? x : r;
    r = (x == 0.0f) ? -PDP_INF : r;
    r = (x < 0.0f) ? PDP_NAN : r;
    return is_nan ? x : r;
}

/* ---- lean variants for hot loops --------------------------------------------------------------
 * Same results as the general functions on their stated domain (checked bit for bit by the tests);
 * they only drop selects for inputs the caller rules out. */
/* torch.max(a, c) / torch.min(a, c) for a constant c that is not NaN: NaN in a propagates */
PDP_HD float pdp_max_c(float a, float c) { return (a < c) ? c : a; }
PDP_HD float pdp_min_c(float a, float c) { return (a > c) ? c : a; }

/* p * 2^n, correctly rounded also into the denormal range (|n| <= 160) */
PDP_HD float pdp_scale2(float p, int n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_ldexpf(p, n);                          /* v_ldexp_f32: one correctly rounded instruction */
#else
    const int n1 = n >> 1, n2 = n - n1;
    return (p * pdp_pow2i(n1)) * pdp_pow2i(n2);
#endif
}

/* e^x for x <= 30 or NaN (the argument of safe_exp after its clamp) */
PDP_HD float pdp_expf_le30(float x)
{
#ifdef PDP_FAST_DEVICE
    { const float r = pdp_hw_expf(pdp_fmaxf_dev(x, -150.0f)); return (x != x) ? x : r; }
#endif
    const float xc = (x < -104.5f) ? -104.5f : x;          /* NaN stays NaN and is returned unchanged below */
    const float t = xc * 1.44269504088896341f;
    const float nf = (t + 12582912.0f) - 12582912.0f;
    float r = fmaf(nf, -0.693359375f, xc);
    r = fmaf(nf, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, z, r);
    p = p + 1.0f;
    float res = pdp_scale2(p, (x != x) ? 0 : (int)nf);
    res = (x < -104.0f) ? 0.0f : res;
    return (x != x) ? x : res;
}

/* log(x) for finite x > 0 (denormals included) or NaN */
PDP_HD float pdp_logf_pos(float x)
{
#ifdef PDP_FAST_DEVICE
    return pdp_hw_logf(x);
#endif
    const uint32_t u0 = pdp_f2bits(x);
    const int den = u0 < 0x00800000u;
    const float xn = x * (den ? 8388608.0f : 1.0f);
    const uint32_t u = pdp_f2bits(xn);
    int e = (den ? -23 : 0) + (int)(u >> 23) - 126;
    float m = pdp_bits2f((u & 0x007fffffu) | 0x3f000000u);
    const int lt = m < 0.707106781186547524f;
    e = e - lt;
    m = (lt ? (m + m) : m) - 1.0f;
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = fmaf(y, m, -1.1514610310e-1f);
    y = fmaf(y, m, 1.1676998740e-1f);
    y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f);
    y = fmaf(y, m, -1.6668057665e-1f);
    y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f);
    y = fmaf(y, m, 3.3333331174e-1f);
    y = (y * m) * z;
    const float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(fe, 0.693359375f, r);
    return (x != x) ? x : r;
}
/* ---- select-free forms for the hot loops --------------------------------------------------------
 * On gfx950 a v_cmp + v_cndmask pair costs about four FMAs (the compare result travels through an SGPR pair), so the
 * per-edge code avoids selects: clamps are v_max_f32, mantissa / exponent come from v_frexp_*, the "m < sqrt(1/2)"
 * decision is integer arithmetic on the bit pattern, and a NaN argument is re-injected at the end as x - x (exactly 0
 * for finite x).  Same results as the general functions on the stated domains (tests/test_hip_ops.py,
 * tests/test_oracle_golden.py). */
PDP_HD float pdp_fmaxf(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaxf(a, b);                           /* v_max_f32: the non-NaN operand wins */
#else
    return (a != a) ? b : ((b != b) ? a : ((a < b) ? b : a));
#endif
}

/* log(max(x, eps)) for a finite-or-NaN x and eps > 0 (the reference's safe_log on everything the SP update feeds it) */
PDP_HD float pdp_safe_log_fin(float x, float eps)
{
#ifdef PDP_FAST_DEVICE
    return pdp_hw_logf(pdp_fmaxf(x, eps)) + (x - x);
#endif
    const float xm = pdp_fmaxf(x, eps);                     /* NaN -> eps here, NaN again through the last line */
#if defined(__HIP_DEVICE_COMPILE__)
    float m = __builtin_amdgcn_frexp_mantf(xm);             /* xm = m * 2^e, m in [0.5, 1), exact, denormals included */
    int e = __builtin_amdgcn_frexp_expf(xm);
#else
    const uint32_t u0 = pdp_f2bits(xm);
    const int den = u0 < 0x00800000u;
    const float xn = xm * (den ? 8388608.0f : 1.0f);
    const uint32_t u = pdp_f2bits(xn);
    int e = (den ? -23 : 0) + (int)(u >> 23) - 126;
    float m = pdp_bits2f((u & 0x007fffffu) | 0x3f000000u);
#endif
    const uint32_t lt = (pdp_f2bits(m) - 0x3f3504f3u) >> 31;   /* m < 0.70710677f: positive floats order like their bit patterns */
    e = e - (int)lt;
    m = pdp_scale2(m, (int)lt) - 1.0f;                       /* (lt ? m + m : m) - 1 */
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = fmaf(y, m, -1.1514610310e-1f);
    y = fmaf(y, m, 1.1676998740e-1f);
    y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f);
    y = fmaf(y, m, -1.6668057665e-1f);
    y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f);
    y = fmaf(y, m, 3.3333331174e-1f);
    y = (y * m) * z;
    const float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(fe, 0.693359375f, r);
    return r + (x - x);
}

/* e^x for a finite x <= 30 or NaN (the reference's safe_exp wherever its argument is a finite sum of logs).
 * No "x < -104 -> 0" select: p * 2^n rounds to zero by itself there (e^-104 = 0.486 * 2^-149). */
PDP_HD float pdp_expf_fin_le30(float x)
{
#ifdef PDP_FAST_DEVICE
    return pdp_hw_expf(pdp_fmaxf(x, -150.0f)) + (x - x);
#endif
    const float xc = pdp_fmaxf(x, -104.5f);
    const float t = xc * 1.44269504088896341f;
    const float nf = (t + 12582912.0f) - 12582912.0f;
    float r = fmaf(nf, -0.693359375f, xc);
    r = fmaf(nf, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, z, r);
    p = p + 1.0f;
    return pdp_scale2(p, (int)nf) + (x - x);
}
/* safe_exp with the lean pieces (valid for every input: the clamp bounds the argument) */
PDP_HD float pdp_safe_exp_fast(float x) { return pdp_expf_le30(pdp_min_c(x, 30.0f)); }

/* ---- activation functions of the neural plug-ins (finite or NaN arguments; select-free like the forms above) ------------ */
PDP_HD float pdp_fminf(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fminf(a, b);                           /* v_min_f32: the non-NaN operand wins */
#else
    return (a != a) ? b : ((b != b) ? a : ((a > b) ? b : a));
#endif
}

/* e^min(x, hi) for any finite x or NaN: underflows to 0; with hi = 89 it overflows to +inf (p * 2^128) like expf */
PDP_HD float pdp_expf_fin_hi(float x, float hi)
{
#ifdef PDP_FAST_DEVICE
    return pdp_hw_expf(pdp_fminf(pdp_fmaxf(x, -150.0f), hi)) + (x - x);
#endif
    const float xc = pdp_fminf(pdp_fmaxf(x, -104.5f), hi);
    const float t = xc * 1.44269504088896341f;
    const float nf = (t + 12582912.0f) - 12582912.0f;
    float r = fmaf(nf, -0.693359375f, xc);
    r = fmaf(nf, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, z, r);
    p = p + 1.0f;
    return pdp_scale2(p, (int)nf) + (x - x);
}
PDP_HD float pdp_expf_fin(float x) { return pdp_expf_fin_hi(x, 89.0f); }

/* torch F.logsigmoid(x) = min(x, 0) - log1p(exp(-|x|)).  With t = e^-|x| in (0, 1]: log1p(t) = t * P(t), P the degree-8 polynomial
 * fitted to log1p(t) / t at Chebyshev nodes of [0, 1] (truncation 3e-8 relative; with fp32 Horner rounding < 3 ulp over the whole range,
 * tests/test_oracle_golden.py checks it against log1p in double).  For tiny t the product is t itself.  No log, no exponent extraction:
 * 27 instead of 50 VALU instructions per element on gfx950, which matters because f32 MFMA and VALU time add up on a SIMD. */
PDP_HD float pdp_logsigmoidf(float x)
{
#ifdef PDP_FAST_DEVICE
    {   /* min(x, 0) - log1p(e^-|x|): e^-|x| underflows to 0 past 87 (v_exp_f32 flushes), where log1p of it is below 1e-38 anyway */
        const float t = __builtin_amdgcn_exp2f(-1.44269504088896341f * pdp_abs(x));
        return fmaf(-0.693147180559945309f, __builtin_amdgcn_logf(1.0f + t), pdp_fminf(x, 0.0f));
    }
#endif
    const float t = pdp_expf_fin_le30(-pdp_abs(x));
    float p = 5.253457930e-03f;
    p = fmaf(p, t, -2.958850749e-02f);
    p = fmaf(p, t, 7.836166769e-02f);
    p = fmaf(p, t, -1.367477030e-01f);
    p = fmaf(p, t, 1.911143064e-01f);
    p = fmaf(p, t, -2.484436929e-01f);
    p = fmaf(p, t, 3.331927061e-01f);
    p = fmaf(p, t, -4.999950230e-01f);
    p = fmaf(p, t, 1.0f);
    return pdp_fminf(x, 0.0f) - p * t;                      /* a NaN x is dropped by the min and carried by t */
}

/* 1 / d for 1 <= d <= 2^126 (or NaN).  On the device: v_rcp_f32 (1 ulp) + one Newton step + the residual correction of the compiler's own
 * division sequence, without its scaling / fix-up instructions, which only matter outside this range -- 8 instead of 12 instructions.  The
 * result is the correctly rounded quotient for EVERY float of the range: tests/test_hip_ops.py::test_reciprocal_exhaustive compares all
 * 1.06e9 of them with the IEEE division, which is what the host side (the oracle) computes here. */
PDP_HD float pdp_rcp_ge1(float d)
{
#ifdef PDP_FAST_DEVICE
    return __builtin_amdgcn_rcpf(d);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(d);
    const float e = fmaf(-d, r, 1.0f);
    r = fmaf(e, r, r);
    const float rem = fmaf(-d, r, 1.0f);
    float q = fmaf(rem, r, r);
    const float rem2 = fmaf(-d, q, 1.0f);
    q = fmaf(rem2, r, q);
    return q;
#else
    return 1.0f / d;
#endif
}

/* torch.sigmoid(x) = 1 / (1 + exp(-x)).  The exponent is clamped at 87 so that the denominator stays in the range of pdp_rcp_ge1: for
 * x < -87 the result is 1.6e-38 where torch's decays on to 0 through the denormals -- a difference of at most 1.6e-38. */
PDP_HD float pdp_sigmoidf(float x)
{
#ifdef PDP_FAST_DEVICE
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * pdp_fmaxf(x, -87.0f))) + (x - x);
#endif
    return pdp_rcp_ge1(1.0f + pdp_expf_fin_hi(-x, 87.0f));
}

/* tanh(x) = em / (em + 2), em = e^{2|x|} - 1 without cancellation: with 2|x| = n ln2 + r the polynomial part q = e^r - 1 is
 * the exact answer for n == 0.  |x| is clamped at 10 (tanh(10) rounds to 1). */
PDP_HD float pdp_tanhf(float x)
{
#ifdef PDP_FAST_DEVICE
    {   /* 1 - 2 / (e^{2|x|} + 1): absolute error about 1e-7 (the relative accuracy of the parity form near 0 is not kept; the value feeds a
         * sign and an arg-max of magnitudes in the np-d-np scorer head, util.py:250-251) */
        const float a = pdp_fminf(pdp_abs(x), 10.0f);
        const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.88539008177792681f * a) + 1.0f);
        const float v = 1.0f - (r + r);
        return pdp_bits2f(pdp_f2bits(v) | (pdp_f2bits(x) & 0x80000000u)) + (x - x);
    }
#endif
    const float a = pdp_fminf(pdp_abs(x), 10.0f);
    const float y = a + a;
    const float t = y * 1.44269504088896341f;
    const float nf = (t + 12582912.0f) - 12582912.0f;
    float r = fmaf(nf, -0.693359375f, y);
    r = fmaf(nf, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float q = fmaf(p, z, r);                          /* e^r - 1 */
    const int n = (int)nf;
    const float big = pdp_scale2(q + 1.0f, n) - 1.0f;
    const float em = (n == 0) ? q : big;
    const float v = em / (em + 2.0f);
    return pdp_bits2f(pdp_f2bits(v) | (pdp_f2bits(x) & 0x80000000u)) + (x - x);
}

/* GRU candidate gate: tanh(x) = sign(x) (1 - 2 / (e^{2|x|} + 1)), |x| clamped at 10 (tanh(10) rounds to 1).  Absolute error < 1.2e-7 (half an ulp of 1 from the
 * division and from the subtraction each, plus a quarter of the exponential's relative error); near 0 the RELATIVE error is not that of a
 * cancellation-free form, which the GRU does not need: the value only enters h' = (h - n) z + n.  32 instead of 45 VALU instructions on gfx950
 * (no expm1 reconstruction, no n == 0 select). */
PDP_HD float pdp_tanhf_abs(float x)
{
#ifdef PDP_FAST_DEVICE
    {
        const float a = pdp_fminf(pdp_abs(x), 10.0f);
        const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.88539008177792681f * a) + 1.0f);
        const float v = 1.0f - (r + r);
        return pdp_bits2f(pdp_f2bits(v) | (pdp_f2bits(x) & 0x80000000u)) + (x - x);
    }
#endif
    const float a = pdp_fminf(pdp_abs(x), 10.0f);
    const float t = pdp_expf_fin(a + a);
    const float r = pdp_rcp_ge1(t + 1.0f);                  /* t + 1 in [2, e^20 + 1] */
    const float v = 1.0f - (r + r);
    return pdp_bits2f(pdp_f2bits(v) | (pdp_f2bits(x) & 0x80000000u)) + (x - x);
}

/* ---- the reference's clamped forms ------------------------------------------------------- */
/* safe_log(x) = log(max(x, eps))  (reference: pdp_propagate.py:133-134, pdp_predict.py:149-150) */
PDP_HD float pdp_safe_log(float x, float eps) { return pdp_logf(pdp_max(x, eps)); }
/* safe_exp(x) = exp(min(x, 30))   (reference: pdp_propagate.py:136-137, util.py:277-280) */
PDP_HD float pdp_safe_exp(float x) { return pdp_expf(pdp_min(x, 30.0f)); }

#define PDP_SP_EPS      1e-40f   /* SurveyPropagator eps (pdp_propagate.py:124), fp32 denormal */
#define PDP_SCORER_EPS  1e-10f   /* SurveyScorer eps (pdp_predict.py:141) */

/* ---- Philox4x32-10 counter RNG (device-side random numbers for Walk-SAT / random fill) ----
 * Stateless: value = f(seed, stream, step, index), so the CPU oracle reproduces it bit-for-bit in
 * any order.  Returns a float in [0, 1) with 24 random bits (same grid as torch.rand fp32). */
PDP_HD uint32_t pdp_mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

PDP_HD uint32_t pdp_philox_u32(uint64_t seed, uint32_t stream, uint32_t step, uint32_t index)
{
    uint32_t c0 = index, c1 = step, c2 = stream, c3 = 0x5044502du;   /* "PDP-" */
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = pdp_mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = pdp_mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

PDP_HD float pdp_philox_uniform(uint64_t seed, uint32_t stream, uint32_t step, uint32_t index)
{
    return (float)(pdp_philox_u32(seed, stream, step, index) >> 8) * 5.9604644775390625e-8f;  /* 2^-24 */
}

#define PDP_RNG_STREAM_FILL   1u   /* IdentityPredictor random fill (pdp_predict.py:121-126) */
#define PDP_RNG_STREAM_WSVAR  2u   /* Walk-SAT per-variable draw  (solver.py:457) */
#define PDP_RNG_STREAM_WSCOIN 3u   /* Walk-SAT per-instance coin  (solver.py:460) */
#define PDP_RNG_STREAM_REINF  4u   /* Reinforce decimation coin   (pdp_decimate.py:218) */

#endif /* PDP_MATH_H */
