// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs transcendental ops on gfx950 (one wave64 per SIMD and four).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    const float2v av = {a, a}, bv = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                p0 = __builtin_elementwise_fma(p0, av, bv); p1 = __builtin_elementwise_fma(p1, av, bv);
                p2 = __builtin_elementwise_fma(p2, av, bv); p3 = __builtin_elementwise_fma(p3, av, bv);
                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
                x4 = __builtin_amdgcn_exp2f(x4); x5 = __builtin_amdgcn_exp2f(x5); x6 = __builtin_amdgcn_exp2f(x6); x7 = __builtin_amdgcn_exp2f(x7);
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
                x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
            }
        } else if (MODE == 5) {                 // ONE dependent chain per wave
#pragma unroll
            for (int u = 0; u < 64; ++u) x0 = __builtin_fmaf(x0, a, b);
        } else if (MODE == 6) {                 // TWO dependent chains per wave
#pragma unroll
            for (int u = 0; u < 32; ++u) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); }
        } else if (MODE == 7) {                 // IEEE division (div_scale x2, rcp, 4 fma, div_fmas, div_fixup), 8 independent
#pragma unroll
            for (int u = 0; u < 8; ++u) { x0 = x0 / a; x1 = x1 / a; x2 = x2 / a; x3 = x3 / a; x4 = x4 / a; x5 = x5 / a; x6 = x6 / a; x7 = x7 / a; }
        } else if (MODE == 8) {                 // compare + select through an SGPR pair (v_cmp_*_e64 + v_cndmask_e64), 8 independent
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = (x0 < a) ? b : x0 + 1.0f; x1 = (x1 < a) ? b : x1 + 1.0f; x2 = (x2 < a) ? b : x2 + 1.0f; x3 = (x3 < a) ? b : x3 + 1.0f;
                x4 = (x4 < a) ? b : x4 + 1.0f; x5 = (x5 < a) ? b : x5 + 1.0f; x6 = (x6 < a) ? b : x6 + 1.0f; x7 = (x7 < a) ? b : x7 + 1.0f;
            }
        } else if (MODE == 9) {                 // v_frexp_mant_f32
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_amdgcn_frexp_mantf(x0) + a; x1 = __builtin_amdgcn_frexp_mantf(x1) + a; x2 = __builtin_amdgcn_frexp_mantf(x2) + a; x3 = __builtin_amdgcn_frexp_mantf(x3) + a;
                x4 = __builtin_amdgcn_frexp_mantf(x4) + a; x5 = __builtin_amdgcn_frexp_mantf(x5) + a; x6 = __builtin_amdgcn_frexp_mantf(x6) + a; x7 = __builtin_amdgcn_frexp_mantf(x7) + a;
            }
        } else if (MODE == 10) {                // v_frexp_exp_i32_f32 + v_cvt_f32_i32
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = (float)__builtin_amdgcn_frexp_expf(x0) + a; x1 = (float)__builtin_amdgcn_frexp_expf(x1) + a; x2 = (float)__builtin_amdgcn_frexp_expf(x2) + a; x3 = (float)__builtin_amdgcn_frexp_expf(x3) + a;
                x4 = (float)__builtin_amdgcn_frexp_expf(x4) + a; x5 = (float)__builtin_amdgcn_frexp_expf(x5) + a; x6 = (float)__builtin_amdgcn_frexp_expf(x6) + a; x7 = (float)__builtin_amdgcn_frexp_expf(x7) + a;
            }
        } else if (MODE == 11) {                // fmaxf (v_max_f32, maybe + canonicalize) + add
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_fmaxf(x0, a) + b; x1 = __builtin_fmaxf(x1, a) + b; x2 = __builtin_fmaxf(x2, a) + b; x3 = __builtin_fmaxf(x3, a) + b;
                x4 = __builtin_fmaxf(x4, a) + b; x5 = __builtin_fmaxf(x5, a) + b; x6 = __builtin_fmaxf(x6, a) + b; x7 = __builtin_fmaxf(x7, a) + b;
            }
        } else if (MODE == 12) {                // integer sub + shift + and (3 ops)
            unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1), u2 = __float_as_uint(x2), u3 = __float_as_uint(x3);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                u0 = ((u0 - 0x3f3504f3u) >> 3) & 0x7fffff1u; u1 = ((u1 - 0x3f3504f3u) >> 3) & 0x7fffff1u; u2 = ((u2 - 0x3f3504f3u) >> 3) & 0x7fffff1u; u3 = ((u3 - 0x3f3504f3u) >> 3) & 0x7fffff1u;
                asm volatile("" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            }
            x0 = __uint_as_float(u0); x1 = __uint_as_float(u1); x2 = __uint_as_float(u2); x3 = __uint_as_float(u3);
        } else if (MODE == 13) {                // ONE dependent v_pk_fma chain per wave
#pragma unroll
            for (int u = 0; u < 64; ++u) { p0 = __builtin_elementwise_fma(p0, av, bv); asm volatile("" : "+v"(p0)); }
        } else if (MODE == 14) {                // TWO dependent v_pk_fma chains per wave
#pragma unroll
            for (int u = 0; u < 32; ++u) { p0 = __builtin_elementwise_fma(p0, av, bv); p1 = __builtin_elementwise_fma(p1, av, bv); asm volatile("" : "+v"(p0), "+v"(p1)); }
        } else if (MODE == 15) {                // FOUR dependent scalar fma chains per wave
#pragma unroll
            for (int u = 0; u < 16; ++u) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b); }
        } else if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_ldexpf(x0, i); x1 = __builtin_ldexpf(x1, i); x2 = __builtin_ldexpf(x2, i); x3 = __builtin_ldexpf(x3, i);
                x4 = __builtin_ldexpf(x4, i); x5 = __builtin_ldexpf(x5, i); x6 = __builtin_ldexpf(x6, i); x7 = __builtin_ldexpf(x7, i);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
static void run(const char *name, int waves_per_simd, double ops_per_inst)
{
    float *out; hipMalloc(&out, sizeof(float) * 256 * 4096);
    const int iters = 20000, blocks = 256 * waves_per_simd;     // 256 CUs x (4 waves = 1 per SIMD) per block of 256 threads
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 0.999f, 0.001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_wave = (double)iters * 8 * (MODE == 1 ? 4 : 8);   // modes 7/8: per source-level operation
    // cycles per instruction per wave, assuming 2.4 GHz and waves_per_simd waves sharing each SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / (insts_per_wave * waves_per_simd);
    printf("%-22s waves/SIMD=%d  %.2f ms  -> %.2f cycles per wave-instruction (%.0f lane-ops/inst)\n", name, waves_per_simd, ms, cyc, ops_per_inst * 64);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_fma_f32", 1, 1); run<1>("v_pk_fma_f32", 1, 2); run<2>("v_exp_f32", 1, 1); run<3>("v_rcp_f32", 1, 1); run<4>("v_ldexp_f32", 1, 1); }
        if (w == 2) { run<0>("v_fma_f32", 2, 1); run<1>("v_pk_fma_f32", 2, 2); run<2>("v_exp_f32", 2, 1); run<3>("v_rcp_f32", 2, 1); run<4>("v_ldexp_f32", 2, 1); }
        if (w == 1) { run<5>("fma 1 dep chain", 1, 1); run<6>("fma 2 dep chains", 1, 1); run<7>("x / a (IEEE)", 1, 1); run<8>("cmp+select+add", 1, 1); }
        if (w == 2) { run<5>("fma 1 dep chain", 2, 1); run<6>("fma 2 dep chains", 2, 1); run<7>("x / a (IEEE)", 2, 1); run<8>("cmp+select+add", 2, 1); }
        if (w == 4) { run<5>("fma 1 dep chain", 4, 1); run<6>("fma 2 dep chains", 4, 1); run<7>("x / a (IEEE)", 4, 1); run<8>("cmp+select+add", 4, 1); }
        if (w == 4) { run<9>("frexp_mant + add", 4, 1); run<10>("frexp_exp+cvt+add", 4, 1); run<11>("fmaxf + add", 4, 1); run<12>("int sub,shr,and (x4)", 4, 1); }
        if (w == 4) { run<13>("pk_fma 1 dep chain", 4, 2); run<14>("pk_fma 2 dep chains", 4, 2); run<15>("fma 4 dep chains", 4, 1); }
        if (w == 2) { run<13>("pk_fma 1 dep chain", 2, 2); run<14>("pk_fma 2 dep chains", 2, 2); run<15>("fma 4 dep chains", 2, 1); }
        if (w == 4) { run<0>("v_fma_f32", 4, 1); run<1>("v_pk_fma_f32", 4, 2); run<2>("v_exp_f32", 4, 1); run<3>("v_rcp_f32", 4, 1); run<4>("v_ldexp_f32", 4, 1); }
    }
    return 0;
}
