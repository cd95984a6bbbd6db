// Micro-benchmark: do v_mfma_f32_32x32x2_f32 and fp32 VALU work overlap on one gfx950 SIMD, or does their time add?
//
//   (i)   M    one wave per SIMD, back-to-back independent fp32 MFMAs (4 accumulator chains)
//   (ii)  V    one wave per SIMD, independent VALU work: v_fma_f32 (8 chains), v_exp_f32, or the log-sigmoid body's instruction mix
//   (iii) M|V  two waves per SIMD: one does (i), its partner does (ii)      -> max(M, V) if the pipes overlap, M + V if MFMA takes VALU issue
//   (iv)  MV   one wave per SIMD, both interleaved in program order (1 MFMA, then k VALU)
//   (v)   MV|MV two waves per SIMD, each interleaved (what k_gru_pipe / k_agg_* do today)
//   the same five with v_mfma_f32_32x32x16_bf16 in place of the fp32 MFMA (the guide's overlap statement was measured on bf16)
//
// One workgroup per CU (256 workgroups), no memory traffic; time = s_memtime shader cycles of wave 0 .. wave 7, max over the workgroup's waves
// of one CU.  build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { V_FMA = 0, V_EXP = 1, V_MIX = 2 };

template <bool BF16>
__device__ __forceinline__ void mfma4(f32x16 &a0, f32x16 &a1, f32x16 &a2, f32x16 &a3, float x, float y, bf16x8 bx, bf16x8 by)
{
    if constexpr (BF16) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n"
                     "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(bx), "v"(by));
    } else {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n v_mfma_f32_32x32x2_f32 %1, %4, %5, %1\n"
                     "v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n v_mfma_f32_32x32x2_f32 %3, %4, %5, %3"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
    }
}
template <bool BF16>
__device__ __forceinline__ void mfma1(f32x16 &a, float x, float y, bf16x8 bx, bf16x8 by)
{
    if constexpr (BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a) : "v"(bx), "v"(by));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
}

// 8 VALU instructions on 8 independent chains
template <int KIND>
__device__ __forceinline__ void valu8(float (&c)[8], float p, float q)
{
    if constexpr (KIND == V_FMA) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(p), "v"(q));
    } else if constexpr (KIND == V_EXP) {
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                     "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
    } else {
        // the parity build's log-sigmoid body is ~ 2/3 fma / mul / add, 1/6 integer / compare-select, 1/6 others: 5 fma + 1 and + 1 cndmask + 1 max
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_and_b32 %2, %2, %8\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_cndmask_b32 %4, %4, %9, vcc\n v_fma_f32 %5, %5, %8, %9\n v_max_f32 %6, %6, %8\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(p), "v"(q) : "vcc");
    }
}

// ROLE: 0 = MFMA only, 1 = VALU only, 2 = interleaved (4 MFMAs, each followed by K8 groups of 8 VALU)
// SPLIT: waves 0-3 take role A, waves 4-7 role B (a 256-thread launch has only waves 0-3)
template <int ROLE_A, int ROLE_B, int KIND, int K8, bool BF16>
__global__ void __launch_bounds__(512) k(float *out, int iters, unsigned long long *clk)
{
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? ROLE_A : ROLE_B;
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = r; a1[r] = -r; a2[r] = 0.5f * r; a3[r] = 1.0f; }
    float c[8];
    for (int r = 0; r < 8; ++r) c[r] = 1e-3f * (threadIdx.x + r);
    const float x = threadIdx.x * 1e-3f, y = 0.5f, p = 0.999f, q = 1e-4f;
    bf16x8 bx, by;
    for (int r = 0; r < 8; ++r) { bx[r] = (__bf16)(0.01f * r); by[r] = (__bf16)0.5f; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0) {
        for (int i = 0; i < iters; ++i) mfma4<BF16>(a0, a1, a2, a3, x, y, bx, by);
    } else if (role == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int g = 0; g < 4 * K8; ++g) valu8<KIND>(c, p, q);
        }
    } else {
        for (int i = 0; i < iters; ++i) {
            mfma1<BF16>(a0, x, y, bx, by);
#pragma unroll
            for (int g = 0; g < K8; ++g) valu8<KIND>(c, p, q);
            mfma1<BF16>(a1, x, y, bx, by);
#pragma unroll
            for (int g = 0; g < K8; ++g) valu8<KIND>(c, p, q);
            mfma1<BF16>(a2, x, y, bx, by);
#pragma unroll
            for (int g = 0; g < K8; ++g) valu8<KIND>(c, p, q);
            mfma1<BF16>(a3, x, y, bx, by);
#pragma unroll
            for (int g = 0; g < K8; ++g) valu8<KIND>(c, p, q);
        }
    }
    // the MFMA results must have landed before the clock is read
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    for (int r = 0; r < 8; ++r) s += c[r];
    asm volatile("" :: "v"(s));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
}

struct Res { double mfma_wave_cycles, valu_wave_cycles, ms; };

template <int ROLE_A, int ROLE_B, int KIND, int K8, bool BF16>
static Res run(int threads, int iters)
{
    float *out; unsigned long long *clk;
    const int grid = 256;
    hipMalloc(&out, grid * 512 * 4); hipMalloc(&clk, grid * 8 * 8); hipMemset(clk, 0, grid * 8 * 8);
    hipLaunchKernelGGL((k<ROLE_A, ROLE_B, KIND, K8, BF16>), dim3(grid), dim3(threads), 0, 0, out, 16, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ROLE_A, ROLE_B, KIND, K8, BF16>), dim3(grid), dim3(threads), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[256 * 8];
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    // median over the CUs of the slowest wave of each role
    double a[256], b[256];
    for (int g = 0; g < grid; ++g) {
        a[g] = (double)std::max(std::max(h[g * 8], h[g * 8 + 1]), std::max(h[g * 8 + 2], h[g * 8 + 3]));
        b[g] = threads > 256 ? (double)std::max(std::max(h[g * 8 + 4], h[g * 8 + 5]), std::max(h[g * 8 + 6], h[g * 8 + 7])) : 0.0;
    }
    std::sort(a, a + grid); std::sort(b, b + grid);
    hipFree(out); hipFree(clk);
    return Res{a[grid / 2] / iters, b[grid / 2] / iters, (double)ms};
}

template <int KIND, int K8, bool BF16>
static void table(const char *kind_name, int iters)
{
    // per loop iteration: 4 MFMAs and 4 * K8 * 8 VALU instructions
    const Res m = run<0, 0, KIND, K8, BF16>(256, iters);
    const Res v = run<1, 1, KIND, K8, BF16>(256, iters);
    const Res mv = run<0, 1, KIND, K8, BF16>(512, iters);
    const Res il = run<2, 2, KIND, K8, BF16>(256, iters);
    const Res il2 = run<2, 2, KIND, K8, BF16>(512, iters);
    const Res mm = run<0, 0, KIND, K8, BF16>(512, iters);
    const Res vv = run<1, 1, KIND, K8, BF16>(512, iters);
    const double M = m.mfma_wave_cycles, V = v.mfma_wave_cycles;
    const double both = std::max(mv.mfma_wave_cycles, mv.valu_wave_cycles);
    printf("\n%s MFMA, VALU = %s, %d VALU per MFMA   (cycles per loop iteration = 4 MFMAs + %d VALU; one workgroup per CU)\n",
           BF16 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32", kind_name, 8 * K8, 32 * K8);
    printf("  (i)   M alone, 1 wave/SIMD                     %8.1f   = %.1f cycles per MFMA\n", M, M / 4);
    printf("  (ii)  V alone, 1 wave/SIMD                     %8.1f   = %.2f cycles per VALU instruction\n", V, V / (32 * K8));
    printf("  (iii) M wave beside V wave on each SIMD         M wave %8.1f, V wave %8.1f   max(M,V) = %.1f, M+V = %.1f  -> overlap fraction %.2f\n",
           mv.mfma_wave_cycles, mv.valu_wave_cycles, std::max(M, V), M + V, (M + V - both) / std::min(M, V));
    printf("  (iv)  interleaved in one wave, 1 wave/SIMD     %8.1f   -> overlap fraction %.2f\n", il.mfma_wave_cycles, (M + V - il.mfma_wave_cycles) / std::min(M, V));
    printf("  (v)   interleaved, 2 waves/SIMD (per 2 loops)  %8.1f   = %.1f per loop -> overlap fraction %.2f\n", std::max(il2.mfma_wave_cycles, il2.valu_wave_cycles),
           std::max(il2.mfma_wave_cycles, il2.valu_wave_cycles) / 2, (M + V - std::max(il2.mfma_wave_cycles, il2.valu_wave_cycles) / 2) / std::min(M, V));
    printf("  (vi)  M beside M, 2 waves/SIMD                 %8.1f   (2 M = %.1f)      (vii) V beside V %8.1f   (2 V = %.1f)\n",
           std::max(mm.mfma_wave_cycles, mm.valu_wave_cycles), 2 * M, std::max(vv.mfma_wave_cycles, vv.valu_wave_cycles), 2 * V);
}

int main()
{
    const int it = 20000;
    printf("overlap fraction: 1 = the shorter of the two is fully hidden (time = max), 0 = the times add\n");
    table<V_FMA, 2, false>("v_fma_f32", it);     // 16 VALU per MFMA: V = half of M at 2 cycles per VALU
    table<V_FMA, 4, false>("v_fma_f32", it);     // 32 VALU per MFMA: V = M at 2 cycles
    table<V_MIX, 2, false>("log-sigmoid mix", it);
    table<V_EXP, 1, false>("v_exp_f32", it);
    table<V_FMA, 2, true>("v_fma_f32", it);      // bf16 MFMA (16 passes = 32 cycles .. 64 cycles): the case the guide measured
    table<V_FMA, 4, true>("v_fma_f32", it);
    return 0;
}
