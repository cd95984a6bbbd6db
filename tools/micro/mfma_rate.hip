// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 on gfx950 -- one dependent chain per wave, two alternating chains per wave,
// with one or two waves per SIMD, with and without an operand stream (one L2 weight load + one LDS read per MFMA) -- and the shader clock
// under that load (s_memtime = shader clock, s_memrealtime = 100 MHz).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void k(float *out, const float *w, int iters, unsigned long long *clk)
{
    __shared__ float lds[8192];
    for (int j = threadIdx.x; j < 8192; j += blockDim.x) lds[j] = j * 1e-4f;
    __syncthreads();
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = r; a1[r] = -r; }
    float x = threadIdx.x * 1e-3f, y = 0.5f;
    const int l = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, 1 << 22, 0x00020000);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 64; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 32; ++u) { a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0); }
        } else if (MODE == 2) {            // operand stream: 64 weight loads + 64 LDS reads requested one batch ahead
            float bw[64], aw[64];
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                bw[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, l * 4, ((i & 63) * 64 + u) * 1024, 0));
                aw[u] = lds[(l * 65 + u + i) & 8191];
            }
#pragma unroll
            for (int u = 0; u < 64; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[u], bw[u], a0, 0, 0, 0);
        }
    }
    if constexpr (MODE >= 20) {            // as below with 2 chunks of lookahead, no address arithmetic: 20 = weights only, 21 = LDS only, 22 = both
        constexpr int LA = MODE >= 30 ? MODE - 30 : 2, CH = 10, NS = LA + 1;
        constexpr bool WL = MODE != 21, AL = MODE != 20, STREAM = MODE >= 23;
        int soff = 0;
        float bw[NS][CH], aw[2][CH];
        const float *lp = lds + l * 65;
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
            for (int u = 0; u < CH; ++u) bw[q][u] = y;
#pragma unroll
        for (int u = 0; u < CH; ++u) { aw[0][u] = x; aw[1][u] = x; }
        for (int i = 0; i < iters * 64 / (CH * NS * 2); ++i) {
#pragma unroll
            for (int q = 0; q < NS * 2; ++q) {
                if (STREAM) { soff += CH * 1920; if (soff >= 151 * 1920) soff = 0; }       // a 290 KB weight matrix walked row pair by row pair
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    if (WL) bw[(q + LA) % NS][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, l * 4 + (l >> 5) * 832, (STREAM ? soff : q * CH * 1024) + u * (STREAM ? 1920 : 1024), 0));
                    if (AL) aw[(q + 1) & 1][u] = lp[2 * (q * CH + u)];
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[q & 1][u], bw[q % NS][u], a0, 0, 0, 0);
                asm volatile("" : "+v"(a0));
#pragma unroll
                for (int u = 0; u < CH; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); if (WL) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); if (AL) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if constexpr (MODE >= 10) {                      // software pipeline: chunks of 10 MFMAs, weights LA chunks ahead, LDS one chunk ahead (MODE = 10 + LA)
        constexpr int LA = MODE - 10, CH = 10, NS = LA + 1;
        float bw[NS][CH], aw[2][CH];
        int soff = 0;
#pragma unroll
        for (int q = 0; q < LA; ++q)
#pragma unroll
            for (int u = 0; u < CH; ++u) bw[q][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, l * 4, (q * CH + u) * 1024, 0));
#pragma unroll
        for (int u = 0; u < CH; ++u) aw[0][u] = lds[(l * 65 + u) & 8191];
        for (int i = 0; i < iters * 64 / (CH * NS * 2); ++i) {
#pragma unroll
            for (int q = 0; q < NS * 2; ++q) {
                soff = (soff + CH * 1024) & ((1 << 22) - 1);
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    bw[(q + LA) % NS][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, l * 4, soff + u * 1024, 0));
                    aw[(q + 1) & 1][u] = lds[(l * 65 + u + q * CH + i) & 8191];
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[q & 1][u], bw[q % NS][u], a0, 0, 0, 0);
                asm volatile("" : "+v"(a0));
#pragma unroll
                for (int u = 0; u < CH; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
static void run(const char *name, int threads, int iters)
{
    float *out, *w; unsigned long long *clk;
    const int grid = 256;
    hipMalloc(&out, grid * threads * 4); hipMalloc(&w, 1 << 22); hipMemset(w, 0, 1 << 22); hipMalloc(&clk, grid * 16);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, out, w, 10, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, out, w, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512]; hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    const double waves_per_simd = threads / 256.0, n = MODE >= 30 ? (double)(iters * 64 / (20 * (MODE - 29))) * (20 * (MODE - 29)) : MODE >= 20 ? (double)(iters * 64 / 60) * 60 : MODE >= 10 ? (double)(iters * 64 / (10 * (MODE - 9) * 2)) * (10 * (MODE - 9) * 2) : 64.0 * iters;
    printf("%-44s %d waves/SIMD: %6.1f shader cycles per MFMA and wave, %6.1f per SIMD; clock %.0f MHz; %.3f ms\n", name, (int)waves_per_simd,
           (double)h[0] / n, (double)h[0] / n / waves_per_simd, (double)h[0] / ((double)h[1] / 100.0), ms);
    hipFree(out); hipFree(w); hipFree(clk);
}

int main()
{
    const int it = 20000;
    run<0>("one dependent chain", 256, it); run<0>("one dependent chain", 512, it);
    run<1>("two alternating chains", 256, it); run<1>("two alternating chains", 512, it);
    run<2>("dependent chain + L2 load + LDS read / MFMA", 256, it); run<2>("dependent chain + L2 load + LDS read / MFMA", 512, it);
    run<20>("pipelined, weight loads only, no VALU", 256, it); run<21>("pipelined, LDS reads only, no VALU", 256, it); run<22>("pipelined, both, no VALU", 256, it);
    run<22>("pipelined, both, no VALU", 512, it);
    run<23>("same, weights streamed from L2 (2 ahead)", 256, it); run<31>("same, weights streamed from L2 (1 ahead)", 256, it);
    run<33>("same, weights streamed from L2 (3 ahead)", 256, it); run<35>("same, weights streamed from L2 (5 ahead)", 256, it);
    run<23>("same, weights streamed from L2 (2 ahead)", 512, it);
    run<11>("pipelined, weights 1 chunk of 10 ahead", 256, it); run<12>("pipelined, weights 2 chunks ahead", 256, it);
    run<13>("pipelined, weights 3 chunks ahead", 256, it); run<15>("pipelined, weights 5 chunks ahead", 256, it);
    run<12>("pipelined, weights 2 chunks ahead", 512, it);
    return 0;
}
