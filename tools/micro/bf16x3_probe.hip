// Probe for the "three-term bf16 product" idea: accuracy of C = A B (32 x 32 x K per wave) computed as
//   (1) v_mfma_f32_32x32x2_f32 chain (what the kernels use),
//   (2) three v_mfma_f32_32x32x16_bf16 per 16 k on operands split x = hi + lo (bf16 each): hi hi + hi lo + lo hi,
//   (3) one bf16 MFMA on the rounded operands (what plain bf16 would give),
// against a double-precision reference, and the issue rate of the bf16 instruction next to the f32 one.
// build + run: hipcc -O3 --offload-arch=gfx950 tools/micro/bf16x3_probe.hip -o /tmp/bf16x3 && /tmp/bf16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline __bf16 to_bf16(float x) { return (__bf16)x; }          // round to nearest even

template <int MODE>
__global__ void k_acc(const float *A /*[32][K]*/, const float *B /*[K][32]*/, int K, float *C /*[32][32]*/)
{
    const int l = threadIdx.x & 63, i = l & 31;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (MODE == 1) {
        const int kh = l >> 5;
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + kh], B[(k + kh) * 32 + i], acc, 0, 0, 0);
    } else {
        const int kq = l >> 5;
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, al, bh, bl;
            for (int e = 0; e < 8; ++e) {
                const float a = A[i * K + k + 8 * kq + e], b = B[(k + 8 * kq + e) * 32 + i];
                ah[e] = to_bf16(a); al[e] = to_bf16(a - (float)ah[e]);
                bh[e] = to_bf16(b); bl[e] = to_bf16(b - (float)bh[e]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            if (MODE == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            }
        }
    }
    const int kh = l >> 5;
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i] = acc[r];
}

template <int MODE>
__global__ void k_rate(float *out, int iters, unsigned long long *clk)
{
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = r; a1[r] = -r; a2[r] = 1; a3[r] = 2; }
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.5f + threadIdx.x * 1e-3f); y[e] = (__bf16)(0.25f); }
    const float xf = threadIdx.x * 1e-3f, yf = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, yf, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(yf, xf, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, yf, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(yf, xf, a3, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a3, 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

int main()
{
    const int K = 256;
    std::vector<float> A(32 * K), B(K * 32), C(32 * 32);
    srand(1);
    for (auto &v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
    for (auto &v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.4f;
    std::vector<double> R(32 * 32, 0.0);
    double rmax = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 32 + j]; R[i * 32 + j] = s; rmax = fmax(rmax, fabs(s)); }
    float *dA, *dB, *dC; unsigned long long *clk;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&clk, 8);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    const char *names[4] = {"", "f32 MFMA chain (32x32x2)", "bf16 x 3: hi hi + hi lo + lo hi (32x32x16)", "bf16 x 1 (rounded operands)"};
    for (int mode = 1; mode <= 3; ++mode) {
        if (mode == 1) hipLaunchKernelGGL(k_acc<1>, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
        if (mode == 2) hipLaunchKernelGGL(k_acc<2>, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
        if (mode == 3) hipLaunchKernelGGL(k_acc<3>, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0;
        for (int q = 0; q < 32 * 32; ++q) emax = fmax(emax, fabs((double)C[q] - R[q]));
        printf("%-46s max |error| = %.3e = %.2e of the largest |C| (K = %d)\n", names[mode], emax, emax / rmax, K);
    }
    float *out; hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(1024), dim3(256), 0, 0, out, iters, clk);
            else hipLaunchKernelGGL(k_rate<1>, dim3(1024), dim3(256), 0, 0, out, iters, clk);
            hipDeviceSynchronize();
        }
        unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("%s: %.1f shader cycles per MFMA and wave (one wave per SIMD; 4 independent accumulators)\n", mode == 0 ? "v_mfma_f32_32x32x2_f32 " : "v_mfma_f32_32x32x16_bf16", (double)c / (iters * 64.0));
    }
    return 0;
}
