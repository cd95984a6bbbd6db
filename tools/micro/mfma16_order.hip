// Does v_mfma_f32_16x16x4_f32 accumulate its four k-steps in ascending order with one rounding each, like a chain of fmaf?
// (v_mfma_f32_32x32x2_f32 does: that is what makes the neural kernels bit-exact against the oracle's fmaf chains.)
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/micro/mfma16_order.hip -o gpurun_out/mfma16_order   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k16(const float *A, const float *B, const float *bias, int K, float *C)   // A [16][K], B [K][16], C [16][16]
{
    const int l = threadIdx.x;
    f32x4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = bias[l % 16];
    for (int c = 0; c < K / 4; ++c) {
        const float a = A[(l % 16) * K + 4 * c + l / 16];
        const float b = B[(4 * c + l / 16) * 16 + l % 16];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) C[(4 * (l / 16) + i) * 16 + l % 16] = acc[i];
}

__global__ void k32(const float *A, const float *B, const float *bias, int K, float *C)   // A [32][K], B [K][32], C [32][32]
{
    const int l = threadIdx.x;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = bias[l % 32];
    for (int c = 0; c < K / 2; ++c) {
        const float a = A[(l % 32) * K + 2 * c + l / 32];
        const float b = B[(2 * c + l / 32) * 32 + l % 32];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) C[(8 * (i / 4) + 4 * (l / 32) + (i % 4)) * 32 + l % 32] = acc[i];
}

int main()
{
    const int K = 152;
    for (int which = 0; which < 2; ++which) {
        const int M = which ? 32 : 16;
        std::vector<float> A(M * K), B(K * M), bias(M), C(M * M);
        srand(7 + which);
        for (auto &x : A) x = (float)rand() / RAND_MAX * 2.0f - 1.0f;
        for (auto &x : B) x = ((float)rand() / RAND_MAX * 2.0f - 1.0f) * 0.3f;
        for (auto &x : bias) x = (float)rand() / RAND_MAX - 0.5f;
        float *dA, *dB, *db, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&db, bias.size() * 4); hipMalloc(&dC, C.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice);
        if (which) hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, db, K, dC);
        else hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, db, K, dC);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        int bad_seq = 0, bad_pair = 0;
        for (int r = 0; r < M; ++r)
            for (int c = 0; c < M; ++c) {
                float s = bias[c];
                for (int k = 0; k < K; ++k) s = fmaf(A[r * K + k], B[k * M + c], s);
                if (s != C[r * M + c]) ++bad_seq;
                // alternative: the products of one instruction summed first, then added
                float s2 = bias[c];
                const int step = which ? 2 : 4;
                for (int k = 0; k < K; k += step) { float t = 0.0f; for (int j = 0; j < step; ++j) t = fmaf(A[r * K + k + j], B[(k + j) * M + c], t); s2 += t; }
                if (s2 != C[r * M + c]) ++bad_pair;
            }
        printf("%s: %d of %d elements differ from the k-ordered fmaf chain (%d from the grouped-sum alternative)\n",
               which ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", bad_seq, M * M, bad_pair);
    }
    return 0;
}
