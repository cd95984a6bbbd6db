// pdp_train.hip -- training path of the neural PDP solver: forward-with-saved-activations and backward of
// the building blocks of NeuralMessagePasser / MessageAggregator (reference: src/pdp/nn/pdp_propagate.py:47-95, src/pdp/nn/util.py:51-77),
// NeuralDecimator's GRU cells (src/pdp/nn/pdp_decimate.py:51-87, torch.nn.GRUCell), NeuralPredictor + Perceptron
// (src/pdp/nn/pdp_predict.py:49-91, src/pdp/trainer.py:20-29) and the energy loss (src/pdp/nn/util.py:178-197), i.e. what
// loss.backward() in FactorGraphTrainerBase._train_batch (src/pdp/factorgraph/base.py:149-182) runs through.
//
// The reference gets its backward pass from torch autograd over sparse-mm / addmm / gru_cell.  Here every differentiable step has a
// forward and a backward entry point; pdp/nn/train_ops.py wraps each pair in a torch.autograd.Function, so the graph bookkeeping
// (which tensors are alive, gradient accumulation into the nn.Parameters, clip_grad_norm_, the optimizer the caller hands to train())
// stays with PyTorch while all arithmetic of the path is native:
//   * dense layers on the fp32 matrix cores: one generic tiled GEMM kernel (v_mfma_f32_32x32x2_f32, 64 x 64 tiles through LDS, any
//     transposition, split-K with a deterministic second pass for the weight gradients, whose reduction runs over all edges)
//   * per-row sums over the by-variable / by-clause CSR of the problem (ordered, no atomics) and their adjoints
//   * the GRU gate arithmetic and its derivative, activation derivatives from the saved OUTPUTS (logsigmoid' = 1 - exp(y), sigmoid' =
//     y (1 - y), tanh' = 1 - y^2, relu' = [y > 0]), the loss gradient with respect to the prediction.
// Inference keeps its fused, bit-exact kernels (pdp_neural.hip); training results are compared with the reference's autograd within a
// tolerance (another summation order than MKL's sgemm), tests/test_train_gpu.py.
#include "pdp_common.hpp"

#define ST(s) ((hipStream_t)(s))
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { TACT_NONE = 0, TACT_LOGSIGMOID = 1, TACT_RELU = 2, TACT_SIGMOID = 3, TACT_TANH = 4, TACT_ACCUMULATE = 0x100 /* k_gemm: C += ... */ };

__device__ __forceinline__ float tact(float v, int act)
{
    switch (act) {
    case TACT_LOGSIGMOID: return pdp_logsigmoidf(v);
    case TACT_RELU: return v > 0.0f ? v : ((v != v) ? v : 0.0f);
    case TACT_SIGMOID: return pdp_sigmoidf(v);
    case TACT_TANH: return pdp_tanhf(v);
    default: return v;
    }
}
// derivative of the activation with respect to its argument, from the activation's OUTPUT y
__device__ __forceinline__ float tact_grad(float y, int act)
{
    switch (act) {
    case TACT_LOGSIGMOID: return 1.0f - pdp_expf(y);            // d/dz log sigma(z) = 1 - sigma(z),  sigma(z) = exp(y)
    case TACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
    case TACT_SIGMOID: return y * (1.0f - y);
    case TACT_TANH: return 1.0f - y * y;
    default: return 1.0f;
    }
}

// ---- generic GEMM on the fp32 matrix cores ---------------------------------------------------------------------------------------------
// C[M,N] = sum_k A(m,k) B(k,n);  A is stored [M,K] (TA = false, leading dimension lda) or [K,M] (TA = true);  B is stored [K,N]
// (TB = false) or [N,K] (TB = true).  64 x 64 output tile per 256-thread workgroup, four waves with one 32 x 32 accumulator block each,
// K in slabs of 32 through LDS.  gridDim.z > 1: split-K, slab range z of the K dimension goes to partial[z] (summed by k_splitk_reduce).
// Epilogue (gridDim.z == 1): + bias[n], activation, optional element-wise factor (dropout mask, or the incoming gradient's act').
#define GK 32
#ifndef GEMM_MINB
#define GEMM_MINB 6                                        // 80 registers per lane: six workgroups per CU
#endif
// One operand tile of a workgroup: BX values of x (rows or columns of the output) by the workgroup's k range, read through a buffer
// descriptor based at the tile.  The per-thread offset is computed once; a slab and the thread's j-th element of it advance SCALAR offsets;
// rows past the range are clipped by the descriptor (loads return 0).  KROW: element (x, k) is stored at P[k ld + x] (k is the row), else
// at P[x ld + k].  (The first form of this kernel recomputed 64-bit addresses and bounds for every element of every slab.)
template <bool KROW, int BX>
struct GemmTile {
    static constexpr int PER = BX * GK / 256;              // elements per thread and slab
    __amdgpu_buffer_rsrc_t rs;
    int voff, kk0;
    __device__ __forceinline__ void init(const float *P, int64_t ld, int x0, int X, int64_t kbase, int64_t kend, int tid)
    {
        const int xs = X - x0 < BX ? X - x0 : BX;          // valid values of x in this tile (>= 1)
        const float *base = KROW ? P + kbase * ld + x0 : P + (int64_t)x0 * ld + kbase;
        // bytes the tile may touch: k-row layout (kend - kbase) rows of the matrix, the last one up to xs; k-minor layout xs rows, the last
        // one up to the k range (a row x >= xs starts at xs ld >= (xs - 1) ld + range because range <= K <= ld: clipped)
        const int64_t floats = KROW ? (kend - kbase - 1) * ld + xs : (int64_t)(xs - 1) * ld + (kend - kbase);
        rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, kend > kbase ? (int)(floats * (int64_t)sizeof(float)) : 0, 0x00020000);
        // element j of thread tid: t = tid + 256 j;  k-row: x = t % BX (the same for every j), k = t / BX;  k-minor: k = t % GK (the same), x = t / GK
        if (KROW) { const int x = tid % BX; kk0 = tid / BX; voff = x < xs ? (int)((kk0 * ld + x) * (int64_t)sizeof(float)) : (int)0x80000000; }     // x past the matrix: 2 GB + (scalar offsets < 2 GB) is past every tile
        else { kk0 = tid % GK; voff = (int)(((tid / GK) * ld + kk0) * (int64_t)sizeof(float)); }
    }
    // slab at krel (relative to kbase); left = valid k values from krel on
    __device__ __forceinline__ void fetch(float (&r)[PER], int64_t krel, int64_t ld, int64_t left) const
    {
        const int so = (int)((KROW ? krel * ld : krel) * (int64_t)sizeof(float));
        const int step = (int)((KROW ? (256 / BX) * ld : (256 / GK) * ld) * (int64_t)sizeof(float));      // from element j to j + 1
        // k past the range is masked explicitly in both layouts (k-minor: it is the next row's start; k-row: the slab offset travels in the
        // scalar offset, which the descriptor's range check need not cover).  Only the LAST slab of a range can hold such a k, and `left`
        // is uniform: full slabs take the loop without the per-element test (with the test in every slab the two k-row GEMMs -- dX and
        // the split-K dW -- ran 2.5-3.5 x slower: 113 -> 200 ms per training step)
        if (left >= GK) {
#pragma unroll
            for (int j = 0; j < PER; ++j) r[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, so + j * step, 0));
        } else {
            const int lim = (int)(left < GK ? left : GK);
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, so + j * step, 0));
                r[j] = (kk0 + (KROW ? j * (256 / BX) : 0) < lim) ? v : 0.0f;
            }
        }
    }
};

// A 256-thread workgroup (2 x 2 waves) computes a BM x BN output tile, every wave (BM / 64) x (BN / 64) blocks of 32 x 32 (the host side
// instantiates 64 x 64 only, see gemm()).
template <bool TA, bool TB, int BM, int BN>
__global__ void __launch_bounds__(256, GEMM_MINB) k_gemm(int M, int N, int64_t K, const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
                                              float *__restrict__ C, int64_t ldc, const float *__restrict__ bias, int act, float *__restrict__ partial)
{
    constexpr int RM = BM / 64, RN = BN / 64;
    __shared__ float As[BM][GK + 1];
    __shared__ float Bs[GK][BN + 1];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wm = (w & 1) * (BM / 2), wn = (w >> 1) * (BN / 2);      // this wave's corner inside the tile
    const int i = l & 31, kh = l >> 5;
    bool live[RM][RN];                                     // wave-uniform: the block has a row < M and a column < N
    bool any = false;
    f32x16 acc[RM][RN];
#pragma unroll
    for (int bm = 0; bm < RM; ++bm)
#pragma unroll
        for (int bn = 0; bn < RN; ++bn) {
            live[bm][bn] = m0 + wm + 32 * bm < M && n0 + wn + 32 * bn < N;
            any = any || live[bm][bn];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bm][bn][r] = 0.0f;
        }
    const int64_t slabs = (K + GK - 1) / GK;
    const int64_t per = (slabs + gridDim.z - 1) / gridDim.z;
    const int64_t s_begin = per * blockIdx.z, s_end = (s_begin + per < slabs) ? s_begin + per : slabs;
    const int64_t kbase = s_begin * GK, kend = s_end * GK < K ? s_end * GK : K;
    GemmTile<TA, BM> ta;
    GemmTile<!TB, BN> tb;
    ta.init(A, lda, m0, M, kbase, kend, tid);
    tb.init(B, ldb, n0, N, kbase, kend, tid);
    // the next slab's operand elements are fetched into registers while the MFMAs of the current one run (the k order of the
    // accumulation -- one ascending chain per output element -- does not depend on the slab or tile size)
    float ra[GemmTile<TA, BM>::PER], rb[GemmTile<!TB, BN>::PER];
    auto fetch = [&](int64_t sl) {
        const int64_t krel = (sl - s_begin) * GK, left = kend - kbase - krel;
        ta.fetch(ra, krel, lda, left);
        tb.fetch(rb, krel, ldb, left);
    };
    auto deposit = [&]() {
#pragma unroll
        for (int j = 0; j < GemmTile<TA, BM>::PER; ++j) {
            const int t = tid + 256 * j;
            if (TA) As[t % BM][t / BM] = ra[j]; else As[t / GK][t % GK] = ra[j];
        }
#pragma unroll
        for (int j = 0; j < GemmTile<!TB, BN>::PER; ++j) {
            const int t = tid + 256 * j;
            if (TB) Bs[t % GK][t / GK] = rb[j]; else Bs[t / BN][t % BN] = rb[j];
        }
    };
    if (s_begin < s_end) { fetch(s_begin); deposit(); }
    __syncthreads();
    for (int64_t sl = s_begin; sl < s_end; ++sl) {
        const bool more = sl + 1 < s_end;
        if (more) fetch(sl + 1);
        // blocks outside [M, N] (the 129th column of a [.., 128 + sign] operand opens a block of its own) and k-steps past K (zeros in LDS)
        // issue nothing: they would only add exact zeros
        if (any) {
            const int64_t left = K - sl * GK;
            const int steps = left >= GK ? GK / 2 : (int)((left + 1) >> 1);
            auto kstep = [&](int s2) {
                float av[RM], bv[RN];
#pragma unroll
                for (int bm = 0; bm < RM; ++bm) av[bm] = As[wm + 32 * bm + i][2 * s2 + kh];
#pragma unroll
                for (int bn = 0; bn < RN; ++bn) bv[bn] = Bs[2 * s2 + kh][wn + 32 * bn + i];
#pragma unroll
                for (int bm = 0; bm < RM; ++bm)
#pragma unroll
                    for (int bn = 0; bn < RN; ++bn)
                        if (live[bm][bn]) acc[bm][bn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[bm], bv[bn], acc[bm][bn], 0, 0, 0);
            };
            if (steps == GK / 2) {
#pragma unroll
                for (int s2 = 0; s2 < GK / 2; ++s2) kstep(s2);
            } else {
                for (int s2 = 0; s2 < steps; ++s2) kstep(s2);
            }
        }
        __syncthreads();
        if (more) deposit();
        __syncthreads();
    }
#pragma unroll
    for (int bm = 0; bm < RM; ++bm)
#pragma unroll
        for (int bn = 0; bn < RN; ++bn) {
            const int col = n0 + wn + 32 * bn + i;
            if (!live[bm][bn] || col >= N) continue;
            const float bc = (gridDim.z == 1 && bias) ? bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * bm + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < M) {
                    if (gridDim.z > 1) partial[((int64_t)blockIdx.z * M + row) * N + col] = acc[bm][bn][r];
                    else {
                        float *cp = C + (int64_t)row * ldc + col;
                        const float v = tact(acc[bm][bn][r] + bc, act & 0xff);
                        *cp = (act & TACT_ACCUMULATE) ? *cp + v : v;
                    }
                }
            }
        }
}

// ---- row-stripe GEMM for the products whose M is the number of rows of the batch (millions) and whose N, K are layer widths -------------
// C[M,N] = act(A[M,K] B + bias);  TB: B(k,n) = W[n ldw + k] (the layer's forward, W = weight [N,K]);  else B(k,n) = W[k ldw + n] (dX = dZ W).
// The B operand of a chunk of NB x 32 output columns lives in LDS for the whole launch (Bs[n][k], k contiguous, row pitch KP with KP / 4 odd:
// the ds_read_b128 of a lane group falls on 16 different 16-byte slots); workgroups are persistent and walk stripes of WAVES x 32 rows, wave w
// owning rows 32 w .. 32 w + 31 of the stripe: its A operand is used by no other wave, so it goes from global memory straight to the
// registers the MFMAs read -- no LDS round trip, no barrier in the loop.  A lane (row i, half kh) loads the float4 A[i][8 g + 4 kh .. + 3] of
// group g (8 k values); the four v_mfma_f32_32x32x2 steps of a group take k = 8 g + s from the lanes of half 0 and k = 8 g + 4 + s from half 1,
// for A and B alike (the order of the k terms inside a group is a permutation of the ascending one: training is compared within a
// tolerance).  Four groups (one 128-byte line per row) are in flight while the previous four are multiplied.  K past the matrix: zeros in Bs,
// and the last, partial group of A is loaded element by element.
struct RowsExtra { const float *xs; const float *ws; int64_t ws_ld; };        // rank-one term of the row-stripe GEMM: C += xs[row] * ws[col * ws_ld]
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// ACT: the epilogue's activation as a compile-time constant (one inlined copy of one function); -1: the run-time switch
template <bool TB, int NB, int ACT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 4) k_gemm_rows(int M, int N, int K, const float *__restrict__ A, int64_t lda, const float *__restrict__ W, int64_t ldw,
                                                   float *__restrict__ C, int64_t ldc, const float *__restrict__ bias, int act_rt, int KP, RowsExtra ex)
{
    extern __shared__ float Bs[];                          // [NB * 32][KP], then the chunk's bias [NB * 32] and rank-one column vector [NB * 32]
    constexpr int NT = 64 * WAVES, ROWS = 32 * WAVES;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, i = l & 31, kh = l >> 5;
    const int n0 = blockIdx.y * NB * 32;
    if (TB) {
        for (int idx = tid; idx < NB * 32 * KP; idx += NT) {
            const int n = idx / KP, k = idx - n * KP;
            Bs[idx] = (n0 + n < N && k < K) ? W[(int64_t)(n0 + n) * ldw + k] : 0.0f;
        }
    } else {
        for (int idx = tid; idx < NB * 32 * KP; idx += NT) {
            const int k = idx / (NB * 32), n = idx - k * (NB * 32);
            Bs[n * KP + k] = (n0 + n < N && k < K) ? W[(int64_t)k * ldw + n0 + n] : 0.0f;
        }
    }
    float *biasS = Bs + NB * 32 * KP;
    float *wsS = biasS + NB * 32;                            // ex: C += xs[row] * ws[col] (the operand's separate last column, pdp_train_linear_s)
    if (tid < NB * 32) {
        biasS[tid] = (bias && n0 + tid < N) ? bias[n0 + tid] : 0.0f;
        wsS[tid] = (ex.ws && n0 + tid < N) ? ex.ws[(int64_t)(n0 + tid) * ex.ws_ld] : 0.0f;
    }
    __syncthreads();
    const int G = (K + 7) >> 3;                            // groups of 8 k
    const int SSF = (K >> 3) >> 2;                         // super-slabs of four whole groups
    const int TG = G - 4 * SSF;                            // 0 .. 4 groups behind them (the last one may be partial): the tail slab, index SSF
    const int SS = SSF + (TG > 0 ? 1 : 0);
    const int stripes = (M + ROWS - 1) / ROWS;
    const float *brow = Bs + (size_t)i * KP + 4 * kh;       // + 32 nb KP + 8 g
    // Every slab is fetched by the same four 16-byte buffer loads (the number of loads in flight never depends on the data, so the counter the
    // compiler waits on is exact): descriptor based at the stripe's first row and ending with the matrix (rows past M and bytes past the
    // matrix read 0), per-lane offset = the lane's row and half, scalar offset = the slab.  What a tail slab reads past K is the start of the
    // next row: masked to 0 below before it meets the (zero) B entries -- a NaN there must not reach this row.
    const int voff = (int)((((int64_t)(32 * w + i)) * lda + 4 * kh) * (int64_t)sizeof(float));
    auto load = [&](f32x4 (&r)[4], int st, int ss) {
        const int64_t row0 = (int64_t)st * ROWS;
        int64_t bytes = ((int64_t)(M - 1 - row0) * lda + K) * (int64_t)sizeof(float);
        bytes = bytes < 0x7fffffff ? bytes : 0x7fffffff;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(A + row0 * lda), 0, (int)bytes, 0x00020000);
        // the slab offset travels as the scalar offset; the tail slab of the matrix's last row reads past the matrix and is clipped by the
        // descriptor: on gfx950 the range check covers voffset + soffset + immediate byte by byte (tools/micro/buffer_soffset_check.hip --
        // LLVM documents the scalar offset as outside the check on other targets; this library is built for gfx950 only)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)((32 * ss + 8 * j) * sizeof(float)), 0));
    };
    // bit (4 j + c): element c of group j of the tail slab lies inside K
    uint32_t tail_ok = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) if (32 * SSF + 8 * j + 4 * kh + c < K) tail_ok |= 1u << (4 * j + c);
#ifdef GEMM_ROWS_PRIO
    { const int pr = ((w >> 2) & 1) | ((blockIdx.x & 1) << 1);         // (wave-uniform)
      if (pr == 0) __builtin_amdgcn_s_setprio(0); else if (pr == 1) __builtin_amdgcn_s_setprio(1); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
#endif
    f32x4 cur[4], nxt[4];
    f32x16 acc[NB];
    // The MFMAs are fed (B, A): a block is C^T, lane (row i, half kh) holds columns 8 q + 4 kh + c (q = r >> 2, c = r & 3) of its row -- four
    // consecutive floats per register quad, stored as 16 bytes.  (Fed (A, B) a lane holds one column of 16 rows: 64 dword stores per block
    // row instead of 16 float4 ones made the kernel 25-40 % slower.  Staging the block through LDS for 128-byte row pieces was slower again.)
    auto epilogue = [&](int st) {
        const int row = st * ROWS + 32 * w + i;
        if (row < M) {
            const float xs = ex.xs ? ex.xs[row] : 0.0f;
            float *crow = C + (int64_t)row * ldc + n0 + 4 * kh;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = n0 + 32 * nb + 8 * q + 4 * kh;
                    if (col >= N) continue;
                    const f32x4 bq = *(const f32x4 *)(biasS + 32 * nb + 8 * q + 4 * kh), wq = *(const f32x4 *)(wsS + 32 * nb + 8 * q + 4 * kh);
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = tact(fmaf(xs, wq[c], acc[nb][4 * q + c]) + bq[c], ACT >= 0 ? ACT : act_rt);
                    float *dst = crow + 32 * nb + 8 * q;
#ifdef GEMM_ROWS_NOSTORE
                    asm volatile("" :: "v"(v));
                    if (act_rt != 77) continue;
#endif
                    if (col + 3 < N) *(f32x4u *)dst = v;
                    else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (col + c < N) dst[c] = v[c];
                    }
                }
        }
    };
    auto group = [&](const f32x4 &a, int g) {
        f32x4 bv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = *(const f32x4 *)(brow + (size_t)32 * nb * KP + 8 * g);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[nb][c], a[c], acc[nb], 0, 0, 0);
    };
    int st = blockIdx.x;
    if (st < stripes) load(cur, st, 0);
    for (; st < stripes; st += gridDim.x) {
        // the slab after (st, ss): the next one of this stripe, or the first one of the workgroup's next stripe (none left: this one again)
        auto load_next = [&](int ss) {
            const bool last = ss + 1 == SS;
            const int st2 = !last ? st : (st + (int)gridDim.x < stripes ? st + (int)gridDim.x : st);
            load(nxt, st2, last ? 0 : ss + 1);
        };
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
        for (int ss = 0; ss < SSF; ++ss) {
            load_next(ss);
            __builtin_amdgcn_sched_barrier(0);             // the loads are issued here, 64 NB / 4 MFMAs ahead of their use -- not sunk to it
#pragma unroll
            for (int j = 0; j < 4; ++j) group(cur[j], 4 * ss + j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
        }
        if (TG > 0) {
            load_next(SSF);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) cur[j][c] = (tail_ok >> (4 * j + c) & 1u) ? cur[j][c] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < TG) group(cur[j], 4 * SSF + j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
        }
        epilogue(st);
    }
}

// C[i] = sum over the slices z of partial[z][i].  A workgroup takes 32 output elements; its 256 threads are 8 groups, group g adds the
// slices z = g, g + 8, ... in ascending order (four loads in flight), and the eight group sums are folded in a fixed order through LDS: the
// order of the additions depends on (splits) only.  (One thread per element walking all slices was latency-bound: 0.24 ms for 1 024 slices
// of a 100 x 129 gradient.)
#define RED_G 8
#define RED_E 32
__global__ void __launch_bounds__(RED_G * RED_E) k_splitk_reduce(int64_t MN, int splits, const float *__restrict__ partial, float *__restrict__ C)
{
    __shared__ float red[RED_G * RED_E];
    const int e = threadIdx.x % RED_E, g = threadIdx.x / RED_E;
    const int64_t i = (int64_t)blockIdx.x * RED_E + e;
    float acc = 0.0f;
    if (i < MN) {
        const float *p = partial + i;
        int z = g;
        for (; z + 3 * RED_G < splits; z += 4 * RED_G) {
            const float a0 = p[(int64_t)z * MN], a1 = p[(int64_t)(z + RED_G) * MN], a2 = p[(int64_t)(z + 2 * RED_G) * MN], a3 = p[(int64_t)(z + 3 * RED_G) * MN];
            acc = (((acc + a0) + a1) + a2) + a3;
        }
        for (; z < splits; z += RED_G) acc = acc + p[(int64_t)z * MN];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && i < MN) {
        for (int j = 1; j < RED_G; ++j) acc = acc + red[j * RED_E + e];
        C[i] = acc;
    }
}
static int reduce_grid(int64_t MN) { return (int)((MN + RED_E - 1) / RED_E); }

// column sums of Z [R,N] in two deterministic passes: partial[s][n] over a slice of rows, then their sum (k_splitk_reduce).  A workgroup takes a
// slice; its 256 threads are 256 / C row groups x C columns (C = min(N, 256) per grid column), every thread walks its rows in ascending
// order and the row groups are folded in a fixed order through LDS -- the order of the additions depends on (R, N, slices) only.
#define COLSUM_SLICES 1024
__global__ void __launch_bounds__(256) k_colsum_partial(int64_t R, int N, const float *__restrict__ Z, int slices, float *__restrict__ partial)
{
    __shared__ float red[256];
    const int C = N < 256 ? N : 256;                       // columns of this grid column
    const int groups = 256 / C, g = threadIdx.x / C, c = threadIdx.x % C;
    const int n = blockIdx.x * 256 + c;
    const int64_t per = (R + slices - 1) / slices, r0 = per * blockIdx.y, r1 = (r0 + per < R) ? r0 + per : R;
    float acc = 0.0f;
    if (g < groups && n < N)
#pragma unroll 4
        for (int64_t r = r0 + g; r < r1; r += groups) acc = acc + Z[r * N + n];        // (unrolled: the loads run ahead, the additions keep their order)
    red[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && n < N) {
        for (int j = 1; j < groups; ++j) acc = acc + red[j * C + c];
        partial[(int64_t)blockIdx.y * N + n] = acc;
    }
}
static int colsum_slices(int64_t R) { int64_t s = (R + 255) / 256; if (s > COLSUM_SLICES) s = COLSUM_SLICES; return (int)(s < 1 ? 1 : s); }

// dZ = dY * act'(Y) (element-wise), in place into out
__global__ void k_act_backward(int64_t n, const float *__restrict__ dY, const float *__restrict__ Y, int act, float *__restrict__ out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = dY[i] * tact_grad(Y[i], act);
}

// dZ = dY * act'(Y) AND the column sums of dZ (the bias gradient) in the same pass over the rows: the thread layout of k_colsum_partial
// (256 / C row groups x C columns, every thread walks its rows of a slice in ascending order), partial[slice][n] summed by k_splitk_reduce.
// dZ may alias dY (act none: nothing is written then).
// xs (may be NULL): a second set of sums weighted by xs[row] -- the gradient of the weight column that multiplies the operand's separate
// last column (pdp_train_linear_s): partial2[slice][n].
__global__ void __launch_bounds__(256) k_act_backward_colsum(int64_t R, int N, const float *__restrict__ dY, const float *__restrict__ Y, int act, float *__restrict__ dZ,
                                                             int slices, float *__restrict__ partial, const float *__restrict__ xs, float *__restrict__ partial2)
{
    __shared__ float red[256], red2[256];
    const int C = N < 256 ? N : 256;
    const int groups = 256 / C, g = threadIdx.x / C, c = threadIdx.x % C;
    const int n = blockIdx.x * 256 + c;
    const int64_t per = (R + slices - 1) / slices, r0 = per * blockIdx.y, r1 = (r0 + per < R) ? r0 + per : R;
    float acc = 0.0f, acc2 = 0.0f;
    if (g < groups && n < N) {
        if (act == TACT_NONE && dZ == dY) {
#pragma unroll 4
            for (int64_t r = r0 + g; r < r1; r += groups) { const float v = dY[r * N + n]; acc = acc + v; if (xs) acc2 = fmaf(v, xs[r], acc2); }
        } else {
#pragma unroll 4
            for (int64_t r = r0 + g; r < r1; r += groups) {
                const float v = dY[r * N + n] * tact_grad(Y[r * N + n], act);
                dZ[r * N + n] = v;
                acc = acc + v;
                if (xs) acc2 = fmaf(v, xs[r], acc2);
            }
        }
    }
    red[threadIdx.x] = acc; red2[threadIdx.x] = acc2;
    __syncthreads();
    if (g == 0 && n < N) {
        for (int j = 1; j < groups; ++j) { acc = acc + red[j * C + c]; acc2 = acc2 + red2[j * C + c]; }
        partial[(int64_t)blockIdx.y * N + n] = acc;
        if (xs) partial2[(int64_t)blockIdx.y * N + n] = acc2;
    }
}
// dW [N, K + 1] = [ dWx [N, K] | dws [N] ]
__global__ void k_assemble_dw(int N, int K, const float *__restrict__ dWx, const float *__restrict__ dws, float *__restrict__ dW)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)N * (K + 1); i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / (K + 1)), k = (int)(i % (K + 1));
        dW[i] = k < K ? dWx[(int64_t)n * K + k] : dws[n];
    }
}
static int act_colsum_slices(int64_t R)
{
    static const int cap = [] { const char *e = getenv("PDP_COLSUM_SLICES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 2048; }();
    int64_t s = (R + 63) / 64; if (s > cap) s = cap; return (int)(s < 1 ? 1 : s);
}

static int grid1d(int64_t n) { int64_t g = (n + 255) / 256; if (g > 16384) g = 16384; return (int)(g < 1 ? 1 : g); }

template <bool TA, bool TB, int BM, int BN>
static int gemm_tiled(int M, int N, int64_t K, const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, const float *bias, int act,
                      int splits, float *partial, hipStream_t st)
{
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, splits > 1 ? splits : 1);
    {   // a workgroup addresses its operand tiles with 32-bit byte offsets from the tile's base
        const int64_t z = grid.z, slabs = (K + GK - 1) / GK, krange = ((slabs + z - 1) / z) * GK;
        const int64_t ea = TA ? krange * lda : (int64_t)BM * lda + krange, eb = TB ? (int64_t)BN * ldb + krange : krange * ldb;
        PDP_REQUIRE(ea < ((int64_t)1 << 29) && eb < ((int64_t)1 << 29), "GEMM operand tile past 2 GB: use more split-K slices");
    }
    hipLaunchKernelGGL((k_gemm<TA, TB, BM, BN>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias, act, partial);
    if (splits > 1) hipLaunchKernelGGL(k_splitk_reduce, dim3(reduce_grid((int64_t)M * N)), dim3(RED_G * RED_E), 0, st, (int64_t)M * N, splits, (const float *)partial, C);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}
// Tile shape: 64 x 64 for every call.  Measured on the 1 M-edge training step (tools/train_time.py, hidden 128): 128 x 128 tiles (half the
// operand traffic per flop, but 220 registers per lane = two workgroups per CU) take 140 ms per step against 123 ms -- the kernel is bound by
// how many workgroups overlap their load -> LDS -> barrier chains, not by operand bandwidth.
template <bool TA, bool TB>
static int gemm(int M, int N, int64_t K, const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, const float *bias, int act,
                int splits, float *partial, hipStream_t st)
{
    return gemm_tiled<TA, TB, 64, 64>(M, N, K, A, lda, B, ldb, C, ldc, bias, act, splits, partial, st);
}

// split-K of the weight gradients (K = rows of the batch, the output is a few 64 x 64 tiles): slices of >= 1024 rows, at most 1024 of them --
// the slices are what fills the chip (at 128 slices of 8 192 rows a CU held two or three workgroups, each waiting on its own loads)
static int pick_splits(int64_t R) { int64_t s = (R + 1023) / 1024; if (s > 1024) s = 1024; return (int)(s < 1 ? 1 : s); }

// scratch for split-K partials and column sums: one block per (device, stream), grown on demand.  Kernels enqueued earlier on that stream may
// still read the old block when it has to grow, so the stream is drained before the block goes back to the pool; work on another stream or
// device never sees this block.
#include <map>
#include <mutex>
#include <utility>
namespace {
struct TrainScratch { float *ptr; size_t floats; };
std::mutex g_train_mu;
std::map<std::pair<int, hipStream_t>, TrainScratch> g_train_scratch;
}
static float *train_scratch(size_t floats, hipStream_t st)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_train_mu);
    TrainScratch &sc = g_train_scratch[std::make_pair(dev, st)];
    if (sc.floats < floats) {
        if (sc.ptr) {
            if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
            pdp_dev_free(sc.ptr);
        }
        sc.ptr = nullptr; sc.floats = 0;
        if (pdp_dev_alloc((void **)&sc.ptr, floats * sizeof(float)) != PDP_OK) return nullptr;
        sc.floats = floats;
    }
    return sc.ptr;
}

// The products over the rows of the batch (forward and dX) on the row-stripe kernel when the B chunk fits the LDS twice per CU; else the tiled one.
#ifndef GEMM_ROWS_WAVES
#define GEMM_ROWS_WAVES 8                 // 512 threads, two workgroups per CU: four waves per SIMD
#endif
static size_t gemm_rows_lds(int nb, int KP) { return ((size_t)nb * 32 * KP + 2 * nb * 32) * sizeof(float); }
template <bool TB, int NB, int ACT>
static int gemm_rows_launch(int M, int N, int K, const float *A, int64_t lda, const float *W, int64_t ldw, float *C, int64_t ldc, const float *bias, int act,
                            int chunks, int KP, hipStream_t st, RowsExtra ex)
{
    const size_t lds = gemm_rows_lds(NB, KP);
    // (set on every launch, like the solver kernels do: the attribute belongs to the function ON THE CURRENT DEVICE, and a process-wide
    //  "already set" flag would leave a second device -- or a second thread racing the first -- with the 64 KB default)
    PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_gemm_rows<TB, NB, ACT, GEMM_ROWS_WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int stripes = (M + 32 * GEMM_ROWS_WAVES - 1) / (32 * GEMM_ROWS_WAVES);
#ifndef GEMM_ROWS_GRID
#define GEMM_ROWS_GRID 3                 // column chunks > 1: three times the resident workgroups (measured 1 / 2 / 3 / 6: 71 / 80 / 91 / 88 TFLOP/s at 129 x 384); one chunk: the resident count
#endif
    static const int grid_mul = [] { const char *e = getenv("PDP_GEMM_ROWS_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : GEMM_ROWS_GRID; }();
    static const bool grid_all = getenv("PDP_GEMM_ROWS_GRID_ALL") != nullptr;
    int gx = ((chunks > 1 || grid_all ? grid_mul : 1) * (16 / GEMM_ROWS_WAVES) * pdp_device_cus() + chunks - 1) / chunks;
    if (gx > stripes) gx = stripes;
    hipLaunchKernelGGL((k_gemm_rows<TB, NB, ACT, GEMM_ROWS_WAVES>), dim3(gx, chunks), dim3(64 * GEMM_ROWS_WAVES), lds, st, M, N, K, A, lda, W, ldw, C, ldc, bias, act, KP, ex);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}
template <bool TB, int NB>
static int gemm_rows_act(int M, int N, int K, const float *A, int64_t lda, const float *W, int64_t ldw, float *C, int64_t ldc, const float *bias, int act,
                         int chunks, int KP, hipStream_t st, RowsExtra ex)
{
    if (act == TACT_NONE) return gemm_rows_launch<TB, NB, TACT_NONE>(M, N, K, A, lda, W, ldw, C, ldc, bias, act, chunks, KP, st, ex);
    if (TB && act == TACT_LOGSIGMOID) return gemm_rows_launch<TB, NB, TB ? TACT_LOGSIGMOID : TACT_NONE>(M, N, K, A, lda, W, ldw, C, ldc, bias, act, chunks, KP, st, ex);
    if (ex.xs) return PDP_ERR_UNSUPPORTED;                                                             // (callers ask gemm_rows_fits first)
    return gemm<false, TB>(M, N, K, A, lda, W, ldw, C, ldc, bias, act, 1, nullptr, st);        // relu / sigmoid / tanh: the predictor's small layers
}
// does the row-stripe kernel take this product (else the tiled one does)?
static bool gemm_rows_fits(int M, int N, int K, int act, bool TB)
{
    static const bool tiled_only = [] { const char *e = getenv("PDP_TRAIN_GEMM"); return e && !strcmp(e, "tiled"); }();
    const int blocks = (N + 31) / 32, chunks = (blocks + 3) / 4, nb = (blocks + chunks - 1) / chunks;
    const int KP = ((K + 7) / 8) * 8 + 4;
    if (tiled_only || M < 4096 || gemm_rows_lds(nb, KP) * (16 / GEMM_ROWS_WAVES) > 160 * 1024) return false;
    return act == TACT_NONE || (TB && act == TACT_LOGSIGMOID);
}
template <bool TB>
static int gemm_rows(int M, int N, int K, const float *A, int64_t lda, const float *W, int64_t ldw, float *C, int64_t ldc, const float *bias, int act, hipStream_t st,
                     RowsExtra ex = RowsExtra{nullptr, nullptr, 0})
{
    const int blocks = (N + 31) / 32, chunks = (blocks + 3) / 4, nb = (blocks + chunks - 1) / chunks;
    const int KP = ((K + 7) / 8) * 8 + 4;
    if (!gemm_rows_fits(M, N, K, act, TB)) {
        if (ex.xs) return PDP_ERR_UNSUPPORTED;
        return gemm<false, TB>(M, N, K, A, lda, W, ldw, C, ldc, bias, act, 1, nullptr, st);
    }
    switch (nb) {
    case 1: return gemm_rows_act<TB, 1>(M, N, K, A, lda, W, ldw, C, ldc, bias, act, chunks, KP, st, ex);
    case 2: return gemm_rows_act<TB, 2>(M, N, K, A, lda, W, ldw, C, ldc, bias, act, chunks, KP, st, ex);
    case 3: return gemm_rows_act<TB, 3>(M, N, K, A, lda, W, ldw, C, ldc, bias, act, chunks, KP, st, ex);
    default: return gemm_rows_act<TB, 4>(M, N, K, A, lda, W, ldw, C, ldc, bias, act, chunks, KP, st, ex);
    }
}

// replaces: nn.Linear + activation as used by MessageAggregator / Perceptron (util.py:56,74; trainer.py:28-29): Y = act(X W^T + b)
extern "C" int pdp_train_linear(const float *X, int64_t R, int K, int64_t ldx, const float *W, const float *b, int N, int act, float *Y, void *stream)
{
    PDP_REQUIRE(X && W && Y && R >= 0 && K > 0 && N > 0 && R < (int64_t)1 << 31, "bad argument");
    if (R == 0) return PDP_OK;
    return gemm_rows<true>((int)R, N, K, X, ldx, W, K, Y, N, b, act, ST(stream));
}

// its adjoint: dZ = dY * act'(Y);  dX = dZ W  (NULL: not needed);  dW = dZ^T X;  db = column sums of dZ (NULL: no bias).
// dZ [R,N] is caller-provided scratch (may alias dY).
extern "C" int pdp_train_linear_backward(const float *dY, const float *Y, const float *X, int64_t R, int K, int64_t ldx, const float *W, int N, int act,
                                         float *dZ, float *dX, int64_t lddx, float *dW, float *db, void *stream)
{
    PDP_REQUIRE(dY && Y && X && W && dZ && dW && R >= 0 && R < (int64_t)1 << 31, "bad argument");
    hipStream_t st = ST(stream);
    if (R == 0) {
        PDP_HIP_CHECK(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)N * K, st));
        if (db) PDP_HIP_CHECK(hipMemsetAsync(db, 0, sizeof(float) * (size_t)N, st));
        return PDP_OK;
    }
    const int splits = pick_splits(R);
    const int cs = act_colsum_slices(R);
    float *scr = train_scratch((size_t)splits * (size_t)N * K + (size_t)cs * N + 16, st);
    if (!scr) return PDP_ERR_HIP;
    if (db) {
        // one pass: dZ and the bias gradient's partial column sums
        float *part = scr + (size_t)splits * N * K;
        hipLaunchKernelGGL(k_act_backward_colsum, dim3((N + 255) / 256, cs), dim3(256), 0, st, R, N, dY, Y, act, dZ, cs, part, (const float *)nullptr, (float *)nullptr);
        hipLaunchKernelGGL(k_splitk_reduce, dim3(reduce_grid(N)), dim3(RED_G * RED_E), 0, st, (int64_t)N, cs, (const float *)part, db);
    } else if (!(act == TACT_NONE && dZ == dY))
        hipLaunchKernelGGL(k_act_backward, dim3(grid1d(R * N)), dim3(256), 0, st, R * N, dY, Y, act, dZ);
    int s;
    if (dX) { s = gemm_rows<false>((int)R, K, N, dZ, N, W, K, dX, lddx, nullptr, TACT_NONE, st); if (s != PDP_OK) return s; }
    s = gemm<true, false>(N, K, R, dZ, N, X, ldx, dW, K, nullptr, TACT_NONE, splits, scr, st);
    if (s != PDP_OK) return s;
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- Linear on an operand whose last column lives apart: Y = act([X | xs] W^T + b), X [R,K], xs [R], W [N, K + 1] -----------------------------
// The training path's layers read [state | edge sign]: the reference concatenates (pdp_propagate.py:66-67, util.py:71-72); here the K-wide
// block goes through the row-stripe GEMM (K = 128: four whole slabs, no tail) and the sign column is a rank-one term of its epilogue.
extern "C" int pdp_train_linear_s_supported(int64_t R, int K, int N, int act)
{
    return R < ((int64_t)1 << 31) && gemm_rows_fits((int)R, N, K, act, true) && gemm_rows_fits((int)R, K, N, TACT_NONE, false) ? 1 : 0;
}
extern "C" int pdp_train_linear_s(const float *X, const float *xs, int64_t R, int K, int64_t ldx, const float *W, const float *b, int N, int act, float *Y, void *stream)
{
    PDP_REQUIRE(X && xs && W && Y && R >= 0 && K > 0 && N > 0, "bad argument");
    PDP_REQUIRE(pdp_train_linear_s_supported(R, K, N, act), "shape outside the row-stripe kernel (ask pdp_train_linear_s_supported; concatenate and use pdp_train_linear)");
    if (R == 0) return PDP_OK;
    return gemm_rows<true>((int)R, N, K, X, ldx, W, K + 1, Y, N, b, act, ST(stream), RowsExtra{xs, W + K, (int64_t)K + 1});
}
// adjoint: dZ = dY * act'(Y);  dX [R,K] = dZ W[:, :K];  dW [N, K + 1] = [ dZ^T X | dZ^T xs ];  db = column sums (NULL: no bias).  No gradient for xs.
extern "C" int pdp_train_linear_s_backward(const float *dY, const float *Y, const float *X, const float *xs, int64_t R, int K, int64_t ldx, const float *W, int N,
                                           int act, float *dZ, float *dX, int64_t lddx, float *dW, float *db, void *stream)
{
    PDP_REQUIRE(dY && Y && X && xs && W && dZ && dW && R >= 0 && R < (int64_t)1 << 31, "bad argument");
    hipStream_t st = ST(stream);
    if (R == 0) {
        PDP_HIP_CHECK(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)N * (K + 1), st));
        if (db) PDP_HIP_CHECK(hipMemsetAsync(db, 0, sizeof(float) * (size_t)N, st));
        return PDP_OK;
    }
    const int splits = pick_splits(R);
    const int cs = act_colsum_slices(R);
    float *scr = train_scratch((size_t)splits * (size_t)N * K + (size_t)N * K + 2 * (size_t)cs * N + 2 * (size_t)N + 16, st);
    if (!scr) return PDP_ERR_HIP;
    float *dWx = scr + (size_t)splits * N * K, *part = dWx + (size_t)N * K, *part2 = part + (size_t)cs * N, *dbs = part2 + (size_t)cs * N, *dws = dbs + N;
    hipLaunchKernelGGL(k_act_backward_colsum, dim3((N + 255) / 256, cs), dim3(256), 0, st, R, N, dY, Y, act, dZ, cs, part, xs, part2);
    hipLaunchKernelGGL(k_splitk_reduce, dim3(reduce_grid(N)), dim3(RED_G * RED_E), 0, st, (int64_t)N, cs, (const float *)part, db ? db : dbs);
    hipLaunchKernelGGL(k_splitk_reduce, dim3(reduce_grid(N)), dim3(RED_G * RED_E), 0, st, (int64_t)N, cs, (const float *)part2, dws);
    int s;
    if (dX) { s = gemm_rows<false>((int)R, K, N, dZ, N, W, K + 1, dX, lddx, nullptr, TACT_NONE, st); if (s != PDP_OK) return s; }
    s = gemm<true, false>(N, K, R, dZ, N, X, ldx, dWx, K, nullptr, TACT_NONE, splits, scr, st);
    if (s != PDP_OK) return s;
    hipLaunchKernelGGL(k_assemble_dw, dim3(grid1d((int64_t)N * (K + 1))), dim3(256), 0, st, N, K, (const float *)dWx, (const float *)dws, dW);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- per-row sums over the problem's CSR and their adjoints (MessageAggregator's torch.mm(mask, state), util.py:60-69) ------------------
__global__ void k_trow_sum(int64_t R, int A, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ row_edges, const float *__restrict__ x,
                           float *__restrict__ out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < R * A; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / A; const int c = (int)(i % A);
        float acc = 0.0f;
        for (int k = row_ptr[r]; k < row_ptr[r + 1]; ++k) acc = acc + x[(int64_t)row_edges[k] * A + c];       // ascending edge id
        out[i] = acc;
    }
}
// out[e] = rows[row(e)] - (x ? x[e] : 0)
__global__ void k_trow_spread(int64_t E, int A, const int32_t *__restrict__ edge_row, const float *__restrict__ rows, const float *__restrict__ x,
                              float *__restrict__ out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < E * A; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / A; const int c = (int)(i % A);
        const float v = rows[(int64_t)edge_row[e] * A + c];
        out[i] = x ? v - x[i] : v;
    }
}

// x [E,A] -> out [rows,A]: the ordered sum over the edges of every variable (by_variable != 0) or clause.  Forward of the
// include_self_message=True aggregator; first half of the exclude-self one and of both adjoints.
extern "C" int pdp_train_row_sum(pdp_problem *p, int by_variable, const float *x, int A, float *out, void *stream)
{
    PDP_REQUIRE(p && x && out && A > 0, "bad argument");
    const int64_t R = by_variable ? p->V : p->F;
    hipLaunchKernelGGL(k_trow_sum, dim3(grid1d(R * A)), dim3(256), 0, ST(stream), R, A, by_variable ? p->nv_ptr : p->nf_ptr,
                       by_variable ? p->nv_edges : p->nf_edges, x, out);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}
// rows [rows,A] -> out [E,A]: out[e] = rows[row(e)] - x[e] (x NULL: plain gather).  With x = the summed edge values this completes the
// exclude-self aggregation (util.py:63-69); with x = the incoming gradient, its adjoint; with x NULL the adjoint of the plain row sum.
extern "C" int pdp_train_row_spread(pdp_problem *p, int by_variable, const float *rows, const float *x, int A, float *out, void *stream)
{
    PDP_REQUIRE(p && rows && out && A > 0, "bad argument");
    const int32_t *edge_row = by_variable ? p->graph_map : p->graph_map + p->E;
    hipLaunchKernelGGL(k_trow_spread, dim3(grid1d((int64_t)p->E * A)), dim3(256), 0, ST(stream), (int64_t)p->E, A, edge_row, rows, x, out);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- GRU cell (torch.nn.GRUCell as used by NeuralDecimator, pdp_decimate.py:38-41,70-83) ---------------------------------------------------
// gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh (gate order r, z, n);  r = sigma(gi_r + gh_r), z = sigma(gi_z + gh_z),
// n = tanh(gi_n + r gh_n),  h' = (1 - z) n + z h.  saved [R,4H] = r | z | n | gh_n.
__global__ void k_gru_point(int64_t R, int H, const float *__restrict__ gi, const float *__restrict__ gh, const float *__restrict__ h,
                            float *__restrict__ hnew, float *__restrict__ saved)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < R * H; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / H; const int c = (int)(i % H);
        const float *a = gi + e * 3 * H, *b = gh + e * 3 * H;
        const float r = pdp_sigmoidf(a[c] + b[c]);
        const float z = pdp_sigmoidf(a[H + c] + b[H + c]);
        const float ghn = b[2 * H + c];
        const float n = pdp_tanhf(a[2 * H + c] + r * ghn);
        hnew[i] = (1.0f - z) * n + z * h[i];
        float *sv = saved + e * 4 * H;
        sv[c] = r; sv[H + c] = z; sv[2 * H + c] = n; sv[3 * H + c] = ghn;
    }
}
// The same thread layout as k_act_backward_colsum: thread (row group g, column c) walks the rows of its slice, so the four column sums the
// bias gradients are made of -- dr, dz, dn (b_ih = dr | dz | dn) and dn r (b_hh = dr | dz | dn r) -- come out of the pass that writes dgi /
// dgh, as partial[slice][4][H].
__global__ void __launch_bounds__(256) k_gru_point_backward(int64_t R, int H, const float *__restrict__ dhn, const float *__restrict__ saved, const float *__restrict__ h,
                                                            float *__restrict__ dgi, float *__restrict__ dgh, float *__restrict__ dh, int slices, float *__restrict__ partial,
                                                            const float *__restrict__ xs /* NULL, or the input's separate last column: + dr xs | dz xs | dn xs */)
{
    __shared__ float red[7][256];
    const int Cw = H < 256 ? H : 256;
    const int groups = 256 / Cw, g = threadIdx.x / Cw, cl = threadIdx.x % Cw;
    const int c = blockIdx.x * 256 + cl;
    const int64_t per = (R + slices - 1) / slices, r0 = per * blockIdx.y, r1 = (r0 + per < R) ? r0 + per : R;
    float s_r = 0.0f, s_z = 0.0f, s_n = 0.0f, s_nr = 0.0f, x_r = 0.0f, x_z = 0.0f, x_n = 0.0f;
    const int NS = xs ? 7 : 4;
    if (g < groups && c < H) {
#pragma unroll 2
        for (int64_t e = r0 + g; e < r1; e += groups) {
            const float *sv = saved + e * 4 * H;
            const float r = sv[c], z = sv[H + c], n = sv[2 * H + c], ghn = sv[3 * H + c];
            const int64_t i = e * H + c;
            const float gd = dhn[i];
            const float dn_pre = gd * (1.0f - z) * (1.0f - n * n);
            const float dz_pre = gd * (h[i] - n) * z * (1.0f - z);
            const float dr_pre = dn_pre * ghn * r * (1.0f - r);
            const float dnr = dn_pre * r;
            float *a = dgi + e * 3 * H, *b = dgh + e * 3 * H;
            a[c] = dr_pre; b[c] = dr_pre;
            a[H + c] = dz_pre; b[H + c] = dz_pre;
            a[2 * H + c] = dn_pre; b[2 * H + c] = dnr;
            dh[i] = gd * z;                              // the direct path; the W_hh path is added by the caller's GEMM (accumulating epilogue)
            s_r = s_r + dr_pre; s_z = s_z + dz_pre; s_n = s_n + dn_pre; s_nr = s_nr + dnr;
            if (xs) { const float xv = xs[e]; x_r = fmaf(dr_pre, xv, x_r); x_z = fmaf(dz_pre, xv, x_z); x_n = fmaf(dn_pre, xv, x_n); }
        }
    }
    red[0][threadIdx.x] = s_r; red[1][threadIdx.x] = s_z; red[2][threadIdx.x] = s_n; red[3][threadIdx.x] = s_nr;
    red[4][threadIdx.x] = x_r; red[5][threadIdx.x] = x_z; red[6][threadIdx.x] = x_n;
    __syncthreads();
    if (g == 0 && c < H) {
        for (int j = 1; j < groups; ++j) {
            s_r = s_r + red[0][j * Cw + cl]; s_z = s_z + red[1][j * Cw + cl]; s_n = s_n + red[2][j * Cw + cl]; s_nr = s_nr + red[3][j * Cw + cl];
            x_r = x_r + red[4][j * Cw + cl]; x_z = x_z + red[5][j * Cw + cl]; x_n = x_n + red[6][j * Cw + cl];
        }
        float *p = partial + (int64_t)blockIdx.y * NS * H;
        p[c] = s_r; p[H + c] = s_z; p[2 * H + c] = s_n; p[3 * H + c] = s_nr;
        if (xs) { p[4 * H + c] = x_r; p[5 * H + c] = x_z; p[6 * H + c] = x_n; }
    }
}

// scratch [R, 6H]: gi | gh
extern "C" int pdp_train_gru(const float *x, const float *h, const float *W_ih, const float *W_hh, const float *b_ih, const float *b_hh, int64_t R, int Kx, int H,
                             float *hnew, float *saved, float *scratch, void *stream)
{
    PDP_REQUIRE(x && h && W_ih && W_hh && b_ih && b_hh && hnew && saved && scratch && R < (int64_t)1 << 31, "bad argument");
    if (R == 0) return PDP_OK;
    hipStream_t st = ST(stream);
    float *gi = scratch, *gh = scratch + (size_t)R * 3 * H;
    int s = gemm_rows<true>((int)R, 3 * H, Kx, x, Kx, W_ih, Kx, gi, 3 * H, b_ih, TACT_NONE, st); if (s != PDP_OK) return s;
    s = gemm_rows<true>((int)R, 3 * H, H, h, H, W_hh, H, gh, 3 * H, b_hh, TACT_NONE, st); if (s != PDP_OK) return s;
    hipLaunchKernelGGL(k_gru_point, dim3(grid1d(R * H)), dim3(256), 0, st, R, H, (const float *)gi, (const float *)gh, h, hnew, saved);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}
// scratch [R, 6H]: dgi | dgh (callers may still hand over [R, 7H])
extern "C" int pdp_train_gru_backward(const float *dhnew, const float *saved, const float *x, const float *h, const float *W_ih, const float *W_hh, int64_t R,
                                      int Kx, int H, float *dx, float *dh, float *dW_ih, float *dW_hh, float *db_ih, float *db_hh, float *scratch, void *stream)
{
    PDP_REQUIRE(dhnew && saved && x && h && W_ih && W_hh && dx && dh && dW_ih && dW_hh && db_ih && db_hh && scratch && R < (int64_t)1 << 31, "bad argument");
    hipStream_t st = ST(stream);
    if (R == 0) {
        PDP_HIP_CHECK(hipMemsetAsync(dW_ih, 0, sizeof(float) * 3 * (size_t)H * Kx, st)); PDP_HIP_CHECK(hipMemsetAsync(dW_hh, 0, sizeof(float) * 3 * (size_t)H * H, st));
        PDP_HIP_CHECK(hipMemsetAsync(db_ih, 0, sizeof(float) * 3 * (size_t)H, st)); PDP_HIP_CHECK(hipMemsetAsync(db_hh, 0, sizeof(float) * 3 * (size_t)H, st));
        return PDP_OK;
    }
    float *dgi = scratch, *dgh = scratch + (size_t)R * 3 * H;
    const int splits = pick_splits(R);
    const size_t wmax = (size_t)3 * H * (Kx > H ? Kx : H);
    const int cs = act_colsum_slices(R);
    float *scr = train_scratch((size_t)splits * wmax + (size_t)cs * 4 * H + 4 * H + 16, st);
    if (!scr) return PDP_ERR_HIP;
    float *part = scr + (size_t)splits * wmax, *sums = part + (size_t)cs * 4 * H;          // sums [4 H] = dr | dz | dn | dn r
    hipLaunchKernelGGL(k_gru_point_backward, dim3((H + 255) / 256, cs), dim3(256), 0, st, R, H, dhnew, saved, h, dgi, dgh, dh, cs, part, (const float *)nullptr);
    hipLaunchKernelGGL(k_splitk_reduce, dim3(reduce_grid(4 * H)), dim3(RED_G * RED_E), 0, st, (int64_t)4 * H, cs, (const float *)part, sums);
    PDP_HIP_CHECK(hipMemcpyAsync(db_ih, sums, sizeof(float) * 3 * (size_t)H, hipMemcpyDeviceToDevice, st));
    PDP_HIP_CHECK(hipMemcpyAsync(db_hh, sums, sizeof(float) * 2 * (size_t)H, hipMemcpyDeviceToDevice, st));
    PDP_HIP_CHECK(hipMemcpyAsync(db_hh + 2 * (size_t)H, sums + 3 * (size_t)H, sizeof(float) * (size_t)H, hipMemcpyDeviceToDevice, st));
    int s = gemm_rows<false>((int)R, Kx, 3 * H, dgi, 3 * H, W_ih, Kx, dx, Kx, nullptr, TACT_NONE, st); if (s != PDP_OK) return s;
    // dh += dgh W_hh: the tiled kernel's accumulating epilogue (dh holds the direct path)
    s = gemm<false, false>((int)R, H, 3 * H, dgh, 3 * H, W_hh, H, dh, H, nullptr, TACT_NONE | TACT_ACCUMULATE, 1, nullptr, st); if (s != PDP_OK) return s;
    s = gemm<true, false>(3 * H, Kx, R, dgi, 3 * H, x, Kx, dW_ih, Kx, nullptr, TACT_NONE, splits, scr, st); if (s != PDP_OK) return s;
    s = gemm<true, false>(3 * H, H, R, dgh, 3 * H, h, H, dW_hh, H, nullptr, TACT_NONE, splits, scr, st); if (s != PDP_OK) return s;
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// The same adjoint for a cell whose input is [state [R,H'] | xs [R]] held apart (W_ih [3H, H' + 1]): dstate = dgi W_ih[:, :H'] (one column
// block less than the concatenated form), dW_ih = [ dgi^T state | dgi^T xs ] -- the last column comes out of the pointwise pass as three more
// column sums -- and xs (the edge sign) has no gradient.  scratch [R, 6H].
extern "C" int pdp_train_gru_backward_s(const float *dhnew, const float *saved, const float *state, const float *xs, const float *h, const float *W_ih,
                                        const float *W_hh, int64_t R, int Ks, int H, float *dstate, float *dh, float *dW_ih, float *dW_hh, float *db_ih, float *db_hh,
                                        float *scratch, void *stream)
{
    PDP_REQUIRE(dhnew && saved && state && xs && h && W_ih && W_hh && dstate && dh && dW_ih && dW_hh && db_ih && db_hh && scratch && R < (int64_t)1 << 31, "bad argument");
    hipStream_t st = ST(stream);
    const int Kx = Ks + 1;
    if (R == 0) {
        PDP_HIP_CHECK(hipMemsetAsync(dW_ih, 0, sizeof(float) * 3 * (size_t)H * Kx, st)); PDP_HIP_CHECK(hipMemsetAsync(dW_hh, 0, sizeof(float) * 3 * (size_t)H * H, st));
        PDP_HIP_CHECK(hipMemsetAsync(db_ih, 0, sizeof(float) * 3 * (size_t)H, st)); PDP_HIP_CHECK(hipMemsetAsync(db_hh, 0, sizeof(float) * 3 * (size_t)H, st));
        return PDP_OK;
    }
    float *dgi = scratch, *dgh = scratch + (size_t)R * 3 * H;
    const int splits = pick_splits(R);
    const size_t wmax = (size_t)3 * H * (Ks > H ? Ks : H);
    const int cs = act_colsum_slices(R);
    float *scr = train_scratch((size_t)splits * wmax + wmax + (size_t)cs * 7 * H + 7 * H + 16, st);
    if (!scr) return PDP_ERR_HIP;
    float *dWx = scr + (size_t)splits * wmax, *part = dWx + wmax, *sums = part + (size_t)cs * 7 * H;       // sums [7 H] = dr | dz | dn | dn r | dr xs | dz xs | dn xs
    hipLaunchKernelGGL(k_gru_point_backward, dim3((H + 255) / 256, cs), dim3(256), 0, st, R, H, dhnew, saved, h, dgi, dgh, dh, cs, part, xs);
    hipLaunchKernelGGL(k_splitk_reduce, dim3(reduce_grid(7 * H)), dim3(RED_G * RED_E), 0, st, (int64_t)7 * H, cs, (const float *)part, sums);
    PDP_HIP_CHECK(hipMemcpyAsync(db_ih, sums, sizeof(float) * 3 * (size_t)H, hipMemcpyDeviceToDevice, st));
    PDP_HIP_CHECK(hipMemcpyAsync(db_hh, sums, sizeof(float) * 2 * (size_t)H, hipMemcpyDeviceToDevice, st));
    PDP_HIP_CHECK(hipMemcpyAsync(db_hh + 2 * (size_t)H, sums + 3 * (size_t)H, sizeof(float) * (size_t)H, hipMemcpyDeviceToDevice, st));
    int s = gemm_rows<false>((int)R, Ks, 3 * H, dgi, 3 * H, W_ih, Kx, dstate, Ks, nullptr, TACT_NONE, st); if (s != PDP_OK) return s;
    s = gemm<false, false>((int)R, H, 3 * H, dgh, 3 * H, W_hh, H, dh, H, nullptr, TACT_NONE | TACT_ACCUMULATE, 1, nullptr, st); if (s != PDP_OK) return s;
    s = gemm<true, false>(3 * H, Ks, R, dgi, 3 * H, state, Ks, dWx, Ks, nullptr, TACT_NONE, splits, scr, st); if (s != PDP_OK) return s;
    hipLaunchKernelGGL(k_assemble_dw, dim3(grid1d((int64_t)3 * H * Kx)), dim3(256), 0, st, 3 * H, Ks, (const float *)dWx, (const float *)(sums + 4 * (size_t)H), dW_ih);
    s = gemm<true, false>(3 * H, H, R, dgh, 3 * H, h, H, dW_hh, H, nullptr, TACT_NONE, splits, scr, st); if (s != PDP_OK) return s;
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- gradient of the energy loss with respect to the prediction (SatLossEvaluator.forward, util.py:178-197; forward: pdp_sat_loss) -------------
// ev = s x_v + (1 - s) / 2, w = exp(coeff ev); per clause nom = sum w ev, den = sum w, d = den / max(nom, eps) - 1, cv = 1 + d^k,
// term = log(max(cv, eps)), loss = mean of the terms.  torch.max(a, eps) passes the gradient to a where a > eps.
__global__ void __launch_bounds__(PDP_NT) k_sat_loss_grad(PView pv, const float *pred, float coeff, float eps, int sharpness, float upstream_over_F,
                                                        float *gnom, float *gden, float *dpred)
{
    const Inst I = load_inst(pv, blockIdx.x);
    pred += I.v0; gnom += I.f0; gden += I.f0; dpred += I.v0;
    for (int c = threadIdx.x; c < I.m; c += blockDim.x) {
        float nom = 0.0f, den = 0.0f;
        for (int k = I.f_ptr[c]; k < I.f_ptr[c + 1]; ++k) {
            const int e = I.f_edges[k];
            const float s = (float)I.sgn[e];
            const float ev = s * pred[I.e_var[e]] + (1.0f - s) / 2.0f;
            const float w = pdp_expf(coeff * ev);
            nom = nom + w * ev; den = den + w;
        }
        const float nm = pdp_max_c(nom, eps);
        const float d = den / nm - 1.0f;
        float pw1 = 1.0f;                                 // d^(k-1)
        for (int j = 1; j < sharpness; ++j) pw1 = pw1 * d;
        const float cv = 1.0f + pw1 * d;
        const float dterm = (cv > eps) ? upstream_over_F / cv : 0.0f;
        const float dd = dterm * (float)sharpness * pw1;
        gden[c] = dd / nm;
        gnom[c] = (nom > eps) ? -dd * den / (nm * nm) : 0.0f;
    }
    __syncthreads();
    for (int v = threadIdx.x; v < I.n; v += blockDim.x) {
        float acc = 0.0f;
        for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
            const int e = I.v_edges[k];
            const int c = I.e_fn[e];
            const float s = (float)I.sgn[e];
            const float ev = s * pred[v] + (1.0f - s) / 2.0f;
            const float w = pdp_expf(coeff * ev);
            const float dev = gnom[c] * (w + coeff * w * ev) + gden[c] * (coeff * w);
            acc = acc + s * dev;
        }
        dpred[v] = acc;
    }
}
extern "C" int pdp_sat_loss_grad(pdp_problem *p, const float *pred, float coeff, float eps, int sharpness, float upstream, float *dpred, void *stream)
{
    PDP_REQUIRE(p && pred && dpred, "NULL argument");
    PDP_REQUIRE(sharpness >= 1, "loss_sharpness must be a positive integer");
    hipLaunchKernelGGL(k_sat_loss_grad, dim3(p->B), dim3(PDP_NT), 0, ST(stream), make_view(p), pred, coeff, eps, sharpness, upstream / (float)p->F, p->ws_f[0],
                       p->ws_f[1], dpred);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- adjoint of the adaptor form of the SP sweep (model type p-nd-np; forward: pdp_sp_propagate_adapted, reference pdp_propagate.py:163-221) ------
// Forward, per edge e = (variable i, clause a), sign s, edge mask m (1 when absent), instance mask 1 (the training path has no active mask):
//   x = xlog m,  S_a = sum_{e in a} x,  eta' = X(S_a - x)                               X(t) = exp(min(t, 30))
//   y = L(1 - eta_in) m,  P_i / N_i = sums of y over the positive / negative edges of i     L(t) = log(max(t, 1e-40))
//   same = (s > 0 ? P : N) - y + c1,  opp = (s > 0 ? N : P) + c2   (c1, c2: constants of the external force, no gradient through sign())
//   A = X(same), B = X(opp), dc = X(same + opp), qu = A (1 - B), qs = B (1 - A), tot = qu + qs + dc, q = [qu, qs, dc] / tot
// torch.max / torch.min pass the gradient to the argument that wins (none on the clamped side).  One workgroup per instance: the clause
// sums and the two signed variable sums of the forward are rebuilt, then the two adjoint row sums, all ordered (no atomics).
__global__ void __launch_bounds__(PDP_NT) k_sp_adapted_backward(PView pv, const float *xlog, const float *fs2, const float *emask, float pi,
                                                               const float *gq, const float *geta, float *dxlog, float *deta_in,
                                                               float *xs, float *ys, float *ds, float *dop, float *Sw, float *Rw, float *Pw, float *Nw,
                                                               float *dPw, float *dNw)
{
    const Inst I = load_inst(pv, blockIdx.x);
    const int tid = threadIdx.x, nt = blockDim.x;
    xlog += I.e0; fs2 += 2 * (size_t)I.e0; gq += 3 * (size_t)I.e0; geta += I.e0; dxlog += I.e0; deta_in += I.e0;
    xs += I.e0; ys += I.e0; ds += I.e0; dop += I.e0; Sw += I.f0; Rw += I.f0; Pw += I.v0; Nw += I.v0; dPw += I.v0; dNw += I.v0;
    const float *em = emask ? emask + I.e0 : nullptr;
    const float L0 = pdp_safe_log(1.0f - pi * 0.0f, PDP_SP_EPS), L1 = pdp_safe_log(1.0f - pi * 1.0f, PDP_SP_EPS);
    for (int e = tid; e < I.e; e += nt) {
        const float m = em ? em[e] : 1.0f;
        xs[e] = xlog[e] * m;
        ys[e] = pdp_safe_log(1.0f - fs2[2 * e], PDP_SP_EPS) * m;
    }
    __syncthreads();
    for (int c = tid; c < I.m; c += nt) {
        float acc = 0.0f;
        for (int k = I.f_ptr[c]; k < I.f_ptr[c + 1]; ++k) acc = acc + xs[I.f_edges[k]];
        Sw[c] = acc;
    }
    for (int v = tid; v < I.n; v += nt) {
        float P = 0.0f, N = 0.0f;
        for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
            const int e = I.v_edges[k];
            if (I.sgn[e] == 1) P = P + ys[e]; else N = N + ys[e];
        }
        Pw[v] = P; Nw[v] = N;
    }
    __syncthreads();
    // per edge: d agg (kept in dxlog until the clause sums are there), d same, d opp
    for (int e = tid; e < I.e; e += nt) {
        const int v = I.e_var[e], c = I.e_fn[e];
        const float s = (float)I.sgn[e];
        const float agg = Sw[c] - xs[e];
        dxlog[e] = (agg < 30.0f) ? geta[e] * pdp_expf(agg) : 0.0f;
        const float force = fs2[2 * e + 1];
        const float same = ((s > 0.0f) ? Pw[v] : Nw[v]) - ys[e] + ((force == s) ? L1 : L0);
        const float opp = ((s > 0.0f) ? Nw[v] : Pw[v]) + ((force == -s) ? L1 : L0);
        const float A = pdp_safe_exp(same), Bv = pdp_safe_exp(opp), dc = pdp_safe_exp(same + opp);
        const float qu = A * (1.0f - Bv), qs = Bv * (1.0f - A);
        const float tot = (qu + qs) + dc;
        const float g0 = gq[3 * e], g1 = gq[3 * e + 1], g2 = gq[3 * e + 2];
        const float dot = (g0 * qu + g1 * qs + g2 * dc) / tot;
        const float dqu = (g0 - dot) / tot, dqs = (g1 - dot) / tot, ddc = (g2 - dot) / tot;
        const float dA = dqu * (1.0f - Bv) - dqs * Bv, dB = dqs * (1.0f - A) - dqu * A;
        const float gdc = (same + opp < 30.0f) ? ddc * dc : 0.0f;
        ds[e] = ((same < 30.0f) ? dA * A : 0.0f) + gdc;
        dop[e] = ((opp < 30.0f) ? dB * Bv : 0.0f) + gdc;
    }
    __syncthreads();
    for (int c = tid; c < I.m; c += nt) {
        float acc = 0.0f;
        for (int k = I.f_ptr[c]; k < I.f_ptr[c + 1]; ++k) acc = acc + dxlog[I.f_edges[k]];
        Rw[c] = acc;
    }
    for (int v = tid; v < I.n; v += nt) {
        float dP = 0.0f, dN = 0.0f;
        for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
            const int e = I.v_edges[k];
            if (I.sgn[e] == 1) { dP = dP + ds[e]; dN = dN + dop[e]; } else { dN = dN + ds[e]; dP = dP + dop[e]; }
        }
        dPw[v] = dP; dNw[v] = dN;
    }
    __syncthreads();
    for (int e = tid; e < I.e; e += nt) {
        const int v = I.e_var[e], c = I.e_fn[e];
        const float m = em ? em[e] : 1.0f;
        const float dagg = dxlog[e];
        dxlog[e] = (Rw[c] - dagg) * m;                                   // x enters every OTHER edge of its clause
        const float dy = (((I.sgn[e] == 1) ? dPw[v] : dNw[v]) - ds[e]) * m;
        const float om = 1.0f - fs2[2 * e];
        deta_in[e] = (om > PDP_SP_EPS) ? -dy / om : 0.0f;
    }
}

extern "C" int pdp_train_sp_adapted_backward(pdp_problem *p, const float *xlog, const float *fs2, const float *edge_mask, float pi, const float *g_q,
                                             const float *g_eta, float *d_xlog, float *d_eta_in, void *stream)
{
    PDP_REQUIRE(p && xlog && fs2 && g_q && g_eta && d_xlog && d_eta_in, "NULL argument");
    PDP_REQUIRE(d_xlog != g_eta && d_xlog != xlog, "outputs must not alias inputs");
    hipLaunchKernelGGL(k_sp_adapted_backward, dim3(p->B), dim3(PDP_NT), 0, ST(stream), make_view(p), xlog, fs2, edge_mask, pi, g_q, g_eta, d_xlog, d_eta_in,
                       p->ws_e[0], p->ws_e[1], p->ws_e[2], p->ws_e[3], p->ws_f[0], p->ws_f[1], p->ws_v[0], p->ws_v[1], p->ws_v[2], p->ws_v[3]);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}
