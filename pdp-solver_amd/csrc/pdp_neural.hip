// pdp_neural.hip -- the neural PDP operators on the fp32 matrix cores of gfx950.
// replaces: MessageAggregator.forward (reference: src/pdp/nn/util.py:51-77), NeuralMessagePasser.forward
// (src/pdp/nn/pdp_propagate.py:47-95), NeuralDecimator.forward / nn.GRUCell (src/pdp/nn/pdp_decimate.py:51-87),
// NeuralPredictor.forward + Perceptron head (src/pdp/nn/pdp_predict.py:49-91, src/pdp/trainer.py:20-29).
//
// Every per-edge MLP / GRU layer is a [64-edge tile] x [K] x [N] product on v_mfma_f32_32x32x2_f32.  That instruction
// is a k-ordered fp32 fmaf chain (MI355X_MICROARCH.md), so the results equal the CPU oracle's
// `acc = bias; acc = fmaf(x[k], w[k], acc)` loops bit for bit; zero padding of K adds fmaf(0,0,acc) = acc.
// A workgroup (8 waves) keeps the tile's activations in LDS between layers (row stride odd -> conflict-free column
// reads for the A operand), weights arrive pre-transposed / zero padded as Wt[Kp][Np] so the B operand is a coalesced
// 128-byte row segment per half-wave; the activation (exact logsigmoid / sigmoid / tanh from pdp_math.h) is applied on
// the accumulator registers.  Row sums between the two halves of an aggregator are sequential in ascending edge id.
//
// Kernel map (DESIGN.md section 4.4 has the measurements behind each choice):
//   generic shapes      k_agg_pre / k_agg_pre_res, k_row_sum, k_agg_post, k_predict_rows, k_gru
//   hidden 128 and 150  k_agg_pre_wave (a wave owns a 32-edge tile through both layers), k_agg_post_pf (prefetched chains; at 150 the
//                       full tiles run k_agg_post_wave, a wave per tile)
//   hidden 128          k_gru_pipe (in-wave pipelined MFMA chains and activation slices; also the 4- / 3-input cells of p-nd-np)
//   hidden 150          k_gru_wave (a wave owns a 32-edge tile through all five column blocks: one software pipeline per tile)
// What bounds them: on gfx950 the f32 MFMA and the VALU share issue time on a SIMD -- their times add up whichever wave issues them --
// so beyond keeping every MFMA's operands in registers ahead of time, instruction count is what counts.
#include "pdp_common.hpp"

#define ST(s) ((hipStream_t)(s))
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TM 64             // edges (rows) per tile
#define NWAVES 8
#define NTN (NWAVES * 64)

enum { ACT_NONE = 0, ACT_LOGSIGMOID = 1, ACT_RELU = 2, ACT_SIGMOID = 3, ACT_TANH = 4 };

__device__ __forceinline__ float act_apply(float v, int act)
{
    switch (act) {
    case ACT_LOGSIGMOID: return pdp_logsigmoidf(v);
    case ACT_RELU: return v > 0.0f ? v : ((v != v) ? v : 0.0f);
    case ACT_SIGMOID: return pdp_sigmoidf(v);
    case ACT_TANH: return pdp_tanhf(v);
    default: return v;
    }
}

// logsigmoid of a live column, 0 for a padded one -- as a select: left to itself the compiler branches around the activation per element
// (an exec-mask region each, which also keeps the elements' dependent chains from interleaving); the empty asm pins the value in front of it
__device__ __forceinline__ float logsigmoid_or_zero(float x, bool live)
{
    float v = pdp_logsigmoidf(x);
    asm volatile("" : "+v"(v));
    return live ? v : 0.0f;
}

// Two activations side by side on the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): every step is the same IEEE
// operation per element as pdp_expf_fin_le30 / pdp_logsigmoidf of include/pdp_math.h, so the results are the same bits
// (tests/test_hip_neural.py compares whole operators with the oracle's scalar forms).  A packed instruction occupies the issue slot of
// two plain ones on gfx950 (tools/micro/pk_rate.hip), so this is not half the time: what it buys is 35 instead of 54 instructions and
// fewer live registers per pair, i.e. scheduling freedom -- measured per aggregator call: hidden 150 16.5 -> 15.8 / 17.0 -> 16.4 ms,
// hidden 128 13.9 -> 13.8 / 14.65 -> 14.5 ms.  The GRU's activation slices stay scalar: packed they were 7 % slower at hidden 128
// (22.2 -> 23.8 ms; their instruction groups are tuned to the MFMA chunks) and no faster at 150.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
// (the bits of a vector ELEMENT go through a by-value float: __builtin_bit_cast applied to `v.y` itself reads element 0 with this hipcc)
__device__ __forceinline__ int f2i(float f) { return __builtin_bit_cast(int, f); }
// the row mask of k_edge_active ends with the stop word of a device-driven loop (rowmask + E is the same address for a launch on a tail of the
// rows: its mask pointer is advanced by what its row count is short of); workgroup-uniform
__device__ __forceinline__ bool loop_stopped(const float *rowmask, int E)
{
    return rowmask && __builtin_amdgcn_readfirstlane((int)__builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(rowmask) + E)) != 0;
}
// the same word for a kernel of many short workgroups: requested at entry, looked at in front of the first store -- tested at entry it is
// one more DEPENDENT round trip in front of every tile's gathers
__device__ __forceinline__ uint32_t loop_stop_word(const float *rowmask, int E)
{
    return rowmask ? (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(rowmask) + E)) : 0u;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// e^x after the clamp (xc = the clamped argument, x = the argument itself for the NaN / infinity rule of the scalar forms)
__device__ __forceinline__ f32x2 pk_exp_clamped(f32x2 xc, f32x2 x)
{
    const f32x2 t = xc * 1.44269504088896341f;
    const f32x2 nf = (t + 12582912.0f) - 12582912.0f;
    f32x2 r = pk_fma(nf, (f32x2)(-0.693359375f), xc);
    r = pk_fma(nf, (f32x2)(2.12194440e-4f), r);
    const f32x2 z = r * r;
    f32x2 p = (f32x2)(1.9875691500e-4f);
    p = pk_fma(p, r, (f32x2)(1.3981999507e-3f));
    p = pk_fma(p, r, (f32x2)(8.3334519073e-3f));
    p = pk_fma(p, r, (f32x2)(4.1665795894e-2f));
    p = pk_fma(p, r, (f32x2)(1.6666665459e-1f));
    p = pk_fma(p, r, (f32x2)(5.0000001201e-1f));
    p = pk_fma(p, z, r);
    p = p + 1.0f;
    const f32x2 res = {__builtin_ldexpf(p.x, (int)nf.x), __builtin_ldexpf(p.y, (int)nf.y)};
    return res + (x - x);                                   // NaN / infinite argument -> NaN, as in the scalar forms
}
__device__ __forceinline__ f32x2 pk_expf_fin_le30(f32x2 x)                                  // pdp_expf_fin_le30
{
    return pk_exp_clamped((f32x2){pdp_fmaxf(x.x, -104.5f), pdp_fmaxf(x.y, -104.5f)}, x);
}
#ifdef PDP_FAST_MATH
// the opt-in fast build (include/pdp_math.h): the scalar pdp_logsigmoidf on the transcendental unit, element by element (a packed form has
// nothing to pack: the work is two v_exp_f32 / v_log_f32 pairs)
__device__ __forceinline__ f32x2 pk_logsigmoid(f32x2 x) { return (f32x2){pdp_logsigmoidf(x.x), pdp_logsigmoidf(x.y)}; }
#else
__device__ __forceinline__ f32x2 pk_logsigmoid(f32x2 x)
{
    const f32x2 t = pk_expf_fin_le30((f32x2){-pdp_abs(x.x), -pdp_abs(x.y)});
    f32x2 p = (f32x2)(5.253457930e-03f);
    p = pk_fma(p, t, (f32x2)(-2.958850749e-02f));
    p = pk_fma(p, t, (f32x2)(7.836166769e-02f));
    p = pk_fma(p, t, (f32x2)(-1.367477030e-01f));
    p = pk_fma(p, t, (f32x2)(1.911143064e-01f));
    p = pk_fma(p, t, (f32x2)(-2.484436929e-01f));
    p = pk_fma(p, t, (f32x2)(3.331927061e-01f));
    p = pk_fma(p, t, (f32x2)(-4.999950230e-01f));
    p = pk_fma(p, t, (f32x2)(1.0f));
    const f32x2 mn = {pdp_fminf(x.x, 0.0f), pdp_fminf(x.y, 0.0f)};
    return mn - p * t;
}
#endif
__device__ __forceinline__ f32x2 pk_logsigmoid_or_zero(float x0, float x1, bool live)
{
    f32x2 v = pk_logsigmoid((f32x2){x0, x1});
    asm volatile("" : "+v"(v));
    return live ? v : (f32x2)(0.0f);
}

// Two 32x32 output blocks (both 32-row halves of the tile, same 32 columns):
//   acc[mb][r] = bias[col];  acc[mb] += A[32*mb + i][k] * Wt[k][32*nb + j],  k ascending.
// The B fragment (one float per lane and k-step, a coalesced 128-byte row segment per half-wave) is shared by the two
// row blocks and prefetched into registers one 16-step chunk ahead (double buffered), so the L2 latency of the weight
// stream is covered by 32 MFMAs (2048 cycles) instead of stalling every k-step.
template <int MB, int BCH>
__device__ __forceinline__ void mfma_chain(const float *A, int lda, int Kp, const float *__restrict__ Wt, int Np, int nb, int mb0,
                                           const float *__restrict__ bias, f32x16 (&acc)[MB])
{
    const int l = threadIdx.x & 63, i = l & 31, kh = l >> 5;
    const float b0 = bias ? bias[32 * nb + i] : 0.0f;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = b0;
    const float *ap = A + (32 * mb0 + i) * lda + kh;
    const float *bp = Wt + (size_t)kh * Np + 32 * nb + i;
    const int steps = Kp >> 1;
    const size_t bstride = (size_t)2 * Np;
    // both operands of a chunk of BCH k-steps sit in registers before its MFMAs issue: the weight fragments (L2) and the LDS operands of
    // the next chunk are requested while the current chunk runs (double buffered)
    float bufA[BCH], bufB[BCH], aA[MB][BCH], aB[MB][BCH];
    int s0 = 0;
    if (steps >= BCH) {
#pragma unroll
        for (int s = 0; s < BCH; ++s) {
            bufA[s] = bp[(size_t)s * bstride];
#pragma unroll
            for (int m = 0; m < MB; ++m) aA[m][s] = ap[32 * m * lda + 2 * s];
        }
    }
    while (s0 + BCH <= steps) {
        const bool more1 = s0 + 2 * BCH <= steps;
        if (more1) {
#pragma unroll
            for (int s = 0; s < BCH; ++s) {
                bufB[s] = bp[(size_t)(s0 + BCH + s) * bstride];
#pragma unroll
                for (int m = 0; m < MB; ++m) aB[m][s] = ap[32 * m * lda + 2 * (s0 + BCH + s)];
            }
        }
#pragma unroll
        for (int s = 0; s < BCH; ++s) {
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA[m][s], bufA[s], acc[m], 0, 0, 0);
        }
        s0 += BCH;
        if (!more1) break;
        const bool more2 = s0 + 2 * BCH <= steps;
        if (more2) {
#pragma unroll
            for (int s = 0; s < BCH; ++s) {
                bufA[s] = bp[(size_t)(s0 + BCH + s) * bstride];
#pragma unroll
                for (int m = 0; m < MB; ++m) aA[m][s] = ap[32 * m * lda + 2 * (s0 + BCH + s)];
            }
        }
#pragma unroll
        for (int s = 0; s < BCH; ++s) {
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(aB[m][s], bufB[s], acc[m], 0, 0, 0);
        }
        s0 += BCH;
        if (!more2) break;
    }
    for (int s = s0; s < steps; ++s) {
        const float bv = bp[(size_t)s * bstride];
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[32 * m * lda + 2 * s], bv, acc[m], 0, 0, 0);
    }
}
// one 32x32 block of a layer on an LDS tile with the weights streamed from L2 (generic shapes)
__device__ __forceinline__ f32x16 mfma_block(const float *A, int lda, int Kp, const float *__restrict__ Wt, int Np, int nb, int mb,
                                             const float *__restrict__ bias)
{
    f32x16 acc[1];
    mfma_chain<1, 8>(A, lda, Kp, Wt, Np, nb, mb, bias, acc);
    return acc[0];
}
// both operands in LDS (resident weights): batches of U k-steps, the operands of the next batch are read while the MFMAs of the
// current one run (an MFMA chain is serial: 64 cycles per step, far longer than an LDS round trip per batch)
template <int U>
__device__ __forceinline__ f32x16 mfma_block_lds(const float *A, int lda, int Kp, const float *W, int Np, int nb, int mb, const float *__restrict__ bias)
{
    const int l = threadIdx.x & 63, i = l & 31, kh = l >> 5;
    const float b0 = bias ? bias[32 * nb + i] : 0.0f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b0;
    const float *ap = A + (32 * mb + i) * lda + kh;
    const float *bp = W + kh * Np + 32 * nb + i;
    const int steps = Kp >> 1, full = steps / U;
    float a0[U], w0[U], a1[U], w1[U];
    if (full > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) { a0[u] = ap[2 * u]; w0[u] = bp[2 * u * Np]; }
    }
    int g = 0;
    for (; g + 1 < full; g += 2) {
        const int k1 = 2 * U * (g + 1);
#pragma unroll
        for (int u = 0; u < U; ++u) { a1[u] = ap[k1 + 2 * u]; w1[u] = bp[(k1 + 2 * u) * Np]; }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], w0[u], acc, 0, 0, 0);
        if (g + 2 < full) {
            const int k2 = 2 * U * (g + 2);
#pragma unroll
            for (int u = 0; u < U; ++u) { a0[u] = ap[k2 + 2 * u]; w0[u] = bp[(k2 + 2 * u) * Np]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], w1[u], acc, 0, 0, 0);
    }
    if (g < full) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], w0[u], acc, 0, 0, 0);
    }
    for (int s2 = full * U; s2 < steps; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s2], bp[2 * s2 * Np], acc, 0, 0, 0);
    return acc;
}

// lds_layer with LDS-resident weights
__device__ __forceinline__ int acc_row(int r, int l);
__device__ __forceinline__ void lds_layer_res(const float *in, int ldi, int Kp, const float *W, int Np, const float *bias, int n_valid, int act,
                                              float *out, int ldo);

__device__ __forceinline__ void mfma_pair(const float *A, int lda, int Kp, const float *__restrict__ Wt, int Np, int nb,
                                          const float *__restrict__ bias, f32x16 &acc0, f32x16 &acc1)
{
    acc0 = mfma_block(A, lda, Kp, Wt, Np, nb, 0, bias);
    acc1 = mfma_block(A, lda, Kp, Wt, Np, nb, 1, bias);
}

// C/D layout of the 32x32 MFMA: reg r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31
__device__ __forceinline__ int acc_row(int r, int l) { return (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); }

// layer on an LDS tile: out[row][col] = act(bias + in[row][:] . Wt[:, col]); columns >= n_valid are written as 0
__device__ __forceinline__ void lds_layer(const float *in, int ldi, int Kp, const float *Wt, int Np, const float *bias, int n_valid, int act,
                                          float *out, int ldo)
{
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int nblocks = (Np / 32) * (TM / 32);
    for (int blk = wave; blk < nblocks; blk += NWAVES) {
        const int nb = blk / (TM / 32), mb = blk % (TM / 32);
        const f32x16 acc = mfma_block(in, ldi, Kp, Wt, Np, nb, mb, bias);
        const int col = 32 * nb + (l & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * mb + acc_row(r, l);
            out[row * ldo + col] = (col < n_valid) ? act_apply(acc[r], act) : 0.0f;
        }
    }
}

__device__ __forceinline__ void lds_layer_res(const float *in, int ldi, int Kp, const float *W, int Np, const float *bias, int n_valid, int act,
                                              float *out, int ldo)
{
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int nblocks = (Np / 32) * (TM / 32);
    for (int blk = wave; blk < nblocks; blk += NWAVES) {
        const int nb = blk / (TM / 32), mb = blk % (TM / 32);
        const f32x16 acc = mfma_block_lds<8>(in, ldi, Kp, W, Np, nb, mb, bias);
        const int col = 32 * nb + (l & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * mb + acc_row(r, l);
            out[row * ldo + col] = (col < n_valid) ? act_apply(acc[r], act) : 0.0f;
        }
    }
}

// ---- model description handed over by the host (all weights pre-transposed and zero padded: Wt[Kp][Np]) ------------
struct AggW {      // MessageAggregator
    const float *Wt1m, *b1m, *Wt2m, *Wt1a, *b1a, *Wt2a;
    int din;       // state width + 1 (edge sign)            -> Kp1 = even(din)
    int m1, a, g, out;            // valid widths
    int Kp1, Np1, Kp2, Np2, Kp3, Np3, Kp4, Np4;
    int fd;        // 1: edge sign appended to the aggregated vector (include_self = False), 0: not
    int no_tail16; // PDP_NEURAL_NO_TAIL16: every layer in 32-column blocks (A/B runs of the 16-column tail blocks)
};
struct GruW {
    const float *Wt_ih, *Wt_hh, *b_ih, *b_hh;     // Wt_ih [Kpx][3*Hp], Wt_hh [Kph][3*Hp], biases [3*Hp]
    int dx, H, Kpx, Kph, Hp;
};
struct HeadW { const float *Wt1, *b1, *w2; int H, C, Kp, Np, out_act; };   // Perceptron: Wt1 [Kp][Np], w2 [C]

static inline int even_up(int x) { return (x + 1) & ~1; }
static inline int pad32(int x) { return (x + 31) & ~31; }

// ---- kernel 1: aggregator pre-transform on edge tiles:  h2 = logsig(W2m logsig(W1m [state ‖ s] + b1m)) * edge_mask ------
__global__ void __launch_bounds__(NTN) k_agg_pre(int E, const float *__restrict__ state, int sd, const float *__restrict__ sign,
                                                 const float *__restrict__ emask, AggW w, float *__restrict__ h2out)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ld0 = w.Kp1 + 1, ld1 = w.Np1 + 1;
    float *X = sm, *H1 = sm + TM * ld0;
    const int e0 = blockIdx.x * TM;
    for (int idx = threadIdx.x; idx < TM * w.Kp1; idx += NTN) {
        const int r = idx / w.Kp1, c = idx % w.Kp1;
        const int e = e0 + r;
        float v = 0.0f;
        if (e < E) v = (c < sd) ? state[(size_t)e * sd + c] : (c == sd ? sign[e] : 0.0f);
        X[r * ld0 + c] = v;
    }
    __syncthreads();
    lds_layer(X, ld0, w.Kp1, w.Wt1m, w.Np1, w.b1m, w.m1, ACT_LOGSIGMOID, H1, ld1);
    __syncthreads();
    // second layer straight to HBM (compact [E, a]), masked by the edge mask (util.py:57-58)
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int nblocks = (w.Np2 / 32) * (TM / 32);
    for (int blk = wave; blk < nblocks; blk += NWAVES) {
        const int nb = blk / (TM / 32), mb = blk % (TM / 32);
        const f32x16 acc = mfma_block(H1, ld1, w.Kp2, w.Wt2m, w.Np2, nb, mb, nullptr);
        const int col = 32 * nb + (l & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = e0 + 32 * mb + acc_row(r, l);
            if (e < E && col < w.a) {
                float v = pdp_logsigmoidf(acc[r]);
                if (emask) v = v * emask[e];
                h2out[(size_t)e * w.a + col] = v;
            }
        }
    }
}

// ---- kernel 2: row sums (per variable / clause), sequential in ascending edge id ------------------------------------------
__global__ void k_row_sum(int R, int A, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ row_edges,
                          const float *__restrict__ h2, float *__restrict__ agg)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)R * A) return;
    const int r = (int)(idx / A), c = (int)(idx % A);
    float acc = 0.0f;
    int k = row_ptr[r];
    const int k1 = row_ptr[r + 1];
    for (; k + 4 <= k1; k += 4) {                          // four gathers in flight, added in edge order
        const int e0 = row_edges[k], e1 = row_edges[k + 1], e2 = row_edges[k + 2], e3 = row_edges[k + 3];
        const float v0 = h2[(size_t)e0 * A + c], v1 = h2[(size_t)e1 * A + c], v2 = h2[(size_t)e2 * A + c], v3 = h2[(size_t)e3 * A + c];
        acc = acc + v0; acc = acc + v1; acc = acc + v2; acc = acc + v3;
    }
    if (k + 2 <= k1) {
        const int e0 = row_edges[k], e1 = row_edges[k + 1];
        const float v0 = h2[(size_t)e0 * A + c], v1 = h2[(size_t)e1 * A + c];
        acc = acc + v0; acc = acc + v1; k += 2;
    }
    if (k < k1) acc = acc + h2[(size_t)row_edges[k] * A + c];
    agg[idx] = acc;
}
// the same sums with two columns per thread (A even: the rows are 8-byte aligned): half the threads, index loads and gather instructions
// for the same bytes -- the kernel waits on HBM 90 % of its time with one float per thread and row
__global__ void k_row_sum2(int R, int A2 /* A / 2 */, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ row_edges,
                           const float2 *__restrict__ h2, float2 *__restrict__ agg)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)R * A2) return;
    const int r = (int)(idx / A2), c = (int)(idx % A2);
    float ax = 0.0f, ay = 0.0f;
    int k = row_ptr[r];
    const int k1 = row_ptr[r + 1];
    for (; k + 8 <= k1; k += 8) {                          // (variable rows: a dozen edges) eight gathers in flight
        int e[8]; float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = row_edges[k + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = h2[(size_t)e[j] * A2 + c];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ax = ax + v[j].x; ay = ay + v[j].y; }
    }
    for (; k + 4 <= k1; k += 4) {
        const int e0 = row_edges[k], e1 = row_edges[k + 1], e2 = row_edges[k + 2], e3 = row_edges[k + 3];
        const float2 v0 = h2[(size_t)e0 * A2 + c], v1 = h2[(size_t)e1 * A2 + c], v2 = h2[(size_t)e2 * A2 + c], v3 = h2[(size_t)e3 * A2 + c];
        ax = ax + v0.x; ay = ay + v0.y; ax = ax + v1.x; ay = ay + v1.y; ax = ax + v2.x; ay = ay + v2.y; ax = ax + v3.x; ay = ay + v3.y;
    }
    if (k + 2 <= k1) {
        const int e0 = row_edges[k], e1 = row_edges[k + 1];
        const float2 v0 = h2[(size_t)e0 * A2 + c], v1 = h2[(size_t)e1 * A2 + c];
        ax = ax + v0.x; ay = ay + v0.y; ax = ax + v1.x; ay = ay + v1.y; k += 2;
    }
    if (k < k1) { const float2 v = h2[(size_t)row_edges[k] * A2 + c]; ax = ax + v.x; ay = ay + v.y; }
    agg[idx] = make_float2(ax, ay);
}
static void launch_row_sum(int R, int A, const int32_t *row_ptr, const int32_t *row_edges, const float *h2, float *agg, hipStream_t st)
{
    const bool two = A % 2 == 0 && ((uintptr_t)h2 & 7) == 0 && ((uintptr_t)agg & 7) == 0;
    pdp_note_kernel(PDP_TK_ROW_SUM, two ? "k_row_sum2" : "k_row_sum");
    if (two)
        hipLaunchKernelGGL(k_row_sum2, dim3((unsigned)(((int64_t)R * (A / 2) + 255) / 256)), dim3(256), 0, st, R, A / 2, row_ptr, row_edges,
                           reinterpret_cast<const float2 *>(h2), reinterpret_cast<float2 *>(agg));
    else
        hipLaunchKernelGGL(k_row_sum, dim3((unsigned)(((int64_t)R * A + 255) / 256)), dim3(256), 0, st, R, A, row_ptr, row_edges, h2, agg);
}

// ---- kernel 3: aggregator post-transform on edge tiles (include_self = False) ------------------------------------------------
// r = agg[row(e)] - h2[e] * edge_mask ; [r ‖ s] -> logsig(W2a logsig(W1a . + b1a)) ; out = mask * new + (1 - mask) * old
__global__ void __launch_bounds__(NTN) k_agg_post(int E, const float *__restrict__ agg, const int32_t *__restrict__ edge_row,
                                                  const float *__restrict__ h2, const float *__restrict__ sign,
                                                  const float *__restrict__ emask, const float *__restrict__ rowmask /*[E] or NULL*/,
                                                  const float *__restrict__ old, AggW w, float *__restrict__ out)
{
    if (loop_stopped(rowmask, E)) return;                  // a device-driven loop has ended: this sweep writes nothing (k_edge_active)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ld0 = w.Kp3 + 1, ld1 = w.Np3 + 1;
    float *Rt = sm, *G1 = sm + TM * ld0;
    const int e0 = blockIdx.x * TM;
    for (int idx = threadIdx.x; idx < TM * w.Kp3; idx += NTN) {
        const int r = idx / w.Kp3, c = idx % w.Kp3;
        const int e = e0 + r;
        float v = 0.0f;
        if (e < E) {
            if (c < w.a) {
                const float own = emask ? h2[(size_t)e * w.a + c] * emask[e] : h2[(size_t)e * w.a + c];
                v = (0.0f + agg[(size_t)edge_row[e] * w.a + c]) - own;
            } else if (c == w.a && w.fd) v = sign[e];
        }
        Rt[r * ld0 + c] = v;
    }
    __syncthreads();
    lds_layer(Rt, ld0, w.Kp3, w.Wt1a, w.Np3, w.b1a, w.g, ACT_LOGSIGMOID, G1, ld1);
    __syncthreads();
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int nblocks = (w.Np4 / 32) * (TM / 32);
    for (int blk = wave; blk < nblocks; blk += NWAVES) {
        const int nb = blk / (TM / 32), mb = blk % (TM / 32);
        const f32x16 acc = mfma_block(G1, ld1, w.Kp4, w.Wt2a, w.Np4, nb, mb, nullptr);
        const int col = 32 * nb + (l & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = e0 + 32 * mb + acc_row(r, l);
            if (e < E && col < w.out) {
                const float nv = pdp_logsigmoidf(acc[r]);
                const float mk = rowmask ? rowmask[e] : 1.0f;
                out[(size_t)e * w.out + col] = mk * nv + (1.0f - mk) * old[(size_t)e * w.out + col];
            }
        }
    }
}

// ---- resident-weight form of kernel 1 --------------------------------------------------------------------------------------------------
// The weight matrices of an aggregator half are small (W1m + W2m = 92 KB at hidden 128): fetching the B operand from L2 inside the
// MFMA chain left the matrix cores idle for most of the time (one k-step per L2 round trip).  Here a workgroup copies both
// matrices into LDS once and then walks over many edge tiles (persistent grid, one workgroup per CU); the next tile's input
// rows are fetched into registers while the current tile runs, and dropped into the X buffer once layer 1 no longer needs it.
#define PRE_R (TM / NWAVES)    /* rows of a tile per wave */
#define PRE_C 3                /* 64-column passes per row (input width <= 192) */

__device__ __forceinline__ void copy_to_lds(float *dst, const float *__restrict__ src, int n)
{
    for (int i = threadIdx.x; i < n; i += NTN) dst[i] = src[i];
}

__global__ void __launch_bounds__(NTN) k_agg_pre_res(int E, const float *__restrict__ state, int sd, const float *__restrict__ sign,
                                                     const float *__restrict__ emask, AggW w, float *__restrict__ h2out, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ld0 = w.Kp1 + 1, ld1 = w.Np1 + 1;
    float *W1 = sm, *W2 = W1 + w.Kp1 * w.Np1, *X = W2 + w.Kp2 * w.Np2, *H1 = X + TM * ld0;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    copy_to_lds(W1, w.Wt1m, w.Kp1 * w.Np1);
    copy_to_lds(W2, w.Wt2m, w.Kp2 * w.Np2);
    float pre[PRE_R][PRE_C];
    auto fetch = [&](int tile) {
        const int e0 = tile * TM;
#pragma unroll
        for (int jr = 0; jr < PRE_R; ++jr) {
            const int e = e0 + wave + NWAVES * jr;
#pragma unroll
            for (int jc = 0; jc < PRE_C; ++jc) {
                const int c = l + 64 * jc;
                float v = 0.0f;
                if (e < E && c < w.Kp1) v = (c < sd) ? state[(size_t)e * sd + c] : (c == sd ? sign[e] : 0.0f);
                pre[jr][jc] = v;
            }
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int jr = 0; jr < PRE_R; ++jr)
#pragma unroll
            for (int jc = 0; jc < PRE_C; ++jc) { const int c = l + 64 * jc; if (c < w.Kp1) X[(wave + NWAVES * jr) * ld0 + c] = pre[jr][jc]; }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) { fetch(tile); deposit(); }
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                   // X (and, the first time, the weights) are in place
        const int next = tile + gridDim.x;
        if (next < ntiles) fetch(next);                   // in flight during layer 1
        lds_layer_res(X, ld0, w.Kp1, W1, w.Np1, w.b1m, w.m1, ACT_LOGSIGMOID, H1, ld1);
        __syncthreads();                                   // H1 complete, X free
        if (next < ntiles) deposit();
        const int e0 = tile * TM;
        const int nblocks = (w.Np2 / 32) * (TM / 32);
        for (int blk = wave; blk < nblocks; blk += NWAVES) {
            const int nb = blk / (TM / 32), mb = blk % (TM / 32);
            const f32x16 acc = mfma_block_lds<8>(H1, ld1, w.Kp2, W2, w.Np2, nb, mb, nullptr);
            const int col = 32 * nb + (l & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = e0 + 32 * mb + acc_row(r, l);
                if (e < E && col < w.a) {
                    float v = pdp_logsigmoidf(acc[r]);
                    if (emask) v = v * emask[e];
                    h2out[(size_t)e * w.a + col] = v;
                }
            }
        }
    }
}

// ---- wave-private form of kernel 1 ---------------------------------------------------------------------------------------------------------
// In the workgroup-tile kernels above the MFMA chains fetch their LDS operands just in time and the 64-column second layer occupies only
// half of the waves between two barriers.  Here a wave owns a 32-edge tile from the input rows to the stored result: its input block and
// (over the same LDS bytes) the hidden layer live in a private LDS region, each layer is one straight-line MFMA sequence with both
// operands prefetched, and there is no workgroup barrier at all.  Measured on config 3 (tools/neural_ops_time.py): 10.4 -> 8.2 ms; the
// pieces are additive (chains 3.2 ms + activations 2.1 ms + rest 3.1 ms): f32 MFMA and VALU work of the two waves of a SIMD do not
// overlap, so what is left is instruction count.  The same form of the post-transform was built and measured: 12.8 ms against 12.5 ms for
// k_agg_post (its gather / previous-state / store streams do not hide behind 2 waves per SIMD), so that kernel stays as it is.
// Per element the arithmetic is the same chain and the same activation, so results are unchanged.
#define WT 32              /* edges per wave tile */

// NB column blocks of one layer as one straight-line sequence of NB * STEPS MFMAs (block after block, each block one k-ordered chain):
// the A operands (LDS) are read one 8-step chunk ahead, the weight fragments (L2, buffer loads: lane offset in a VGPR, k-step offset as a
// scalar) two chunks ahead, so no MFMA waits for an operand and the prefetch runs on across the block boundaries.
// SWAP: the MFMAs are fed (weights, activations) -- the block comes out transposed: lane (i, kh) then holds ROW i of the tile, its register
// r = 4 q + c the column 8 q + 4 kh + c (four consecutive columns per register quad: 16-byte loads / stores of the surrounding rows).  Every
// element is the same chain of the same products in the same k order (a * b = b * a), so the values are bit-identical; no bias in this form.
template <int STEPS, int NB, int NP, bool SWAP = false>
__device__ __forceinline__ void wave_chains(const float *a /* LDS: row (lane & 31) of the block, column lane >> 5 */, __amdgpu_buffer_rsrc_t wr,
                                            const float *__restrict__ bias, f32x16 (&acc)[NB])
{
    constexpr int T = STEPS * NB, CH = 8, NC = (T + CH - 1) / CH;
    const int l = threadIdx.x & 63, i = l & 31, kh = l >> 5;
    const int voff = (kh * NP + i) * (int)sizeof(float);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float b0 = bias ? bias[32 * nb + i] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = b0;
    }
    auto wl = [&](int t) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, voff + (t / STEPS) * 32 * (int)sizeof(float),
                                                                             (t % STEPS) * 2 * NP * (int)sizeof(float), 0));
    };
    float aa[2][CH], bb[3][CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        if (j < T) { bb[0][j] = wl(j); aa[0][j] = a[2 * (j % STEPS)]; }
        if (CH + j < T) bb[1][j] = wl(CH + j);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int t2 = (c + 2) * CH + j, t1 = (c + 1) * CH + j;
            if (t2 < T) bb[(c + 2) % 3][j] = wl(t2);
            if (t1 < T) aa[(c + 1) & 1][j] = a[2 * (t1 % STEPS)];
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int t = c * CH + j;
            if (t < T) acc[t / STEPS] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(bb[c % 3][j], aa[c & 1][j], acc[t / STEPS], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_32x32x2f32(aa[c & 1][j], bb[c % 3][j], acc[t / STEPS], 0, 0, 0);
        }
        // pure arithmetic is not ordered against scheduling barriers by itself: tie the chunk's accumulators to this point
        asm volatile("" : "+v"(acc[(c * CH) / STEPS]));
        if ((c * CH + CH - 1) / STEPS != (c * CH) / STEPS && (c * CH + CH - 1) / STEPS < NB) asm volatile("" : "+v"(acc[(c * CH + CH - 1) / STEPS]));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// A 16-column tail block of a layer for a 32-row wave tile: two 16x16 output blocks (rows 0-15 and 16-31, columns c0 .. c0+15) on
// v_mfma_f32_16x16x4_f32, which accumulates its four k-steps in ascending order with one rounding each (tools/micro/mfma16_order.hip: 0 of 256
// elements differ from the fmaf chain), so the block equals the oracle's chain like the 32-wide blocks do.  A 100-wide layer is then three
// 32-wide blocks + this tail (112 columns) instead of four blocks (128): 66 half-size MFMAs instead of 65 full-size ones and half of the last
// block's activations.  Operands as in wave_chains: LDS one 8-step chunk ahead, weights (same Wt[Kp][NP] matrix; rows past Kp are clipped to
// zero by the descriptor) two chunks ahead.  ST = k-steps of four, the LDS rows hold zeros up to column 4 * ST.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int ST, int NP, int CH = 8>
__device__ __forceinline__ void wave_tail16(const float *X /* LDS tile, row stride ld */, int ld, __amdgpu_buffer_rsrc_t wr, int c0,
                                            const float *__restrict__ bias, f32x4 (&acc)[2])
{
    constexpr int NC = (ST + CH - 1) / CH;
    const int l = threadIdx.x & 63, j = l & 15, kq = l >> 4;
    const float b0 = bias ? bias[c0 + j] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[0][r] = b0; acc[1][r] = b0; }
    const float *a0 = X + j * ld + kq, *a1 = X + (16 + j) * ld + kq;
    const int voff = (kq * NP + c0 + j) * (int)sizeof(float);
    auto wl = [&](int t) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, voff, t * 4 * NP * (int)sizeof(float), 0)); };
    float aa[2][2][CH], bb[3][CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) {
        if (q < ST) { bb[0][q] = wl(q); aa[0][0][q] = a0[4 * q]; aa[0][1][q] = a1[4 * q]; }
        if (CH + q < ST) bb[1][q] = wl(CH + q);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int t2 = (c + 2) * CH + q, t1 = (c + 1) * CH + q;
            if (t2 < ST) bb[(c + 2) % 3][q] = wl(t2);
            if (t1 < ST) { aa[(c + 1) & 1][0][q] = a0[4 * t1]; aa[(c + 1) & 1][1][q] = a1[4 * t1]; }
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int t = c * CH + q;
            if (t < ST) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[c & 1][0][q], bb[c % 3][q], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[c & 1][1][q], bb[c % 3][q], acc[1], 0, 0, 0);
            }
        }
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Specialised on the layer shapes (S = k-steps, NB = 32-column blocks); other shapes use the workgroup-tile kernels above.
// k_agg_pre_wave: the input row is [128 or 150 message floats, edge sign, zero pad]: two or three dwords per lane and row, and the whole next tile is
// fetched into registers right after the current one has been dropped into LDS, so its HBM latency hides behind both layers.
template <int S1, int NB1, int S2, int NB2, bool TAIL16>
__global__ void __launch_bounds__(NTN) k_agg_pre_wave(int E, const float *__restrict__ state, const float *__restrict__ sign,
                                                      const float *__restrict__ emask, AggW w, float *__restrict__ h2out, int ntiles)
{
    constexpr int SD = 2 * S1 - 2, CG = (SD + 63) / 64;    // message width (input row = SD floats + sign + pad), 64-column groups of a row
    static_assert(CG <= 3, "input rows of at most 192 floats");
    constexpr int ST = (2 * S1 + 3) / 4;                   // TAIL16: k-steps of four of the first layer's tail block (the row is zero up to 4 ST)
    constexpr int KW = TAIL16 ? 4 * ST : 2 * S1;
    constexpr int ld = (KW > 32 * NB1 ? KW : 32 * NB1) | 1;
    constexpr int NBF = TAIL16 ? NB1 - 1 : NB1;            // full 32-column blocks of the first layer
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, i = l & 31, kh = l >> 5;
    float *X = sm + wave * WT * ld;                        // this wave's region: input block, then the hidden layer
    const __amdgpu_buffer_rsrc_t w1 = __builtin_amdgcn_make_buffer_rsrc((void *)w.Wt1m, 0, 2 * S1 * 32 * NB1 * (int)sizeof(float), 0x00020000);
    const __amdgpu_buffer_rsrc_t w2 = __builtin_amdgcn_make_buffer_rsrc((void *)w.Wt2m, 0, 2 * S2 * 32 * NB2 * (int)sizeof(float), 0x00020000);
    // All tile traffic goes through buffer accesses with a per-tile base (scalar arithmetic), the lane offset in one VGPR and the row offset
    // as a scalar; rows past E are clipped by the descriptor (loads return 0, stores are dropped) -- no per-element 64-bit addresses or
    // bounds tests in the vector ALU, whose time adds to the MFMA time.
    auto tile_rsrc = [&](const float *base, int e0, int row_bytes) {
        const int rows = E - e0 < WT ? E - e0 : WT;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(base + (size_t)e0 * (row_bytes / (int)sizeof(float))), 0, rows * row_bytes, 0x00020000);
    };
    float pv[WT][CG];
    float psg = 0.0f;
    auto fetch = [&](int tile) {
        const int e0 = tile * WT;
        const __amdgpu_buffer_rsrc_t sb = tile_rsrc(state, e0, SD * (int)sizeof(float)), gb = tile_rsrc(sign, e0, (int)sizeof(float));
#pragma unroll
        for (int r = 0; r < WT; ++r) {
            // (dword loads, columns l, 64 + l, ...: the b64 / b128 forms of the raw buffer load builtin come out of this hipcc as ONE
            //  buffer_load_dword -- checked in the ISA; a column past the row reads the next row's start and is not deposited)
#pragma unroll
            for (int j = 0; j < CG; ++j)
                pv[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sb, l * 4 + 256 * j, r * SD * (int)sizeof(float), 0));
        }
        psg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gb, l * 4, 0, 0));       // lanes >= 32 read past the tile: 0
    };
    const int stride = gridDim.x * NWAVES;
    int tile = blockIdx.x * NWAVES + wave;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += stride) {
        const int e0 = tile * WT;
#pragma unroll
        for (int r = 0; r < WT; ++r)
#pragma unroll
            for (int j = 0; j < CG; ++j)
                if (64 * j + l < SD) X[r * ld + 64 * j + l] = pv[r][j];
        if (l < WT) {
            X[l * ld + SD] = psg;
#pragma unroll
            for (int c = SD + 1; c < KW; ++c) X[l * ld + c] = 0.0f;       // zero pad up to the k range the chains read
        }
        f32x16 acc[NBF];
        wave_chains<S1, NBF, 32 * NB1>(X + i * ld + kh, w1, w.b1m, acc);
        f32x4 tl[2];
        if constexpr (TAIL16) wave_tail16<ST, 32 * NB1>(X, ld, w1, 32 * NBF, w.b1m, tl);
        // HBM requests go out here, in front of the long activation phase: vector-memory results return in issue order, so a weight load of
        // the next chain issued behind them would otherwise wait for a full HBM round trip
        if (tile + stride < ntiles) fetch(tile + stride);
        float em[16];
        if (emask) {
            const __amdgpu_buffer_rsrc_t eb = tile_rsrc(emask, e0, (int)sizeof(float));
#pragma unroll
            for (int r = 0; r < 16; ++r) em[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(eb, 16 * kh, ((r & 3) + 8 * (r >> 2)) * 4, 0));
        }
        // every chain has consumed its operands (LDS operations of a wave complete in order): the hidden layer replaces the input block
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb) {
            const int col = 32 * nb + i;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 v = pk_logsigmoid_or_zero(acc[nb][r], acc[nb][r + 1], col < w.m1);
                X[acc_row(r, l) * ld + col] = v.x; X[acc_row(r + 1, l) * ld + col] = v.y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (TAIL16) {
            // C layout of the 16x16 block: register r of lane l holds row 4 (l / 16) + r, column l % 16
            const int col = 32 * NBF + (l & 15);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 v = pk_logsigmoid_or_zero(tl[hb][r], tl[hb][r + 1], col < w.m1);
                    X[(16 * hb + 4 * (l >> 4) + r) * ld + col] = v.x; X[(16 * hb + 4 * (l >> 4) + r + 1) * ld + col] = v.y;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x16 ac2[NB2];
        wave_chains<S2, NB2, 32 * NB2>(X + i * ld + kh, w2, nullptr, ac2);
        const int rowb = w.a * (int)sizeof(float);                           // h2 rows are a floats wide
        const __amdgpu_buffer_rsrc_t hb = tile_rsrc(h2out, e0, rowb);
        const int lo = 4 * kh * rowb + i * (int)sizeof(float);
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) {
            if (32 * nb + i < w.a) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 v = pk_logsigmoid((f32x2){ac2[nb][r], ac2[nb][r + 1]});
                    if (emask) v = v * (f32x2){em[r], em[r + 1]};
                    __builtin_amdgcn_raw_buffer_store_b32(f2i(v.x), hb, lo + nb * 128, ((r & 3) + 8 * (r >> 2)) * rowb, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(f2i(v.y), hb, lo + nb * 128, (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * rowb, 0);
                }
            }
        }
    }
}

// ---- kernel 3 with prefetched chains (config 3's shapes) ---------------------------------------------------------------------------------
// (Round 5 measured the same kernel on 32-edge tiles with four waves per workgroup -- six workgroups per CU, barriers over four waves: 6.91 / 6.96
//  -> 6.88 / 6.94 ms per call, noise; not kept.)
// Same workgroup-tile structure as k_agg_post (three workgroups per CU hide its gather / previous-state / store streams), but every wave's
// block is a straight-line wave_chains sequence: both operands prefetched instead of fetched just in time.
template <int S3, int NB3, int S4, int NB4>
__global__ void __launch_bounds__(NTN, 6) k_agg_post_pf(int E, const float *__restrict__ agg, const int32_t *__restrict__ edge_row,
                                                        const float *__restrict__ h2, const float *__restrict__ sign,
                                                        const float *__restrict__ emask, const float *__restrict__ rowmask,
                                                        const float *__restrict__ old, AggW w, float *__restrict__ out)
{
    const uint32_t stop_word = loop_stop_word(rowmask, E);  // a device-driven loop has ended: this sweep writes nothing (k_edge_active); tested behind the hidden layer
    static_assert(NB3 * 2 == NWAVES && NB4 * 2 >= NWAVES, "one 32x32 block per wave in the hidden layer, one or two in the output layer");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int ld0 = 2 * S3 + 1, ld1 = 32 * NB3 + 1;
    float *Rt = sm, *G1 = sm + TM * ld0;
    const int e0 = blockIdx.x * TM;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, i = l & 31, kh = l >> 5;
    // tile rows: a wave takes 8 rows, lane c holds column c (2 S3 <= 64); the row's indices are wave-uniform (scalar loads, scalar address
    // arithmetic) -- the element-indexed loop of k_agg_post spends ~1000 VALU instructions per wave and tile on divisions and 64-bit
    // addresses, and VALU time adds to MFMA time on this chip
    static_assert(2 * S3 <= 64, "one lane per input column");
    // all the tile's HBM requests go out before the first result is used: the gathered rows first (results return in issue order and the
    // hidden layer waits for these), then the previous-state rows of this wave's first output block, which are only needed at the very end
    // (round 6: written with `if (e < E) { ... if (l < w.a) { loads } }` per row, this compiled to s_load / wait / two loads / s_waitcnt vmcnt(1) PER
    //  ROW -- the loads sat in exec-masked blocks whose joins wait, and one register was reused for every row: eight dependent round trips
    //  in front of every tile, the 39 % of parked wave cycles of this kernel.  Now straight-line code: the eight rows' scalar loads, then
    //  sixteen unconditional vector loads on clamped addresses (a lane past the row reads its last column, a row past E reads row E - 1;
    //  both are discarded below))
    float g_hv[TM / NWAVES], g_ag[TM / NWAVES], g_sg[TM / NWAVES], g_em[TM / NWAVES];
    {
        const int lc = l < w.a ? l : w.a - 1;
        const float *const emp = emask ? emask : sign;        // (always a valid address: the value is dropped when there is no edge mask)
        int ec[TM / NWAVES], rowc[TM / NWAVES];
#pragma unroll
        for (int jr = 0; jr < TM / NWAVES; ++jr) {
            const int e = e0 + wave + NWAVES * jr;
            ec[jr] = e < E ? e : E - 1;
            rowc[jr] = edge_row[ec[jr]];
            g_sg[jr] = sign[ec[jr]];
            const float em = emp[ec[jr]];
            g_em[jr] = emask ? em : 1.0f;
        }
#pragma unroll
        for (int jr = 0; jr < TM / NWAVES; ++jr) {
            g_hv[jr] = h2[(size_t)ec[jr] * w.a + lc];
            g_ag[jr] = agg[(size_t)rowc[jr] * w.a + lc];
        }
    }
    const int ROWB = w.out * (int)sizeof(float);
    const int rows = E - e0 < TM ? E - e0 : TM;
    const __amdgpu_buffer_rsrc_t pb = __builtin_amdgcn_make_buffer_rsrc((void *)(old + (size_t)e0 * w.out), 0, rows * ROWB, 0x00020000);
    // QUADS (a 128-wide output: NB4 == 4): the output layer's blocks come out transposed (wave_chains SWAP) -- a lane owns one row and, per
    // register quad, four consecutive columns, so the previous state arrives in four 16-byte loads, the result leaves in four 16-byte stores
    // and the row mask is one load (was: 16 + 16 + 16 dword accesses per block; this kernel issues ~76 MFMAs per wave and tile, so its
    // vector-memory instruction count is what the matrix pipe waits behind)
    constexpr bool QUADS = NB4 == 4;
    float po0[16];
    {
        const int nb4 = wave >> 1, mb4 = wave & 1, col4 = 32 * nb4 + i;
        if constexpr (QUADS) {
            const int lo4 = (32 * mb4 + i) * ROWB + (32 * nb4 + 4 * kh) * (int)sizeof(float);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pb, lo4, q * 32, 0));
#pragma unroll
                for (int c = 0; c < 4; ++c) po0[4 * q + c] = v[c];
            }
        } else {
        const int lo = (32 * mb4 + 4 * kh) * ROWB + col4 * (int)sizeof(float);
#pragma unroll
        for (int r = 0; r < 16; ++r) po0[r] = col4 < w.out ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pb, lo, ((r & 3) + 8 * (r >> 2)) * ROWB, 0)) : 0.0f;
        }
    }
#pragma unroll
    for (int jr = 0; jr < TM / NWAVES; ++jr) {
        const int r = wave + NWAVES * jr, e = e0 + r;
        // (selects, not branches: a load whose only use sits in a conditional block is sunk into it, behind the block's wait)
        const float own = emask ? g_hv[jr] * g_em[jr] : g_hv[jr];
        const float inner = (0.0f + g_ag[jr]) - own;
        float v = (l < w.a) ? inner : ((l == w.a && w.fd) ? g_sg[jr] : 0.0f);
        v = (e < E) ? v : 0.0f;
        if (l < 2 * S3) Rt[r * ld0 + l] = v;
    }
    const int nb = wave >> 1, mb = wave & 1, col = 32 * nb + i;
    const __amdgpu_buffer_rsrc_t w3 = __builtin_amdgcn_make_buffer_rsrc((void *)(w.Wt1a + 32 * nb), 0, (2 * S3 * 32 * NB3 - 32 * nb) * (int)sizeof(float), 0x00020000);
    __syncthreads();
    // hidden layer g wide (100): the waves of the last column block take a 16-column tail (columns 96 .. 111) when that covers the layer --
    // half the MFMA cycles and activations of a full block; they still wait at the barrier, but the two other workgroups of the CU get the
    // issue slots (the output layer reads columns < g only)
    const bool tail16 = nb == NB3 - 1 && w.g <= 32 * (NB3 - 1) + 16 && (2 * S3) % 4 == 0 && !w.no_tail16;
    if (__builtin_amdgcn_readfirstlane(tail16 ? 1 : 0)) {
        const __amdgpu_buffer_rsrc_t w3t = __builtin_amdgcn_make_buffer_rsrc((void *)w.Wt1a, 0, 2 * S3 * 32 * NB3 * (int)sizeof(float), 0x00020000);
        f32x4 tl[2];
        wave_tail16<(2 * S3) / 4, 32 * NB3, 4>(Rt + 32 * mb * ld0, ld0, w3t, 32 * (NB3 - 1), w.b1a, tl);      // (short chunks: this kernel lives in 80 registers)
        const int colt = 32 * (NB3 - 1) + (l & 15);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2 v = pk_logsigmoid_or_zero(tl[hb][r], tl[hb][r + 1], colt < w.g);
                G1[(32 * mb + 16 * hb + 4 * (l >> 4) + r) * ld1 + colt] = v.x; G1[(32 * mb + 16 * hb + 4 * (l >> 4) + r + 1) * ld1 + colt] = v.y;
            }
    } else {
        f32x16 acc[1];
        wave_chains<S3, 1, 32 * NB3>(Rt + (32 * mb + i) * ld0 + kh, w3, w.b1a + 32 * nb, acc);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 v = pk_logsigmoid_or_zero(acc[0][r], acc[0][r + 1], col < w.g);
            G1[(32 * mb + acc_row(r, l)) * ld1 + col] = v.x; G1[(32 * mb + acc_row(r + 1, l)) * ld1 + col] = v.y;
        }
    }
    __syncthreads();
    if (stop_word) return;                                  // (workgroup-uniform; every barrier of the kernel is behind us)
    // blend operands and result through buffer accesses with a per-tile base (lane offset in a VGPR, row offset as a scalar, rows past E
    // clipped by the descriptor); the previous state is requested before the last layer's chain.  The output is w.out = 128 or 150 floats
    // wide: with 5 column blocks (150) the 10 blocks of the tile go round the 8 waves twice.
    const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)e0 * w.out), 0, rows * ROWB, 0x00020000);
    const __amdgpu_buffer_rsrc_t mb_ = __builtin_amdgcn_make_buffer_rsrc((void *)(rowmask ? rowmask + e0 : old), 0, rows * (int)sizeof(float), 0x00020000);
    if constexpr (QUADS) {
        // one block per wave (2 NB4 == NWAVES)
        const int nb4 = wave >> 1, mb4 = wave & 1;
        const __amdgpu_buffer_rsrc_t w4 = __builtin_amdgcn_make_buffer_rsrc((void *)(w.Wt2a + 32 * nb4), 0, (2 * S4 * 32 * NB4 - 32 * nb4) * (int)sizeof(float), 0x00020000);
        const int lo4 = (32 * mb4 + i) * ROWB + (32 * nb4 + 4 * kh) * (int)sizeof(float);
        f32x16 acc[1];
        wave_chains<S4, 1, 32 * NB4, true>(G1 + (32 * mb4 + i) * ld1 + kh, w4, nullptr, acc);
        const float mk = rowmask ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mb_, (32 * mb4 + i) * (int)sizeof(float), 0, 0)) : 1.0f;
        const f32x2 m2 = {mk, mk};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 n0 = pk_logsigmoid((f32x2){acc[0][4 * q], acc[0][4 * q + 1]}), n1 = pk_logsigmoid((f32x2){acc[0][4 * q + 2], acc[0][4 * q + 3]});
            const f32x2 b0 = m2 * n0 + (1.0f - m2) * (f32x2){po0[4 * q], po0[4 * q + 1]}, b1 = m2 * n1 + (1.0f - m2) * (f32x2){po0[4 * q + 2], po0[4 * q + 3]};
            const f32x4 v = {b0.x, b0.y, b1.x, b1.y};
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), ob, lo4, q * 32, 0);
        }
    } else {
    for (int blk = wave; blk < 2 * NB4; blk += NWAVES) {
        const int nb4 = blk >> 1, mb4 = blk & 1, col4 = 32 * nb4 + i;
        const bool live = col4 < w.out;                   // the last block of a 150-wide output has 22 live columns
        const __amdgpu_buffer_rsrc_t w4 = __builtin_amdgcn_make_buffer_rsrc((void *)(w.Wt2a + 32 * nb4), 0, (2 * S4 * 32 * NB4 - 32 * nb4) * (int)sizeof(float), 0x00020000);
        const int lo = (32 * mb4 + 4 * kh) * ROWB + col4 * (int)sizeof(float), lm = (32 * mb4 + 4 * kh) * (int)sizeof(float);
        float po[16], mk[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            po[r] = blk == wave ? po0[r] : (live ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pb, lo, ((r & 3) + 8 * (r >> 2)) * ROWB, 0)) : 0.0f);
        f32x16 acc[1];
        wave_chains<S4, 1, 32 * NB4>(G1 + (32 * mb4 + i) * ld1 + kh, w4, nullptr, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            mk[r] = rowmask ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mb_, lm, ((r & 3) + 8 * (r >> 2)) * (int)sizeof(float), 0)) : 1.0f;
        if (live) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 nv = pk_logsigmoid((f32x2){acc[0][r], acc[0][r + 1]});
                const f32x2 m2 = {mk[r], mk[r + 1]}, bl = m2 * nv + (1.0f - m2) * (f32x2){po[r], po[r + 1]};
                __builtin_amdgcn_raw_buffer_store_b32(f2i(bl.x), ob, lo, ((r & 3) + 8 * (r >> 2)) * ROWB, 0);
                __builtin_amdgcn_raw_buffer_store_b32(f2i(bl.y), ob, lo, (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * ROWB, 0);
            }
        }
    }
    }
}

// ---- kernel 3 with a WAVE as the unit of work (hidden 150) ---------------------------------------------------------------------------------
// The form that made the hidden-150 GRU fast, tried on the post-transform: a wave owns a 32-edge tile (four waves per workgroup, one per
// SIMD, 512 registers per lane), both layers run over all their column blocks as straight-line chains whose weights are requested two
// chunks ahead across the layer and tile boundaries, and everything the tile reads from HBM is requested at the start of an activation
// phase a whole phase before it is used (the gathered rows of the NEXT tile and the previous-state rows of this one in front of the hidden
// layer's logsigmoids, the first two chunks of the next chain's weights in front of them: vector-memory results return in issue order).
// Bit-identical.  At hidden 128 SLOWER than k_agg_post_pf: 8.5 against 7.5 ms per call (at hidden 150, where that kernel's ten output blocks
// take two rounds of its eight waves, 9.8 against 10.2 ms).  Measured with one piece compiled out: MFMA chains + tile
// bookkeeping 4.6 ms (floor 3.1), logsigmoids 2.8, HBM requests 1.1 (their round trip is longer than the 3.6 us activation phase when all
// waves of the chip burst at once; moving the row gather in front of the second activation phase changes nothing), stores 0.3 -- this
// kernel's tile is too short (304 MFMAs, 8 us) for one wave per SIMD to cover its own memory traffic, which six waves per SIMD do for free.
template <int STEPS, int NB, int NP>
struct WaveChains {
    static constexpr int T = STEPS * NB, CH = 8, NC = (T + CH - 1) / CH;
    float aa[2][CH], bb[3][CH];
    __amdgpu_buffer_rsrc_t wr;
    const float *a;            // LDS: row (lane & 31) of the tile, column lane >> 5
    int voff;
    __device__ __forceinline__ float wl(int t) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, voff + (t / STEPS) * 32 * (int)sizeof(float),
                                                                             (t % STEPS) * 2 * NP * (int)sizeof(float), 0));
    }
    __device__ __forceinline__ void head_w() {         // weights of the first two chunks
#pragma unroll
        for (int j = 0; j < CH; ++j) { if (j < T) bb[0][j] = wl(j); if (CH + j < T) bb[1][j] = wl(CH + j); }
    }
    __device__ __forceinline__ void head_a() {         // LDS operands of the first chunk (the tile must be in LDS)
#pragma unroll
        for (int j = 0; j < CH; ++j) if (j < T) aa[0][j] = a[2 * (j % STEPS)];
    }
    __device__ __forceinline__ void run(const float *__restrict__ bias, f32x16 (&acc)[NB]) {
        const int i = threadIdx.x & 31;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float b0 = bias ? bias[32 * nb + i] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = b0;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int t2 = (c + 2) * CH + j, t1 = (c + 1) * CH + j;
                if (t2 < T) bb[(c + 2) % 3][j] = wl(t2);
                if (t1 < T) aa[(c + 1) & 1][j] = a[2 * (t1 % STEPS)];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int t = c * CH + j;
                if (t < T) acc[t / STEPS] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[c & 1][j], bb[c % 3][j], acc[t / STEPS], 0, 0, 0);
            }
            // tie the chunk's accumulators to this point (accumulation registers: see gru_phase2), requests between the MFMAs
            asm volatile("" : "+a"(acc[(c * CH) / STEPS]));
            if ((c * CH + CH - 1) / STEPS != (c * CH) / STEPS && (c * CH + CH - 1) / STEPS < NB) asm volatile("" : "+a"(acc[(c * CH + CH - 1) / STEPS]));
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (c * CH + j < T) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// Four waves per workgroup, one per SIMD (512 registers per lane): separate input / hidden regions, everything of the next tile requested a phase ahead.
#define PW_NW 4
template <int S3, int NB3, int S4, int NB4, bool TAIL16>
__global__ void __launch_bounds__(64 * PW_NW) k_agg_post_wave(int E, const float *__restrict__ agg, int agg_rows, const int32_t *__restrict__ edge_row,
                                                       const float *__restrict__ h2, const float *__restrict__ sign,
                                                       const float *__restrict__ emask, const float *__restrict__ rowmask,
                                                       const float *__restrict__ old, AggW w, float *__restrict__ out, int ntiles /* full 32-edge tiles */)
{
    if (loop_stopped(rowmask, E)) return;                  // a device-driven loop has ended: this sweep writes nothing (k_edge_active)
    static_assert(2 * S3 <= 64, "one lane per input column");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int ld0 = 2 * S3 + 1, ld1 = 32 * NB3 + 1, WR = WT * (ld0 + ld1) + WT;
    static_assert(ld0 <= ld1, "the input block fits the hidden layer's region");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, i = l & 31, kh = l >> 5;
    float *Rt = sm + wave * WR, *G1 = Rt + WT * ld0, *Mk = G1 + WT * ld1;
    const int A = w.a, ROWA = A * (int)sizeof(float), ROWB = w.out * (int)sizeof(float);
    constexpr int NBF = TAIL16 ? NB3 - 1 : NB3;           // full 32-column blocks of the hidden layer (TAIL16: + a 16-column tail, see wave_tail16)
    static_assert(!TAIL16 || (2 * S3) % 4 == 0, "the tail walks the input row in k-steps of four");
    WaveChains<S3, NBF, 32 * NB3> c3;
    WaveChains<S4, NB4, 32 * NB4> c4;
    c3.wr = __builtin_amdgcn_make_buffer_rsrc((void *)w.Wt1a, 0, 2 * S3 * 32 * NB3 * (int)sizeof(float), 0x00020000);
    c4.wr = __builtin_amdgcn_make_buffer_rsrc((void *)w.Wt2a, 0, 2 * S4 * 32 * NB4 * (int)sizeof(float), 0x00020000);
    c3.voff = (kh * 32 * NB3 + i) * (int)sizeof(float); c4.voff = (kh * 32 * NB4 + i) * (int)sizeof(float);
    c3.a = Rt + i * ld0 + kh; c4.a = G1 + i * ld1 + kh;
    const __amdgpu_buffer_rsrc_t ab = __builtin_amdgcn_make_buffer_rsrc((void *)agg, 0, agg_rows * ROWA, 0x00020000);
    auto tile_rsrc = [&](const void *base, int e0, int row_bytes) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)base + (size_t)e0 * row_bytes), 0, WT * row_bytes, 0x00020000);
    };
    // registers of the tile under way: gathered rows (agg, own h2), edge sign / edge mask / row mask / row ids (one per lane < 32)
    float pa[WT], ph[WT], psg = 0.0f, pem = 1.0f, pmk = 1.0f;
    int er_next = 0;                                       // row ids of the tile whose rows are gathered next
    auto fetch_ids = [&](int tile) { er_next = __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc(edge_row, tile * WT, 4), l * 4, 0, 0); };
    auto fetch_rows = [&](int tile) {                      // uses er_next
        const int e0 = tile * WT;
        const __amdgpu_buffer_rsrc_t hb = tile_rsrc(h2, e0, ROWA);
#pragma unroll
        for (int r = 0; r < WT; ++r) {
            const int row = __builtin_amdgcn_readlane(er_next, r);
            pa[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ab, l * 4, row * ROWA, 0));
            ph[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hb, l * 4, r * ROWA, 0));
        }
        psg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc(sign, e0, 4), l * 4, 0, 0));
        if (emask) pem = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc(emask, e0, 4), l * 4, 0, 0));
        if (rowmask) pmk = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc(rowmask, e0, 4), l * 4, 0, 0));
    };
    const bool in_a = l < A, in_s = l == A && w.fd;        // lane = column of the input row: aggregated message, edge sign, zero pad
    auto deposit = [&]() {
        if (l < 2 * S3) {                                  // one divergent region; inside it selects only
#pragma unroll
            for (int r = 0; r < WT; ++r) {
                const float hv = ph[r];
                const float own = emask ? hv * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pem), r)) : hv;
                const float sg = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, psg), r));
                Rt[r * ld0 + l] = in_a ? (0.0f + pa[r]) - own : (in_s ? sg : 0.0f);
            }
            if (l < WT) Mk[l] = pmk;
        }
    };
    const int stride = gridDim.x * PW_NW;
    int tile = blockIdx.x * PW_NW + wave;
    if (tile < ntiles) {
        fetch_ids(tile);
        c3.head_w();
        fetch_rows(tile);
        if (tile + stride < ntiles) fetch_ids(tile + stride);
        deposit();
    }
    for (; tile < ntiles; tile += stride) {
        const int e0 = tile * WT;
        c3.head_a();
        f32x16 acc3[NBF];
        c3.run(w.b1a, acc3);
        f32x4 tl3[2];
        if constexpr (TAIL16) wave_tail16<(2 * S3) / 4, 32 * NB3>(Rt, ld0, c3.wr, 32 * NBF, w.b1a, tl3);
        // in front of the hidden layer's activations: the first weights of the output layer, then everything that comes from HBM
        c4.head_w();
        const __amdgpu_buffer_rsrc_t pb = tile_rsrc(old, e0, ROWB), ob = tile_rsrc(out, e0, ROWB);
        const int lo = 4 * kh * ROWB + i * (int)sizeof(float);
        float po[NB4][16];
        auto load_po = [&](int nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                po[nb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pb, (32 * nb + i < w.out) ? lo + nb * 128 : 0x40000000, ((r & 3) + 8 * (r >> 2)) * ROWB, 0));
        };
#pragma unroll
        for (int nb = 0; nb < NB4; ++nb) load_po(nb);
        const bool more = tile + stride < ntiles;
        if (more) { fetch_rows(tile + stride); if (tile + 2 * stride < ntiles) fetch_ids(tile + 2 * stride); }
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb) {
            const int col = 32 * nb + i;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 v = pk_logsigmoid_or_zero(acc3[nb][r], acc3[nb][r + 1], col < w.g);
                G1[acc_row(r, l) * ld1 + col] = v.x; G1[acc_row(r + 1, l) * ld1 + col] = v.y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (TAIL16) {
            const int colt = 32 * NBF + (l & 15);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 v = pk_logsigmoid_or_zero(tl3[hb][r], tl3[hb][r + 1], colt < w.g);
                    G1[(16 * hb + 4 * (l >> 4) + r) * ld1 + colt] = v.x; G1[(16 * hb + 4 * (l >> 4) + r + 1) * ld1 + colt] = v.y;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        c4.head_a();
        f32x16 acc4[NB4];
        c4.run(nullptr, acc4);
        c3.head_w();                                       // the next tile's first weights, in front of the stores
        float mk[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) mk[r] = rowmask ? Mk[4 * kh + (r & 3) + 8 * (r >> 2)] : 1.0f;
#pragma unroll
        for (int nb = 0; nb < NB4; ++nb) {
            const int so = (32 * nb + i < w.out) ? lo + nb * 128 : 0x40000000;      // a column past the row is stored nowhere
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 nv = pk_logsigmoid((f32x2){acc4[nb][r], acc4[nb][r + 1]});
                const f32x2 m2 = {mk[r], mk[r + 1]}, bl = m2 * nv + (1.0f - m2) * (f32x2){po[nb][r], po[nb][r + 1]};
                __builtin_amdgcn_raw_buffer_store_b32(f2i(bl.x), ob, so, ((r & 3) + 8 * (r >> 2)) * ROWB, 0);
                __builtin_amdgcn_raw_buffer_store_b32(f2i(bl.y), ob, so, (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * ROWB, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) deposit();                               // every chain has consumed its operands (LDS operations of a wave complete in order)
    }
}

// ---- kernel 4: predictor tail on variable tiles (include_self = True) + Perceptron head --------------------------------------------
__global__ void __launch_bounds__(NTN) k_predict_rows(int V, const float *__restrict__ agg, AggW w, HeadW hd, float *__restrict__ pred)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ld0 = w.Kp3 + 1, ld1 = w.Np3 + 1, ld2 = w.Np4 + 1, ld3 = hd.Np + 1;
    float *Rt = sm, *G1 = Rt + TM * ld0, *O = G1 + TM * ld1, *C1 = O + TM * ld2;
    const int v0 = blockIdx.x * TM;
    for (int idx = threadIdx.x; idx < TM * w.Kp3; idx += NTN) {
        const int r = idx / w.Kp3, c = idx % w.Kp3;
        Rt[r * ld0 + c] = (v0 + r < V && c < w.a) ? agg[(size_t)(v0 + r) * w.a + c] : 0.0f;
    }
    __syncthreads();
    lds_layer(Rt, ld0, w.Kp3, w.Wt1a, w.Np3, w.b1a, w.g, ACT_LOGSIGMOID, G1, ld1);
    __syncthreads();
    lds_layer(G1, ld1, w.Kp4, w.Wt2a, w.Np4, nullptr, w.out, ACT_LOGSIGMOID, O, ld2);
    __syncthreads();
    lds_layer(O, ld2, hd.Kp, hd.Wt1, hd.Np, hd.b1, hd.C, ACT_RELU, C1, ld3);
    __syncthreads();
    if (threadIdx.x < TM && v0 + threadIdx.x < V) {
        float acc = 0.0f;
        for (int k = 0; k < hd.C; ++k) acc = fmaf(C1[threadIdx.x * ld3 + k], hd.w2[k], acc);
        pred[v0 + threadIdx.x] = act_apply(acc, hd.out_act);
    }
}

// ---- kernel 4 on the shapes of the shipped configurations (52 -> 100 -> 128, head 128 -> <= 64 -> 1) ------------------------------------
// Same arithmetic as k_predict_rows (every layer the k-ordered fmaf chain of its column), organised like k_agg_post_pf: one 32x32 block per
// wave and layer with both operands prefetched, packed logsigmoids, the hidden layer's last column block as a 16-column tail -- and the head's
// 128 -> 64 layer, which has only four 32x32 blocks for eight waves, as eight pairs of 16x16 blocks on v_mfma_f32_16x16x4_f32 (one pair per
// wave; the instruction adds its four k-steps in ascending order with one rounding each, tools/micro/mfma16_order.hip).  The input tile
// shares its LDS with the output layer's result and the hidden layer with the head's: 66 KB, two workgroups per CU (the generic kernel keeps
// four tiles: 96 KB, one).
template <int S3, int NB3, int S4, int NB4>
__global__ void __launch_bounds__(NTN, 4) k_predict_rows_pf(int V, const float *__restrict__ agg, AggW w, HeadW hd, float *__restrict__ pred)
{
    static_assert(NB3 * 2 == NWAVES && NB4 * 2 == NWAVES, "one 32x32 block per wave in both aggregator layers");
    static_assert(2 * S3 <= 64, "one lane per input column");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int ld0 = 2 * S3 + 1, ld1 = 32 * NB3 + 1, ld2 = 32 * NB4 + 1, ld3 = 64 + 1;
    float *O = sm, *Rt = sm, *G1 = sm + TM * ld2, *C1 = G1;          // Rt is dead when O is written, G1 when C1 is
    const int v0 = blockIdx.x * TM;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, i = l & 31, kh = l >> 5;
    // the tile's rows are one contiguous piece of [V, a]: a wave takes 8 rows, lane c column c
#pragma unroll
    for (int jr = 0; jr < TM / NWAVES; ++jr) {
        const int r = wave + NWAVES * jr, v = v0 + r;
        float x = 0.0f;
        if (v < V && l < w.a) x = agg[(size_t)v * w.a + l];
        if (l < 2 * S3) Rt[r * ld0 + l] = x;
    }
    const int nb = wave >> 1, mb = wave & 1, col = 32 * nb + i;
    // (the predictor's aggregator has no sign column: 50 input columns; the k-steps past the matrix read zeros through the clipped descriptor)
    const __amdgpu_buffer_rsrc_t w3 = __builtin_amdgcn_make_buffer_rsrc((void *)(w.Wt1a + 32 * nb), 0, (w.Kp3 * 32 * NB3 - 32 * nb) * (int)sizeof(float), 0x00020000);
    __syncthreads();
    const bool tail16 = nb == NB3 - 1 && w.g <= 32 * (NB3 - 1) + 16 && (2 * S3) % 4 == 0 && !w.no_tail16;
    f32x4 tl[2];
    if (__builtin_amdgcn_readfirstlane(tail16 ? 1 : 0)) {
        const __amdgpu_buffer_rsrc_t w3t = __builtin_amdgcn_make_buffer_rsrc((void *)w.Wt1a, 0, w.Kp3 * 32 * NB3 * (int)sizeof(float), 0x00020000);
        wave_tail16<(2 * S3) / 4, 32 * NB3, 4>(Rt + 32 * mb * ld0, ld0, w3t, 32 * (NB3 - 1), w.b1a, tl);
        const int colt = 32 * (NB3 - 1) + (l & 15);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2 v = pk_logsigmoid_or_zero(tl[hb][r], tl[hb][r + 1], colt < w.g);
                G1[(32 * mb + 16 * hb + 4 * (l >> 4) + r) * ld1 + colt] = v.x; G1[(32 * mb + 16 * hb + 4 * (l >> 4) + r + 1) * ld1 + colt] = v.y;
            }
        // (the columns 112 .. 127 of this block are never read: the output layer's K is 2 S4 <= 112)
    } else {
        f32x16 acc[1];
        wave_chains<S3, 1, 32 * NB3>(Rt + (32 * mb + i) * ld0 + kh, w3, w.b1a + 32 * nb, acc);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 v = pk_logsigmoid_or_zero(acc[0][r], acc[0][r + 1], col < w.g);
            G1[(32 * mb + acc_row(r, l)) * ld1 + col] = v.x; G1[(32 * mb + acc_row(r + 1, l)) * ld1 + col] = v.y;
        }
    }
    __syncthreads();                                         // G1 complete, Rt dead
    {
        const __amdgpu_buffer_rsrc_t w4 = __builtin_amdgcn_make_buffer_rsrc((void *)(w.Wt2a + 32 * nb), 0, (2 * S4 * 32 * NB4 - 32 * nb) * (int)sizeof(float), 0x00020000);
        f32x16 acc[1];
        wave_chains<S4, 1, 32 * NB4>(G1 + (32 * mb + i) * ld1 + kh, w4, nullptr, acc);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 v = pk_logsigmoid_or_zero(acc[0][r], acc[0][r + 1], col < w.out);
            O[(32 * mb + acc_row(r, l)) * ld2 + col] = v.x; O[(32 * mb + acc_row(r + 1, l)) * ld2 + col] = v.y;
        }
    }
    __syncthreads();                                         // O complete, G1 dead
    {
        // head layer 128 -> hd.Np (= 64) columns: wave = (row half, 16-column group)
        const int rh = wave & 1, cg = wave >> 1;
        const __amdgpu_buffer_rsrc_t w5 = __builtin_amdgcn_make_buffer_rsrc((void *)hd.Wt1, 0, 32 * NB4 * 64 * (int)sizeof(float), 0x00020000);
        wave_tail16<(32 * NB4) / 4, 64>(O + 32 * rh * ld2, ld2, w5, 16 * cg, hd.b1, tl);
        const int colh = 16 * cg + (l & 15);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = tl[hb][r];
                C1[(32 * rh + 16 * hb + 4 * (l >> 4) + r) * ld3 + colh] = (colh < hd.C) ? (v > 0.0f ? v : ((v != v) ? v : 0.0f)) : 0.0f;
            }
    }
    __syncthreads();
    if (threadIdx.x < TM && v0 + threadIdx.x < V) {
        float acc = 0.0f;
        for (int k = 0; k < hd.C; ++k) acc = fmaf(C1[threadIdx.x * ld3 + k], hd.w2[k], acc);
        pred[v0 + threadIdx.x] = act_apply(acc, hd.out_act);
    }
}

// ---- kernel 5: GRU cell on edge tiles ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NTN) k_gru(int E, const float *__restrict__ state, const float *__restrict__ sign,
                                             const float *__restrict__ hprev, const float *__restrict__ rowmask, GruW g, float *__restrict__ out, int ntiles)
{
    if (loop_stopped(rowmask, E)) return;                  // a device-driven loop has ended: this sweep writes nothing (k_edge_active)
    // Persistent over edge tiles: the rows of the next tile are fetched into registers (8 rows per wave, all loads in flight at once)
    // while the six MFMA chains of the current tile run, and dropped into LDS between the two barriers that separate tiles.
    // (Staging the weights through LDS in k-chunks shared by all waves was tried and is slower here: 37 vs 35 ms at config 3 --
    // the kernel is limited by the serial MFMA -> epilogue sequence of each wave at two waves per SIMD, not by the weight stream.)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ldx = g.Kpx + 1, ldh = g.Kph + 1;
    float *X = sm, *Hs = sm + TM * ldx;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    float px[PRE_R][PRE_C], ph[PRE_R][PRE_C];
    auto fetch = [&](int tile) {
        const int e0 = tile * TM;
#pragma unroll
        for (int jr = 0; jr < PRE_R; ++jr) {
            const int e = e0 + wave + NWAVES * jr;
#pragma unroll
            for (int jc = 0; jc < PRE_C; ++jc) {
                const int c = l + 64 * jc;
                float v = 0.0f, hv = 0.0f;
                if (e < E && c < g.Kpx) v = (c < g.dx) ? state[(size_t)e * g.dx + c] : (c == g.dx ? sign[e] : 0.0f);
                if (e < E && c < g.H) hv = hprev[(size_t)e * g.H + c];
                px[jr][jc] = v; ph[jr][jc] = hv;
            }
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int jr = 0; jr < PRE_R; ++jr)
#pragma unroll
            for (int jc = 0; jc < PRE_C; ++jc) {
                const int c = l + 64 * jc, r = wave + NWAVES * jr;
                if (c < g.Kpx) X[r * ldx + c] = px[jr][jc];
                if (c < g.Kph) Hs[r * ldh + c] = ph[jr][jc];
            }
    };
    const int hb = g.Hp / 32;
    const int N3 = 3 * g.Hp;
    int tile = blockIdx.x;
    if (tile < ntiles) { fetch(tile); deposit(); }
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < ntiles) fetch(next);
        const int e0 = tile * TM;
        for (int blk = wave; blk < hb * 2; blk += NWAVES) {
            const int nb = blk >> 1, mb = blk & 1;
            const int col = 32 * nb + (l & 31);
            // gate order r, z, n (torch.nn.GRUCell); input and hidden products stay separate sums (hgates + igates)
            f32x16 ia[1], ha[1], rg, zg;
            mfma_chain<1, 8>(X, ldx, g.Kpx, g.Wt_ih, N3, nb, mb, g.b_ih, ia);
            mfma_chain<1, 8>(Hs, ldh, g.Kph, g.Wt_hh, N3, nb, mb, g.b_hh, ha);
#pragma unroll
            for (int r = 0; r < 16; ++r) rg[r] = pdp_sigmoidf(ha[0][r] + ia[0][r]);
            mfma_chain<1, 8>(X, ldx, g.Kpx, g.Wt_ih, N3, hb + nb, mb, g.b_ih, ia);
            mfma_chain<1, 8>(Hs, ldh, g.Kph, g.Wt_hh, N3, hb + nb, mb, g.b_hh, ha);
#pragma unroll
            for (int r = 0; r < 16; ++r) zg[r] = pdp_sigmoidf(ha[0][r] + ia[0][r]);
            mfma_chain<1, 8>(X, ldx, g.Kpx, g.Wt_ih, N3, 2 * hb + nb, mb, g.b_ih, ia);
            mfma_chain<1, 8>(Hs, ldh, g.Kph, g.Wt_hh, N3, 2 * hb + nb, mb, g.b_hh, ha);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * mb + acc_row(r, l);
                const int e = e0 + row;
                if (e < E && col < g.H) {
                    const float ng = pdp_tanhf_abs(ia[0][r] + ha[0][r] * rg[r]);
                    const float hp = Hs[row * ldh + col];
                    const float hnew = (hp - ng) * zg[r] + ng;
                    const float mk = rowmask ? rowmask[e] : 1.0f;
                    out[(size_t)e * g.H + col] = mk * hnew + (1.0f - mk) * hp;
                }
            }
        }
        __syncthreads();                                   // every wave is done with X / Hs
        if (next < ntiles) deposit();
    }
}

// ---- kernel 5b: the same cell, software pipelined inside each wave (hidden width 128) -------------------------------------------------
// In k_gru the two waves of a SIMD run in lock step (two barriers per tile): both issue their MFMA chains, then both run their
// activation epilogues, so the matrix pipe idles during every epilogue (rocprofv3: MFMA pipe 46 % busy).  Here every gate is one
// straight-line block in which the k-steps of the *next* gate's two chains alternate with slices of the *previous* gate's
// activations (one accumulator register = one slice per 1/16 of the k-steps):
//     r chains  ||  tanh + blend + store of the previous tile      (carried across the tile boundary in registers)
//     z chains  ||  sigmoid of the r gate
//     n chains  ||  sigmoid of the z gate
// The arithmetic per element is unchanged (same chains, same activation functions), so the result is bit-identical to k_gru.
template <int SX, int SH, int NV, class Epi>
__device__ __forceinline__ void gru_phase(const float *xa, const float *ha, __amdgpu_buffer_rsrc_t wi, __amdgpu_buffer_rsrc_t wh,
                                          int voff /* byte offset of this lane's column in a weight row pair */, int wstride /* bytes per k-step */,
                                          float bi, float bh, f32x16 &ai, f32x16 &ah, Epi &&epi)
{
    // weight rows through buffer loads: lane offset in a VGPR, k-step offset as a scalar -- no per-step 64-bit address registers
    auto wload = [&](int s2) -> float {
        return (s2 < SX) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wi, voff, s2 * wstride, 0))
                         : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wh, voff, (s2 - SX) * wstride, 0));
    };
    auto aload = [&](int s2) -> float { return (s2 < SX) ? xa[2 * s2] : ha[2 * (s2 - SX)]; };
    constexpr int S = SX + SH, CH = (S + 15) / 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ai[r] = bi; ah[r] = bh; }
    float bb[2][CH], aa[2][CH];                            // both operands of a chunk are fetched while the previous chunk runs
#pragma unroll
    for (int j = 0; j < CH; ++j)
        if (j < (S / 16)) { bb[0][j] = wload(j); aa[0][j] = aload(j); }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int lo = c * S / 16, hi = (c + 1) * S / 16, hi2 = (c + 2) * S / 16;
        if (c < 15) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int s = hi + j;
                if (s < hi2) { bb[(c + 1) & 1][j] = wload(s); aa[(c + 1) & 1][j] = aload(s); }
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int s = lo + j;
            if (s < hi) {
                if (s < SX) ai = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[c & 1][j], bb[c & 1][j], ai, 0, 0, 0);
                else ah = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[c & 1][j], bb[c & 1][j], ah, 0, 0, 0);
            }
        }
        // activation slices two at a time (every other chunk): the two elements' dependent chains interleave (22.3 -> 22.0 ms; four at a
        // time: 22.1 ms)
        if (c & 1) { epi(c - 1); epi(c); }
        // (pure arithmetic is not ordered against the scheduling barriers by itself: the empty asm statements here and in the
        //  activation slices tie the chunk's results to this point of the instruction stream)
        asm volatile("" : "+v"(ai), "+v"(ah));
        // issue order inside the chunk: one MFMA, NV VALU instructions of the activation slice, ... (measured: few long VALU runs beat many
        // short ones -- NV = 4/3: 26.5 ms, 8/6: 24.8 ms, 26/16: 24.0 ms per call; every VALU run between two dependent MFMAs costs a fixed delay)
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (lo + j < hi) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); if (NV > 0 && (c & 1)) __builtin_amdgcn_sched_group_barrier(0x002, 2 * NV, 0); }
        __builtin_amdgcn_sched_barrier(0);                 // and nothing moves across chunks (keeps the loads of later chunks from piling up)
    }
}

// SAVE (training forward, pdp_train_gru_fused): the gates the adjoint needs leave with the result -- saved [E, 4 H] = r | z | n | gh_n
// (what k_gru_point writes after two GEMMs that store and re-read gi, gh [E, 3 H] each).
template <int SX, bool MASK, bool SAVE = false>
__global__ void __launch_bounds__(NTN) k_gru_pipe(int E, const float *__restrict__ state, const float *__restrict__ sign,
                                                  const float *__restrict__ hprev, const float *__restrict__ rowmask, GruW g,
                                                  float *__restrict__ out, int ntiles /* full tiles only */, float *__restrict__ saved = nullptr)
{
    if (MASK && loop_stopped(rowmask, E)) return;          // a device-driven loop has ended: this sweep writes nothing (k_edge_active)
    constexpr int SH = 64, H = 128;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int ldx = 2 * SX + 1, ldh = 2 * SH + 1;
    // two tile buffers (X, Hs), used alternately: the next tile's rows are dropped into the other buffer as soon as they have arrived, while
    // this tile still computes -- one barrier per tile, and the deposit is off the critical path
    constexpr int TB = TM * (ldx + ldh);
    float *X = sm, *Hs = sm + TM * ldx, *Mk = sm + 2 * TB;        // Mk [3][TM]: row masks of the previous, the current and the next tile
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    // input row = [message, edge sign, zero pad] of 2 SX floats.  WIDE (np-nd-np): the message is H = 128 floats, 512 bytes per edge = one
    // dwordx2 per lane and row, the sign column by wave 0.  Narrow (p-nd-np: 3 or 2 survey columns): lane c loads column c of the row.
    constexpr bool WIDE = 2 * SX - 2 == H;
    static_assert(WIDE || 2 * SX <= 64, "narrow input rows: one lane per column");
    const int dx = g.dx;
    // The next tile's rows come in by LDS-DMA (global_load_lds_dword: no register round trip, no deposit pass), issued by ONE wave per SIMD
    // (waves 0-3, sixteen rows each).  Vector-memory results return in order, so a wave that has an HBM request in flight waits ~2 us at its
    // next weight fragment: when all eight waves fetched, both waves of every SIMD stood there at the start of every tile and the matrix
    // pipe idled (round 6: ~7 % of a tile); now the SIMD's other wave has only L2 weight loads on its counter and keeps the pipe busy.
    // The narrow input row (p-nd-np: 3 or 4 columns) still goes through registers (one dword per row).
    const bool fetcher = wave < 4;
    float pxn[PRE_R];
    float psg = 0.0f;                                     // wave 0: edge sign (WIDE), wave 1: row mask
    auto dma4 = [&](float *lds_row /* wave-uniform */, const float *src /* per lane */) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)lds_row, 4, 0, 0);
    };
    auto fetch = [&](int tile, int par_dst) {
        const int e0 = tile * TM;
        float *Xb = X + par_dst * TB, *Hb = Hs + par_dst * TB;
        if (fetcher) {
#pragma unroll
            for (int jr = 0; jr < TM / 4; ++jr) {
                const int r = wave + 4 * jr;
                const size_t e = (size_t)(e0 + r);
                if constexpr (WIDE) { dma4(Xb + r * ldx, state + e * H + l); dma4(Xb + r * ldx + 64, state + e * H + 64 + l); }
                dma4(Hb + r * ldh, hprev + e * H + l); dma4(Hb + r * ldh + 64, hprev + e * H + 64 + l);
            }
        }
        if constexpr (!WIDE) {
#pragma unroll
            for (int jr = 0; jr < PRE_R; ++jr) {
                const size_t e = (size_t)(e0 + wave + NWAVES * jr);
                pxn[jr] = (l < dx) ? state[e * dx + l] : (l == dx ? sign[e] : 0.0f);
            }
        }
        if (WIDE && wave == 0) psg = sign[e0 + l];
        if (wave == 1) psg = MASK ? rowmask[e0 + l] : 1.0f;
    };
    auto deposit = [&](int par, int mslot) {
        float *Xb = X + par * TB;
        if constexpr (!WIDE) {
#pragma unroll
            for (int jr = 0; jr < PRE_R; ++jr) { const int r = wave + NWAVES * jr; if (l < 2 * SX) Xb[r * ldx + l] = pxn[jr]; }
        }
        if (WIDE && wave == 0) Xb[l * ldx + H] = psg;
        if (wave == 1) Mk[mslot * TM + l] = psg;
    };
    if (threadIdx.x < TM) {
        if (WIDE) { X[threadIdx.x * ldx + H + 1] = 0.0f; X[TB + threadIdx.x * ldx + H + 1] = 0.0f; }   // zero pad column (never overwritten)
        Mk[2 * TM + threadIdx.x] = 0.0f;                    // the "previous tile" row of the first pass (rows 0 and 1 are deposited before they are read)
    }
    const int nb = wave >> 1, mb = wave & 1, i = l & 31, kh = l >> 5;
    const int col = 32 * nb + i;
    const int N3 = 3 * H;
    const int ws = 2 * N3 * (int)sizeof(float);
    const float *xa = X + (32 * mb + i) * ldx + kh, *ha = Hs + (32 * mb + i) * ldh + kh;
    const __amdgpu_buffer_rsrc_t wi = __builtin_amdgcn_make_buffer_rsrc((void *)g.Wt_ih, 0, 2 * SX * N3 * (int)sizeof(float), 0x00020000);
    const __amdgpu_buffer_rsrc_t wh = __builtin_amdgcn_make_buffer_rsrc((void *)g.Wt_hh, 0, 2 * SH * N3 * (int)sizeof(float), 0x00020000);
    const int voff = (kh * N3 + col) * (int)sizeof(float);
    const int ooff = ((32 * mb + 4 * kh) * H + col) * (int)sizeof(float);
    const int soff = ((32 * mb + 4 * kh) * 4 * H + col) * (int)sizeof(float);        // the same element of a saved row (4 H wide)
    const float bir = g.b_ih[col], biz = g.b_ih[H + col], bin = g.b_ih[2 * H + col];
    const float bhr = g.b_hh[col], bhz = g.b_hh[H + col], bhn = g.b_hh[2 * H + col];
    const int row0 = 32 * mb + 4 * kh;                    // acc_row(r, l) = row0 - 32 mb + (r & 3) + 8 (r >> 2)
    f32x16 tq, zg, hq;                                    // carried: tanh argument, update gate, previous hidden value
#pragma unroll
    for (int r = 0; r < 16; ++r) { tq[r] = 0.0f; zg[r] = 0.0f; hq[r] = 0.0f; }
    int tile = blockIdx.x;
    if (tile < ntiles) { fetch(tile, 0); deposit(0, 0); }
    int e_prev = tile * TM;                               // first pass: the slices store zeros where this lane stores its results later
    int par = 0, ms = 0;                                  // tile buffer of the current tile; Mk[ms] = its masks, Mk[(ms + 2) % 3] = the previous tile's
    for (; tile < ntiles; tile += gridDim.x, par ^= 1, ms = (ms + 1) % 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's LDS-DMA requests of the tile have landed; the barrier publishes them
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < ntiles) fetch(next, par ^ 1);            // (nobody reads the other buffer during this tile)
        const float *xa_b = xa + par * TB, *ha_b = ha + par * TB;
        f32x16 ai, ah, rg;
        const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)e_prev * H), 0, TM * H * (int)sizeof(float), 0x00020000);
        const float *mp = Mk + ((ms + 2) % 3) * TM + row0;
        const __amdgpu_buffer_rsrc_t sbp = __builtin_amdgcn_make_buffer_rsrc((void *)(saved + (SAVE ? (size_t)e_prev * 4 * H : 0)), 0, SAVE ? TM * 4 * H * (int)sizeof(float) : 0, 0x00020000);
        gru_phase<SX, SH, 26>(xa_b, ha_b, wi, wh, voff, ws, bir, bhr, ai, ah, [&](int c) {
            const int ro = (c & 3) + 8 * (c >> 2);
            const float ng = pdp_tanhf_abs(tq[c]);
            const float hnew = (hq[c] - ng) * zg[c] + ng;
            const float mk = mp[ro];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, mk * hnew + (1.0f - mk) * hq[c]), ob, ooff, ro * H * (int)sizeof(float), 0);
            if (SAVE) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ng), sbp, soff + 2 * H * (int)sizeof(float), ro * 4 * H * (int)sizeof(float), 0);
        });
#pragma unroll
        for (int r = 0; r < 16; ++r) rg[r] = ah[r] + ai[r];
        if (next < ntiles) deposit(par ^ 1, (ms + 1) % 3);   // nobody reads the other buffer (or that mask row) during this tile
        gru_phase<SX, SH, 16>(xa_b, ha_b, wi, wh, voff + H * (int)sizeof(float), ws, biz, bhz, ai, ah, [&](int c) { float v = pdp_sigmoidf(rg[c]); asm volatile("" : "+v"(v)); rg[c] = v; });
#pragma unroll
        for (int r = 0; r < 16; ++r) zg[r] = ah[r] + ai[r];
        gru_phase<SX, SH, 16>(xa_b, ha_b, wi, wh, voff + 2 * H * (int)sizeof(float), ws, bin, bhn, ai, ah, [&](int c) { float v = pdp_sigmoidf(zg[c]); asm volatile("" : "+v"(v)); zg[c] = v; });
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            tq[r] = ai[r] + ah[r] * rg[r];
            hq[r] = Hs[par * TB + (row0 + (r & 3) + 8 * (r >> 2)) * ldh + col];
        }
        if (SAVE) {
            const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc((void *)(saved + (size_t)tile * TM * 4 * H), 0, TM * 4 * H * (int)sizeof(float), 0x00020000);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = ((r & 3) + 8 * (r >> 2)) * 4 * H * (int)sizeof(float);
                const float vr = rg[r], vz = zg[r], vg = ah[r];     // (scalars first: __builtin_bit_cast of a vector ELEMENT takes element 0 with this compiler)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vr), sb, soff, ro, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vz), sb, soff + H * (int)sizeof(float), ro, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vg), sb, soff + 3 * H * (int)sizeof(float), ro, 0);
            }
        }
        e_prev = tile * TM;
    }
    if (blockIdx.x < ntiles) {
        const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)e_prev * H), 0, TM * H * (int)sizeof(float), 0x00020000);
        const float *mp = Mk + ((ms + 2) % 3) * TM + row0;   // ms was advanced once more when the loop ended
        const __amdgpu_buffer_rsrc_t sbp = __builtin_amdgcn_make_buffer_rsrc((void *)(saved + (SAVE ? (size_t)e_prev * 4 * H : 0)), 0, SAVE ? TM * 4 * H * (int)sizeof(float) : 0, 0x00020000);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int ro = (c & 3) + 8 * (c >> 2);
            const float ng = pdp_tanhf_abs(tq[c]);
            const float hnew = (hq[c] - ng) * zg[c] + ng;
            const float mk = mp[ro];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, mk * hnew + (1.0f - mk) * hq[c]), ob, ooff, ro * H * (int)sizeof(float), 0);
            if (SAVE) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ng), sbp, soff + 2 * H * (int)sizeof(float), ro * 4 * H * (int)sizeof(float), 0);
        }
    }
}

#ifdef PDP_FAST_MATH
// ---- kernel 5b', opt-in fast build only: the hidden-128 cell on three-term bf16 products -------------------------------------------------
// The f32 MFMA runs at the f32 vector rate: a gate of this cell is 129 v_mfma_f32_32x32x2_f32 per wave.  Here both operands are split into
// bf16 high + low parts (x = hi + lo, 16 mantissa bits together) and a k-block of 16 is hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_bf16
// with fp32 accumulation: 51 instructions per gate.  tools/micro/bf16x3_probe.hip: 3.7e-6 of the largest |C| on a 32 x 32 x 256 product
// (the fp32 chain: 4.8e-7; operands merely rounded to bf16: 1.9e-3) -- inside the tolerances the fast build is gated with
// (tests/test_fast_build_gpu.py), never in the parity build, whose oracle chain is the fp32 fma.
// Layout: the tile rows are split ONCE, when they are deposited in LDS (hi and lo bf16 take the 4 bytes the float took), and every wave of
// the tile reads ready 16-byte fragments; the weights are split per call into [k / 8][3 H][8] so that a lane's fragment is one 16-byte load.
// The previous hidden value of the blend is the caller's fp32 row (re-read from L2), not the split one.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define BF3_XS 152        /* bf16 per row of the input part: 144 used (128 message + sign + zero pad), row stride 76 dwords = 19 x 16 bytes (odd: no 16-byte bank conflicts) */
#define BF3_HS 136        /* bf16 per row of the hidden part: 128 used, 68 dwords = 17 x 16 bytes */
#define BF3_KX 144
__device__ __forceinline__ uint32_t bf3_pack_hi(float a, float b, float &ra, float &rb)
{
    const __bf16 ha = (__bf16)a, hb = (__bf16)b;            // round to nearest even
    ra = a - (float)ha; rb = b - (float)hb;                  // exact: the residual of a rounding fits a float
    return (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
}
__device__ __forceinline__ uint32_t bf3_pack(float a, float b)
{
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)a) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)b) << 16);
}
// Wt [K][N3] fp32 (k-major, the layout of the f32 kernels) -> hi / lo [Kpad / 8][N3][8] bf16, zero rows behind K
__global__ void k_bf3_split_weights(const float *__restrict__ Wt, int K, int Kpad, int N3, uint32_t *__restrict__ hi, uint32_t *__restrict__ lo)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (Kpad / 8) * N3) return;
    const int k8 = idx / N3, col = idx - k8 * N3;
    uint32_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k0 = 8 * k8 + 2 * j;
        const float a = k0 < K ? Wt[(size_t)k0 * N3 + col] : 0.0f, b = k0 + 1 < K ? Wt[(size_t)(k0 + 1) * N3 + col] : 0.0f;
        float ra, rb;
        h[j] = bf3_pack_hi(a, b, ra, rb); l[j] = bf3_pack(ra, rb);
    }
    reinterpret_cast<uint4 *>(hi)[idx] = make_uint4(h[0], h[1], h[2], h[3]);
    reinterpret_cast<uint4 *>(lo)[idx] = make_uint4(l[0], l[1], l[2], l[3]);
}

// WIDE: the 129-wide input of np-nd-np (message + sign), split like the hidden part.  Narrow (p-nd-np: 2 or 3 survey columns + sign = K 4): the
// input part stays two f32 MFMA steps on an fp32 tile column block (nothing to gain on K = 4), the hidden part runs on the split products.
template <bool WIDE, bool MASK>
__global__ void __launch_bounds__(NTN) k_gru_bf3(int E, const float *__restrict__ state, const float *__restrict__ sign, const float *__restrict__ hprev,
                                                 const float *__restrict__ rowmask, const uint32_t *__restrict__ wxh, const uint32_t *__restrict__ wxl,
                                                 const uint32_t *__restrict__ whh, const uint32_t *__restrict__ whl, const float *__restrict__ Wt_ih, int dx,
                                                 const float *__restrict__ b_ih, const float *__restrict__ b_hh, float *__restrict__ out, int ntiles /* full tiles only */)
{
    if (MASK && loop_stopped(rowmask, E)) return;          // a device-driven loop has ended: this sweep writes nothing (k_edge_active)
    constexpr int H = 128, N3 = 3 * H;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // two tile buffers, used alternately (as in k_gru_pipe); per buffer Xh | Xl | Hh | Hl (narrow: an fp32 [TM][5] block in place of Xh | Xl)
    constexpr int XW = WIDE ? TM * BF3_XS : 320;             // dwords of the input part of a buffer
    constexpr int TBW = XW + TM * BF3_HS;                    // dwords per buffer
    uint32_t *const T32 = reinterpret_cast<uint32_t *>(sm);
    float *const Mk = sm + 2 * TBW;                          // [3][TM] row masks of the previous, the current and the next tile
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    // the next tile's rows travel in two halves (rows jr < 4 during the first gate, the others during the second): 256 registers per wave
    // hold the accumulators, the carried gates and the operand fragments, and any spill costs the launch a scratch allocation
    constexpr int PH = PRE_R / 2;
    float2 px[PH], ph[PH];
    float psg = 0.0f;
    auto fetch = [&](int tile, int half) {
        const int e0 = tile * TM;
#pragma unroll
        for (int jr = 0; jr < PH; ++jr) {
            const size_t e = (size_t)(e0 + wave + NWAVES * (jr + PH * half));
            if constexpr (WIDE) px[jr] = reinterpret_cast<const float2 *>(state + e * H)[l];
            else px[jr].x = (l < dx) ? state[e * dx + l] : (l == dx ? sign[e] : 0.0f);
            ph[jr] = reinterpret_cast<const float2 *>(hprev + e * H)[l];
        }
        if (half == 0) {
            if (WIDE && wave == 0) psg = sign[e0 + l];
            if (wave == 1) psg = MASK ? rowmask[e0 + l] : 1.0f;
        }
    };
    auto deposit = [&](int par, int mslot, int half) {
        uint32_t *Xh = T32 + par * TBW, *Xl = Xh + TM * (BF3_XS / 2), *Hh = T32 + par * TBW + XW, *Hl = Hh + TM * (BF3_HS / 2);
#pragma unroll
        for (int jr = 0; jr < PH; ++jr) {
            const int r = wave + NWAVES * (jr + PH * half);
            float ra, rb;
            if constexpr (WIDE) { Xh[r * (BF3_XS / 2) + l] = bf3_pack_hi(px[jr].x, px[jr].y, ra, rb); Xl[r * (BF3_XS / 2) + l] = bf3_pack(ra, rb); }
            else { if (l < 4) reinterpret_cast<float *>(Xh)[r * 5 + l] = px[jr].x; }
            Hh[r * (BF3_HS / 2) + l] = bf3_pack_hi(ph[jr].x, ph[jr].y, ra, rb); Hl[r * (BF3_HS / 2) + l] = bf3_pack(ra, rb);
        }
        if (half == 0) {
            if (WIDE && wave == 0) Xh[l * (BF3_XS / 2) + H / 2] = bf3_pack(psg, 0.0f);    // columns 128 (the sign: +-1 is a bf16) and 129
            if (wave == 1) Mk[mslot * TM + l] = psg;
        }
    };
    if constexpr (WIDE) {
        // the pad columns 130 .. 151 of both parts and the low part of the sign column: zero, never overwritten
        for (int idx = threadIdx.x; idx < 2 * TM * 12; idx += NTN) {
            const int b = idx / (TM * 12), r = (idx / 12) % TM, c = idx % 12;
            uint32_t *Xh = T32 + b * TBW, *Xl = Xh + TM * (BF3_XS / 2);
            if (c > 0) Xh[r * (BF3_XS / 2) + H / 2 + c] = 0u;
            Xl[r * (BF3_XS / 2) + H / 2 + c] = 0u;
        }
    }
    if (threadIdx.x < TM) Mk[2 * TM + threadIdx.x] = 0.0f;
    const int nb = wave >> 1, mb = wave & 1, i = l & 31, kq = l >> 5;
    const int col = 32 * nb + i;
    const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void *)wxh, 0, WIDE ? (BF3_KX / 8) * N3 * 16 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void *)wxl, 0, WIDE ? (BF3_KX / 8) * N3 * 16 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rhh = __builtin_amdgcn_make_buffer_rsrc((void *)whh, 0, (H / 8) * N3 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rhl = __builtin_amdgcn_make_buffer_rsrc((void *)whl, 0, (H / 8) * N3 * 16, 0x00020000);
    const int wstep = 2 * N3 * 16;                            // bytes per k-block of 16 (two k8 rows)
    const int ooff = ((32 * mb + 4 * kq) * H + col) * (int)sizeof(float);
    const float bir = b_ih[col], biz = b_ih[H + col], bin = b_ih[2 * H + col];
    const float bhr = b_hh[col], bhz = b_hh[H + col], bhn = b_hh[2 * H + col];
    const int row0 = 32 * mb + 4 * kq;
    // narrow: the four weight values of this lane per gate (k = kq and 2 + kq), kept in registers
    float wn[3][2];
    if constexpr (!WIDE) {
#pragma unroll
        for (int g = 0; g < 3; ++g) { wn[g][0] = Wt_ih[(size_t)kq * N3 + g * H + col]; wn[g][1] = Wt_ih[(size_t)(2 + kq) * N3 + g * H + col]; }
    }
    f32x16 tq, zg;                                         // carried: tanh argument, update gate (the previous hidden value is re-read per slice)
#pragma unroll
    for (int r = 0; r < 16; ++r) { tq[r] = 0.0f; zg[r] = 0.0f; }
    int tile = blockIdx.x;
    if (tile < ntiles) { fetch(tile, 0); deposit(0, 0, 0); fetch(tile, 1); deposit(0, 0, 1); }
    int e_prev = tile * TM;
    int par = 0, ms = 0;
    for (; tile < ntiles; tile += gridDim.x, par ^= 1, ms = (ms + 1) % 3) {
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < ntiles) fetch(next, 0);
        const uint16_t *Xh = reinterpret_cast<const uint16_t *>(T32 + par * TBW), *Xl = Xh + TM * BF3_XS;
        const uint16_t *Hh = reinterpret_cast<const uint16_t *>(T32 + par * TBW + XW), *Hl = Hh + TM * BF3_HS;
        const uint16_t *axh = Xh + (32 * mb + i) * BF3_XS + 8 * kq, *axl = Xl + (32 * mb + i) * BF3_XS + 8 * kq;
        const uint16_t *ahh = Hh + (32 * mb + i) * BF3_HS + 8 * kq, *ahl = Hl + (32 * mb + i) * BF3_HS + 8 * kq;
        const float *xf = reinterpret_cast<const float *>(Xh) + (32 * mb + i) * 5 + kq;
        f32x16 ai, ah, rg;
        // one gate: 9 k-blocks of the input part and 8 of the hidden part, alternating (two accumulators: no MFMA waits for the one before
        // it), weight fragments one block ahead, two activation slices of the previous gate per block pair
        auto phase = [&](int gate, float bi, float bh, auto &&epi) {
            const int voff = (kq * N3 + gate * H + col) * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ai[r] = bi; ah[r] = bh; }
            // the weight fragments (L2) are requested one block ahead, the tile fragments (LDS) when the block before has issued: the
            // activation slices that follow cover the LDS round trip, and the registers of a second copy are not there (256 per wave)
            bf16x8 a_xh, a_xl, a_hh, a_hl, b_xh[2], b_xl[2], b_hh[2], b_hl[2];
            constexpr int NSX = WIDE ? BF3_KX / 16 : 0, NS = WIDE ? BF3_KX / 16 : H / 16;
            auto loadB = [&](int s, int q) {
                if (s < NSX) {
                    b_xh[q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rxh, voff, s * wstep, 0));
                    b_xl[q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rxl, voff, s * wstep, 0));
                }
                if (s < H / 16) {
                    b_hh[q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rhh, voff, s * wstep, 0));
                    b_hl[q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rhl, voff, s * wstep, 0));
                }
            };
            auto loadA = [&](int s) {
                if (s < NSX) { a_xh = *reinterpret_cast<const bf16x8 *>(axh + 16 * s); a_xl = *reinterpret_cast<const bf16x8 *>(axl + 16 * s); }
                if (s < H / 16) { a_hh = *reinterpret_cast<const bf16x8 *>(ahh + 16 * s); a_hl = *reinterpret_cast<const bf16x8 *>(ahl + 16 * s); }
            };
            loadB(0, 0); loadA(0);
            if constexpr (!WIDE) {
                ai = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[0], wn[gate][0], ai, 0, 0, 0);
                ai = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[2], wn[gate][1], ai, 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int q = s & 1;
                if (s + 1 < NS) loadB(s + 1, q ^ 1);
                const bool xpart = s < NSX, hpart = s < H / 16;
                if (xpart) ai = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_xh, b_xh[q], ai, 0, 0, 0);
                if (hpart) ah = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hh, b_hh[q], ah, 0, 0, 0);
                if (xpart) ai = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_xh, b_xl[q], ai, 0, 0, 0);
                if (hpart) ah = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hh, b_hl[q], ah, 0, 0, 0);
                if (xpart) ai = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_xl, b_xh[q], ai, 0, 0, 0);
                if (hpart) ah = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hl, b_hh[q], ah, 0, 0, 0);
                asm volatile("" : "+v"(ai), "+v"(ah));
                if (s + 1 < NS) loadA(s + 1);
                if (hpart) { epi(2 * s); epi(2 * s + 1); }
                __builtin_amdgcn_sched_barrier(0);            // nothing moves across blocks: later blocks' loads would pile up in registers
            }
        };
        const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)e_prev * H), 0, TM * H * (int)sizeof(float), 0x00020000);
        // the previous tile's hidden rows (fp32, from L2): slice c's value is requested two slices earlier
        const __amdgpu_buffer_rsrc_t hb = __builtin_amdgcn_make_buffer_rsrc((void *)(hprev + (size_t)e_prev * H), 0, TM * H * (int)sizeof(float), 0x00020000);
        auto hload = [&](int c) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hb, ooff, ((c & 3) + 8 * (c >> 2)) * H * (int)sizeof(float), 0)); };
        float hq2[2] = {hload(0), hload(1)};
        const float *mp = Mk + ((ms + 2) % 3) * TM + row0;
        phase(0, bir, bhr, [&](int c) {
            const int ro = (c & 3) + 8 * (c >> 2);
            const float hqc = hq2[c & 1];
            if (c + 2 < 16) hq2[c & 1] = hload(c + 2);
            const float ng = pdp_tanhf_abs(tq[c]);
            const float hnew = (hqc - ng) * zg[c] + ng;
            const float mk = mp[ro];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, mk * hnew + (1.0f - mk) * hqc), ob, ooff, ro * H * (int)sizeof(float), 0);
        });
#pragma unroll
        for (int r = 0; r < 16; ++r) rg[r] = ah[r] + ai[r];
        if (next < ntiles) { deposit(par ^ 1, (ms + 1) % 3, 0); fetch(next, 1); }
        phase(1, biz, bhz, [&](int c) { float v = pdp_sigmoidf(rg[c]); asm volatile("" : "+v"(v)); rg[c] = v; });
#pragma unroll
        for (int r = 0; r < 16; ++r) zg[r] = ah[r] + ai[r];
        if (next < ntiles) deposit(par ^ 1, (ms + 1) % 3, 1);
        phase(2, bin, bhn, [&](int c) { float v = pdp_sigmoidf(zg[c]); asm volatile("" : "+v"(v)); zg[c] = v; });
#pragma unroll
        for (int r = 0; r < 16; ++r) tq[r] = ai[r] + ah[r] * rg[r];
        e_prev = tile * TM;
    }
    if (blockIdx.x < ntiles) {
        const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)e_prev * H), 0, TM * H * (int)sizeof(float), 0x00020000);
        const __amdgpu_buffer_rsrc_t hb = __builtin_amdgcn_make_buffer_rsrc((void *)(hprev + (size_t)e_prev * H), 0, TM * H * (int)sizeof(float), 0x00020000);
        const float *mp = Mk + ((ms + 2) % 3) * TM + row0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int ro = (c & 3) + 8 * (c >> 2);
            const float hqc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hb, ooff, ro * H * (int)sizeof(float), 0));
            const float ng = pdp_tanhf_abs(tq[c]);
            const float hnew = (hqc - ng) * zg[c] + ng;
            const float mk = mp[ro];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, mk * hnew + (1.0f - mk) * hqc), ob, ooff, ro * H * (int)sizeof(float), 0);
        }
    }
}

// ---- the aggregator's second half (hidden 128) on three-term bf16 products, fast build only ---------------------------------------------
// k_agg_post_pf's workgroup tile (64 edges, 8 waves, three workgroups per CU) with both layers as split products: the 52-wide input block
// and the 100-wide hidden layer live in LDS as bf16 high + low parts (the bytes of the float), every wave's block is 4 (7) k-blocks of 16
// x three matrix instructions instead of 26 (50) f32 steps.  The 16-column tail of the f32 kernel is not needed: a full 32-column block of
// the hidden layer is 12 instructions here.  Output block transposed (operands swapped) for 16-byte stores, as in the f32 kernel.
#define BF3_RS 72         /* bf16 per row of the input block: 64 used (52 + zero pad), 36 dwords = 9 x 16 bytes */
#define BF3_GS 120        /* bf16 per row of the hidden layer: 112 used (100 + zero pad), 60 dwords = 15 x 16 bytes */
template <int NS, bool SWAP>
__device__ __forceinline__ f32x16 bf3_block(const uint16_t *ah, const uint16_t *al, __amdgpu_buffer_rsrc_t wh, __amdgpu_buffer_rsrc_t wl, int voff, int wstep, float bias)
{
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias;
    bf16x8 bh[2], bl[2];
    bh[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wh, voff, 0, 0));
    bl[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wl, voff, 0, 0));
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int q = s & 1;
        if (s + 1 < NS) {
            bh[q ^ 1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wh, voff, (s + 1) * wstep, 0));
            bl[q ^ 1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wl, voff, (s + 1) * wstep, 0));
        }
        const bf16x8 xh = *reinterpret_cast<const bf16x8 *>(ah + 16 * s), xl = *reinterpret_cast<const bf16x8 *>(al + 16 * s);
        if (SWAP) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[q], xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[q], xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[q], xl, acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh[q], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl[q], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh[q], acc, 0, 0, 0);
        }
    }
    return acc;
}
// two column blocks of one row block at once: the tile fragments are read once per k-block and the two accumulators alternate, so that no
// matrix instruction waits for the one before it (a wave alone on its chains: the wave-private kernels run two waves per SIMD)
template <int NS, bool SWAP>
__device__ __forceinline__ void bf3_block2(const uint16_t *ah, const uint16_t *al, __amdgpu_buffer_rsrc_t wh, __amdgpu_buffer_rsrc_t wl, int voff0, int voff1, int wstep,
                                           f32x16 &acc0, f32x16 &acc1)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    bf16x8 bh[2][2], bl[2][2];
    auto loadB = [&](int s, int q) {
        bh[q][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wh, voff0, s * wstep, 0));
        bl[q][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wl, voff0, s * wstep, 0));
        bh[q][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wh, voff1, s * wstep, 0));
        bl[q][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wl, voff1, s * wstep, 0));
    };
    loadB(0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int q = s & 1;
        if (s + 1 < NS) loadB(s + 1, q ^ 1);
        const bf16x8 xh = *reinterpret_cast<const bf16x8 *>(ah + 16 * s), xl = *reinterpret_cast<const bf16x8 *>(al + 16 * s);
        if (SWAP) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[q][0], xh, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[q][1], xh, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[q][0], xh, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[q][1], xh, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[q][0], xl, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[q][1], xl, acc1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh[q][0], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh[q][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl[q][0], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl[q][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh[q][0], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh[q][1], acc1, 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void bf3_store1(uint16_t *hi, uint16_t *lo, float v)
{
    const __bf16 h = (__bf16)v;
    *hi = __builtin_bit_cast(uint16_t, h); *lo = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)h));
}
__global__ void __launch_bounds__(NTN, 6) k_agg_post_bf3(int E, const float *__restrict__ agg, const int32_t *__restrict__ edge_row,
                                                         const float *__restrict__ h2, const float *__restrict__ sign, const float *__restrict__ emask,
                                                         const float *__restrict__ rowmask, const float *__restrict__ old, AggW w,
                                                         const uint32_t *__restrict__ w3h, const uint32_t *__restrict__ w3l,
                                                         const uint32_t *__restrict__ w4h, const uint32_t *__restrict__ w4l, float *__restrict__ out)
{
    if (loop_stopped(rowmask, E)) return;                  // a device-driven loop has ended: this sweep writes nothing (k_edge_active)
    constexpr int K3 = 64, K4 = 112, N = 128;               // padded k ranges, columns of both layers (100 -> 128, 128)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    uint16_t *Rh = reinterpret_cast<uint16_t *>(sm), *Rl = Rh + TM * BF3_RS, *Gh = Rl + TM * BF3_RS, *Gl = Gh + TM * BF3_GS;
    const int e0 = blockIdx.x * TM;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, i = l & 31, kq = l >> 5;
    float g_hv[TM / NWAVES], g_ag[TM / NWAVES], g_sg[TM / NWAVES], g_em[TM / NWAVES];
#pragma unroll
    for (int jr = 0; jr < TM / NWAVES; ++jr) {
        const int e = e0 + wave + NWAVES * jr;
        g_hv[jr] = 0.0f; g_ag[jr] = 0.0f; g_sg[jr] = 0.0f; g_em[jr] = 1.0f;
        if (e < E) {
            const int row = edge_row[e];
            g_sg[jr] = sign[e];
            if (emask) g_em[jr] = emask[e];
            if (l < w.a) { g_hv[jr] = h2[(size_t)e * w.a + l]; g_ag[jr] = agg[(size_t)row * w.a + l]; }
        }
    }
    const int ROWB = w.out * (int)sizeof(float);
    const int rows = E - e0 < TM ? E - e0 : TM;
    const __amdgpu_buffer_rsrc_t pb = __builtin_amdgcn_make_buffer_rsrc((void *)(old + (size_t)e0 * w.out), 0, rows * ROWB, 0x00020000);
    const int nb = wave >> 1, mb = wave & 1, col = 32 * nb + i;
    const int lo4 = (32 * mb + i) * ROWB + (32 * nb + 4 * kq) * (int)sizeof(float);
    float po0[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pb, lo4, q * 32, 0));
#pragma unroll
        for (int c = 0; c < 4; ++c) po0[4 * q + c] = v[c];
    }
#pragma unroll
    for (int jr = 0; jr < TM / NWAVES; ++jr) {
        const int r = wave + NWAVES * jr, e = e0 + r;
        float v = 0.0f;
        if (e < E) {
            if (l < w.a) {
                const float own = emask ? g_hv[jr] * g_em[jr] : g_hv[jr];
                v = (0.0f + g_ag[jr]) - own;
            } else if (l == w.a && w.fd) v = g_sg[jr];
        }
        bf3_store1(Rh + r * BF3_RS + l, Rl + r * BF3_RS + l, v);            // columns 52 .. 63: zero
    }
    __syncthreads();
    {
        const __amdgpu_buffer_rsrc_t r3h = __builtin_amdgcn_make_buffer_rsrc((void *)w3h, 0, (K3 / 8) * N * 16, 0x00020000);
        const __amdgpu_buffer_rsrc_t r3l = __builtin_amdgcn_make_buffer_rsrc((void *)w3l, 0, (K3 / 8) * N * 16, 0x00020000);
        const float b0 = col < w.g ? w.b1a[col] : 0.0f;
        const f32x16 acc = bf3_block<K3 / 16, false>(Rh + (32 * mb + i) * BF3_RS + 8 * kq, Rl + (32 * mb + i) * BF3_RS + 8 * kq, r3h, r3l, (kq * N + col) * 16, 2 * N * 16, b0);
        if (col < K4) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 v = pk_logsigmoid_or_zero(acc[r], acc[r + 1], col < w.g);
                bf3_store1(Gh + (32 * mb + acc_row(r, l)) * BF3_GS + col, Gl + (32 * mb + acc_row(r, l)) * BF3_GS + col, v.x);
                bf3_store1(Gh + (32 * mb + acc_row(r + 1, l)) * BF3_GS + col, Gl + (32 * mb + acc_row(r + 1, l)) * BF3_GS + col, v.y);
            }
        }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)e0 * w.out), 0, rows * ROWB, 0x00020000);
    const __amdgpu_buffer_rsrc_t mb_ = __builtin_amdgcn_make_buffer_rsrc((void *)(rowmask ? rowmask + e0 : old), 0, rows * (int)sizeof(float), 0x00020000);
    const __amdgpu_buffer_rsrc_t r4h = __builtin_amdgcn_make_buffer_rsrc((void *)w4h, 0, (K4 / 8) * N * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r4l = __builtin_amdgcn_make_buffer_rsrc((void *)w4l, 0, (K4 / 8) * N * 16, 0x00020000);
    // operands swapped: lane (i, kq) holds ROW 32 mb + i of the tile, register 4 q + c the column 32 nb + 8 q + 4 kq + c
    const f32x16 acc = bf3_block<K4 / 16, true>(Gh + (32 * mb + i) * BF3_GS + 8 * kq, Gl + (32 * mb + i) * BF3_GS + 8 * kq, r4h, r4l, (kq * N + col) * 16, 2 * N * 16, 0.0f);
    const float mk = rowmask ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mb_, (32 * mb + i) * (int)sizeof(float), 0, 0)) : 1.0f;
    const f32x2 m2 = {mk, mk};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2 n0 = pk_logsigmoid((f32x2){acc[4 * q], acc[4 * q + 1]}), n1 = pk_logsigmoid((f32x2){acc[4 * q + 2], acc[4 * q + 3]});
        const f32x2 b0 = m2 * n0 + (1.0f - m2) * (f32x2){po0[4 * q], po0[4 * q + 1]}, b1 = m2 * n1 + (1.0f - m2) * (f32x2){po0[4 * q + 2], po0[4 * q + 3]};
        const f32x4 v = {b0.x, b0.y, b1.x, b1.y};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), ob, lo4, q * 32, 0);
    }
}

// ---- the aggregator's first half (hidden 128) on three-term bf16 products, fast build only -----------------------------------------------
// k_agg_pre_wave's form -- a wave owns a 32-edge tile from its input rows to the stored [E, 50] result, no workgroup barrier -- with the
// 130-wide input block and the 100-wide hidden layer as bf16 high + low parts in the wave's LDS region (the hidden layer replaces the input
// block, as there).  First layer: four 32-column blocks x 9 k-blocks x three matrix instructions, transposed (operands swapped) so that a
// lane holds four consecutive columns of a row and the split hidden values leave in 8-byte stores; second layer two blocks x 7 k-blocks.
__global__ void __launch_bounds__(NTN) k_agg_pre_bf3(int E, const float *__restrict__ state, const float *__restrict__ sign, const float *__restrict__ emask, AggW w,
                                                     const uint32_t *__restrict__ w1h, const uint32_t *__restrict__ w1l, const uint32_t *__restrict__ w2h,
                                                     const uint32_t *__restrict__ w2l, float *__restrict__ h2out, int ntiles)
{
    constexpr int SD = 128, K1 = BF3_KX, K2 = 112, N1 = 128, N2 = 64;
    constexpr int REG = WT * BF3_XS * 2;                     // dwords of a wave's region: input block hi | lo (the hidden layer, 2 x WT x BF3_GS bf16, is smaller)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, i = l & 31, kq = l >> 5;
    uint32_t *const X32 = reinterpret_cast<uint32_t *>(sm) + wave * (REG / 2);
    uint16_t *const Xh = reinterpret_cast<uint16_t *>(X32), *const Xl = Xh + WT * BF3_XS;
    uint16_t *const Hh = Xh, *const Hl = Xh + WT * BF3_GS;
    const __amdgpu_buffer_rsrc_t r1h = __builtin_amdgcn_make_buffer_rsrc((void *)w1h, 0, (K1 / 8) * N1 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1l = __builtin_amdgcn_make_buffer_rsrc((void *)w1l, 0, (K1 / 8) * N1 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2h = __builtin_amdgcn_make_buffer_rsrc((void *)w2h, 0, (K2 / 8) * N2 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2l = __builtin_amdgcn_make_buffer_rsrc((void *)w2l, 0, (K2 / 8) * N2 * 16, 0x00020000);
    auto tile_rsrc = [&](const float *base, int e0, int row_bytes) {
        const int rows = E - e0 < WT ? E - e0 : WT;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(base + (size_t)e0 * (row_bytes / (int)sizeof(float))), 0, rows * row_bytes, 0x00020000);
    };
    float2 pv[WT];
    float psg = 0.0f;
    auto fetch = [&](int tile) {
        const int e0 = tile * WT;
#pragma unroll
        for (int r = 0; r < WT; ++r) pv[r] = (e0 + r < E) ? reinterpret_cast<const float2 *>(state + (size_t)(e0 + r) * SD)[l] : make_float2(0.0f, 0.0f);
        psg = (l < WT && e0 + l < E) ? sign[e0 + l] : 0.0f;
    };
    // first layer's bias per register of the transposed block: column 32 nb + 8 q + 4 kq + c
    const int stride = gridDim.x * NWAVES;
    int tile = blockIdx.x * NWAVES + wave;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += stride) {
        const int e0 = tile * WT;
#pragma unroll
        for (int r = 0; r < WT; ++r) {
            float ra, rb;
            X32[r * (BF3_XS / 2) + l] = bf3_pack_hi(pv[r].x, pv[r].y, ra, rb);
            X32[WT * (BF3_XS / 2) + r * (BF3_XS / 2) + l] = bf3_pack(ra, rb);
        }
        if (l < WT) {
            // columns 128 (the sign: +-1 is a bf16) .. 143: the hidden layer of the previous tile stood here
            uint4 z = make_uint4(bf3_pack(psg, 0.0f), 0u, 0u, 0u), zz = make_uint4(0u, 0u, 0u, 0u);
            uint4 *hrow = reinterpret_cast<uint4 *>(Xh + l * BF3_XS + SD), *lrow = reinterpret_cast<uint4 *>(Xl + l * BF3_XS + SD);
            hrow[0] = z; hrow[1] = zz; lrow[0] = zz; lrow[1] = zz;
        }
        f32x16 acc[4];
        bf3_block2<K1 / 16, true>(Xh + i * BF3_XS + 8 * kq, Xl + i * BF3_XS + 8 * kq, r1h, r1l, (kq * N1 + i) * 16, (kq * N1 + 32 + i) * 16, 2 * N1 * 16, acc[0], acc[1]);
        bf3_block2<K1 / 16, true>(Xh + i * BF3_XS + 8 * kq, Xl + i * BF3_XS + 8 * kq, r1h, r1l, (kq * N1 + 64 + i) * 16, (kq * N1 + 96 + i) * 16, 2 * N1 * 16, acc[2], acc[3]);
        if (tile + stride < ntiles) fetch(tile + stride);
        float em[16];
        if (emask) {
            const __amdgpu_buffer_rsrc_t eb = tile_rsrc(emask, e0, (int)sizeof(float));
#pragma unroll
            for (int r = 0; r < 16; ++r) em[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(eb, 16 * kq, ((r & 3) + 8 * (r >> 2)) * 4, 0));
        }
        // every chain has consumed its operands (LDS operations of a wave complete in order): the hidden layer replaces the input block.
        // Transposed blocks: lane (i, kq) holds row i, register 4 q + c the column 32 nb + 8 q + 4 kq + c -- four columns per 8-byte store
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * nb + 8 * q + 4 * kq;
                if (c0 < K2) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(w.b1m + c0);
                    const f32x2 v0 = pk_logsigmoid_or_zero(acc[nb][4 * q] + b4[0], acc[nb][4 * q + 1] + b4[1], c0 < w.m1);
                    const f32x2 v1 = pk_logsigmoid_or_zero(acc[nb][4 * q + 2] + b4[2], acc[nb][4 * q + 3] + b4[3], c0 + 2 < w.m1);
                    float r0, r1, r2, r3;
                    const uint32_t h0 = bf3_pack_hi(v0.x, v0.y, r0, r1), h1 = bf3_pack_hi(v1.x, v1.y, r2, r3);
                    *reinterpret_cast<uint2 *>(Hh + i * BF3_GS + c0) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(Hl + i * BF3_GS + c0) = make_uint2(bf3_pack(r0, r1), bf3_pack(r2, r3));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x16 ac2[2];
        bf3_block2<K2 / 16, false>(Hh + i * BF3_GS + 8 * kq, Hl + i * BF3_GS + 8 * kq, r2h, r2l, (kq * N2 + i) * 16, (kq * N2 + 32 + i) * 16, 2 * N2 * 16, ac2[0], ac2[1]);
        const int rowb = w.a * (int)sizeof(float);                           // h2 rows are a floats wide
        const __amdgpu_buffer_rsrc_t hb = tile_rsrc(h2out, e0, rowb);
        const int lo = 4 * kq * rowb + i * (int)sizeof(float);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            if (32 * nb + i < w.a) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 v = pk_logsigmoid((f32x2){ac2[nb][r], ac2[nb][r + 1]});
                    if (emask) v = v * (f32x2){em[r], em[r + 1]};
                    __builtin_amdgcn_raw_buffer_store_b32(f2i(v.x), hb, lo + nb * 128, ((r & 3) + 8 * (r >> 2)) * rowb, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(f2i(v.y), hb, lo + nb * 128, (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * rowb, 0);
                }
            }
        }
    }
}
#endif

// ---- kernel 5c: the pipelined cell with a WAVE as the unit of work (hidden widths whose column blocks do not divide eight waves) --------
// Hidden 150 (the reference's shipped np-nd-np predict config) has five 32-column blocks: the ten blocks of a 64-edge tile leave two of
// eight waves busy in a second round, whatever the window.  Here a wave owns a 32-edge tile from its rows to the stored result and walks
// ALL column blocks itself: four waves per workgroup (one per SIMD, the tile of each is 38 KB of LDS), every SIMD carries the same
// 3 x NBK chains per tile, nothing is exchanged and there is no workgroup barrier.  With one wave per SIMD nobody covers a wait, so the
// operand stream never stops: the 3 x NBK phases of a tile are ONE software pipeline -- LDS operands one chunk ahead, weight fragments two
// chunks ahead, across the phase and block boundaries (the first weights of the next tile are requested in the last chunks of this one) --
// with the previous gate's activation slices scheduled between the MFMAs as in k_gru_pipe; the result of a block is finished inside the
// next block's first phase, the last block's while the next tile's rows are in flight.  Same chains, same activation functions per
// element as k_gru: bit-identical results.
template <int SX, int SH>
struct GruStream {
    static constexpr int S = SX + SH, CH = (S + 15) / 16;
    float bb[3][CH], aa[2][CH];
    __amdgpu_buffer_rsrc_t wi, wh;
    const float *xa, *ha;
    int ws;
    __device__ __forceinline__ float wload(int voff, int s2) const {
        return (s2 < SX) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wi, voff, s2 * ws, 0))
                         : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wh, voff, (s2 - SX) * ws, 0));
    }
    __device__ __forceinline__ float aload(int s2) const { return (s2 < SX) ? xa[2 * s2] : ha[2 * (s2 - SX)]; }
    // chunk c of a phase = k-steps [c S / 16, (c + 1) S / 16)
    // (slot and c are constants after unrolling: the arrays live in registers)
    __device__ __forceinline__ void wchunk(int slot, int voff, int c) {
        const int lo = c * S / 16, hi = (c + 1) * S / 16;
#pragma unroll
        for (int j = 0; j < CH; ++j) if (lo + j < hi) bb[slot][j] = wload(voff, lo + j);
    }
    __device__ __forceinline__ void achunk(int slot, int c) {
        const int lo = c * S / 16, hi = (c + 1) * S / 16;
#pragma unroll
        for (int j = 0; j < CH; ++j) if (lo + j < hi) aa[slot][j] = aload(lo + j);
    }
};

// one gate of one block: 16 chunks.  BASE = number of the phase's first chunk modulo 3 (the weight slots rotate through the phases).
// Requests, in chunk c: the LDS operands of chunk c + 1 and the weights of chunk c + 2 -- of this phase, or of the next one (voff_next;
// a_next: its LDS operands can be read already, i.e. it belongs to the same tile).
template <int SX, int SH, int NV, int BASE, class Epi>
__device__ __forceinline__ void gru_phase2(GruStream<SX, SH> &st, int voff, int voff_next, bool a_next, float bi, float bh,
                                           f32x16 &ai, f32x16 &ah, Epi &&epi)
{
    constexpr int S = SX + SH, CH = (S + 15) / 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ai[r] = bi; ah[r] = bh; }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int lo = c * S / 16, hi = (c + 1) * S / 16;
        if (c + 2 < 16) st.wchunk((BASE + c + 2) % 3, voff, c + 2);
        else st.wchunk((BASE + c + 2) % 3, voff_next, c + 2 - 16);
        if (c + 1 < 16) st.achunk((c + 1) & 1, c + 1);
        else if (a_next) st.achunk(0, 0);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int s = lo + j;
            if (s < hi) {
                const float bv = st.bb[(BASE + c) % 3][j], av = st.aa[c & 1][j];
                if (s < SX) ai = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, ai, 0, 0, 0);
                else ah = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, ah, 0, 0, 0);
            }
        }
        // activation slices two at a time, as in gru_phase (one, four, eight at a time: 35.1 / 34.2 / 34.0 against 33.7 ms)
        if (c & 1) { epi(c - 1); epi(c); }
        // (with one wave per SIMD the accumulators live in the accumulation registers: the tie must not pull them into VGPRs -- a "+v" here
        //  costs 64 register moves per chunk, measured 93 instead of 64 cycles per MFMA)
        asm volatile("" : "+a"(ai), "+a"(ah));
        // issue order inside the chunk: one MFMA, one weight load and one LDS read of the chunks ahead, then the VALU run of the activation
        // slice -- the requests go out between the (dependent, 64-cycle) MFMAs instead of in front of them
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (lo + j < hi) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (NV > 0 && (c & 1)) __builtin_amdgcn_sched_group_barrier(0x002, 2 * NV, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int SX, int SH, int NBK, bool MASK>
__global__ void __launch_bounds__(256) k_gru_wave(int E, const float *__restrict__ state, const float *__restrict__ sign,
                                                  const float *__restrict__ hprev, const float *__restrict__ rowmask, GruW g,
                                                  float *__restrict__ out, int ntiles /* full 32-edge tiles only */)
{
    if (MASK && loop_stopped(rowmask, E)) return;          // a device-driven loop has ended: this sweep writes nothing (k_edge_active)
    constexpr int H = 2 * SX - 2;                          // input row = [H message floats, edge sign, zero pad]
    static_assert(2 * SH == H || 2 * SH == H + 1, "hidden rows of H floats (+ one zero when H is odd)");
    constexpr int HP = 32 * NBK, N3 = 3 * HP, CG = (H + 63) / 64;
    constexpr int ldx = 2 * SX + 1, ldh = 2 * SH + 1;
    constexpr int WR = WT * (ldx + ldh) + WT;              // floats per wave: X, Hs, row masks
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, i = l & 31, kh = l >> 5;
    float *X = sm + wave * WR, *Hs = X + WT * ldx, *Mk = Hs + WT * ldh;
    constexpr int rowb = H * (int)sizeof(float);
    GruStream<SX, SH> st;
    st.wi = __builtin_amdgcn_make_buffer_rsrc((void *)g.Wt_ih, 0, 2 * SX * N3 * (int)sizeof(float), 0x00020000);
    st.wh = __builtin_amdgcn_make_buffer_rsrc((void *)g.Wt_hh, 0, 2 * SH * N3 * (int)sizeof(float), 0x00020000);
    st.ws = 2 * N3 * (int)sizeof(float);
    st.xa = X + i * ldx + kh; st.ha = Hs + i * ldh + kh;
    auto tile_rsrc = [&](const float *base, int e0, int row_bytes) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(base + (size_t)e0 * (row_bytes / (int)sizeof(float))), 0, WT * row_bytes, 0x00020000);
    };
    float px[WT][CG], ph[WT][CG];
    float psg = 0.0f, pmk = 1.0f;
    auto fetch = [&](int tile) {
        const int e0 = tile * WT;
        const __amdgpu_buffer_rsrc_t sb = tile_rsrc(state, e0, rowb), hb = tile_rsrc(hprev, e0, rowb), gb = tile_rsrc(sign, e0, (int)sizeof(float));
#pragma unroll
        for (int r = 0; r < WT; ++r)
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                // (columns l, 64 + l, ...: a column past the row reads the next row's start, or 0 past the tile, and is not deposited)
                px[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sb, l * 4 + 256 * j, r * rowb, 0));
                ph[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hb, l * 4 + 256 * j, r * rowb, 0));
            }
        psg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gb, l * 4, 0, 0));       // lanes >= 32: past the tile, 0
        if (MASK) pmk = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc(rowmask, e0, (int)sizeof(float)), l * 4, 0, 0));
    };
    auto deposit = [&]() {
#pragma unroll
        for (int r = 0; r < WT; ++r)
#pragma unroll
            for (int j = 0; j < CG; ++j)
                if (64 * j + l < H) { X[r * ldx + 64 * j + l] = px[r][j]; Hs[r * ldh + 64 * j + l] = ph[r][j]; }
        if (l < WT) {
            X[l * ldx + H] = psg; X[l * ldx + H + 1] = 0.0f;
            if (2 * SH > H) Hs[l * ldh + H] = 0.0f;
            Mk[l] = pmk;
        }
    };
    const int row0 = 4 * kh;                               // acc_row(r, l) = row0 + (r & 3) + 8 (r >> 2)
    const int voff0 = (kh * N3 + i) * (int)sizeof(float);  // this lane's column of block 0, gate r
    f32x16 tq, zg, hq;                                     // carried from a block to the next: tanh argument, update gate, previous hidden value
    const int stride = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    if (tile < ntiles) {
        fetch(tile);
        st.wchunk(0, voff0, 0); st.wchunk(1, voff0, 1);
    }
    for (; tile < ntiles; tile += stride) {
        deposit();
        st.achunk(0, 0);
        const __amdgpu_buffer_rsrc_t ob = tile_rsrc(out, tile * WT, rowb);
        int ooff = 0;                                      // where the carried block goes (byte offset of this lane's column in row row0)
        auto finish = [&](int c) {
            const int ro = (c & 3) + 8 * (c >> 2);
            const float ng = pdp_tanhf_abs(tq[c]);
            const float hnew = (hq[c] - ng) * zg[c] + ng;
            const float mk = MASK ? Mk[row0 + ro] : 1.0f;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, mk * hnew + (1.0f - mk) * hq[c]), ob, ooff, ro * rowb, 0);
        };
        for (int nb = 0; nb < NBK; ++nb) {
            const int col = 32 * nb + i;
            const int voff = voff0 + 32 * nb * (int)sizeof(float);
            const bool last = nb == NBK - 1;
            const float bir = g.b_ih[col], biz = g.b_ih[HP + col], bin = g.b_ih[2 * HP + col];
            const float bhr = g.b_hh[col], bhz = g.b_hh[HP + col], bhn = g.b_hh[2 * HP + col];
            f32x16 ai, ah, rg;
            if (nb == 0) gru_phase2<SX, SH, 0, 0>(st, voff, voff + HP * (int)sizeof(float), true, bir, bhr, ai, ah, [&](int) {});
            else gru_phase2<SX, SH, 26, 0>(st, voff, voff + HP * (int)sizeof(float), true, bir, bhr, ai, ah, finish);
#pragma unroll
            for (int r = 0; r < 16; ++r) rg[r] = ah[r] + ai[r];
            gru_phase2<SX, SH, 16, 1>(st, voff + HP * (int)sizeof(float), voff + 2 * HP * (int)sizeof(float), true, biz, bhz, ai, ah,
                                      [&](int c) { float v = pdp_sigmoidf(rg[c]); asm volatile("" : "+v"(v)); rg[c] = v; });
#pragma unroll
            for (int r = 0; r < 16; ++r) zg[r] = ah[r] + ai[r];
            // the stream runs on into the next block, or (weights only) into block 0 of the next tile
            gru_phase2<SX, SH, 16, 2>(st, voff + 2 * HP * (int)sizeof(float), last ? voff0 : voff + 32 * (int)sizeof(float), !last, bin, bhn, ai, ah,
                                      [&](int c) { float v = pdp_sigmoidf(zg[c]); asm volatile("" : "+v"(v)); zg[c] = v; });
            const int hc = col < H ? col : H - 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                tq[r] = ai[r] + ah[r] * rg[r];
                hq[r] = Hs[(row0 + (r & 3) + 8 * (r >> 2)) * ldh + hc];
            }
            // a column past H is stored nowhere: an offset past the tile's bytes is dropped by the descriptor
            ooff = col < H ? (row0 * H + col) * (int)sizeof(float) : 0x40000000;
        }
        // the chains of this tile are done with X / Hs (LDS operations of a wave complete in order): request the next tile's rows and
        // finish the last block while they are in flight
        if (tile + stride < ntiles) fetch(tile + stride);
#pragma unroll
        for (int c = 0; c < 16; ++c) finish(c);
    }
}

// mask per edge from the per-instance active mask (K1: two chained sparse products in the reference).  The word behind the mask, out[E], carries
// the stop word of a device-driven loop (pdp_loop_*, FL_LOOP_STOP) to the kernels that blend a new state with the old one under this mask:
// once every instance has left the loop they return at once (loop_stopped) -- the sweeps a captured graph still replays are exact no-ops,
// where `0 * new + 1 * old` would turn a -0 into +0 and carry a NaN of `new` over.
__global__ void k_edge_active(int E, const int32_t *__restrict__ gm, const int32_t *__restrict__ var_inst, const uint8_t *__restrict__ amask,
                              float *__restrict__ out, const uint32_t *__restrict__ stop)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<uint32_t *>(out)[E] = stop ? *stop : 0u;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x)
        out[e] = amask ? (0.0f + (0.0f + (float)amask[var_inst[gm[e]]])) : 1.0f;
}

// ---- C ABI ----------------------------------------------------------------------------------------------------------------------------
static AggW make_agg(const pdp_agg_desc *d)
{
    AggW w;
    w.Wt1m = d->Wt1m; w.b1m = d->b1m; w.Wt2m = d->Wt2m; w.Wt1a = d->Wt1a; w.b1a = d->b1a; w.Wt2a = d->Wt2a;
    w.din = d->din; w.m1 = d->m1; w.a = d->a; w.g = d->g; w.out = d->out; w.fd = d->fd;
    w.Kp1 = even_up(d->din); w.Np1 = pad32(d->m1);
    w.Kp2 = even_up(d->m1); w.Np2 = pad32(d->a);
    w.Kp3 = even_up(d->a + d->fd); w.Np3 = pad32(d->g);
    w.Kp4 = even_up(d->g); w.Np4 = pad32(d->out);
    w.no_tail16 = getenv("PDP_NEURAL_NO_TAIL16") != nullptr ? 1 : 0;
    return w;
}

static int set_lds(const void *fn, size_t bytes)
{
    if (bytes > 160 * 1024 - 512) { pdp_set_error("neural kernel tile needs %zu bytes of LDS", bytes); return PDP_ERR_UNSUPPORTED; }
    PDP_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PDP_OK;
}

static float *neural_ws(pdp_problem *p, int slot, size_t floats);
#ifdef PDP_FAST_MATH
// split weights of the three-term bf16 kernels: part 0 the GRU's, 1 the aggregator's second half, 2 its first half.  One buffer per (device,
// stream), allocated at first use and kept for the life of the process: a SATProblem lives for one forward, and a hipMalloc / hipFree pair
// per forward stalls the device far longer than the kernels it serves (measured: 40 ms per np-nd-np iteration on some boxes).  The splits
// are written and read in stream order, so calls on one stream never see each other's weights.
#include <mutex>
#include <vector>
static uint32_t *bf3_workspace(pdp_problem *, int part, hipStream_t st)
{
    const size_t g = 2 * ((size_t)(144 / 8) * 384 * 4 + (size_t)(128 / 8) * 384 * 4), q = 2 * ((size_t)(64 / 8) * 128 * 4 + (size_t)(112 / 8) * 128 * 4);
    const size_t r = 2 * ((size_t)(144 / 8) * 128 * 4 + (size_t)(112 / 8) * 64 * 4);
    struct Slot { int dev; hipStream_t st; uint32_t *base; };
    static std::mutex mu;
    static std::vector<Slot> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    uint32_t *base = nullptr;
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const Slot &s : slots) if (s.dev == dev && s.st == st) base = s.base;
        if (!base) {
            if (hipMalloc((void **)&base, (g + q + r) * sizeof(uint32_t)) != hipSuccess) return nullptr;
            slots.push_back(Slot{dev, st, base});
        }
    }
    return part == 0 ? base : (part == 1 ? base + g : base + g + q);
}
#endif

#define LDS_RES_LIMIT (160 * 1024 - 512)
// PDP_NEURAL_GENERIC=1: every operator on its generic tile kernel (k_agg_pre / k_agg_post / k_gru) -- the cross-check the full-size tests run
// against the specialised kernels, bit for bit
static bool generic_forced() { return getenv("PDP_NEURAL_GENERIC") != nullptr; }
static int persistent_grid()
{
    if (const char *e = getenv("PDP_NEURAL_GRID")) { const int v = atoi(e); if (v > 0) return v; }   // tests: many tiles per workgroup on small inputs
    return pdp_device_cus();
}

// aggregator pre-transform: wave-private form (default); resident-weight persistent form when both matrices and the tile fit the LDS, tile-per-workgroup form otherwise
static int launch_agg_pre(int E, const float *state, const float *sign, const float *edge_mask, const AggW &w, float *h2, hipStream_t st, uint32_t *bf3_ws = nullptr)
{
    pdp_timed_scope timed(PDP_TK_AGG_PRE, st);
    const int tiles = (E + TM - 1) / TM;
    const size_t lds1 = sizeof(float) * (size_t)TM * ((w.Kp1 + 1) + (w.Np1 + 1));
    const size_t res1 = sizeof(float) * ((size_t)w.Kp1 * w.Np1 + (size_t)w.Kp2 * w.Np2) + lds1;
    const bool shape128 = w.din - 1 == 128 && w.Kp1 == 130, shape150 = w.din - 1 == 150 && w.Kp1 == 152;
#ifdef PDP_FAST_MATH
    if (!generic_forced() && shape128 && w.Np1 == 128 && w.Kp2 == 100 && w.Np2 == 64 && w.m1 <= 112 && (w.m1 & 3) == 0 && !getenv("PDP_AGG_NO_BF16X3")) {
        // three-term bf16 products (k_agg_pre_bf3): both layers' weights split per call (workspace passed in by the caller)
        const size_t w1 = (size_t)(BF3_KX / 8) * 128 * 4, w2 = (size_t)(112 / 8) * 64 * 4;
        uint32_t *w1h = bf3_ws, *w1l = w1h + w1, *w2h = w1l + w1, *w2l = w2h + w2;
        if (bf3_ws) {
            hipLaunchKernelGGL(k_bf3_split_weights, dim3(((BF3_KX / 8) * 128 + 255) / 256), dim3(256), 0, st, w.Wt1m, w.Kp1, BF3_KX, 128, w1h, w1l);
            hipLaunchKernelGGL(k_bf3_split_weights, dim3(((112 / 8) * 64 + 255) / 256), dim3(256), 0, st, w.Wt2m, w.Kp2, 112, 64, w2h, w2l);
            const size_t ldsb = (size_t)NWAVES * WT * BF3_XS * 2 * 2;
            int s = set_lds((const void *)k_agg_pre_bf3, ldsb); if (s != PDP_OK) return s;
            const int wt = (E + WT - 1) / WT, need = (wt + NWAVES - 1) / NWAVES;
            const int grid = need < persistent_grid() ? need : persistent_grid();
            pdp_note_kernel(PDP_TK_AGG_PRE, "k_agg_pre_bf3");
            hipLaunchKernelGGL(k_agg_pre_bf3, dim3(grid), dim3(NTN), ldsb, st, E, state, sign, edge_mask, w, w1h, w1l, w2h, w2l, h2, wt);
            return PDP_OK;
        }
    }
#endif
    if (!generic_forced() && (shape128 || shape150) && w.Np1 == 128 && w.Kp2 == 100 && w.Np2 == 64) {
        // hidden 128 (BASELINE configs) or 150 (the reference's shipped predict config) with the 100 / 50 inner widths
        // the first layer is 100 wide: three 32-column blocks + a 16-column tail (112 columns) instead of four blocks; PDP_NEURAL_NO_TAIL16
        // keeps the four-block form (A/B runs)
        const bool tail16 = w.m1 <= 112 && !w.no_tail16;
        const int ldw = shape128 ? (tail16 ? 133 : 131) : 153;
        const size_t ldsw = sizeof(float) * (size_t)NWAVES * WT * ldw;
        const void *fn = shape128 ? (tail16 ? (const void *)k_agg_pre_wave<65, 4, 50, 2, true> : (const void *)k_agg_pre_wave<65, 4, 50, 2, false>)
                                  : (tail16 ? (const void *)k_agg_pre_wave<76, 4, 50, 2, true> : (const void *)k_agg_pre_wave<76, 4, 50, 2, false>);
        int s = set_lds(fn, ldsw); if (s != PDP_OK) return s;
        const int wt = (E + WT - 1) / WT, need = (wt + NWAVES - 1) / NWAVES;
        const int grid = need < persistent_grid() ? need : persistent_grid();
        pdp_note_kernel(PDP_TK_AGG_PRE, shape128 ? (tail16 ? "k_agg_pre_wave<65, 4, 50, 2, true>" : "k_agg_pre_wave<65, 4, 50, 2, false>")
                                                 : (tail16 ? "k_agg_pre_wave<76, 4, 50, 2, true>" : "k_agg_pre_wave<76, 4, 50, 2, false>"));
        if (shape128 && tail16) hipLaunchKernelGGL((k_agg_pre_wave<65, 4, 50, 2, true>), dim3(grid), dim3(NTN), ldsw, st, E, state, sign, edge_mask, w, h2, wt);
        else if (shape128) hipLaunchKernelGGL((k_agg_pre_wave<65, 4, 50, 2, false>), dim3(grid), dim3(NTN), ldsw, st, E, state, sign, edge_mask, w, h2, wt);
        else if (tail16) hipLaunchKernelGGL((k_agg_pre_wave<76, 4, 50, 2, true>), dim3(grid), dim3(NTN), ldsw, st, E, state, sign, edge_mask, w, h2, wt);
        else hipLaunchKernelGGL((k_agg_pre_wave<76, 4, 50, 2, false>), dim3(grid), dim3(NTN), ldsw, st, E, state, sign, edge_mask, w, h2, wt);
    } else if (!generic_forced() && res1 <= LDS_RES_LIMIT && w.Kp1 <= 64 * PRE_C) {
        int s = set_lds((const void *)k_agg_pre_res, res1); if (s != PDP_OK) return s;
        const int grid = tiles < persistent_grid() ? tiles : persistent_grid();
        pdp_note_kernel(PDP_TK_AGG_PRE, "k_agg_pre_res");
        hipLaunchKernelGGL(k_agg_pre_res, dim3(grid), dim3(NTN), res1, st, E, state, w.din - 1, sign, edge_mask, w, h2, tiles);
    } else {
        int s = set_lds((const void *)k_agg_pre, lds1); if (s != PDP_OK) return s;
        pdp_note_kernel(PDP_TK_AGG_PRE, "k_agg_pre");
        hipLaunchKernelGGL(k_agg_pre, dim3(tiles), dim3(NTN), lds1, st, E, state, w.din - 1, sign, edge_mask, w, h2);
    }
    return PDP_OK;
}

// replaces: MessageAggregator.forward with include_self_message=False as used by NeuralMessagePasser
// (pdp_propagate.py:77-78,88-89): by_variable != 0 aggregates over the variable of each edge, else over its clause.
// state [E, din-1]; edge_mask [E] or NULL; active_mask uint8 [B] or NULL; old [E, out] = state blended in where inactive.
extern "C" int pdp_neural_aggregate_edges(pdp_problem *p, const pdp_agg_desc *d, int by_variable, const float *state,
                                          const float *edge_mask, const uint8_t *active_mask, const float *old, float *out, void *stream)
{
    PDP_REQUIRE(p && d && state && old && out, "NULL argument");
    hipStream_t st = ST(stream);
    const AggW w = make_agg(d);
    const int E = p->E, R = by_variable ? p->V : p->F;
    float *h2 = neural_ws(p, 0, (size_t)E * w.a), *agg = neural_ws(p, 1, (size_t)R * w.a), *rowmask = neural_ws(p, 2, (size_t)E + 4);
    if (!h2 || !agg || !rowmask) return PDP_ERR_HIP;
    const int tiles = (E + TM - 1) / TM;
    const size_t lds1 = sizeof(float) * (size_t)TM * ((w.Kp1 + 1) + (w.Np1 + 1));
    const size_t lds3 = sizeof(float) * (size_t)TM * ((w.Kp3 + 1) + (w.Np3 + 1));
    int s = set_lds((const void *)k_agg_pre, lds1); if (s != PDP_OK) return s;
    s = set_lds((const void *)k_agg_post, lds3); if (s != PDP_OK) return s;
    hipLaunchKernelGGL(k_edge_active, dim3(1024), dim3(256), 0, st, E, p->graph_map, p->var_inst, active_mask, rowmask, p->flags + FL_LOOP_STOP);
#ifdef PDP_FAST_MATH
    s = launch_agg_pre(E, state, p->edge_sign, edge_mask, w, h2, st, bf3_workspace(p, 2, st)); if (s != PDP_OK) return s;
#else
    s = launch_agg_pre(E, state, p->edge_sign, edge_mask, w, h2, st); if (s != PDP_OK) return s;
#endif
    // global CSR rows in ascending edge id: the sorted edge lists of the problem (global ids)
    const int32_t *row_ptr = by_variable ? p->nv_ptr : p->nf_ptr;
    const int32_t *row_edges = by_variable ? p->nv_edges : p->nf_edges;
    { pdp_timed_scope timed(PDP_TK_ROW_SUM, st);
      launch_row_sum(R, w.a, row_ptr, row_edges, h2, agg, st); }
    const int32_t *edge_row = by_variable ? p->graph_map : p->graph_map + E;
    pdp_timed_scope timed_post(PDP_TK_AGG_POST, st);
    const bool shape_pf = !generic_forced() && w.Kp3 == 52 && w.Np3 == 128 && w.Kp4 == 100 && ((w.Np4 == 128 && w.out == 128) || (w.Np4 == 160 && w.out == 150));
    const size_t ldsp = sizeof(float) * (size_t)TM * (53 + 129);
    if (shape_pf && w.out == 150 && (int64_t)R * w.a * 4 < ((int64_t)1 << 31) && E >= WT) {
        // hidden 150: a wave per 32-edge tile on the full tiles (the workgroup-tile kernel's ten output blocks take two rounds of its eight
        // waves there: 10.2 against 9.8 ms per call at 12.6 M edges), the workgroup-tile kernel on the ragged tail
        const int full = E / WT, tail = E - full * WT;
        const size_t ldsw = sizeof(float) * PW_NW * (size_t)(WT * (53 + 129) + WT);
        const int wgs = (full + PW_NW - 1) / PW_NW;
        const int grid = wgs < persistent_grid() ? wgs : persistent_grid();
        const size_t o = (size_t)full * WT;
        const bool tail16 = w.g <= 112 && !w.no_tail16;
        s = set_lds(tail16 ? (const void *)k_agg_post_wave<26, 4, 50, 5, true> : (const void *)k_agg_post_wave<26, 4, 50, 5, false>, ldsw); if (s != PDP_OK) return s;
        pdp_note_kernel(PDP_TK_AGG_POST, tail16 ? "k_agg_post_wave<26, 4, 50, 5, true>" : "k_agg_post_wave<26, 4, 50, 5, false>");
        if (tail16) hipLaunchKernelGGL((k_agg_post_wave<26, 4, 50, 5, true>), dim3(grid), dim3(64 * PW_NW), ldsw, st, E, agg, R, edge_row, h2, p->edge_sign, edge_mask, rowmask, old, w, out, full);
        else hipLaunchKernelGGL((k_agg_post_wave<26, 4, 50, 5, false>), dim3(grid), dim3(64 * PW_NW), ldsw, st, E, agg, R, edge_row, h2, p->edge_sign, edge_mask, rowmask, old, w, out, full);
        if (tail > 0) {
            s = set_lds((const void *)k_agg_post_pf<26, 4, 50, 5>, ldsp); if (s != PDP_OK) return s;
            hipLaunchKernelGGL((k_agg_post_pf<26, 4, 50, 5>), dim3(1), dim3(NTN), ldsp, st, tail, agg, edge_row + o, h2 + o * w.a, p->edge_sign + o,
                               edge_mask ? edge_mask + o : nullptr, rowmask + o, old + o * w.out, w, out + o * w.out);
        }
    } else if (shape_pf) {
        // hidden 128 (BASELINE configs), and hidden 150 when the row offsets pass 31 bits: the 100 / 50 inner widths, prefetched chains
#ifdef PDP_FAST_MATH
        if (w.Np4 == 128 && w.a <= 51 && w.g <= 112 && !getenv("PDP_AGG_NO_BF16X3")) {
            // three-term bf16 products (k_agg_post_bf3): both layers' weights split per call into the workspace behind the GRU's
            const size_t w3 = (size_t)(64 / 8) * 128 * 4, w4 = (size_t)(112 / 8) * 128 * 4;
            uint32_t *wsp = bf3_workspace(p, 1, st);
            if (!wsp) return PDP_ERR_HIP;
            uint32_t *w3h = wsp, *w3l = w3h + w3, *w4h = w3l + w3, *w4l = w4h + w4;
            hipLaunchKernelGGL(k_bf3_split_weights, dim3(((64 / 8) * 128 + 255) / 256), dim3(256), 0, st, w.Wt1a, w.Kp3, 64, 128, w3h, w3l);
            hipLaunchKernelGGL(k_bf3_split_weights, dim3(((112 / 8) * 128 + 255) / 256), dim3(256), 0, st, w.Wt2a, w.Kp4, 112, 128, w4h, w4l);
            const size_t ldsb = (size_t)TM * (BF3_RS + BF3_GS) * 2 * 2;
            s = set_lds((const void *)k_agg_post_bf3, ldsb); if (s != PDP_OK) return s;
            pdp_note_kernel(PDP_TK_AGG_POST, "k_agg_post_bf3");
            hipLaunchKernelGGL(k_agg_post_bf3, dim3(tiles), dim3(NTN), ldsb, st, E, agg, edge_row, h2, p->edge_sign, edge_mask, rowmask, old, w, w3h, w3l, w4h, w4l, out);
        } else
#endif
        if (w.Np4 == 128) {
            s = set_lds((const void *)k_agg_post_pf<26, 4, 50, 4>, ldsp); if (s != PDP_OK) return s;
            pdp_note_kernel(PDP_TK_AGG_POST, "k_agg_post_pf<26, 4, 50, 4>");
            hipLaunchKernelGGL((k_agg_post_pf<26, 4, 50, 4>), dim3(tiles), dim3(NTN), ldsp, st, E, agg, edge_row, h2, p->edge_sign, edge_mask, rowmask, old, w, out);
        } else {
            s = set_lds((const void *)k_agg_post_pf<26, 4, 50, 5>, ldsp); if (s != PDP_OK) return s;
            pdp_note_kernel(PDP_TK_AGG_POST, "k_agg_post_pf<26, 4, 50, 5>");
            hipLaunchKernelGGL((k_agg_post_pf<26, 4, 50, 5>), dim3(tiles), dim3(NTN), ldsp, st, E, agg, edge_row, h2, p->edge_sign, edge_mask, rowmask, old, w, out);
        }
    } else {
        pdp_note_kernel(PDP_TK_AGG_POST, "k_agg_post");
        hipLaunchKernelGGL(k_agg_post, dim3(tiles), dim3(NTN), lds3, st, E, agg, edge_row, h2, p->edge_sign, edge_mask, rowmask, old, w, out);
    }
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// replaces: NeuralDecimator.forward, one direction (pdp_decimate.py:70-75 / 78-83): h' = mask * GRU([state ‖ s], h) + (1 - mask) * h
extern "C" int pdp_neural_gru(pdp_problem *p, const pdp_gru_desc *d, const float *state, const float *h, const uint8_t *active_mask,
                              float *out, void *stream)
{
    PDP_REQUIRE(p && d && state && h && out, "NULL argument");
    PDP_REQUIRE(out != h, "output must not alias the hidden state");
    hipStream_t st = ST(stream);
    GruW g;
    g.Wt_ih = d->Wt_ih; g.Wt_hh = d->Wt_hh; g.b_ih = d->b_ih; g.b_hh = d->b_hh; g.dx = d->dx; g.H = d->H;
    g.Kpx = even_up(d->dx + 1); g.Kph = even_up(d->H); g.Hp = pad32(d->H);
    const int E = p->E;
    float *rowmask = neural_ws(p, 2, (size_t)E + 4);
    if (!rowmask) return PDP_ERR_HIP;
    const size_t lds = sizeof(float) * (size_t)TM * ((g.Kpx + 1) + (g.Kph + 1));
    PDP_REQUIRE(g.Kpx <= 64 * PRE_C && g.Kph <= 64 * PRE_C, "GRU wider than 192 inputs is not supported by the tile prefetch");
    int s = set_lds((const void *)k_gru, lds); if (s != PDP_OK) return s;
    hipLaunchKernelGGL(k_edge_active, dim3(1024), dim3(256), 0, st, E, p->graph_map, p->var_inst, active_mask, rowmask, p->flags + FL_LOOP_STOP);
    pdp_timed_scope timed(PDP_TK_GRU, st);
    const bool plain = generic_forced();
    auto ragged_tail = [&](size_t o, int tail) {
        hipLaunchKernelGGL(k_gru, dim3(1), dim3(NTN), lds, st, tail, state + o * g.dx, p->edge_sign + o, h + o * g.H, rowmask + o, g, out + o * g.H, 1);
    };
    if (!plain && d->H == 128 && (g.Kpx == 130 || g.Kpx == 4)) {
        // hidden width 128 with a 129-wide input (np-nd-np, config 3) or a 4- / 3-wide one (p-nd-np: surveys + sign): pipelined kernel on
        // the full tiles, the plain one on the ragged tail
        const int full = E / TM, tail = E - full * TM;
        if (full > 0) {
            const size_t ldsp = 2 * lds + sizeof(float) * 3 * TM;        // two tile buffers + three mask rows
            const int grid = full < persistent_grid() ? full : persistent_grid();
#ifdef PDP_FAST_MATH
            if (!getenv("PDP_GRU_NO_BF16X3")) {
                // three-term bf16 products (k_gru_bf3): the weights are split per call (99 072 values, microseconds) into a workspace
                const bool wide = g.Kpx == 130;
                const size_t wx = (size_t)(BF3_KX / 8) * 3 * 128 * 4, wh = (size_t)(128 / 8) * 3 * 128 * 4;        // dwords per part
                uint32_t *wsp = bf3_workspace(p, 0, st);
                if (!wsp) return PDP_ERR_HIP;
                uint32_t *wxh = wsp, *wxl = wxh + wx, *whh = wxl + wx, *whl = whh + wh;
                if (wide) hipLaunchKernelGGL(k_bf3_split_weights, dim3(((BF3_KX / 8) * 384 + 255) / 256), dim3(256), 0, st, g.Wt_ih, g.Kpx, BF3_KX, 384, wxh, wxl);
                hipLaunchKernelGGL(k_bf3_split_weights, dim3(((128 / 8) * 384 + 255) / 256), dim3(256), 0, st, g.Wt_hh, g.Kph, 128, 384, whh, whl);
                const size_t ldsb = (size_t)2 * ((wide ? TM * BF3_XS : 320) + TM * BF3_HS) * 4 + sizeof(float) * 3 * TM;
                if (wide) {
                    s = set_lds((const void *)k_gru_bf3<true, true>, ldsb); if (s != PDP_OK) return s;
                    pdp_note_kernel(PDP_TK_GRU, "k_gru_bf3<true, true>");
                    hipLaunchKernelGGL((k_gru_bf3<true, true>), dim3(grid), dim3(NTN), ldsb, st, E, state, p->edge_sign, h, rowmask, wxh, wxl, whh, whl, g.Wt_ih, g.dx, g.b_ih, g.b_hh, out, full);
                } else {
                    s = set_lds((const void *)k_gru_bf3<false, true>, ldsb); if (s != PDP_OK) return s;
                    pdp_note_kernel(PDP_TK_GRU, "k_gru_bf3<false, true>");
                    hipLaunchKernelGGL((k_gru_bf3<false, true>), dim3(grid), dim3(NTN), ldsb, st, E, state, p->edge_sign, h, rowmask, wxh, wxl, whh, whl, g.Wt_ih, g.dx, g.b_ih, g.b_hh, out, full);
                }
            } else
#endif
            if (g.Kpx == 130) {
                s = set_lds((const void *)k_gru_pipe<65, true>, ldsp); if (s != PDP_OK) return s;
                pdp_note_kernel(PDP_TK_GRU, "k_gru_pipe<65, true>");
                hipLaunchKernelGGL((k_gru_pipe<65, true>), dim3(grid), dim3(NTN), ldsp, st, E, state, p->edge_sign, h, rowmask, g, out, full);
            } else {
                s = set_lds((const void *)k_gru_pipe<2, true>, ldsp); if (s != PDP_OK) return s;
                pdp_note_kernel(PDP_TK_GRU, "k_gru_pipe<2, true>");
                hipLaunchKernelGGL((k_gru_pipe<2, true>), dim3(grid), dim3(NTN), ldsp, st, E, state, p->edge_sign, h, rowmask, g, out, full);
            }
        }
        if (tail > 0) ragged_tail((size_t)full * TM, tail);
        PDP_LAUNCH_CHECK();
        return PDP_OK;
    }
    if (!plain && d->H == 150 && g.Kpx == 152) {
        // hidden width 150 (five column blocks): a wave per 32-edge tile, four waves per workgroup, one workgroup per CU
        const int full = E / WT, tail = E - full * WT;
        if (full > 0) {
            const size_t ldsw = sizeof(float) * 4 * (size_t)(WT * ((g.Kpx + 1) + (g.Kph + 1)) + WT);
            const int wgs = (full + 3) / 4;
            const int grid = wgs < persistent_grid() ? wgs : persistent_grid();
            s = set_lds((const void *)k_gru_wave<76, 75, 5, true>, ldsw); if (s != PDP_OK) return s;
            pdp_note_kernel(PDP_TK_GRU, "k_gru_wave<76, 75, 5, true>");
            hipLaunchKernelGGL((k_gru_wave<76, 75, 5, true>), dim3(grid), dim3(256), ldsw, st, E, state, p->edge_sign, h, rowmask, g, out, full);
        }
        if (tail > 0) ragged_tail((size_t)full * WT, tail);
        PDP_LAUNCH_CHECK();
        return PDP_OK;
    }
    const int tiles = (E + TM - 1) / TM;
    int per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_gru, NTN, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    const int grid = tiles < per_cu * persistent_grid() ? tiles : per_cu * persistent_grid();
    pdp_note_kernel(PDP_TK_GRU, "k_gru");
    hipLaunchKernelGGL(k_gru, dim3(grid), dim3(NTN), lds, st, E, state, p->edge_sign, h, rowmask, g, out, tiles);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// Training forward of the hidden-128 cell on the pipelined inference kernel (include/pdp_hip.h): R rows, R a multiple of 64.
extern "C" int pdp_train_gru_fused(const pdp_gru_desc *d, const float *state, const float *sign, const float *h, int64_t R, float *hnew, float *saved,
                                   void *stream)
{
    PDP_REQUIRE(d && state && sign && h && hnew && saved, "NULL argument");
    PDP_REQUIRE(d->H == 128 && (d->dx == 128 || d->dx == 2 || d->dx == 3), "the fused training cells are 129 -> 128 (np-nd-np) and 3 / 4 -> 128 (p-nd-np)");
    PDP_REQUIRE(R >= 0 && R % TM == 0 && R < ((int64_t)1 << 31), "row count must be a multiple of the 64-row tile");
    PDP_REQUIRE(hnew != h, "output must not alias the hidden state");
    if (R == 0) return PDP_OK;
    hipStream_t st = ST(stream);
    GruW g;
    g.Wt_ih = d->Wt_ih; g.Wt_hh = d->Wt_hh; g.b_ih = d->b_ih; g.b_hh = d->b_hh; g.dx = d->dx; g.H = d->H;
    g.Kpx = even_up(d->dx + 1); g.Kph = even_up(d->H); g.Hp = pad32(d->H);
    const size_t lds = sizeof(float) * (size_t)TM * ((g.Kpx + 1) + (g.Kph + 1));
    const size_t ldsp = 2 * lds + sizeof(float) * 3 * TM;
    const int full = (int)(R / TM);
    const int grid = full < persistent_grid() ? full : persistent_grid();
    int s;
    if (g.Kpx == 130) {
        s = set_lds((const void *)k_gru_pipe<65, false, true>, ldsp); if (s != PDP_OK) return s;
        hipLaunchKernelGGL((k_gru_pipe<65, false, true>), dim3(grid), dim3(NTN), ldsp, st, (int)R, state, sign, h, (const float *)nullptr, g, hnew, full, saved);
    } else {
        PDP_REQUIRE(g.Kpx == 4, "narrow cell: 3 or 4 input columns");
        s = set_lds((const void *)k_gru_pipe<2, false, true>, ldsp); if (s != PDP_OK) return s;
        hipLaunchKernelGGL((k_gru_pipe<2, false, true>), dim3(grid), dim3(NTN), ldsp, st, (int)R, state, sign, h, (const float *)nullptr, g, hnew, full, saved);
    }
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// replaces: NeuralPredictor.forward (pdp_predict.py:67-77): variable aggregator with include_self_message=True + classifier head
extern "C" int pdp_neural_predict(pdp_problem *p, const pdp_agg_desc *d, const pdp_head_desc *hd, const float *state, const float *edge_mask,
                                  float *pred, void *stream)
{
    PDP_REQUIRE(p && d && hd && state && pred, "NULL argument");
    hipStream_t st = ST(stream);
    const AggW w = make_agg(d);
    HeadW h;
    h.Wt1 = hd->Wt1; h.b1 = hd->b1; h.w2 = hd->w2; h.H = hd->H; h.C = hd->C; h.out_act = hd->out_act;
    h.Kp = even_up(hd->H); h.Np = pad32(hd->C);
    const int E = p->E, V = p->V;
    float *h2 = neural_ws(p, 0, (size_t)E * w.a), *agg = neural_ws(p, 1, (size_t)V * w.a);
    if (!h2 || !agg) return PDP_ERR_HIP;
    const size_t lds1 = sizeof(float) * (size_t)TM * ((w.Kp1 + 1) + (w.Np1 + 1));
    const size_t lds4 = sizeof(float) * (size_t)TM * ((w.Kp3 + 1) + (w.Np3 + 1) + (w.Np4 + 1) + (h.Np + 1));
    int s = set_lds((const void *)k_agg_pre, lds1); if (s != PDP_OK) return s;
    s = set_lds((const void *)k_predict_rows, lds4); if (s != PDP_OK) return s;
#ifdef PDP_FAST_MATH
    s = launch_agg_pre(E, state, p->edge_sign, edge_mask, w, h2, st, bf3_workspace(p, 2, st)); if (s != PDP_OK) return s;
#else
    s = launch_agg_pre(E, state, p->edge_sign, edge_mask, w, h2, st); if (s != PDP_OK) return s;
#endif
    { pdp_timed_scope timed(PDP_TK_ROW_SUM, st);
      launch_row_sum(V, w.a, p->nv_ptr, p->nv_edges, h2, agg, st); }
    { pdp_timed_scope timed(PDP_TK_PREDICT_HEAD, st);
      const bool shape_pf = !generic_forced() && !getenv("PDP_PREDICT_GENERIC") && (w.Kp3 == 50 || w.Kp3 == 52) && w.Np3 == 128 && w.Kp4 == 100 && w.Np4 == 128 && w.out == 128 &&
                            h.Kp == 128 && h.Np == 64 && w.a <= 52 && w.g <= 128;
      if (shape_pf) {
          const size_t ldsp = sizeof(float) * (size_t)TM * (129 + 129);
          s = set_lds((const void *)k_predict_rows_pf<26, 4, 50, 4>, ldsp); if (s != PDP_OK) return s;
          pdp_note_kernel(PDP_TK_PREDICT_HEAD, "k_predict_rows_pf<26, 4, 50, 4>");
          hipLaunchKernelGGL((k_predict_rows_pf<26, 4, 50, 4>), dim3((V + TM - 1) / TM), dim3(NTN), ldsp, st, V, agg, w, h, pred);
      } else {
          pdp_note_kernel(PDP_TK_PREDICT_HEAD, "k_predict_rows");
          hipLaunchKernelGGL(k_predict_rows, dim3((V + TM - 1) / TM), dim3(NTN), lds4, st, V, agg, w, h, pred);
      } }
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

static float *neural_ws(pdp_problem *p, int slot, size_t floats)
{
    if (p->nws_floats[slot] < floats) {
        if (p->nws[slot]) pdp_dev_free(p->nws[slot]);
        p->nws[slot] = nullptr; p->nws_floats[slot] = 0;
        if (pdp_dev_alloc((void **)&p->nws[slot], floats * sizeof(float)) != PDP_OK) return nullptr;
        p->nws_floats[slot] = floats;
    }
    return p->nws[slot];
}
