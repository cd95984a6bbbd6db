/*
 * pdp_oracle_neural.c -- CPU restatement of the reference's NEURAL PDP operators.  TEST INFRASTRUCTURE ONLY
 * (same rules as pdp_oracle.c).
 *
 *   MessageAggregator.forward      reference: src/pdp/nn/util.py:51-77
 *   NeuralMessagePasser.forward    reference: src/pdp/nn/pdp_propagate.py:47-95
 *   NeuralDecimator.forward        reference: src/pdp/nn/pdp_decimate.py:51-87  (torch nn.GRUCell semantics)
 *   NeuralPredictor.forward        reference: src/pdp/nn/pdp_predict.py:49-91   + Perceptron trainer.py:20-29
 *
 * Every dense product is a k-ordered chain  acc = bias; acc = fmaf(x[k], w[k], acc)  -- bit for bit what the
 * fp32 MFMA instructions of gfx950 compute (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain), so the HIP kernels
 * can be compared with array_equal.  torch's MKL sgemm sums in another order: oracle-vs-reference comparisons
 * (tests/test_oracle_neural.py against tests/golden/trace_neural_*.npz) are tolerance based (rtol 2e-4 / atol 2e-5).
 * Row sums (per variable / clause) are sequential in ascending edge id like every sparse product of the reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>

#include "../include/pdp_math.h"

#define ORC_API __attribute__((visibility("default")))

enum { ACT_NONE = 0, ACT_LOGSIGMOID = 1, ACT_RELU = 2, ACT_SIGMOID = 3, ACT_TANH = 4 };

static inline float act_apply(float v, int act)
{
    switch (act) {
    case ACT_LOGSIGMOID: return pdp_logsigmoidf(v);
    case ACT_RELU: return v > 0.0f ? v : ((v != v) ? v : 0.0f);      /* torch relu keeps NaN */
    case ACT_SIGMOID: return pdp_sigmoidf(v);
    case ACT_TANH: return pdp_tanhf(v);
    default: return v;
    }
}

/* y[r, j] = act(b[j] + sum_k x[r, k] * W[j, k]),  x [R, K] (row stride ldx), W [N, K] row-major (nn.Linear.weight) */
ORC_API void orc_linear(const float *x, int64_t R, int K, int64_t ldx, const float *W, const float *b, int N, int act, float *y, int64_t ldy)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        const float *xr = x + r * ldx;
        float *yr = y + r * ldy;
        for (int j = 0; j < N; ++j) {
            const float *w = W + (int64_t)j * K;
            float acc = b ? b[j] : 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(xr[k], w[k], acc);
            yr[j] = act_apply(acc, act);
        }
    }
}

typedef struct {
    int din, m1, a, fd, g, out;      /* input, mem_hidden, mem_agg_hidden, feature dim (0/1), agg_hidden, output */
    const float *W1m, *b1m, *W2m, *W1a, *b1a, *W2a;
} orc_agg_weights;

/* MessageAggregator.forward.  state [E, din-1] + edge sign as last input column (the reference concatenates
 * edge_feature before calling); rows given by CSR (row_ptr [Rn+1], row_edges [E] ascending edge id) and edge_row [E].
 * include_self = 0: out [E, out] (sum over the OTHER edges of the row, feature = edge sign appended);
 * include_self = 1: out [Rn, out] (feature dim 0).  edge_mask [E] or NULL. */
ORC_API void orc_aggregator(int E, int Rn, const int32_t *row_ptr, const int32_t *row_edges, const int32_t *edge_row,
                            const float *state /*[E, din-1]*/, const float *edge_sign /*[E]*/, const float *edge_mask,
                            int include_self, const orc_agg_weights *w, float *out)
{
    const int din = w->din, sd = din - 1;
    float *xin = (float *)malloc(sizeof(float) * (size_t)E * din);
    float *h1 = (float *)malloc(sizeof(float) * (size_t)E * w->m1);
    float *h2 = (float *)malloc(sizeof(float) * (size_t)E * w->a);
    float *agg = (float *)malloc(sizeof(float) * (size_t)Rn * w->a);
    for (int64_t e = 0; e < E; ++e) {
        memcpy(xin + e * din, state + e * sd, sizeof(float) * (size_t)sd);
        xin[e * din + sd] = edge_sign[e];
    }
    orc_linear(xin, E, din, din, w->W1m, w->b1m, w->m1, ACT_LOGSIGMOID, h1, w->m1);
    orc_linear(h1, E, w->m1, w->m1, w->W2m, NULL, w->a, ACT_LOGSIGMOID, h2, w->a);
    if (edge_mask) for (int64_t e = 0; e < E; ++e) for (int j = 0; j < w->a; ++j) h2[e * w->a + j] = h2[e * w->a + j] * edge_mask[e];
    for (int r = 0; r < Rn; ++r) {
        for (int j = 0; j < w->a; ++j) {
            float acc = 0.0f;
            for (int k = row_ptr[r]; k < row_ptr[r + 1]; ++k) acc = acc + h2[(int64_t)row_edges[k] * w->a + j];
            agg[(int64_t)r * w->a + j] = acc;
        }
    }
    if (include_self) {
        float *g1 = (float *)malloc(sizeof(float) * (size_t)Rn * w->g);
        orc_linear(agg, Rn, w->a, w->a, w->W1a, w->b1a, w->g, ACT_LOGSIGMOID, g1, w->g);
        orc_linear(g1, Rn, w->g, w->g, w->W2a, NULL, w->out, ACT_LOGSIGMOID, out, w->out);
        free(g1);
    } else {
        const int rin = w->a + w->fd;
        float *r = (float *)malloc(sizeof(float) * (size_t)E * rin);
        float *g1 = (float *)malloc(sizeof(float) * (size_t)E * w->g);
        for (int64_t e = 0; e < E; ++e) {
            const float *ar = agg + (int64_t)edge_row[e] * w->a;
            for (int j = 0; j < w->a; ++j) {
                const float own = edge_mask ? h2[e * w->a + j] * edge_mask[e] : h2[e * w->a + j];
                r[e * rin + j] = (0.0f + ar[j]) - own;
            }
            if (w->fd) r[e * rin + w->a] = edge_sign[e];
        }
        orc_linear(r, E, rin, rin, w->W1a, w->b1a, w->g, ACT_LOGSIGMOID, g1, w->g);
        orc_linear(g1, E, w->g, w->g, w->W2a, NULL, w->out, ACT_LOGSIGMOID, out, w->out);
        free(r); free(g1);
    }
    free(xin); free(h1); free(h2); free(agg);
}

/* torch.nn.GRUCell (ATen/native/RNN.cpp): r = sig(h_r + i_r), z = sig(h_z + i_z), n = tanh(i_n + r * h_n),
 * h' = (h - n) * z + n; input x = [state ‖ edge sign].  out = mask * h' + (1 - mask) * h  (pdp_decimate.py:75,83) */
ORC_API void orc_gru(int E, int H, int dx /*state width, input = dx + 1*/, const float *state, const float *edge_sign, const float *h,
                     const float *W_ih /*[3H, dx+1]*/, const float *W_hh /*[3H, H]*/, const float *b_ih, const float *b_hh,
                     const float *mask /*[E] or NULL*/, float *out)
{
    const int din = dx + 1;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; ++e) {
        float *x = (float *)malloc(sizeof(float) * (size_t)din);
        memcpy(x, state + e * dx, sizeof(float) * (size_t)dx);
        x[dx] = edge_sign[e];
        const float *he = h + e * H;
        const float mk = mask ? mask[e] : 1.0f;
        for (int j = 0; j < H; ++j) {
            float ig[3], hg[3];
            for (int gidx = 0; gidx < 3; ++gidx) {
                const float *wi = W_ih + (int64_t)(gidx * H + j) * din;
                const float *wh = W_hh + (int64_t)(gidx * H + j) * H;
                float a = b_ih[gidx * H + j], b2 = b_hh[gidx * H + j];
                for (int k = 0; k < din; ++k) a = fmaf(x[k], wi[k], a);
                for (int k = 0; k < H; ++k) b2 = fmaf(he[k], wh[k], b2);
                ig[gidx] = a; hg[gidx] = b2;
            }
            const float r = pdp_sigmoidf(hg[0] + ig[0]);
            const float z = pdp_sigmoidf(hg[1] + ig[1]);
            const float n = pdp_tanhf_abs(ig[2] + hg[2] * r);
            const float hn = (he[j] - n) * z + n;
            out[e * H + j] = mk * hn + (1.0f - mk) * he[j];
        }
        free(x);
    }
}

/* Perceptron head: sigmoid(W2 relu(W1 x + b1)) (trainer.py:28-29) or tanh for PerceptronTanh (util.py:250-251) */
ORC_API void orc_perceptron(int R, int H, int C, const float *x, const float *W1, const float *b1, const float *W2, int out_act, float *y)
{
    float *hid = (float *)malloc(sizeof(float) * (size_t)R * C);
    orc_linear(x, R, H, H, W1, b1, C, ACT_RELU, hid, C);
    orc_linear(hid, R, C, C, W2, NULL, 1, out_act, y, 1);
    free(hid);
}

/* blend helper: out = mask * a + (1 - mask) * b, row-wise mask */
ORC_API void orc_blend_rows(int64_t R, int W, const float *mask, const float *a, const float *b, float *out)
{
    for (int64_t r = 0; r < R; ++r) {
        const float m = mask ? mask[r] : 1.0f;
        for (int j = 0; j < W; ++j) out[r * W + j] = m * a[r * W + j] + (1.0f - m) * b[r * W + j];
    }
}
