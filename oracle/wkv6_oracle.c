/*
 * wkv6_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, double-precision restatement of the RWKV-6 WKV recurrence that the
 * reference implements in CUDA.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product path
 * (rwkv_lm_ext_amd/) never does.
 *
 * Parity pin: oracle/gen_golden.py runs the reference's own CPU paths
 * (fla/ops/rwkv6/recurrent_naive.py:8-36 with autograd, and
 * src/model_encoder_run.py:30-62) in the build container, checks this file
 * against them and freezes the vectors under tests/golden/.
 *
 * Math followed (per (b,h), N = C/H, i = key/receptance channel, j = value
 * channel, d_t[i] = exp(-exp(w_t[i]))):
 *   forward   y_t[j]        = sum_i r_t[i] * (u[i] k_t[i] v_t[j] + S_t[i][j])   cuda/wkv6_cuda.cu:44-52
 *             S_{t+1}[i][j] = d_t[i] S_t[i][j] + k_t[i] v_t[j]                  cuda/wkv6_cuda.cu:54-57
 *   backward  gr_t[i] = sum_j gy_t[j] (u[i]k_t[i]v_t[j] + S_t[i][j])            cuda/wkv6_cuda.cu:100-109
 *             gk_t[i] = sum_j v_t[j]  (u[i]r_t[i]gy_t[j] + G_t[i][j])           cuda/wkv6_cuda.cu:126-134
 *             gv_t[j] = sum_i k_t[i]  (u[i]r_t[i]gy_t[j] + G_t[i][j])           cuda/wkv6_cuda.cu:149-157
 *             G_{t-1} = d_t (.) G_t + r_t gy_t^T,  G_{T-1} = 0                  cuda/wkv6_cuda.cu:128-132
 *             gu[b][i] = sum_t r_t[i]k_t[i] sum_j v_t[j]gy_t[j]                 cuda/wkv6_cuda.cu:106-112
 *             gw_t[i] = (-exp(w_t[i])) d_t[i] sum_j G_t[i][j] S_t[i][j]         cuda/wkv6_cuda.cu:161-227
 *   w is the RAW decay parameter (the reference's python wrapper forms
 *   ew = -exp(w) at src/model.py:210 and the kernel exponentiates it again at
 *   cuda/wkv6_cuda.cu:26; wkv6state does both in-kernel, cuda/wkv6state_cuda.cu:30).
 *   state variants: S_0[i][j] = s[h][j][i]  (value-major, cuda/wkv6state_cuda.cu:15,24)
 *                   or s[b][h][j][i] for infctx (cuda/wkv6infctx_cuda.cu:15);
 *             gs[b][h][j][i] = dL/dS_0[i][j]                                    cuda/wkv6state_cuda.cu:172-190
 *             final state returned in the same layout                          cuda/wkv6infctx_cuda.cu:65-67
 *   wkv6_bi:  L_b = first t with mask[b][t]==0 (inclusive bound; T-1 when the row
 *             has no zero -- SURVEY.md Q1), y_t = fwd scan over [0..L_b] plus a
 *             reverse-time scan over [0..L_b] with u = 0; y_t = 0 for t > L_b.
 *             cuda/wkv6_bi_cuda.cu:21-111.  The backward here is the exact adjoint
 *             of that forward (the reference's ignores the mask).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXN 256

/* One head, one directional scan over `n` tokens visited in the order
 * tok(p) = t0 + p*dir.  Pointers address element [t=0][channel 0 of this head];
 * consecutive tokens are `C` floats apart.  S is [i][j] row-major, double. */
static void head_forward(int n, int t0, int dir, int C, int N,
                         const float *r, const float *k, const float *v, const float *w,
                         const float *u /* may be NULL => 0 */,
                         double *S, float *y, int accumulate)
{
    double d[MAXN];
    for (int p = 0; p < n; ++p) {
        const long o = (long)(t0 + p * dir) * C;
        const float *rt = r + o, *kt = k + o, *vt = v + o, *wt = w + o;
        for (int i = 0; i < N; ++i) d[i] = exp(-exp((double)wt[i]));
        for (int j = 0; j < N; ++j) {
            double acc = 0.0;
            for (int i = 0; i < N; ++i) {
                const double kv = (double)kt[i] * (double)vt[j];
                const double uu = u ? (double)u[i] : 0.0;
                acc += (double)rt[i] * (uu * kv + S[i * N + j]);
            }
            if (accumulate) y[o + j] += (float)acc; else y[o + j] = (float)acc;
        }
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j)
                S[i * N + j] = d[i] * S[i * N + j] + (double)kt[i] * (double)vt[j];
    }
}

/* Adjoint of head_forward.  S0 is the entry state (double [i][j]); grads are
 * ACCUMULATED into gr/gk/gv/gw (float, same addressing as r) and gu (double[N]);
 * gS0 (double [i][j]) receives dL/dS_0 (overwritten).  Scratch `hist` must hold
 * n*N*N doubles. */
static void head_backward(int n, int t0, int dir, int C, int N,
                          const float *r, const float *k, const float *v, const float *w,
                          const float *u, const double *S0, const float *gy,
                          float *gr, float *gk, float *gv, float *gw,
                          double *gu, double *gS0, double *hist)
{
    const int NN = N * N;
    double d[MAXN];
    /* pass 1: entry state of every step */
    double *S = (double *)malloc(sizeof(double) * NN);
    memcpy(S, S0, sizeof(double) * NN);
    for (int p = 0; p < n; ++p) {
        const long o = (long)(t0 + p * dir) * C;
        memcpy(hist + (long)p * NN, S, sizeof(double) * NN);
        for (int i = 0; i < N; ++i) {
            const double di = exp(-exp((double)w[o + i]));
            for (int j = 0; j < N; ++j)
                S[i * N + j] = di * S[i * N + j] + (double)k[o + i] * (double)v[o + j];
        }
    }
    /* pass 2: reverse sweep with G = dL/d(state after step p) */
    double *G = S; /* reuse */
    memset(G, 0, sizeof(double) * NN);
    for (int p = n - 1; p >= 0; --p) {
        const long o = (long)(t0 + p * dir) * C;
        const double *St = hist + (long)p * NN;
        double vg = 0.0;
        for (int j = 0; j < N; ++j) vg += (double)v[o + j] * (double)gy[o + j];
        for (int i = 0; i < N; ++i) {
            const double ew = -exp((double)w[o + i]);
            d[i] = exp(ew);
            const double uu = u ? (double)u[i] : 0.0;
            const double ri = r[o + i], ki = k[o + i];
            double a_gr = 0.0, a_gk = 0.0, a_gd = 0.0;
            for (int j = 0; j < N; ++j) {
                const double gyj = gy[o + j], vj = v[o + j];
                a_gr += gyj * (uu * ki * vj + St[i * N + j]);
                a_gk += vj * (uu * ri * gyj + G[i * N + j]);
                a_gd += G[i * N + j] * St[i * N + j];
            }
            gr[o + i] += (float)a_gr;
            gk[o + i] += (float)a_gk;
            gw[o + i] += (float)(a_gd * d[i] * ew);
            if (u) gu[i] += ri * ki * vg;
        }
        for (int j = 0; j < N; ++j) {
            const double gyj = gy[o + j];
            double a_gv = 0.0;
            for (int i = 0; i < N; ++i) {
                const double uu = u ? (double)u[i] : 0.0;
                a_gv += (double)k[o + i] * (uu * (double)r[o + i] * gyj + G[i * N + j]);
            }
            gv[o + j] += (float)a_gv;
        }
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j)
                G[i * N + j] = d[i] * G[i * N + j] + (double)r[o + i] * (double)gy[o + j];
    }
    memcpy(gS0, G, sizeof(double) * NN);
    free(S);
}

static void load_state(double *S, const float *s, int N)
{   /* S[i][j] = s[j][i]   (cuda/wkv6state_cuda.cu:15,24) */
    if (!s) { memset(S, 0, sizeof(double) * N * N); return; }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) S[i * N + j] = (double)s[j * N + i];
}
static void store_state(float *s, const double *S, int N)
{
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) s[j * N + i] = (float)S[i * N + j];
}

/* s0: NULL, [H,N,N] (s0_per_batch=0, wkv6state) or [B,H,N,N] (s0_per_batch=1,
 * wkv6infctx).  s_out: NULL or [B,H,N,N] final states. */
int wkv6_oracle_forward(int B, int T, int C, int H,
                        const float *r, const float *k, const float *v, const float *w,
                        const float *u, const float *s0, int s0_per_batch,
                        float *y, float *s_out)
{
    const int N = C / H;
    if (N * H != C || N > MAXN) return -1;
    double *S = (double *)malloc(sizeof(double) * N * N);
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h) {
            const long base = (long)b * T * C + (long)h * N;
            const float *sp = s0 ? s0 + ((long)(s0_per_batch ? b : 0) * H + h) * N * N : NULL;
            load_state(S, sp, N);
            head_forward(T, 0, +1, C, N, r + base, k + base, v + base, w + base,
                         u + (long)h * N, S, y + base, 0);
            if (s_out) store_state(s_out + ((long)b * H + h) * N * N, S, N);
        }
    free(S);
    return 0;
}

/* gu: [B,C] per-batch partials (the reference sums them in python, src/model.py:232);
 * gs: NULL or [B,H,N,N] per-batch dL/dS_0 (src/model.py:181). */
int wkv6_oracle_backward(int B, int T, int C, int H,
                         const float *r, const float *k, const float *v, const float *w,
                         const float *u, const float *s0, int s0_per_batch, const float *gy,
                         float *gr, float *gk, float *gv, float *gw, float *gu, float *gs)
{
    const int N = C / H;
    if (N * H != C || N > MAXN) return -1;
    const int NN = N * N;
    double *S0 = (double *)malloc(sizeof(double) * NN);
    double *gS0 = (double *)malloc(sizeof(double) * NN);
    double *hist = (double *)malloc(sizeof(double) * (size_t)NN * (size_t)(T > 0 ? T : 1));
    double gud[MAXN];
    const long tot = (long)B * T * C;
    memset(gr, 0, sizeof(float) * tot); memset(gk, 0, sizeof(float) * tot);
    memset(gv, 0, sizeof(float) * tot); memset(gw, 0, sizeof(float) * tot);
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h) {
            const long base = (long)b * T * C + (long)h * N;
            const float *sp = s0 ? s0 + ((long)(s0_per_batch ? b : 0) * H + h) * NN : NULL;
            load_state(S0, sp, N);
            memset(gud, 0, sizeof(gud));
            head_backward(T, 0, +1, C, N, r + base, k + base, v + base, w + base, u + (long)h * N,
                          S0, gy + base, gr + base, gk + base, gv + base, gw + base, gud, gS0, hist);
            for (int i = 0; i < N; ++i) gu[(long)b * C + h * N + i] = (float)gud[i];
            if (gs) store_state(gs + ((long)b * H + h) * NN, gS0, N);
        }
    free(S0); free(gS0); free(hist);
    return 0;
}

static int row_last(const int *mask, int b, int T)
{   /* inclusive index of the last token both scans visit */
    for (int t = 0; t < T; ++t) if (mask[(long)b * T + t] == 0) return t;
    return T - 1;
}

int wkv6_oracle_bi_forward(int B, int T, int C, int H, const int *mask,
                           const float *r, const float *k, const float *v, const float *w,
                           const float *u, float *y)
{
    const int N = C / H;
    if (N * H != C || N > MAXN) return -1;
    double *S = (double *)malloc(sizeof(double) * N * N);
    memset(y, 0, sizeof(float) * (size_t)B * T * C);
    for (int b = 0; b < B; ++b) {
        const int n = row_last(mask, b, T) + 1;
        for (int h = 0; h < H; ++h) {
            const long base = (long)b * T * C + (long)h * N;
            load_state(S, NULL, N);
            head_forward(n, 0, +1, C, N, r + base, k + base, v + base, w + base,
                         u + (long)h * N, S, y + base, 0);
            load_state(S, NULL, N);
            head_forward(n, n - 1, -1, C, N, r + base, k + base, v + base, w + base,
                         NULL, S, y + base, 1);
        }
    }
    free(S);
    return 0;
}

int wkv6_oracle_bi_backward(int B, int T, int C, int H, const int *mask,
                            const float *r, const float *k, const float *v, const float *w,
                            const float *u, const float *gy,
                            float *gr, float *gk, float *gv, float *gw, float *gu)
{
    const int N = C / H;
    if (N * H != C || N > MAXN) return -1;
    const int NN = N * N;
    double *S0 = (double *)calloc(NN, sizeof(double));
    double *gS0 = (double *)malloc(sizeof(double) * NN);
    double *hist = (double *)malloc(sizeof(double) * (size_t)NN * (size_t)(T > 0 ? T : 1));
    double gud[MAXN];
    const long tot = (long)B * T * C;
    memset(gr, 0, sizeof(float) * tot); memset(gk, 0, sizeof(float) * tot);
    memset(gv, 0, sizeof(float) * tot); memset(gw, 0, sizeof(float) * tot);
    for (int b = 0; b < B; ++b) {
        const int n = row_last(mask, b, T) + 1;
        for (int h = 0; h < H; ++h) {
            const long base = (long)b * T * C + (long)h * N;
            memset(gud, 0, sizeof(gud));
            head_backward(n, 0, +1, C, N, r + base, k + base, v + base, w + base, u + (long)h * N,
                          S0, gy + base, gr + base, gk + base, gv + base, gw + base, gud, gS0, hist);
            head_backward(n, n - 1, -1, C, N, r + base, k + base, v + base, w + base, NULL,
                          S0, gy + base, gr + base, gk + base, gv + base, gw + base, gud, gS0, hist);
            for (int i = 0; i < N; ++i) gu[(long)b * C + h * N + i] = (float)gud[i];
        }
    }
    free(S0); free(gS0); free(hist);
    return 0;
}
