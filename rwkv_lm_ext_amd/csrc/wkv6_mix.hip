// Memory-bound elementwise kernels either side of the WKV6 operator in the RWKV-6 time-mix block,
// hand-written for gfx950: one pass over HBM each instead of the 5-12 eager PyTorch kernels of the reference.
//
//   ddlerp  (src/model.py:435-448):  xx = shift(x) - x;   out_s = x + xx (.) (maa_s + m_s),  s = 0..NS-1
//           NS = 1, m = null: the input of the first low-rank GEMM (x + xx (.) time_maa_x);
//           NS = 5: xw, xk, xv, xr, xg from the five low-rank corrections m_s (the bmm output, [5,B,T,C] contiguous).
//           NS = 2, m = null: the channel-mix FFN's two inputs x + xx (.) time_maa_k, x + xx (.) time_maa_r (src/model.py:636-638).
//   sqrelu  (src/model.py:640-641):  out = relu(x)^2 on the FFN's key projection;   backward dx = 2 relu(x) dout.
//   sigmul  (src/model.py:643-644):  out = sigmoid(r) (.) kv, the receptance gate on the FFN's value projection;
//           backward dr = dout kv s (1 - s), dkv = dout s.
//   gn_gate (src/model.py:462-468):  out = GroupNorm_H(y) (.) g  -- per-head normalisation of the WKV output over its 64
//           channels (nn.GroupNorm(H, C, eps) on [B*T, C]) fused with the gate multiply that feeds the output GEMM.
// Forward and backward of both; parameter gradients (time_maa_*, ln_x.weight/bias) leave as per-workgroup fp32 partial rows
// that the caller sums (deterministic, no atomics).
//
// Layout: rows of C bf16 channels; a thread owns 4 consecutive channels (8-byte accesses, a 64-channel head = 16 lanes =
// one DPP row, so the GroupNorm statistics are DPP row reductions); one workgroup of C/4 threads per row.
#include "wkv6_common.h"
#include "../../include/wkv6_amd.h"

namespace wkv6 {
namespace {


struct LerpArgs {
    int B, T, C, NS;
    const bf16_t* x;          // [B,T,C]
    const bf16_t* shifted0;   // [B,C] token in front of every row (infctx), or null (zero)
    const int* rev_n;         // [B] or null: the shift runs over the stream "first rev_n[b] tokens reversed, rest in place"
    const bf16_t* m;          // [NS,B,T,C] or null
    const bf16_t* maa;        // [NS,C]
    bf16_t* out;              // [NS,B,T,C]
    // backward
    const bf16_t* dout;       // [NS,B,T,C]
    bf16_t* dx;               // [B,T,C]
    bf16_t* dm;               // [NS,B,T,C] or null
    float* dmaa_part;         // [nparts,NS,C]
    int nparts;
};

__device__ __forceinline__ void ld4(const bf16_t* p, float (&o)[4]) { io4<bf16_t>::load(p, o); }

// Token shift over a partially reversed stream, in the ORIGINAL token order: the bidirectional encoder
// reverses the first n tokens of a row (reverse_x_idx, src/model_ext.py:410-417), runs the time-mix on that stream and
// un-reverses the result.  Stream position of token t: n-1-t for t < n, t otherwise.  prev_tok = the token one stream
// position earlier (-1: the row's leading pad), next_tok = the token one stream position later (-1: none); n = 0: plain.
__device__ __forceinline__ int prev_tok(int t, int n)
{
    if (n <= 0 || t > n) return t - 1;
    if (t < n - 1) return t + 1;
    return t == n ? 0 : -1;                              // t == n: behind the reversed span sits token 0; t == n-1: stream start
}
__device__ __forceinline__ int next_tok(int t, int n, int T)
{
    if (n <= 0 || t >= n) return t + 1 < T ? t + 1 : -1;
    if (t >= 1) return t - 1;
    return n < T ? n : -1;                               // token 0 closes the reversed span
}

template <int NS, bool HAS_M>
__global__ void ddlerp_fwd_kernel(const LerpArgs a)
{
    const long row = blockIdx.x;                         // b*T + t
    const int t = (int)(row % a.T), b = (int)(row / a.T);
    const int c = 4 * threadIdx.x;
    float x[4], xp[4] = {0.f, 0.f, 0.f, 0.f};
    ld4(a.x + row * a.C + c, x);
    const int nrev = a.rev_n ? min(max(a.rev_n[b], 0), a.T) : 0;
    const int tp = prev_tok(t, nrev);
    if (tp >= 0) ld4(a.x + ((long)b * a.T + tp) * a.C + c, xp);
    else if (a.shifted0) ld4(a.shifted0 + (long)b * a.C + c, xp);
    const long plane = (long)a.B * a.T * a.C;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float maa[4], m[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
        ld4(a.maa + (long)s * a.C + c, maa);
        if constexpr (HAS_M) ld4(a.m + s * plane + row * a.C + c, m);
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = fmaf(xp[q] - x[q], maa[q] + m[q], x[q]);
        io4<bf16_t>::store(a.out + s * plane + row * a.C + c, o);
    }
}

// Backward: dx_t = sum_s dout_{s,t} (1 - c_{s,t}) + sum_s dout_{s,t+1} c_{s,t+1}  (c = maa + m; the second term is the adjoint
// of the token shift), dm_{s,t} = dout_{s,t} xx_t, dmaa_s = sum_rows dout_s xx.  Workgroup p handles rows p, p + nparts, ...
template <int NS, bool HAS_M>
__global__ void ddlerp_bwd_kernel(const LerpArgs a)
{
    const int c = 4 * threadIdx.x;
    const long plane = (long)a.B * a.T * a.C, rows = (long)a.B * a.T;
    float maa[NS][4], acc[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        ld4(a.maa + (long)s * a.C + c, maa[s]);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[s][q] = 0.f;
    }
    if (!a.rev_n) {
        // Plain stream: a workgroup walks a contiguous run of rows backwards, so what token t+1 hands to token t -- the part of
        // its blends that came from x_t:  sum_s dout[s][t+1] (maa_s + m_s[t+1])  -- is four floats carried in registers
        // instead of a second read of dout and m (54 -> 34 bytes per token-channel at NS = 5).
        const long per = (rows + gridDim.x - 1) / gridDim.x, r0 = (long)blockIdx.x * per, r1 = min(rows, r0 + per);
        float carry[4] = {0.f, 0.f, 0.f, 0.f};
        if (r1 > r0 && r1 < rows && r1 % a.T != 0) {          // the run ends inside a sequence: fetch the next row's hand-over once
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float dn[4], mn[4] = {0.f, 0.f, 0.f, 0.f};
                ld4(a.dout + s * plane + r1 * a.C + c, dn);
                if constexpr (HAS_M) ld4(a.m + s * plane + r1 * a.C + c, mn);
#pragma unroll
                for (int q = 0; q < 4; ++q) carry[q] = fmaf(dn[q], maa[s][q] + mn[q], carry[q]);
            }
        }
        for (long row = r1 - 1; row >= r0; --row) {
            const int t = (int)(row % a.T), b = (int)(row / a.T);
            float x[4], xp[4] = {0.f, 0.f, 0.f, 0.f}, g[4], own[4] = {0.f, 0.f, 0.f, 0.f};
            ld4(a.x + row * a.C + c, x);
            if (t > 0) ld4(a.x + (row - 1) * a.C + c, xp);
            else if (a.shifted0) ld4(a.shifted0 + (long)b * a.C + c, xp);
            const bool last = t == a.T - 1;                       // nothing behind the last token of a sequence
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = last ? 0.f : carry[q];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float d[4], m[4] = {0.f, 0.f, 0.f, 0.f}, dm[4];
                ld4(a.dout + s * plane + row * a.C + c, d);
                if constexpr (HAS_M) ld4(a.m + s * plane + row * a.C + c, m);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float wgt = maa[s][q] + m[q];
                    g[q] = fmaf(d[q], 1.f - wgt, g[q]);
                    own[q] = fmaf(d[q], wgt, own[q]);
                    dm[q] = d[q] * (xp[q] - x[q]);
                    acc[s][q] += dm[q];
                }
                if constexpr (HAS_M) io4<bf16_t>::store(a.dm + s * plane + row * a.C + c, dm);
            }
            io4<bf16_t>::store(a.dx + row * a.C + c, g);
#pragma unroll
            for (int q = 0; q < 4; ++q) carry[q] = own[q];
        }
    } else
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {       // reversed-span streams: the neighbours are not adjacent rows
        const int t = (int)(row % a.T), b = (int)(row / a.T);
        float x[4], xp[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f};
        ld4(a.x + row * a.C + c, x);
        const int nrev = min(max(a.rev_n[b], 0), a.T);
        const int tp = prev_tok(t, nrev), tn = next_tok(t, nrev, a.T);
        if (tp >= 0) ld4(a.x + ((long)b * a.T + tp) * a.C + c, xp);
        else if (a.shifted0) ld4(a.shifted0 + (long)b * a.C + c, xp);
        if (tn >= 0) {                                        // (summed in the order of the plain path: the two agree bit for bit)
            const long rown = (long)b * a.T + tn;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float dn[4], mn[4] = {0.f, 0.f, 0.f, 0.f};
                ld4(a.dout + s * plane + rown * a.C + c, dn);
                if constexpr (HAS_M) ld4(a.m + s * plane + rown * a.C + c, mn);
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] = fmaf(dn[q], maa[s][q] + mn[q], g[q]);
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float d[4], m[4] = {0.f, 0.f, 0.f, 0.f};
            ld4(a.dout + s * plane + row * a.C + c, d);
            if constexpr (HAS_M) ld4(a.m + s * plane + row * a.C + c, m);
            float dm[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                g[q] = fmaf(d[q], 1.f - (maa[s][q] + m[q]), g[q]);
                dm[q] = d[q] * (xp[q] - x[q]);
                acc[s][q] += dm[q];
            }
            if constexpr (HAS_M) io4<bf16_t>::store(a.dm + s * plane + row * a.C + c, dm);
        }
        io4<bf16_t>::store(a.dx + row * a.C + c, g);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
        *reinterpret_cast<float4*>(a.dmaa_part + ((long)blockIdx.x * NS + s) * a.C + c) =
            make_float4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
}

struct GnArgs {
    long rows;
    int C, H;
    float eps;
    const bf16_t *y, *g, *gamma, *beta;
    bf16_t* out;
    float* stats;             // [rows, H, 2] mean, rstd (saved by the forward for the backward)
    const bf16_t* dout;
    bf16_t *dy, *dg;
    float *dgamma_part, *dbeta_part;   // [nparts, C]
    int nparts;
};

__global__ void gn_gate_fwd_kernel(const GnArgs a)
{
    const long row = blockIdx.x;
    const int c = 4 * threadIdx.x, head = c >> 6;
    float y[4], g[4], ga[4], be[4], o[4];
    ld4(a.y + row * a.C + c, y);
    ld4(a.g + row * a.C + c, g);
    ld4(a.gamma + c, ga);
    ld4(a.beta + c, be);
    const float mean = row_sum16(y[0] + y[1] + y[2] + y[3]) * (1.f / 64.f);
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) var = fmaf(y[q] - mean, y[q] - mean, var);
    var = row_sum16(var) * (1.f / 64.f);                  // biased variance, as nn.GroupNorm
    const float rstd = rsqrtf(var + a.eps);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = fmaf((y[q] - mean) * rstd, ga[q], be[q]) * g[q];
    io4<bf16_t>::store(a.out + row * a.C + c, o);
    if (a.stats && (threadIdx.x & 15) == 0) {
        a.stats[(row * a.H + head) * 2] = mean;
        a.stats[(row * a.H + head) * 2 + 1] = rstd;
    }
}

// no = xhat gamma + beta, out = no g:  dg = dout no;  dno = dout g;  dxhat = dno gamma;
// dy = rstd (dxhat - mean_64(dxhat) - xhat mean_64(dxhat xhat));  dgamma += dno xhat;  dbeta += dno.
__global__ void gn_gate_bwd_kernel(const GnArgs a)
{
    const int c = 4 * threadIdx.x, head = c >> 6;
    float ga[4], be[4], accg[4] = {0.f, 0.f, 0.f, 0.f}, accb[4] = {0.f, 0.f, 0.f, 0.f};
    ld4(a.gamma + c, ga);
    ld4(a.beta + c, be);
    for (long row = blockIdx.x; row < a.rows; row += gridDim.x) {
        float y[4], g[4], d[4], dy[4], dg[4], dxh[4], xh[4];
        ld4(a.y + row * a.C + c, y);
        ld4(a.g + row * a.C + c, g);
        ld4(a.dout + row * a.C + c, d);
        const float mean = a.stats[(row * a.H + head) * 2], rstd = a.stats[(row * a.H + head) * 2 + 1];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            xh[q] = (y[q] - mean) * rstd;
            dg[q] = d[q] * fmaf(xh[q], ga[q], be[q]);
            const float dno = d[q] * g[q];
            accg[q] = fmaf(dno, xh[q], accg[q]);
            accb[q] += dno;
            dxh[q] = dno * ga[q];
            s1 += dxh[q];
            s2 = fmaf(dxh[q], xh[q], s2);
        }
        s1 = row_sum16(s1) * (1.f / 64.f);
        s2 = row_sum16(s2) * (1.f / 64.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) dy[q] = rstd * (dxh[q] - s1 - xh[q] * s2);
        io4<bf16_t>::store(a.dy + row * a.C + c, dy);
        io4<bf16_t>::store(a.dg + row * a.C + c, dg);
    }
    *reinterpret_cast<float4*>(a.dgamma_part + (long)blockIdx.x * a.C + c) = make_float4(accg[0], accg[1], accg[2], accg[3]);
    *reinterpret_cast<float4*>(a.dbeta_part + (long)blockIdx.x * a.C + c) = make_float4(accb[0], accb[1], accb[2], accb[3]);
}

// ---- flat elementwise kernels of the channel-mix FFN: 8 bf16 per lane (16-byte accesses), grid-stride over n / 8 units -----------
struct FlatArgs {
    long n8;                  // number of 8-element units (n / 8)
    const bf16_t *a, *b, *dout;
    bf16_t *out, *da, *db;
};
__device__ __forceinline__ void ld8(const bf16_t* p, float (&o)[8])
{
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    o[0] = bf_lo(raw.x); o[1] = bf_hi(raw.x); o[2] = bf_lo(raw.y); o[3] = bf_hi(raw.y);
    o[4] = bf_lo(raw.z); o[5] = bf_hi(raw.z); o[6] = bf_lo(raw.w); o[7] = bf_hi(raw.w);
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8])
{
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

template <bool BWD> __global__ __launch_bounds__(256) void sqrelu_kernel(const FlatArgs a)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n8; i += (long)gridDim.x * blockDim.x) {
        float x[8], o[8];
        ld8(a.a + 8 * i, x);
        if constexpr (BWD) {
            float d[8];
            ld8(a.dout + 8 * i, d);
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = 2.f * fmaxf(x[q], 0.f) * d[q];
            st8(a.da + 8 * i, o);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float r = fmaxf(x[q], 0.f); o[q] = r * r; }
            st8(a.out + 8 * i, o);
        }
    }
}
template <bool BWD> __global__ __launch_bounds__(256) void sigmul_kernel(const FlatArgs a)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n8; i += (long)gridDim.x * blockDim.x) {
        float r[8], kv[8], o[8];
        ld8(a.a + 8 * i, r);
        ld8(a.b + 8 * i, kv);
        if constexpr (BWD) {
            float d[8], o2[8];
            ld8(a.dout + 8 * i, d);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float s = sigmoidf_(r[q]);
                o[q] = d[q] * kv[q] * s * (1.f - s);
                o2[q] = d[q] * s;
            }
            st8(a.da + 8 * i, o);
            st8(a.db + 8 * i, o2);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = sigmoidf_(r[q]) * kv[q];
            st8(a.out + 8 * i, o);
        }
    }
}
template <typename K> int launch_flat(K kernel, const FlatArgs& a, hipStream_t st)
{
    const long blocks = (a.n8 + 255) / 256;
    hipLaunchKernelGGL(kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? WKV6_OK : (int)e;
}

int check_rows(long rows, int C)
{
    if (rows < 1 || C < 64 || C % 64 != 0 || C / 4 > 1024) return WKV6_EINVAL;
    if (rows * (long)C >= (1L << 40)) return WKV6_EUNSUPPORTED;
    return WKV6_OK;
}

template <int NS, bool HAS_M> void launch_lerp(const LerpArgs& a, bool bwd, hipStream_t st)
{
    if (bwd) hipLaunchKernelGGL((ddlerp_bwd_kernel<NS, HAS_M>), dim3(a.nparts), dim3(a.C / 4), 0, st, a);
    else hipLaunchKernelGGL((ddlerp_fwd_kernel<NS, HAS_M>), dim3((unsigned)((long)a.B * a.T)), dim3(a.C / 4), 0, st, a);
}

int dispatch_lerp(const LerpArgs& a, bool bwd, hipStream_t st)
{
    if (a.NS == 1 && !a.m) launch_lerp<1, false>(a, bwd, st);
    else if (a.NS == 5 && a.m) launch_lerp<5, true>(a, bwd, st);
    else if (a.NS == 1 && a.m) launch_lerp<1, true>(a, bwd, st);
    else if (a.NS == 2 && !a.m) launch_lerp<2, false>(a, bwd, st);
    else return WKV6_EUNSUPPORTED;
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? WKV6_OK : (int)e;
}

}  // namespace
}  // namespace wkv6

using namespace wkv6;

extern "C" {

int wkv6_ddlerp_rev_forward(int B, int T, int C, int NS, const void* x, const void* shifted0, const void* m, const void* maa,
                            const int* rev_n, void* out, void* stream)
{
    if (B < 1 || T < 1) return WKV6_EINVAL;
    if (int rc = check_rows((long)B * T, C)) return rc;
    if (!x || !maa || !out) return WKV6_ENULL;
    LerpArgs a = {};
    a.B = B; a.T = T; a.C = C; a.NS = NS;
    a.x = (const bf16_t*)x; a.shifted0 = (const bf16_t*)shifted0; a.m = (const bf16_t*)m; a.maa = (const bf16_t*)maa;
    a.rev_n = rev_n;
    a.out = (bf16_t*)out;
    return dispatch_lerp(a, false, (hipStream_t)stream);
}
int wkv6_ddlerp_forward(int B, int T, int C, int NS, const void* x, const void* shifted0, const void* m, const void* maa,
                        void* out, void* stream)
{
    return wkv6_ddlerp_rev_forward(B, T, C, NS, x, shifted0, m, maa, nullptr, out, stream);
}

int wkv6_ddlerp_backward(int B, int T, int C, int NS, const void* x, const void* shifted0, const void* m, const void* maa,
                         const void* dout, void* dx, void* dm, float* dmaa_part, int nparts, void* stream)
{
    return wkv6_ddlerp_rev_backward(B, T, C, NS, x, shifted0, m, maa, nullptr, dout, dx, dm, dmaa_part, nparts, stream);
}
int wkv6_ddlerp_rev_backward(int B, int T, int C, int NS, const void* x, const void* shifted0, const void* m, const void* maa,
                             const int* rev_n, const void* dout, void* dx, void* dm, float* dmaa_part, int nparts, void* stream)
{
    if (B < 1 || T < 1 || nparts < 1) return WKV6_EINVAL;
    if (int rc = check_rows((long)B * T, C)) return rc;
    if (!x || !maa || !dout || !dx || !dmaa_part || (m && !dm)) return WKV6_ENULL;
    LerpArgs a = {};
    a.B = B; a.T = T; a.C = C; a.NS = NS;
    a.x = (const bf16_t*)x; a.shifted0 = (const bf16_t*)shifted0; a.m = (const bf16_t*)m; a.maa = (const bf16_t*)maa;
    a.rev_n = rev_n;
    a.dout = (const bf16_t*)dout; a.dx = (bf16_t*)dx; a.dm = (bf16_t*)dm; a.dmaa_part = dmaa_part; a.nparts = nparts;
    return dispatch_lerp(a, true, (hipStream_t)stream);
}

int wkv6_sqrelu_forward(long n, const void* x, void* out, void* stream)
{
    if (n < 8 || n % 8) return WKV6_EINVAL;
    if (!x || !out) return WKV6_ENULL;
    FlatArgs a = {};
    a.n8 = n / 8; a.a = (const bf16_t*)x; a.out = (bf16_t*)out;
    return launch_flat(sqrelu_kernel<false>, a, (hipStream_t)stream);
}
int wkv6_sqrelu_backward(long n, const void* x, const void* dout, void* dx, void* stream)
{
    if (n < 8 || n % 8) return WKV6_EINVAL;
    if (!x || !dout || !dx) return WKV6_ENULL;
    FlatArgs a = {};
    a.n8 = n / 8; a.a = (const bf16_t*)x; a.dout = (const bf16_t*)dout; a.da = (bf16_t*)dx;
    return launch_flat(sqrelu_kernel<true>, a, (hipStream_t)stream);
}
int wkv6_sigmul_forward(long n, const void* r, const void* kv, void* out, void* stream)
{
    if (n < 8 || n % 8) return WKV6_EINVAL;
    if (!r || !kv || !out) return WKV6_ENULL;
    FlatArgs a = {};
    a.n8 = n / 8; a.a = (const bf16_t*)r; a.b = (const bf16_t*)kv; a.out = (bf16_t*)out;
    return launch_flat(sigmul_kernel<false>, a, (hipStream_t)stream);
}
int wkv6_sigmul_backward(long n, const void* r, const void* kv, const void* dout, void* dr, void* dkv, void* stream)
{
    if (n < 8 || n % 8) return WKV6_EINVAL;
    if (!r || !kv || !dout || !dr || !dkv) return WKV6_ENULL;
    FlatArgs a = {};
    a.n8 = n / 8; a.a = (const bf16_t*)r; a.b = (const bf16_t*)kv; a.dout = (const bf16_t*)dout; a.da = (bf16_t*)dr; a.db = (bf16_t*)dkv;
    return launch_flat(sigmul_kernel<true>, a, (hipStream_t)stream);
}

int wkv6_gn_gate_forward(long rows, int C, int H, const void* y, const void* g, const void* gamma, const void* beta,
                         float eps, void* out, float* stats, void* stream)
{
    if (int rc = check_rows(rows, C)) return rc;
    if (H * HEAD != C) return WKV6_EINVAL;
    if (!y || !g || !gamma || !beta || !out) return WKV6_ENULL;
    GnArgs a = {};
    a.rows = rows; a.C = C; a.H = H; a.eps = eps;
    a.y = (const bf16_t*)y; a.g = (const bf16_t*)g; a.gamma = (const bf16_t*)gamma; a.beta = (const bf16_t*)beta;
    a.out = (bf16_t*)out; a.stats = stats;
    hipLaunchKernelGGL(gn_gate_fwd_kernel, dim3((unsigned)rows), dim3(C / 4), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? WKV6_OK : (int)e;
}

int wkv6_gn_gate_backward(long rows, int C, int H, const void* y, const void* g, const void* gamma, const void* beta,
                          const float* stats, const void* dout, void* dy, void* dg, float* dgamma_part, float* dbeta_part,
                          int nparts, void* stream)
{
    if (int rc = check_rows(rows, C)) return rc;
    if (H * HEAD != C || nparts < 1) return WKV6_EINVAL;
    if (!y || !g || !gamma || !beta || !stats || !dout || !dy || !dg || !dgamma_part || !dbeta_part) return WKV6_ENULL;
    GnArgs a = {};
    a.rows = rows; a.C = C; a.H = H;
    a.y = (const bf16_t*)y; a.g = (const bf16_t*)g; a.gamma = (const bf16_t*)gamma; a.beta = (const bf16_t*)beta;
    a.stats = const_cast<float*>(stats); a.dout = (const bf16_t*)dout; a.dy = (bf16_t*)dy; a.dg = (bf16_t*)dg;
    a.dgamma_part = dgamma_part; a.dbeta_part = dbeta_part; a.nparts = nparts;
    hipLaunchKernelGGL(gn_gate_bwd_kernel, dim3(nparts), dim3(C / 4), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? WKV6_OK : (int)e;
}

}  // extern "C"
