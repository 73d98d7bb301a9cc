// Exact-fp32 token-serial WKV6 kernels for gfx950.  Design notes: wkv6_scan.h.
#include "wkv6_scan.h"

namespace wkv6 {
namespace {

constexpr int TB = 16;                 // tokens staged per LDS batch
constexpr int ROW = HEAD;              // floats per staged token row

// ---- CPT-wide channel I/O (CPT = 2 or 4) ----------------------------------------------------------
template <typename T, int CPT> struct ion;
template <typename T> struct ion<T, 4> : io4<T> {};
template <> struct ion<bf16_t, 2> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&o)[2])
    {
        const uint32_t raw = *reinterpret_cast<const uint32_t*>(p);
        o[0] = bf_lo(raw); o[1] = bf_hi(raw);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[2])
    {
        *reinterpret_cast<uint32_t*>(p) = pack_bf2(v[0], v[1]);
    }
};
template <> struct ion<f16_t, 2> {
    typedef f16_t h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ void load(const f16_t* p, float (&o)[2])
    {
        const h2 raw = *reinterpret_cast<const h2*>(p);
        o[0] = (float)raw[0]; o[1] = (float)raw[1];
    }
    static __device__ __forceinline__ void store(f16_t* p, const float (&v)[2])
    {
        const h2 raw = {(f16_t)v[0], (f16_t)v[1]};
        *reinterpret_cast<h2*>(p) = raw;
    }
};
template <> struct ion<float, 2> {
    static __device__ __forceinline__ void load(const float* p, float (&o)[2])
    {
        const float2 raw = *reinterpret_cast<const float2*>(p);
        o[0] = raw.x; o[1] = raw.y;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[2])
    {
        *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
    }
};
template <int CPT> __device__ __forceinline__ void lds_store(float* p, const float (&v)[CPT])
{
    if constexpr (CPT == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
}
template <int CPT> __device__ __forceinline__ void lds_load(const float* p, float (&v)[CPT])
{
    if constexpr (CPT == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
    }
}
// sum over the TPT consecutive lanes that stage one token (TPT = 16 or 32)
template <int TPT> __device__ __forceinline__ float token_sum(float x)
{
    x = row_sum16(x);
    if constexpr (TPT == 32) x += __shfl_xor(x, 16);
    return x;
}
template <int NV> __device__ __forceinline__ float pick(const float (&v)[NV], int s)
{
    float o = v[0];
#pragma unroll
    for (int q = 1; q < NV; ++q) o = (s == q) ? v[q] : o;
    return o;
}

// Per-thread staging geometry shared by the three kernels.
template <int NW> struct Geo {
    static constexpr int NT = NW * 64;
    static constexpr int CPT = TB * ROW / NT;      // channels staged per thread (4 or 2)
    static constexpr int TPT = ROW / CPT;          // threads per token (16 or 32)
    static_assert(CPT == 2 || CPT == 4, "unsupported wave count");
};

template <typename T, int CPT>
__device__ __forceinline__ void load_ew(const ScanArgs& a, long idx, bool valid, float (&ew)[CPT])
{
#pragma unroll
    for (int c = 0; c < CPT; ++c) ew[c] = 0.f;
    if (!valid) return;
    if (a.wkind != 1) {
        lds_load<CPT>(reinterpret_cast<const float*>(a.w) + idx, ew);   // plain (global) vector load (ew, or d when wkind == 2)
    } else {
        float w[CPT];
        ion<T, CPT>::load(reinterpret_cast<const T*>(a.w) + idx, w);
#pragma unroll
        for (int c = 0; c < CPT; ++c) ew[c] = -__expf(w[c]);
    }
}

// =====================================================================================================
// forward:  y_t[j] = sum_i r_t[i] S_t[i][j] + (sum_i r_t[i]u[i]k_t[i]) v_t[j];  S <- d_t (.) S + k_t v_t^T
// (cuda/wkv6_cuda.cu:44-57).  Wave `wv` owns value columns [wv*JPW, (wv+1)*JPW); lane (jb = lane>>4,
// ib = lane&15) owns S[4ib..4ib+3][j0..j0+JR-1].
// =====================================================================================================
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void scan_fwd_kernel(const ScanArgs a)
{
    using G = Geo<NW>;
    constexpr int CPT = G::CPT, TPT = G::TPT, JPW = HEAD / NW, JR = JPW / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const inb = smem;                          // [2][4][TB][ROW]  r,k,d,v
    float* const coef = smem + 2 * 4 * TB * ROW;      // [2][TB] (64 floats reserved)
    float* const ys = coef + 64;                      // [2][TB][ROW]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const T* const gr_ = reinterpret_cast<const T*>(a.r);
    const T* const gk_ = reinterpret_cast<const T*>(a.k);
    const T* const gv_ = reinterpret_cast<const T*>(a.v);
    T* const gy_ = reinterpret_cast<T*>(a.y);
    int ntok = a.T;
    if (a.lens) ntok = min(max(a.lens[b], 0), a.T);
    const RevMap tokmap = make_revmap(a, b, ntok);      // token each tensor holds at scan position p (wkv6_scan.h)
    const long base = (long)b * a.T * a.C + (long)h * HEAD;

    // staging role
    const int spp = tid / TPT, sc0 = (tid % TPT) * CPT;
    float uu[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) uu[c] = 0.f;
    if (a.use_u) ion<T, CPT>::load(reinterpret_cast<const T*>(a.u) + h * HEAD + sc0, uu);

    // compute role
    const int jb = lane >> 4, ib = lane & 15;
    const int i0 = ib * 4, j0 = wv * JPW + jb * JR;
    float S[4][JR];
#pragma unroll
    for (int jj = 0; jj < JR; ++jj) {
        float t4[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.s0) {
            const long so_ = (long)b * a.s0_bstride + ((long)h * HEAD + j0 + jj) * HEAD + i0;
            if (a.state_f32) io4<float>::load(reinterpret_cast<const float*>(a.s0) + so_, t4);
            else io4<T>::load(reinterpret_cast<const T*>(a.s0) + so_, t4);
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) S[ii][jj] = t4[ii];
    }

    float pr[CPT], pk[CPT], pv[CPT], pew[CPT];
    auto load_regs = [&](int q) {
        const int p = q * TB + spp;
        const bool valid = p < ntok;
        [[maybe_unused]] const long idx_r = base + (long)tokmap(p, REV_R) * a.C + sc0, idx_k = base + (long)tokmap(p, REV_K) * a.C + sc0;
        [[maybe_unused]] const long idx_v = base + (long)tokmap(p, REV_V) * a.C + sc0, idx_w = base + (long)tokmap(p, REV_W) * a.C + sc0;
        [[maybe_unused]] const long idx_y = base + (long)tokmap(p, REV_Y) * a.C + sc0, idx_p = base + (long)p * a.C + sc0;   // idx_p: scratch indexed by scan position
#pragma unroll
        for (int c = 0; c < CPT; ++c) { pr[c] = 0.f; pk[c] = 0.f; pv[c] = 0.f; }
        if (valid) {
            ion<T, CPT>::load(gr_ + idx_r, pr);
            ion<T, CPT>::load(gk_ + idx_k, pk);
            ion<T, CPT>::load(gv_ + idx_v, pv);
        }
        load_ew<T, CPT>(a, idx_w, valid, pew);
    };
    auto write_lds = [&](int buf) {
        float* const ib_ = inb + buf * 4 * TB * ROW + spp * ROW + sc0;
        float d[CPT];
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            d[c] = a.wkind == 2 ? pew[c] : __expf(pew[c]);
            part = fmaf(pr[c] * uu[c], pk[c], part);
        }
        lds_store<CPT>(ib_, pr);
        lds_store<CPT>(ib_ + TB * ROW, pk);
        lds_store<CPT>(ib_ + 2 * TB * ROW, d);
        lds_store<CPT>(ib_ + 3 * TB * ROW, pv);
        part = token_sum<TPT>(part);
        if ((tid % TPT) == 0) coef[buf * TB + spp] = part;
    };

    const int nq = (ntok + TB - 1) / TB;
    if (nq > 0) {
        load_regs(0);
        write_lds(0);
    }
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
        const int buf = q & 1;
        if (q + 1 < nq) load_regs(q + 1);
        {   // ---- scan the staged tokens
            const float* const rs = inb + buf * 4 * TB * ROW;
            const float* const ks = rs + TB * ROW;
            const float* const ds = rs + 2 * TB * ROW;
            const float* const vs = rs + 3 * TB * ROW;
            float* const yb = ys + buf * TB * ROW;
            const int nb = min(TB, ntok - q * TB);

            for (int pp = 0; pp < nb; ++pp) {
                float r4[4], k4[4], d4[4], vv[JR], yacc[JR];
                lds_load<4>(rs + pp * ROW + i0, r4);
                lds_load<4>(ks + pp * ROW + i0, k4);
                lds_load<4>(ds + pp * ROW + i0, d4);
                lds_load<JR>(vs + pp * ROW + j0, vv);
#pragma unroll
                for (int jj = 0; jj < JR; ++jj) yacc[jj] = 0.f;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int jj = 0; jj < JR; ++jj) {
                        const float kv = k4[ii] * vv[jj];
                        yacc[jj] = fmaf(r4[ii], S[ii][jj], yacc[jj]);
                        S[ii][jj] = fmaf(S[ii][jj], d4[ii], kv);
                    }
                const float tot = row_reduce(yacc, ib);
                if (ib < JR) {
                    const int jj = row_sel<JR>(ib);
                    yb[pp * ROW + j0 + jj] = fmaf(coef[buf * TB + pp], pick<JR>(vv, jj), tot);
                }
            }
        }
        if (q + 1 < nq) write_lds(buf ^ 1);
        __syncthreads();
        {   // ---- coalesced store of this batch's outputs
            const int p = q * TB + spp;
            if (p < ntok) {
                [[maybe_unused]] const long idx_r = base + (long)tokmap(p, REV_R) * a.C + sc0, idx_k = base + (long)tokmap(p, REV_K) * a.C + sc0;
                [[maybe_unused]] const long idx_v = base + (long)tokmap(p, REV_V) * a.C + sc0, idx_w = base + (long)tokmap(p, REV_W) * a.C + sc0;
                [[maybe_unused]] const long idx_y = base + (long)tokmap(p, REV_Y) * a.C + sc0, idx_p = base + (long)p * a.C + sc0;   // idx_p: scratch indexed by scan position
                float o[CPT];
                lds_load<CPT>(ys + buf * TB * ROW + spp * ROW + sc0, o);
                if (a.accumulate) {
                    float old[CPT];
                    if (a.y_f32) lds_load<CPT>(a.y_f32 + idx_y, old);
                    else ion<T, CPT>::load(gy_ + idx_y, old);
#pragma unroll
                    for (int c = 0; c < CPT; ++c) o[c] += old[c];
                }
                if (a.y_f32 && !a.accumulate) lds_store<CPT>(a.y_f32 + idx_y, o);
                else ion<T, CPT>::store(gy_ + idx_y, o);
            }
        }
    }
    if (a.s_out) {
        const long so_ = ((long)b * a.H + h) * HEAD * HEAD;
#pragma unroll
        for (int jj = 0; jj < JR; ++jj) {
            const float t4[4] = {S[0][jj], S[1][jj], S[2][jj], S[3][jj]};
            if (a.state_f32) io4<float>::store(reinterpret_cast<float*>(a.s_out) + so_ + (long)(j0 + jj) * HEAD + i0, t4);
            else io4<T>::store(reinterpret_cast<T*>(a.s_out) + so_ + (long)(j0 + jj) * HEAD + i0, t4);
        }
    }
    if (a.zero_tail && !a.accumulate) {
        const float z[CPT] = {};
        for (int t = ntok + spp; t < a.T; t += TB) ion<T, CPT>::store(gy_ + base + (long)t * a.C + sc0, z);
    }
}

// =====================================================================================================
// backward sweep S (scan order):  dq_t[i] = sum_j gy_t[j] S_t[i][j]   (cuda/wkv6_cuda.cu:100-109 minus
// its u-term), gr_t = dq_t + u (.) k_t (v_t.gy_t), aux a_t = r_t (.) dq_t, gu += r_t (.) k_t (v_t.gy_t)
// (cuda/wkv6_cuda.cu:106-112).  Wave `wv` owns key rows [wv*IPW, (wv+1)*IPW); lane (irow = lane>>4,
// jl = lane&15) owns S[i0..i0+IR-1][4jl..4jl+3].
// =====================================================================================================
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void scan_bwd_s_kernel(const ScanArgs a)
{
    using G = Geo<NW>;
    constexpr int CPT = G::CPT, TPT = G::TPT, IPW = HEAD / NW, IR = IPW / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const inb = smem;                          // [2][4][TB][ROW]  k,d,v,gy
    float* const dqs = smem + 2 * 4 * TB * ROW;       // [2][TB][ROW]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const T* const gr_ = reinterpret_cast<const T*>(a.r);
    const T* const gk_ = reinterpret_cast<const T*>(a.k);
    const T* const gv_ = reinterpret_cast<const T*>(a.v);
    const T* const ggy = reinterpret_cast<const T*>(a.gy);
    T* const ogr = reinterpret_cast<T*>(a.gr);
    int ntok = a.T;
    if (a.lens) ntok = min(max(a.lens[b], 0), a.T);
    const RevMap tokmap = make_revmap(a, b, ntok);      // token each tensor holds at scan position p (wkv6_scan.h)
    const long base = (long)b * a.T * a.C + (long)h * HEAD;

    const int spp = tid / TPT, sc0 = (tid % TPT) * CPT;
    float uu[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) uu[c] = 0.f;
    if (a.use_u) ion<T, CPT>::load(reinterpret_cast<const T*>(a.u) + h * HEAD + sc0, uu);

    const int irow = lane >> 4, jl = lane & 15;
    const int i0 = wv * IPW + irow * IR, j0 = jl * 4;
    float S[IR][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        float t2[IR];
#pragma unroll
        for (int ii = 0; ii < IR; ++ii) t2[ii] = 0.f;
        if (a.s0)
            ion<T, IR>::load(reinterpret_cast<const T*>(a.s0) + (long)b * a.s0_bstride +
                             ((long)h * HEAD + j0 + jj) * HEAD + i0, t2);
#pragma unroll
        for (int ii = 0; ii < IR; ++ii) S[ii][jj] = t2[ii];
    }

    float pr[CPT], pk[CPT], pv[CPT], pew[CPT], pgy[CPT];
    float nr[CPT], nk[CPT], nvg = 0.f;                 // values of the batch just written to LDS
    float cr[CPT], ck[CPT], cvg = 0.f;                 // values of the batch being scanned
    float gu_acc[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) { gu_acc[c] = 0.f; nr[c] = nk[c] = cr[c] = ck[c] = 0.f; }

    auto load_regs = [&](int q) {
        const int p = q * TB + spp;
        const bool valid = p < ntok;
        [[maybe_unused]] const long idx_r = base + (long)tokmap(p, REV_R) * a.C + sc0, idx_k = base + (long)tokmap(p, REV_K) * a.C + sc0;
        [[maybe_unused]] const long idx_v = base + (long)tokmap(p, REV_V) * a.C + sc0, idx_w = base + (long)tokmap(p, REV_W) * a.C + sc0;
        [[maybe_unused]] const long idx_y = base + (long)tokmap(p, REV_Y) * a.C + sc0, idx_p = base + (long)p * a.C + sc0;   // idx_p: scratch indexed by scan position
#pragma unroll
        for (int c = 0; c < CPT; ++c) { pr[c] = 0.f; pk[c] = 0.f; pv[c] = 0.f; pgy[c] = 0.f; }
        if (valid) {
            ion<T, CPT>::load(gr_ + idx_r, pr);
            ion<T, CPT>::load(gk_ + idx_k, pk);
            ion<T, CPT>::load(gv_ + idx_v, pv);
            ion<T, CPT>::load(ggy + idx_y, pgy);
        }
        load_ew<T, CPT>(a, idx_w, valid, pew);
    };
    auto write_lds = [&](int buf) {
        float* const ib_ = inb + buf * 4 * TB * ROW + spp * ROW + sc0;
        float d[CPT];
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            d[c] = __expf(pew[c]);
            part = fmaf(pv[c], pgy[c], part);
        }
        lds_store<CPT>(ib_, pk);
        lds_store<CPT>(ib_ + TB * ROW, d);
        lds_store<CPT>(ib_ + 2 * TB * ROW, pv);
        lds_store<CPT>(ib_ + 3 * TB * ROW, pgy);
        nvg = token_sum<TPT>(part);
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            nr[c] = pr[c]; nk[c] = pk[c];
            gu_acc[c] = fmaf(pr[c] * pk[c], nvg, gu_acc[c]);
        }
    };

    const int nq = (ntok + TB - 1) / TB;
    if (nq > 0) {
        load_regs(0);
        write_lds(0);
    }
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
        const int buf = q & 1;
#pragma unroll
        for (int c = 0; c < CPT; ++c) { cr[c] = nr[c]; ck[c] = nk[c]; }
        cvg = nvg;
        if (q + 1 < nq) load_regs(q + 1);
        {
            const float* const ks = inb + buf * 4 * TB * ROW;
            const float* const ds = ks + TB * ROW;
            const float* const vs = ks + 2 * TB * ROW;
            const float* const gs = ks + 3 * TB * ROW;
            float* const qb = dqs + buf * TB * ROW;
            const int nb = min(TB, ntok - q * TB);

            for (int pp = 0; pp < nb; ++pp) {
                float kk[IR], dd[IR], v4[4], g4[4], dq[IR];
                lds_load<IR>(ks + pp * ROW + i0, kk);
                lds_load<IR>(ds + pp * ROW + i0, dd);
                lds_load<4>(vs + pp * ROW + j0, v4);
                lds_load<4>(gs + pp * ROW + j0, g4);
#pragma unroll
                for (int ii = 0; ii < IR; ++ii) {
                    dq[ii] = 0.f;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        dq[ii] = fmaf(g4[jj], S[ii][jj], dq[ii]);
                        S[ii][jj] = fmaf(S[ii][jj], dd[ii], kk[ii] * v4[jj]);
                    }
                }
                const float tot = row_reduce(dq, jl);
                if (jl < IR) qb[pp * ROW + i0 + row_sel<IR>(jl)] = tot;
            }
        }
        if (q + 1 < nq) write_lds(buf ^ 1);
        __syncthreads();
        {
            const int p = q * TB + spp;
            if (p < ntok) {
                [[maybe_unused]] const long idx_r = base + (long)tokmap(p, REV_R) * a.C + sc0, idx_k = base + (long)tokmap(p, REV_K) * a.C + sc0;
                [[maybe_unused]] const long idx_v = base + (long)tokmap(p, REV_V) * a.C + sc0, idx_w = base + (long)tokmap(p, REV_W) * a.C + sc0;
                [[maybe_unused]] const long idx_y = base + (long)tokmap(p, REV_Y) * a.C + sc0, idx_p = base + (long)p * a.C + sc0;   // idx_p: scratch indexed by scan position
                float dq[CPT], av[CPT], o[CPT];
                lds_load<CPT>(dqs + buf * TB * ROW + spp * ROW + sc0, dq);
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    av[c] = cr[c] * dq[c];
                    o[c] = fmaf(uu[c] * ck[c], cvg, dq[c]);
                }
                lds_store<CPT>(a.aux + idx_p, av);       // plain (global) vector store
                if (a.accumulate) {
                    float old[CPT];
                    ion<T, CPT>::load(ogr + idx_r, old);
#pragma unroll
                    for (int c = 0; c < CPT; ++c) o[c] += old[c];
                }
                ion<T, CPT>::store(ogr + idx_r, o);
            }
        }
    }
    if (a.gu) {   // per-batch gu partial: sum the staging threads' accumulators over the TB token slots
        __syncthreads();
        lds_store<CPT>(dqs + spp * ROW + sc0, gu_acc);
        __syncthreads();
        if (tid < HEAD) {
            float s = 0.f;
#pragma unroll
            for (int pp = 0; pp < TB; ++pp) s += dqs[pp * ROW + tid];
            const long o = (long)b * a.C + h * HEAD + tid;
            if (sizeof(T) == 4 || a.part_f32) reinterpret_cast<float*>(a.gu)[o] = s;
            else reinterpret_cast<bf16_t*>(a.gu)[o] = (bf16_t)(pack_bf2(s, 0.f) & 0xffffu);
        }
    }
}

// =====================================================================================================
// backward sweep G (reverse scan order):  G <- d_t (.) G + r_t gy_t^T  (cuda/wkv6_cuda.cu:128-132),
//   dk_t[i] = sum_j G[i][j] v_t[j],  gk_t = dk_t + u (.) r_t (v_t.gy_t)          (:126-134)
//   gv_t[j] = sum_i k_t[i] G[i][j] + (sum_i u[i]r_t[i]k_t[i]) gy_t[j]            (:149-157)
//   gw_t = ew_t (.) (sum_{s>t} a_s - sum_{s>=t} b_s),  b_t = k_t (.) dk_t        (replaces :161-227)
//   gs = dL/dS_0 = G after the last step                                          (wkv6state_cuda.cu:172-190)
// Same lane layout as sweep S.
// =====================================================================================================
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void scan_bwd_g_kernel(const ScanArgs a)
{
    using G_ = Geo<NW>;
    constexpr int CPT = G_::CPT, TPT = G_::TPT, IPW = HEAD / NW, IR = IPW / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const inb = smem;                                  // [2][5][TB][ROW]  r,k,d,v,gy
    float* const dks = inb + 2 * 5 * TB * ROW;                // [2][TB][ROW]
    float* const gvs = dks + 2 * TB * ROW;                    // [2][NW][TB][ROW]
    float* const dls = gvs + 2 * NW * TB * ROW;               // [2][TB][ROW]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const T* const gr_ = reinterpret_cast<const T*>(a.r);
    const T* const gk_ = reinterpret_cast<const T*>(a.k);
    const T* const gv_ = reinterpret_cast<const T*>(a.v);
    const T* const ggy = reinterpret_cast<const T*>(a.gy);
    T* const ogr = reinterpret_cast<T*>(a.gr);
    T* const ogk = reinterpret_cast<T*>(a.gk);
    T* const ogv = reinterpret_cast<T*>(a.gv);
    T* const ogw = reinterpret_cast<T*>(a.gw);
    int ntok = a.T;
    if (a.lens) ntok = min(max(a.lens[b], 0), a.T);
    const RevMap tokmap = make_revmap(a, b, ntok);      // token each tensor holds at scan position p (wkv6_scan.h)
    const long base = (long)b * a.T * a.C + (long)h * HEAD;

    const int spp = tid / TPT, sc0 = (tid % TPT) * CPT;
    float uu[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) uu[c] = 0.f;
    if (a.use_u) ion<T, CPT>::load(reinterpret_cast<const T*>(a.u) + h * HEAD + sc0, uu);

    const int irow = lane >> 4, jl = lane & 15;
    const int i0 = wv * IPW + irow * IR, j0 = jl * 4;
    float Gs[IR][4];
#pragma unroll
    for (int ii = 0; ii < IR; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) Gs[ii][jj] = 0.f;

    float pr[CPT], pk[CPT], pv[CPT], pew[CPT], pgy[CPT], pa[CPT];
    // n*: batch just written to LDS; c*: batch being scanned; d*: batch whose gw is still pending
    float nr[CPT], nk[CPT], ngy[CPT], new_[CPT], na[CPT], ncoef = 0.f, nvg = 0.f;
    float cr[CPT], ck[CPT], cgy[CPT], cew[CPT], ca[CPT], ccoef = 0.f, cvg = 0.f;
    float db[CPT], dew[CPT], R[CPT];
    long didx = 0;
    int dbuf = -1;
    bool dvalid = false;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        nr[c] = nk[c] = ngy[c] = new_[c] = na[c] = 0.f;
        cr[c] = ck[c] = cgy[c] = cew[c] = ca[c] = 0.f;
        db[c] = dew[c] = R[c] = 0.f;
    }

    auto load_regs = [&](int q) {
        const int p = q * TB + spp;
        const bool valid = p < ntok;
        [[maybe_unused]] const long idx_r = base + (long)tokmap(p, REV_R) * a.C + sc0, idx_k = base + (long)tokmap(p, REV_K) * a.C + sc0;
        [[maybe_unused]] const long idx_v = base + (long)tokmap(p, REV_V) * a.C + sc0, idx_w = base + (long)tokmap(p, REV_W) * a.C + sc0;
        [[maybe_unused]] const long idx_y = base + (long)tokmap(p, REV_Y) * a.C + sc0, idx_p = base + (long)p * a.C + sc0;   // idx_p: scratch indexed by scan position
#pragma unroll
        for (int c = 0; c < CPT; ++c) { pr[c] = 0.f; pk[c] = 0.f; pv[c] = 0.f; pgy[c] = 0.f; pa[c] = 0.f; }
        if (valid) {
            ion<T, CPT>::load(gr_ + idx_r, pr);
            ion<T, CPT>::load(gk_ + idx_k, pk);
            ion<T, CPT>::load(gv_ + idx_v, pv);
            ion<T, CPT>::load(ggy + idx_y, pgy);
            lds_load<CPT>(a.aux + idx_p, pa);
        }
        load_ew<T, CPT>(a, idx_w, valid, pew);
    };
    auto write_lds = [&](int buf) {
        float* const ib_ = inb + buf * 5 * TB * ROW + spp * ROW + sc0;
        float d[CPT];
        float p1 = 0.f, p2 = 0.f;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            d[c] = __expf(pew[c]);
            p1 = fmaf(pr[c] * uu[c], pk[c], p1);
            p2 = fmaf(pv[c], pgy[c], p2);
        }
        lds_store<CPT>(ib_, pr);
        lds_store<CPT>(ib_ + TB * ROW, pk);
        lds_store<CPT>(ib_ + 2 * TB * ROW, d);
        lds_store<CPT>(ib_ + 3 * TB * ROW, pv);
        lds_store<CPT>(ib_ + 4 * TB * ROW, pgy);
        ncoef = token_sum<TPT>(p1);
        nvg = token_sum<TPT>(p2);
#pragma unroll
        for (int c = 0; c < CPT; ++c) { nr[c] = pr[c]; nk[c] = pk[c]; ngy[c] = pgy[c]; new_[c] = pew[c]; na[c] = pa[c]; }
    };
    // gw of the pending batch: needs every slot's delta of that batch (written before the last barrier)
    auto finish_gw = [&]() {
        if (dbuf < 0) return;
        const float* const dl = dls + dbuf * TB * ROW + sc0;
        float after[CPT], total[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) { after[c] = 0.f; total[c] = 0.f; }
#pragma unroll
        for (int pp = TB - 1; pp >= 0; --pp) {       // later scan positions first (fixed order)
            float t[CPT];
            lds_load<CPT>(dl + pp * ROW, t);
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                if (pp == spp) after[c] = total[c];
                total[c] += t[c];
            }
        }
        if (dvalid) {
            float o[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) o[c] = (R[c] + after[c] - db[c]) * dew[c];
            if (a.accumulate) {
                float old[CPT];
                ion<T, CPT>::load(ogw + didx, old);
#pragma unroll
                for (int c = 0; c < CPT; ++c) o[c] += old[c];
            }
            ion<T, CPT>::store(ogw + didx, o);
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) R[c] += total[c];
        dbuf = -1;
    };

    const int nq = (ntok + TB - 1) / TB;
    if (nq > 0) {
        load_regs(nq - 1);
        write_lds((nq - 1) & 1);
    }
    __syncthreads();
    for (int q = nq - 1; q >= 0; --q) {
        const int buf = q & 1;
#pragma unroll
        for (int c = 0; c < CPT; ++c) { cr[c] = nr[c]; ck[c] = nk[c]; cgy[c] = ngy[c]; cew[c] = new_[c]; ca[c] = na[c]; }
        ccoef = ncoef; cvg = nvg;
        if (q > 0) load_regs(q - 1);
        {
            const float* const rs = inb + buf * 5 * TB * ROW;
            const float* const ks = rs + TB * ROW;
            const float* const ds = rs + 2 * TB * ROW;
            const float* const vs = rs + 3 * TB * ROW;
            const float* const gs = rs + 4 * TB * ROW;
            float* const kb = dks + buf * TB * ROW;
            float* const vb = gvs + (buf * NW + wv) * TB * ROW;
            const int nb = min(TB, ntok - q * TB);
            for (int pp = nb - 1; pp >= 0; --pp) {
                float rr[IR], kk[IR], dd[IR], v4[4], g4[4], gkp[IR], gvp[4];
                lds_load<IR>(rs + pp * ROW + i0, rr);
                lds_load<IR>(ks + pp * ROW + i0, kk);
                lds_load<IR>(ds + pp * ROW + i0, dd);
                lds_load<4>(vs + pp * ROW + j0, v4);
                lds_load<4>(gs + pp * ROW + j0, g4);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) gvp[jj] = 0.f;
#pragma unroll
                for (int ii = 0; ii < IR; ++ii) {
                    gkp[ii] = 0.f;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        gkp[ii] = fmaf(v4[jj], Gs[ii][jj], gkp[ii]);
                        gvp[jj] = fmaf(kk[ii], Gs[ii][jj], gvp[jj]);
                        Gs[ii][jj] = fmaf(Gs[ii][jj], dd[ii], rr[ii] * g4[jj]);
                    }
                }
                const float dk = row_reduce(gkp, jl);
                if (jl < IR) kb[pp * ROW + i0 + row_sel<IR>(jl)] = dk;
                const float gvt = col_reduce(gvp);
                vb[pp * ROW + j0 + col_sel(irow)] = gvt;
            }
        }
        if (q > 0) write_lds(buf ^ 1);
        __syncthreads();
        finish_gw();                                   // batch q+1 (its deltas were complete one barrier ago)
        {
            const int p = q * TB + spp;
            const bool valid = p < ntok;
            [[maybe_unused]] const long idx_r = base + (long)tokmap(p, REV_R) * a.C + sc0, idx_k = base + (long)tokmap(p, REV_K) * a.C + sc0;
            [[maybe_unused]] const long idx_v = base + (long)tokmap(p, REV_V) * a.C + sc0, idx_w = base + (long)tokmap(p, REV_W) * a.C + sc0;
            [[maybe_unused]] const long idx_y = base + (long)tokmap(p, REV_Y) * a.C + sc0, idx_p = base + (long)p * a.C + sc0;   // idx_p: scratch indexed by scan position
            float dk[CPT], gvsum[CPT], dl[CPT];
            lds_load<CPT>(dks + buf * TB * ROW + spp * ROW + sc0, dk);
#pragma unroll
            for (int c = 0; c < CPT; ++c) gvsum[c] = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < NW; ++w_) {
                float t2[CPT];
                lds_load<CPT>(gvs + (buf * NW + w_) * TB * ROW + spp * ROW + sc0, t2);
#pragma unroll
                for (int c = 0; c < CPT; ++c) gvsum[c] += t2[c];
            }
            float ogk_[CPT], ogv_[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                if (!valid) { dk[c] = 0.f; }
                ogk_[c] = fmaf(uu[c] * cr[c], cvg, dk[c]);
                ogv_[c] = fmaf(ccoef, cgy[c], gvsum[c]);
                db[c] = ck[c] * dk[c];
                dl[c] = ca[c] - db[c];
                dew[c] = cew[c];
            }
            lds_store<CPT>(dls + buf * TB * ROW + spp * ROW + sc0, dl);
            dbuf = buf; dvalid = valid; didx = idx_w;
            if (valid) {
                if (a.accumulate) {
                    float o1[CPT], o2[CPT];
                    ion<T, CPT>::load(ogk + idx_k, o1);
                    ion<T, CPT>::load(ogv + idx_v, o2);
#pragma unroll
                    for (int c = 0; c < CPT; ++c) { ogk_[c] += o1[c]; ogv_[c] += o2[c]; }
                }
                ion<T, CPT>::store(ogk + idx_k, ogk_);
                ion<T, CPT>::store(ogv + idx_v, ogv_);
            }
        }
    }
    __syncthreads();
    finish_gw();                                       // batch 0

    if (a.gs) {
        const long so_ = ((long)b * a.H + h) * HEAD * HEAD;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            float t2[IR];
#pragma unroll
            for (int ii = 0; ii < IR; ++ii) t2[ii] = Gs[ii][jj];
            if (sizeof(T) == 4 || a.part_f32) ion<float, IR>::store(reinterpret_cast<float*>(a.gs) + so_ + (long)(j0 + jj) * HEAD + i0, t2);
            else ion<T, IR>::store(reinterpret_cast<T*>(a.gs) + so_ + (long)(j0 + jj) * HEAD + i0, t2);
        }
    }
    if (a.zero_tail && !a.accumulate) {
        const float z[CPT] = {};
        for (int t = ntok + spp; t < a.T; t += TB) {
            const long idx = base + (long)t * a.C + sc0;
            ion<T, CPT>::store(ogr + idx, z);
            ion<T, CPT>::store(ogk + idx, z);
            ion<T, CPT>::store(ogv + idx, z);
            ion<T, CPT>::store(ogw + idx, z);
        }
    }
}

// ---- device self-test of the cross-lane primitives (tests call it once) ---------------------------
__global__ void selftest_kernel(int* result)
{
    const int lane = threadIdx.x & 63;
    int bad = 0;
    // pseudo-random but exactly representable values
    float x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = (float)((lane * 37 + q * 11) % 101) - 50.f;
    {   // col_reduce against shuffles
        const float got = col_reduce(x);
        const float ref = col_reduce_ref(x, lane >> 4);
        if (got != ref) bad |= 1;
    }
    {   // row_reduce<4>
        const float got = row_reduce(x, lane & 15);
        float t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float s = x[q];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
            t[q] = s;
        }
        if (got != pick<4>(t, row_sel<4>(lane & 15))) bad |= 2;
    }
    {   // row_reduce<2>
        const float y2[2] = {x[0], x[3]};
        const float got = row_reduce(y2, lane & 15);
        float t[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float s = y2[q];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
            t[q] = s;
        }
        if (got != pick<2>(t, row_sel<2>(lane & 15))) bad |= 4;
    }
    {   // row_sum16
        float s = x[1];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        if (row_sum16(x[1]) != s) bad |= 8;
    }
    if (bad) atomicOr(result, bad);
}

template <typename K> hipError_t set_lds(K kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

constexpr int NWAVES = 8;

}  // namespace

hipError_t launch_scan_fwd(const ScanArgs& a, int io, hipStream_t st)
{
    constexpr size_t lds = (2 * 4 * TB * ROW + 64 + 2 * TB * ROW) * sizeof(float);
    const dim3 grid(a.B * a.H), block(NWAVES * 64);
    if (io == IO_F32) hipLaunchKernelGGL((scan_fwd_kernel<float, NWAVES>), grid, block, lds, st, a);
    else if (io == IO_F16) hipLaunchKernelGGL((scan_fwd_kernel<f16_t, NWAVES>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((scan_fwd_kernel<bf16_t, NWAVES>), grid, block, lds, st, a);
    return hipGetLastError();
}

hipError_t launch_scan_bwd(const ScanArgs& a, bool io_f32, hipStream_t st)
{
    constexpr size_t lds_s = (2 * 4 * TB * ROW + 2 * TB * ROW) * sizeof(float);
    constexpr size_t lds_g = (2 * 5 * TB * ROW + 2 * TB * ROW + 2 * NWAVES * TB * ROW + 2 * TB * ROW) * sizeof(float);
    const dim3 grid(a.B * a.H), block(NWAVES * 64);
    hipError_t e;
    if (io_f32) {
        if ((e = set_lds(scan_bwd_g_kernel<float, NWAVES>, lds_g)) != hipSuccess) return e;
        hipLaunchKernelGGL((scan_bwd_s_kernel<float, NWAVES>), grid, block, lds_s, st, a);
        hipLaunchKernelGGL((scan_bwd_g_kernel<float, NWAVES>), grid, block, lds_g, st, a);
    } else {
        if ((e = set_lds(scan_bwd_g_kernel<bf16_t, NWAVES>, lds_g)) != hipSuccess) return e;
        hipLaunchKernelGGL((scan_bwd_s_kernel<bf16_t, NWAVES>), grid, block, lds_s, st, a);
        hipLaunchKernelGGL((scan_bwd_g_kernel<bf16_t, NWAVES>), grid, block, lds_g, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_selftest(int* result, hipStream_t st)
{
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, result);
    return hipGetLastError();
}

}  // namespace wkv6
