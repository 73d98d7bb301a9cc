// Chunked MFMA backward of WKV6 for gfx950 (bf16 I/O), two-level form: 64-token chunks, 16-token blocks, 16 identical waves.
//
// Why a second backward (the 12-wave staged kernel is wkv6_chunk_bwd12.hip): that kernel touches, scales, splits and updates
// three 64x64 state copies every 16 tokens, needs a forward-state checkpoint every 32 tokens (8 B of HBM per token-channel) and
// gives each of its three roles ONE wave per SIMD, so every role runs at the latency of a single instruction stream
// (profiles/r03_clock_and_roles.txt: each role alone takes 0.19-0.27 ms of the 0.47 ms).  Here
//   * the states S (forward, at chunk entry: the checkpoint) and G (adjoint, at chunk exit) are touched once per 64 tokens:
//     all interactions inside a chunk go through the ten lower-triangular 16x16 token tiles (the two-level chunking of the
//     reference's alternative backend, fla/ops/rwkv6/chunk.py:174-309), so checkpoints are 64 tokens apart (4 B per
//     token-channel), there is no state rebuild and G exists once;
//   * all 16 waves run the same program (four per SIMD): every phase has four instruction streams per SIMD to interleave.
//
// Block algebra.  p = position in the chunk, I = p >> 4 its block, i key channel, j value channel; lw2_p <= 0 the log2-decay of
// token p (clamped at LW_MIN per token as in the other chunked kernels):
//     C_p  = sum_{q<p} lw2_q  (exclusive; 0 at the chunk start),  P4 = C_64,
//     N_I  = rint(C at token 16 I + 8)   -- an INTEGER reference frame per block and channel,
//     fR_p = 2^{C_p - N_I},  fK_p = 2^{N_I - C_{p+1}},  Rhat = r fR,  Khat = k fK     (exponents span <= 8 tokens + 0.5),
// so for a pair a in block I, b in block J, b < a:  2^{C_a - C_{b+1}} = fR_a 2^{N_I - N_J} fK_b, and the ratio between two
// frames is an exact power of two: a bf16 hi/lo operand pair moves from one frame to another by an integer subtraction on its
// exponent fields (v_pk_sub_u16 clamp: underflow saturates to +0), no re-split, no rounding.  With
//     dA[a][b] = gy_a . v_b (b < a),   vg_a = gy_a . v_a,   A[a][b] = sum_i Rhat_a[i] 2^{N_I-N_J}[i] Khat_b[i],  A[a][a] = sum_i r u k,
//     S = state at chunk entry,  Gop = 2^{P4 - rint(P4)} (.) G  (G = dL/d state at chunk exit),  NE = rint(P4):
//   dq_a[i] = fR_a[i] ( sum_{J<=I} 2^{N_I-N_J}[i] sum_{b in J} dA[a][b] Khat_b[i]  +  2^{N_I}[i] sum_j S[i][j] gy_a[j] )
//   dk_b[i] = fK_b[i] ( sum_{I>=J} 2^{N_I-N_J}[i] sum_{a in I} dA[a][b] Rhat_a[i]  +  2^{NE-N_J}[i] sum_j Gop[i][j] v_b[j] )
//   gv_b[j] = sum_{a>=b} A[a][b] gy_a[j]  +  sum_i (Khat_b[i] 2^{NE-N_J}[i]) Gop[i][j]
//   G_entry = 2^{P4} (.) G + sum_I 2^{N_I} (.) (Rhat_I^T gy_I)
//   gr = dq + vg u k,  gk = dk + vg u r,  gu += vg r k,  gw_t = lw_t ( sum_{s>t} (r_s dq_s - k_s dk_s) - k_t dk_t )
// (the adjoint of cuda/wkv6_cuda.cu:44-57, reference backward cuda/wkv6_cuda.cu:63-227; tools/emulate_bwd64.py evaluates
// exactly these formulas in fp64 against the oracle).  Frame factors that multiply an OUTPUT channel (dq, dk, G) are applied to
// the MFMA results with v_ldexp_f32; only the two products that contract over i (A, and the G term of gv) shift operands.
//
// One 1024-thread workgroup per (batch, head); wave w = (I = w >> 2, q = w & 3); lane (x = lane & 15, g = lane >> 4) owns
// token 16 I + x and the four channels 16 q + 4 g .. +3 -- the C-layout of a 16x16 MFMA tile [channel rows][token columns] --
// for the preparation AND for the gradient epilogue, so r, k, fR, fK, lw stay in registers from one to the other.
// Per chunk, three phases separated by workgroup barriers:
//   A  operands of the chunk -> LDS image (Rhat, Khat hi/lo, v, gy); checkpoint -> S operand image; G -> Gop operand image;
//      gw of the previous chunk (needs the other blocks' totals);
//   B  the ten score tiles and dA tiles (both orientations), once for the workgroup (waves 0..9), as MFMA fragments in LDS;
//   C  per wave: its 16x16 tiles of dq, dk, gv and of the G update, the gradient epilogue, and the decay scan of the next chunk.
#include "wkv6_chunk.h"

namespace wkv6 {

namespace {

using namespace chunk;

constexpr int CHK = 64;                                // tokens per chunk
constexpr int ARR64 = CHK * RSB;                       // one operand array: 64 rows of RSB bytes
enum { C_RH = 0, C_RL, C_KH, C_KL, C_V, C_GY, NC_ARR };
constexpr int L_IMG = 0;                               // bf16 [NC_ARR][64 tokens][RSB/2]
constexpr int L_SOP = L_IMG + NC_ARR * ARR64;          // S operand, transposed: bf16 hi | lo, [64 j][RSB/2] (columns i)
constexpr int L_GOP = L_SOP + 2 * ARR64;               // Gop operand: bf16 hi | lo, [64 i][RSB/2] (columns j)
constexpr int L_SCF = L_GOP + 2 * ARR64;               // score tiles: 10 x [64 lanes] uint4 (bf16x4 hi | bf16x4 lo)
constexpr int L_DAF = L_SCF + 10 * 1024;               // dA tiles: 10 x 2 orientations x [64 lanes] uint4
constexpr int L_CKQ = L_DAF + 20 * 1024;               // checkpoint landing zone (LDS-DMA): 16 pieces of 1 KB, one per wave
constexpr int L_TOT = L_CKQ + 16384;                   // float [4][64]  per-block sums of lw2 (next chunk, written in phase C)
constexpr int L_NI = L_TOT + 1024;                     // float [4][64]  N_I
constexpr int L_PT = L_NI + 1024;                      // float [64]     P4
constexpr int L_NE = L_PT + 256;                       // float [64]     NE = rint(P4)
constexpr int L_VGQ = L_NE + 256;                      // float [4 q][64 tokens]  per-quarter gy.v
constexpr int L_CFQ = L_VGQ + 1024;                    // float [4 q][64 tokens]  per-quarter sum r u k
constexpr int L_TDL = L_CFQ + 1024;                    // float [4][64]  per-block sums of r dq - k dk (gw suffix across blocks)
constexpr int L_GU = L_TDL + 1024;                     // float [4][64]  per-block gu partials (end of the kernel)
constexpr int BWD64_LDS = L_GU + 1024;
static_assert(BWD64_LDS <= 160 * 1024, "LDS budget");

constexpr int DPP_BCAST8 = 0x158;                      // row_newbcast:8

// four independent in-row scans, one step (see wkv6_chunk_bwd12.hip for the wait-state reasoning)
#define WKV6_DPP_STEP4(x, ctrl) asm("s_nop 1\n\t" \
    "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]))

// ---- frame changes of bf16 operands: multiply two packed bf16 by 2^-s0, 2^-s1 (s >= 0) ------------------------------------
__device__ __forceinline__ unsigned pk_sub_sat(unsigned a_, unsigned b_)
{
    unsigned r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a_), "v"(b_));
    return r;
}
__device__ __forceinline__ unsigned shift_pair(unsigned v, unsigned sh)
{   // magnitudes (15 bits each) minus s << 7, saturating at 0; signs kept
    return pk_sub_sat(v & 0x7fff7fffu, sh) | (v & 0x80008000u);
}
__device__ __forceinline__ unsigned pack_shift(float n0, float n1)
{   // n0, n1: integer-valued, >= 0 (a negative value converts to 0); 255 already empties any exponent field
    const unsigned s0 = min((unsigned)n0, 255u), s1 = min((unsigned)n1, 255u);
    return (s0 << 7) | (s1 << 23);
}
__device__ __forceinline__ b8v shift_frag(b8v f, const unsigned (&sh)[4])
{
    uint4 v = __builtin_bit_cast(uint4, f);
    v.x = shift_pair(v.x, sh[0]); v.y = shift_pair(v.y, sh[1]); v.z = shift_pair(v.z, sh[2]); v.w = shift_pair(v.w, sh[3]);
    return __builtin_bit_cast(b8v, v);
}
__device__ __forceinline__ constexpr int tile_id(int I, int J) { return I * (I + 1) / 2 + J; }   // J <= I
__device__ __forceinline__ float4 ldf4(const char* p) { return *reinterpret_cast<const float4*>(p); }

template <bool W_RAW, bool GEN>
__global__ __launch_bounds__(1024) void chunk_bwd64_kernel(const ScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, g = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int I = wid >> 2, q = wid & 3;                       // token block / channel quarter of this wave
    const int bh = blockIdx.x;
    const int b = a.order ? a.order[bh / a.H] : bh / a.H, h = bh % a.H;
    const long base = (long)b * a.T * a.C + (long)h * HEAD;
    const bf16_t* const gr_ = reinterpret_cast<const bf16_t*>(a.r) + base;
    const bf16_t* const gk_ = reinterpret_cast<const bf16_t*>(a.k) + base;
    const bf16_t* const gv_ = reinterpret_cast<const bf16_t*>(a.v) + base;
    const bf16_t* const ggy = reinterpret_cast<const bf16_t*>(a.gy) + base;
    bf16_t* const ogr = reinterpret_cast<bf16_t*>(a.gr) + base;
    bf16_t* const ogk = reinterpret_cast<bf16_t*>(a.gk) + base;
    bf16_t* const ogv = reinterpret_cast<bf16_t*>(a.gv) + base;
    bf16_t* const ogw = reinterpret_cast<bf16_t*>(a.gw) + base;
    int ntok = a.T;
    if (a.lens) ntok = min(max(a.lens[b], 0), a.T);
    const RevMap tokmap = make_revmap(a, b, ntok);
    const unsigned nbytes = ntok > 0 ? (unsigned)(ntok - 1) * a.C * 2u + 128u : 0u;
    const rsrc_t rs_gr = make_rsrc(ogr, nbytes), rs_gk = make_rsrc(ogk, nbytes), rs_gv = make_rsrc(ogv, nbytes), rs_gw = make_rsrc(ogw, nbytes);
    // gradient store of scan position pos, channels ch..ch+3 (same contract as wkv6_chunk_bwd12.hip: emit)
    auto emit = [&](int which, const rsrc_t& rs, bf16_t* out, int pos, unsigned bit, int ch, float (&o)[4]) {
        const unsigned idx = (unsigned)(tokmap(pos, bit) * a.C + ch);
        if constexpr (!GEN) {
            buf_store8(rs, idx * 2u, make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])));
        } else {
            if (pos >= ntok) return;
            float* const side = a.g_f32[which];
            if (side && !a.accumulate) {
                io4<float>::store(side + base + idx, o);
                return;
            }
            if (a.accumulate) {
                float old[4];
                if (side) io4<float>::load(side + base + idx, old);
                else io4<bf16_t>::load(out + idx, old);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += old[e];
            }
            io4<bf16_t>::store(out + idx, o);
        }
    };
#ifdef WKV6_STAMP
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, ts6 = 0;
#endif
#ifdef WKV6_DEBUGBUF
    unsigned long long clk0 = 0, rtc0 = 0, clk1 = 0, rtc1 = 0;
    WKV6_CLK(clk0, rtc0);
#endif

    const int p = 16 * I + x;                                  // this lane's token within a chunk
    const int ch0 = 16 * q + 4 * g;                            // ... and its four channels
    const int nC = (ntok + CHK - 1) / CHK, nCmax = (a.T + CHK - 1) / CHK;
    float uu[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.use_u) io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.u) + h * HEAD + ch0, uu);

    const rsrc_t rs_r = make_rsrc(gr_, nbytes), rs_k = make_rsrc(gk_, nbytes), rs_v = make_rsrc(gv_, nbytes), rs_g = make_rsrc(ggy, nbytes);
    const rsrc_t rs_w = W_RAW ? make_rsrc(reinterpret_cast<const bf16_t*>(a.w) + base, nbytes)
                              : make_rsrc(reinterpret_cast<const float*>(a.w) + base, ntok > 0 ? (unsigned)(ntok - 1) * a.C * 4u + 256u : 0u);
    uint2 pr, pk, pv, pg, pw = make_uint2(0u, 0u);
    float4 pe = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_chunk = [&](int c) {                             // tokens past the end load zeros
        const int pos = c * CHK + p;
        pr = buf_load8(rs_r, (unsigned)(tokmap(pos, REV_R) * a.C + ch0) * 2u);
        pk = buf_load8(rs_k, (unsigned)(tokmap(pos, REV_K) * a.C + ch0) * 2u);
        pv = buf_load8(rs_v, (unsigned)(tokmap(pos, REV_V) * a.C + ch0) * 2u);
        pg = buf_load8(rs_g, (unsigned)(tokmap(pos, REV_Y) * a.C + ch0) * 2u);
        const unsigned iw = (unsigned)(tokmap(pos, REV_W) * a.C + ch0);
        if constexpr (W_RAW) pw = buf_load8(rs_w, iw * 2u);
        else pe = buf_load16f(rs_w, iw * 4u);
    };
    // decay scan of chunk c (from pw / pe): loc = exclusive in-block prefix of lw2, lw2v, lwe; block totals -> L_TOT
    float loc[4], lw2v[4], lwe[4];
    auto decay_scan = [&](int c) {
        const bool valid = c * CHK + p < ntok;
        float lw[4];
        if constexpr (W_RAW) {
            lw[0] = -exp2_fast(LOG2E * bf_lo(pw.x)); lw[1] = -exp2_fast(LOG2E * bf_hi(pw.x));
            lw[2] = -exp2_fast(LOG2E * bf_lo(pw.y)); lw[3] = -exp2_fast(LOG2E * bf_hi(pw.y));
        } else {
            lw[0] = pe.x; lw[1] = pe.y; lw[2] = pe.z; lw[3] = pe.w;
        }
        float inc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lw2v[e] = valid ? fmaxf(lw[e] * LOG2E, LW_MIN2) : 0.f;
            lwe[e] = valid ? lw[e] : 0.f;      // gw multiplier: the true lw (times d_true / d_clamped where the clamp is active)
            inc[e] = lw2v[e];
        }
        if (__builtin_amdgcn_ballot_w64(lwe[0] < LW_MIN || lwe[1] < LW_MIN || lwe[2] < LW_MIN || lwe[3] < LW_MIN)) {   // rare
#pragma unroll
            for (int e = 0; e < 4; ++e) lwe[e] *= exp2_fast(LOG2E * fminf(lwe[e] - LW_MIN, 0.f));
        }
        WKV6_DPP_STEP4(inc, "row_shr:1");                      // inclusive prefix over the 16 tokens of the DPP row
        WKV6_DPP_STEP4(inc, "row_shr:2");
        WKV6_DPP_STEP4(inc, "row_shr:4");
        WKV6_DPP_STEP4(inc, "row_shr:8");
#pragma unroll
        for (int e = 0; e < 4; ++e) loc[e] = inc[e] - lw2v[e];
        if (x == 15) *reinterpret_cast<float4*>(smem + L_TOT + (I * 64 + ch0) * 4) = make_float4(inc[0], inc[1], inc[2], inc[3]);
    };
    // checkpoint of chunk c -> landing zone, 1 KB per wave (piece w of the forward's register dump: forward wave w >> 2, tile w & 3)
    const unsigned ckq_lds = __builtin_amdgcn_readfirstlane(
        (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)(smem + L_CKQ + wid * 1024)));
    auto request_ckpt = [&](int c) {
        const float* const src = a.ckpt + ((long)(b * a.H + h) * nCmax + c) * (HEAD * HEAD) + wid * 256 + lane * 4;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(ckq_lds) : "memory");
    };

    const int troff = (4 * g + (x >> 2)) * RSB + 8 * (x & 3);  // transposed read of a 16-row block: + first row * RSB + 32 * column group
    char* const img = smem + L_IMG;
    f4v Gt = {0.f, 0.f, 0.f, 0.f};                             // G[i = 16 I + x][j = 16 q + 4 g + e]
    float Rc[4] = {0.f, 0.f, 0.f, 0.f}, gu_acc[4] = {0.f, 0.f, 0.f, 0.f};
    float gw_loc[4] = {0.f, 0.f, 0.f, 0.f}, gw_lwe[4] = {0.f, 0.f, 0.f, 0.f};   // gw of the previous chunk, finished one phase later
    // finish gw of chunk cp: the blocks behind this one in the chunk, and everything behind the chunk (Rc)
    auto finish_gw = [&](int cp) {
        float later[4] = {0.f, 0.f, 0.f, 0.f}, all[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int J = 0; J < 4; ++J) {
            const float4 t = ldf4(smem + L_TDL + (J * 64 + ch0) * 4);
            all[0] += t.x; all[1] += t.y; all[2] += t.z; all[3] += t.w;
            if (J > I) { later[0] += t.x; later[1] += t.y; later[2] += t.z; later[3] += t.w; }
        }
        float o_gw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o_gw[e] = (Rc[e] + later[e] + gw_loc[e]) * gw_lwe[e];
            Rc[e] += all[e];
        }
        emit(3, rs_gw, ogw, cp * CHK + p, REV_W, ch0, o_gw);
    };

    if (nC > 0) {
        load_chunk(nC - 1);
        request_ckpt(nC - 1);
        decay_scan(nC - 1);
    }
    __syncthreads();
    for (int c = nC - 1; c >= 0; --c) {
        // =============================== phase A: operands of chunk c ===============================================
        float rv[4], kv[4], fR[4], fK[4], lwc[4];
        WKV6_T(ts0);
        {
            // everything this wave issued a phase ago has landed: the checkpoint DMA (invisible to the compiler's counters) and
            // the previous chunk's gradient stores
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            float t4[4][4];
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                const float4 t = ldf4(smem + L_TOT + (J * 64 + ch0) * 4);
                t4[J][0] = t.x; t4[J][1] = t.y; t4[J][2] = t.z; t4[J][3] = t.w;
            }
            float nI[4], p4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pI = (I > 0 ? t4[0][e] : 0.f) + (I > 1 ? t4[1][e] : 0.f) + (I > 2 ? t4[2][e] : 0.f);
                p4[e] = (t4[0][e] + t4[1][e]) + (t4[2][e] + t4[3][e]);
                const float m = dpp_mov<DPP_BCAST8>(loc[e]);               // in-block prefix at token 8
                const float cmid = pI + m;
                nI[e] = __builtin_rintf(cmid);
                const float phi = cmid - nI[e];                            // |phi| <= 0.5: the frame sits within half a bit of token 8
                fR[e] = exp2_fast((loc[e] - m) + phi);
                fK[e] = exp2_fast((m - (loc[e] + lw2v[e])) - phi);
                lwc[e] = lwe[e];
            }
            if (x == 0) *reinterpret_cast<float4*>(smem + L_NI + (I * 64 + ch0) * 4) = make_float4(nI[0], nI[1], nI[2], nI[3]);
            if (I == 0 && x == 0) {
                *reinterpret_cast<float4*>(smem + L_PT + ch0 * 4) = make_float4(p4[0], p4[1], p4[2], p4[3]);
                *reinterpret_cast<float4*>(smem + L_NE + ch0 * 4) =
                    make_float4(__builtin_rintf(p4[0]), __builtin_rintf(p4[1]), __builtin_rintf(p4[2]), __builtin_rintf(p4[3]));
            }
            rv[0] = bf_lo(pr.x); rv[1] = bf_hi(pr.x); rv[2] = bf_lo(pr.y); rv[3] = bf_hi(pr.y);
            kv[0] = bf_lo(pk.x); kv[1] = bf_hi(pk.x); kv[2] = bf_lo(pk.y); kv[3] = bf_hi(pk.y);
            float rh[4], kh[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { rh[e] = rv[e] * fR[e]; kh[e] = kv[e] * fK[e]; }
            char* const row = img + p * RSB + ch0 * 2;
            uint2 hi, lo;
            split4(rh, hi, lo);
            *reinterpret_cast<uint2*>(row + C_RH * ARR64) = hi; *reinterpret_cast<uint2*>(row + C_RL * ARR64) = lo;
            split4(kh, hi, lo);
            *reinterpret_cast<uint2*>(row + C_KH * ARR64) = hi; *reinterpret_cast<uint2*>(row + C_KL * ARR64) = lo;
            *reinterpret_cast<uint2*>(row + C_V * ARR64) = pv;
            *reinterpret_cast<uint2*>(row + C_GY * ARR64) = pg;
            // per-token sums over this wave's 16 channels: sum r u k (diagonal of A) and gy.v (diagonal of dA)
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) part = fmaf(rv[e] * uu[e], kv[e], part);
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            float pvg = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pg.x), __builtin_bit_cast(bf2, pv.x), 0.f, false);
            pvg = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pg.y), __builtin_bit_cast(bf2, pv.y), pvg, false);
            const float two[4] = {part, pvg, 0.f, 0.f};
            const float red = col_reduce(two);                 // row 0: total of value 0, row 2: total of value 1 (col_sel)
            if (g == 0) *reinterpret_cast<float*>(smem + L_CFQ + (q * 64 + p) * 4) = red;
            if (g == 2) *reinterpret_cast<float*>(smem + L_VGQ + (q * 64 + p) * 4) = red;
            // checkpoint piece `wid` -> S operand image, transposed [j][i]: this lane holds S[i0 .. i0+3][j]
            {
                const float4 s4 = ldf4(smem + L_CKQ + wid * 1024 + lane * 16);
                const int j = 16 * (wid >> 2) + x, i0 = tile_ch(wid & 3) + 8 * g;
                const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
                split4(sv, hi, lo);
                *reinterpret_cast<uint2*>(smem + L_SOP + j * RSB + i0 * 2) = hi;
                *reinterpret_cast<uint2*>(smem + L_SOP + ARR64 + j * RSB + i0 * 2) = lo;
            }
            // Gop = 2^{P4 - rint(P4)} (.) G -> operand image [i][j]
            {
                const int i = 16 * I + x;
                const float pt = (*reinterpret_cast<const float*>(smem + L_TOT + i * 4) + *reinterpret_cast<const float*>(smem + L_TOT + (64 + i) * 4))
                               + (*reinterpret_cast<const float*>(smem + L_TOT + (128 + i) * 4) + *reinterpret_cast<const float*>(smem + L_TOT + (192 + i) * 4));
                const float sc = exp2_fast(pt - __builtin_rintf(pt));
                const float gs4[4] = {Gt[0] * sc, Gt[1] * sc, Gt[2] * sc, Gt[3] * sc};
                split4(gs4, hi, lo);
                *reinterpret_cast<uint2*>(smem + L_GOP + i * RSB + ch0 * 2) = hi;
                *reinterpret_cast<uint2*>(smem + L_GOP + ARR64 + i * RSB + ch0 * 2) = lo;
            }
            if (c + 1 < nC) finish_gw(c + 1);
            // the landing zone has been read back (the split consumed it): request the next checkpoint, then the next chunk's inputs
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (c > 0) {
                request_ckpt(c - 1);
                load_chunk(c - 1);
            }
        }
        WKV6_T(ts1);
        __syncthreads();
        WKV6_T(ts2);
        // =============================== phase B: score and dA tiles, once for the workgroup ==========================
        if (wid < 10) {
            const int tI = (wid >= 1) + (wid >= 3) + (wid >= 6), tJ = wid - tI * (tI + 1) / 2;
            const bool diag = tI == tJ;
            const char* const rowI = img + (16 * tI + x) * RSB + 16 * g;      // + array, + 64 s
            const char* const rowJ = img + (16 * tJ + x) * RSB + 16 * g;
            f4v sc = {0.f, 0.f, 0.f, 0.f}, dab = {0.f, 0.f, 0.f, 0.f}, dba = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                b8v rh = ld_b8(rowI + C_RH * ARR64 + 64 * s), rl = ld_b8(rowI + C_RL * ARR64 + 64 * s);
                const b8v kh = ld_b8(rowJ + C_KH * ARR64 + 64 * s), kl = ld_b8(rowJ + C_KL * ARR64 + 64 * s);
                if (!diag) {   // Rhat of block I into the frame of block J: channels 32 s + 8 g .. +7, shift N_J - N_I >= 0
                    const char* const nI_ = smem + L_NI + (tI * 64 + 32 * s + 8 * g) * 4;
                    const char* const nJ_ = smem + L_NI + (tJ * 64 + 32 * s + 8 * g) * 4;
                    const float4 i0 = ldf4(nI_), i1 = ldf4(nI_ + 16), j0 = ldf4(nJ_), j1 = ldf4(nJ_ + 16);
                    const unsigned sh[4] = {pack_shift(j0.x - i0.x, j0.y - i0.y), pack_shift(j0.z - i0.z, j0.w - i0.w),
                                            pack_shift(j1.x - i1.x, j1.y - i1.y), pack_shift(j1.z - i1.z, j1.w - i1.w)};
                    rh = shift_frag(rh, sh);
                    rl = shift_frag(rl, sh);
                }
                sc = mfma32(rh, kh, sc);                       // A[row a][col b]: lane col b = x, rows a = 4 g + e
                sc = mfma32(rh, kl, sc);
                sc = mfma32(rl, kh, sc);
                const b8v gyr = ld_b8(rowI + C_GY * ARR64 + 64 * s), vr = ld_b8(rowJ + C_V * ARR64 + 64 * s);
                dab = mfma32(gyr, vr, dab);                    // dA[row a][col b]
                dba = mfma32(vr, gyr, dba);                    // dA^T[row b][col a]
            }
            float scm[4], dabm[4], dbam[4];
            if (diag) {
                float cf[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float4 t = ldf4(smem + L_CFQ + (qq * 64 + 16 * tI + 4 * g) * 4);
                    cf[0] += t.x; cf[1] += t.y; cf[2] += t.z; cf[3] += t.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = 4 * g + e;                   // row index; column index = x
                    scm[e] = x < o ? sc[e] : (x == o ? cf[e] : 0.f);       // A[a = o][b = x]: b < a, bonus on the diagonal
                    dabm[e] = x < o ? dab[e] : 0.f;                        // dA[a = o][b = x], strictly lower
                    dbam[e] = o < x ? dba[e] : 0.f;                        // dA^T[b = o][a = x], strictly lower
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { scm[e] = sc[e]; dabm[e] = dab[e]; dbam[e] = dba[e]; }
            }
            uint2 th, tl;
            split4(scm, th, tl);
            *reinterpret_cast<uint4*>(smem + L_SCF + wid * 1024 + lane * 16) = make_uint4(th.x, th.y, tl.x, tl.y);
            split4(dabm, th, tl);
            *reinterpret_cast<uint4*>(smem + L_DAF + (2 * wid) * 1024 + lane * 16) = make_uint4(th.x, th.y, tl.x, tl.y);
            split4(dbam, th, tl);
            *reinterpret_cast<uint4*>(smem + L_DAF + (2 * wid + 1) * 1024 + lane * 16) = make_uint4(th.x, th.y, tl.x, tl.y);
        }
        WKV6_T(ts3);
        __syncthreads();
        WKV6_T(ts4);
        // =============================== phase C: this wave's tiles, epilogue, G update ==============================
        {
            const float4 nI4 = ldf4(smem + L_NI + (I * 64 + ch0) * 4);
            const float nIv[4] = {nI4.x, nI4.y, nI4.z, nI4.w};
            // ---- dq^T[i = ch0 + e][a = x]
            float accq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                if (J > I) continue;
                const s4v khf = tr_read(img + C_KH * ARR64 + 16 * J * RSB + troff + 32 * q);     // Khat[b = 4g+e][i = 16q + x]
                const s4v klf = tr_read(img + C_KL * ARR64 + 16 * J * RSB + troff + 32 * q);
                const uint4 f = *reinterpret_cast<const uint4*>(smem + L_DAF + (2 * tile_id(I, J) + 1) * 1024 + lane * 16);
                const s4v d_hi = __builtin_bit_cast(s4v, make_uint2(f.x, f.y)), d_lo = __builtin_bit_cast(s4v, make_uint2(f.z, f.w));
                f4v o = {0.f, 0.f, 0.f, 0.f};
                o = mfma16(khf, d_hi, o);                      // sum_b Khat[b][i] dA[a][b]
                o = mfma16(khf, d_lo, o);
                o = mfma16(klf, d_hi, o);
                if (J == I) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) accq[e] += o[e];
                } else {
                    const float4 nJ4 = ldf4(smem + L_NI + (J * 64 + ch0) * 4);
                    accq[0] += ldexpf(o[0], (int)(nIv[0] - nJ4.x)); accq[1] += ldexpf(o[1], (int)(nIv[1] - nJ4.y));
                    accq[2] += ldexpf(o[2], (int)(nIv[2] - nJ4.z)); accq[3] += ldexpf(o[3], (int)(nIv[3] - nJ4.w));
                }
            }
            {   // state term: sum_j S[i][j] gy_a[j], four 16-wide slices of j
                f4v o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jg = 0; jg < 4; ++jg) {
                    const s4v shf = tr_read(smem + L_SOP + 16 * jg * RSB + troff + 32 * q);           // S[i = 16q + x][j = 16jg + 4g + e]
                    const s4v slf = tr_read(smem + L_SOP + ARR64 + 16 * jg * RSB + troff + 32 * q);
                    const s4v gyb = *reinterpret_cast<const s4v*>(img + C_GY * ARR64 + p * RSB + (16 * jg + 4 * g) * 2);   // gy[a = x][j]
                    o = mfma16(shf, gyb, o);
                    o = mfma16(slf, gyb, o);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) accq[e] += ldexpf(o[e], (int)nIv[e]);
            }
            // ---- dk^T[i = ch0 + e][b = x]
            float acck[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int I2 = 0; I2 < 4; ++I2) {
                if (I2 < I) continue;
                const s4v rhf = tr_read(img + C_RH * ARR64 + 16 * I2 * RSB + troff + 32 * q);    // Rhat[a = 4g+e][i = 16q + x]
                const s4v rlf = tr_read(img + C_RL * ARR64 + 16 * I2 * RSB + troff + 32 * q);
                const uint4 f = *reinterpret_cast<const uint4*>(smem + L_DAF + (2 * tile_id(I2, I)) * 1024 + lane * 16);
                const s4v d_hi = __builtin_bit_cast(s4v, make_uint2(f.x, f.y)), d_lo = __builtin_bit_cast(s4v, make_uint2(f.z, f.w));
                f4v o = {0.f, 0.f, 0.f, 0.f};
                o = mfma16(rhf, d_hi, o);                      // sum_a Rhat[a][i] dA[a][b]
                o = mfma16(rhf, d_lo, o);
                o = mfma16(rlf, d_hi, o);
                if (I2 == I) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acck[e] += o[e];
                } else {
                    const float4 n2 = ldf4(smem + L_NI + (I2 * 64 + ch0) * 4);
                    acck[0] += ldexpf(o[0], (int)(n2.x - nIv[0])); acck[1] += ldexpf(o[1], (int)(n2.y - nIv[1]));
                    acck[2] += ldexpf(o[2], (int)(n2.z - nIv[2])); acck[3] += ldexpf(o[3], (int)(n2.w - nIv[3]));
                }
            }
            const float4 nE4 = ldf4(smem + L_NE + ch0 * 4);
            {   // G term: sum_j Gop[i][j] v_b[j]
                f4v o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const b8v gh = ld_b8(smem + L_GOP + (16 * q + x) * RSB + (32 * s + 8 * g) * 2);
                    const b8v gl = ld_b8(smem + L_GOP + ARR64 + (16 * q + x) * RSB + (32 * s + 8 * g) * 2);
                    const b8v vr = ld_b8(img + C_V * ARR64 + p * RSB + (32 * s + 8 * g) * 2);
                    o = mfma32(gh, vr, o);
                    o = mfma32(gl, vr, o);
                }
                acck[0] += ldexpf(o[0], (int)(nE4.x - nIv[0])); acck[1] += ldexpf(o[1], (int)(nE4.y - nIv[1]));
                acck[2] += ldexpf(o[2], (int)(nE4.z - nIv[2])); acck[3] += ldexpf(o[3], (int)(nE4.w - nIv[3]));
            }
            // ---- gv^T[j = ch0 + e][b = x]
            f4v ov = {0.f, 0.f, 0.f, 0.f};
            s4v gyf[4];                                        // gy[a = 16 I2 + 4g+e][j = 16q + x]: also the A operand of the G update
#pragma unroll
            for (int I2 = 0; I2 < 4; ++I2) gyf[I2] = tr_read(img + C_GY * ARR64 + 16 * I2 * RSB + troff + 32 * q);
#pragma unroll
            for (int I2 = 0; I2 < 4; ++I2) {
                if (I2 < I) continue;
                const uint4 f = *reinterpret_cast<const uint4*>(smem + L_SCF + tile_id(I2, I) * 1024 + lane * 16);
                const s4v s_hi = __builtin_bit_cast(s4v, make_uint2(f.x, f.y)), s_lo = __builtin_bit_cast(s4v, make_uint2(f.z, f.w));
                ov = mfma16(gyf[I2], s_hi, ov);                // sum_a gy[a][j] A[a][b]
                ov = mfma16(gyf[I2], s_lo, ov);
            }
#pragma unroll
            for (int ig = 0; ig < 4; ++ig) {                   // G term: sum_i Gop[i][j] (Khat_b[i] 2^{NE - N_I}[i])
                const s4v gth = tr_read(smem + L_GOP + 16 * ig * RSB + troff + 32 * q);               // Gop[i = 16ig + 4g+e][j = 16q + x]
                const s4v gtl = tr_read(smem + L_GOP + ARR64 + 16 * ig * RSB + troff + 32 * q);
                uint2 kh2 = *reinterpret_cast<const uint2*>(img + C_KH * ARR64 + p * RSB + (16 * ig + 4 * g) * 2);   // Khat[b = x][i]
                uint2 kl2 = *reinterpret_cast<const uint2*>(img + C_KL * ARR64 + p * RSB + (16 * ig + 4 * g) * 2);
                const float4 ni = ldf4(smem + L_NI + (I * 64 + 16 * ig + 4 * g) * 4), ne = ldf4(smem + L_NE + (16 * ig + 4 * g) * 4);
                const unsigned s0 = pack_shift(ni.x - ne.x, ni.y - ne.y), s1 = pack_shift(ni.z - ne.z, ni.w - ne.w);
                kh2.x = shift_pair(kh2.x, s0); kh2.y = shift_pair(kh2.y, s1);
                kl2.x = shift_pair(kl2.x, s0); kl2.y = shift_pair(kl2.y, s1);
                const s4v khs = __builtin_bit_cast(s4v, kh2), kls = __builtin_bit_cast(s4v, kl2);
                ov = mfma16(gth, khs, ov);
                ov = mfma16(gth, kls, ov);
                ov = mfma16(gtl, khs, ov);
            }
            // ---- G[i = 16 I + x][j = ch0 + e] <- 2^{P4[i]} G + sum_I2 2^{N_I2[i]} sum_a gy[a][j] Rhat[a][i]
            {
                const int i = 16 * I + x;
                const float pt = *reinterpret_cast<const float*>(smem + L_PT + i * 4);
                const float e4 = exp2_fast(pt);
#pragma unroll
                for (int e = 0; e < 4; ++e) Gt[e] *= e4;
#pragma unroll
                for (int I2 = 0; I2 < 4; ++I2) {
                    const s4v rth = tr_read(img + C_RH * ARR64 + 16 * I2 * RSB + troff + 32 * I);    // Rhat[a = 4g+e][i = 16 I + x]
                    const s4v rtl = tr_read(img + C_RL * ARR64 + 16 * I2 * RSB + troff + 32 * I);
                    f4v o = {0.f, 0.f, 0.f, 0.f};
                    o = mfma16(gyf[I2], rth, o);
                    o = mfma16(gyf[I2], rtl, o);
                    const int n2 = (int)*reinterpret_cast<const float*>(smem + L_NI + (I2 * 64 + i) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) Gt[e] += ldexpf(o[e], n2);
                }
            }
            // ---- epilogue: token pos, channels ch0 .. ch0+3
            {
                const int pos = c * CHK + p;
                float vg = 0.f;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) vg += *reinterpret_cast<const float*>(smem + L_VGQ + (qq * 64 + p) * 4);
                float o_gr[4], o_gk[4], dl[4], sfx[4], bt[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dq = fR[e] * accq[e], dk = fK[e] * acck[e];
                    o_gr[e] = fmaf(vg * uu[e], kv[e], dq);
                    o_gk[e] = fmaf(vg * uu[e], rv[e], dk);
                    gu_acc[e] = fmaf(vg * rv[e], kv[e], gu_acc[e]);
                    bt[e] = kv[e] * dk;
                    dl[e] = rv[e] * dq - bt[e];
                    sfx[e] = dl[e];
                }
                WKV6_DPP_STEP4(sfx, "row_shl:1");              // inclusive suffix sums over the later tokens of the block
                WKV6_DPP_STEP4(sfx, "row_shl:2");
                WKV6_DPP_STEP4(sfx, "row_shl:4");
                WKV6_DPP_STEP4(sfx, "row_shl:8");
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gw_loc[e] = (sfx[e] - dl[e]) - bt[e];
                    gw_lwe[e] = lwc[e];
                }
                if (x == 0) *reinterpret_cast<float4*>(smem + L_TDL + (I * 64 + ch0) * 4) = make_float4(sfx[0], sfx[1], sfx[2], sfx[3]);
                emit(0, rs_gr, ogr, pos, REV_R, ch0, o_gr);
                emit(1, rs_gk, ogk, pos, REV_K, ch0, o_gk);
                float o_gv[4] = {ov[0], ov[1], ov[2], ov[3]};
                emit(2, rs_gv, ogv, pos, REV_V, ch0, o_gv);
            }
            if (c > 0) decay_scan(c - 1);
        }
        WKV6_T(ts5);
        __syncthreads();
        WKV6_T(ts6);
        WKV6_ACC(0, ts1, ts0); WKV6_ACC(1, ts2, ts1); WKV6_ACC(2, ts3, ts2); WKV6_ACC(3, ts4, ts3); WKV6_ACC(4, ts5, ts4); WKV6_ACC(5, ts6, ts5);
    }
    if (nC > 0) finish_gw(0);
    // ---- gu [B, C] partials: sum over this wave's tokens, then over the four blocks
    if (a.gu) {
        float s4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) s4[e] = row_sum16(gu_acc[e]);
        if (x == 0) *reinterpret_cast<float4*>(smem + L_GU + (I * 64 + ch0) * 4) = make_float4(s4[0], s4[1], s4[2], s4[3]);
        __syncthreads();
        if (I == 0 && x == 0) {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                const float4 v4 = ldf4(smem + L_GU + (J * 64 + ch0) * 4);
                t[0] += v4.x; t[1] += v4.y; t[2] += v4.z; t[3] += v4.w;
            }
            const long o = (long)b * a.C + h * HEAD + ch0;
            if (a.part_f32) io4<float>::store(reinterpret_cast<float*>(a.gu) + o, t);
            else io4<bf16_t>::store(reinterpret_cast<bf16_t*>(a.gu) + o, t);
        }
    }
    if (a.gs) {   // dL/dS0, layout [j][i]: this lane holds G[i = 16 I + x][j = ch0 + e]
        const long so_ = ((long)b * a.H + h) * HEAD * HEAD + 16 * I + x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long o = so_ + (long)(ch0 + e) * HEAD;
            if (a.part_f32) reinterpret_cast<float*>(a.gs)[o] = Gt[e];
            else reinterpret_cast<bf16_t*>(a.gs)[o] = (bf16_t)(pack_bf2(Gt[e], 0.f) & 0xffffu);
        }
    }
#ifdef WKV6_DEBUGBUF
    WKV6_CLK(clk1, rtc1);
    if (a.aux && lane == 0) {
        unsigned long long* const d = reinterpret_cast<unsigned long long*>(a.aux) + ((long)bh * 16 + wid) * 8;
#ifdef WKV6_STAMP
        for (int i_ = 0; i_ < 6; ++i_) d[i_] = stamp_acc[i_];
#endif
        d[6] = clk1 - clk0;
        d[7] = rtc1 - rtc0;
    }
#endif
    if (GEN && a.zero_tail && !a.accumulate) {
        const float z[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = ntok + (tid >> 4); t < a.T; t += (int)(blockDim.x >> 4)) {
            const unsigned idx = (unsigned)(t * a.C + 4 * (tid & 15));
            io4<bf16_t>::store(ogr + idx, z);
            io4<bf16_t>::store(ogk + idx, z);
            io4<bf16_t>::store(ogv + idx, z);
            io4<bf16_t>::store(ogw + idx, z);
        }
    }
}

template <bool W_RAW, bool GEN> hipError_t launch_bwd64_inst(const ScanArgs& a, hipStream_t st)
{
    constexpr size_t lds = BWD64_LDS;
    static LdsAttrOnce attr;                   // per instantiation and device
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(chunk_bwd64_kernel<W_RAW, GEN>), lds)) return e;
    hipLaunchKernelGGL((chunk_bwd64_kernel<W_RAW, GEN>), dim3(a.B * a.H), dim3(1024), lds, st, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_chunk_bwd64(const ScanArgs& a_, hipStream_t st)
{
    ScanArgs a = a_;
    a.split = 0;
#ifdef WKV6_DEBUGBUF
    a.aux = reinterpret_cast<float*>(g_stamp_buffer);
#endif
    const bool gen = a.accumulate || a.zero_tail || a.g_f32[0] || a.g_f32[1] || a.g_f32[2] || a.g_f32[3];
    if (a.wkind) return gen ? launch_bwd64_inst<true, true>(a, st) : launch_bwd64_inst<true, false>(a, st);
    return gen ? launch_bwd64_inst<false, true>(a, st) : launch_bwd64_inst<false, false>(a, st);
}

}  // namespace wkv6
