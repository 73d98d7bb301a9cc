// Chunked MFMA backward of WKV6 for gfx950 (bf16 I/O), two-level form: 64-token chunks, 16-token blocks, 16 identical waves.
//
// Why a second backward (the 12-wave staged kernel is wkv6_chunk_bwd12k.hip): that kernel touches, scales, splits and updates
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
constexpr int L_NI = L_TOT + 1024;                     // int   [4][64]  N_I
constexpr int L_PT = L_NI + 1024;                      // float [64]     P4
constexpr int L_NE = L_PT + 256;                       // (unused)
constexpr int L_VGQ = L_NE + 256;                      // float [4 q][64 tokens]  per-quarter gy.v
constexpr int L_CFQ = L_VGQ + 1024;                    // float [4 q][64 tokens]  per-quarter sum r u k
constexpr int L_TDL = L_CFQ + 1024;                    // float [4][64]  per-block sums of r dq - k dk (gw suffix across blocks)
constexpr int L_GU = L_TDL;                            // float [4][64]  per-block gu partials (end of the kernel: the region is free then)
constexpr int L_MID = L_TDL + 1024;                    // float [4][64]  in-block prefix of lw2 at token 8 (next chunk, written in phase C)
constexpr int L_SHE = L_MID + 1024;                    // uint [4 blocks][32]  packed exponent shifts N_I - NE, two channels per word
constexpr int L_DD = L_SHE + 4 * 128;                  // int  [5][64]  frame steps: N_I - N_{I-1} (N_{-1} = 0), and NE - N_3   (all <= 0)
constexpr int L_RCX = L_DD + 5 * 256;                  // float [4][64]  gw: sum of r dq - k dk over everything behind block I of the previous chunk
constexpr int L_VG = L_RCX + 1024;                     // float [64]     gy.v per token
constexpr int BWD64_LDS = L_VG + 256;
static_assert(BWD64_LDS <= 160 * 1024, "LDS budget");

// four independent in-row scans, one step (see wkv6_chunk_bwd12k.hip for the wait-state reasoning)
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
__device__ __forceinline__ unsigned pack_shift_i(int n0, int n1)
{   // the same from integers >= 0
    return ((unsigned)min(n0, 255) << 7) | ((unsigned)min(n1, 255) << 23);
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
__device__ __forceinline__ int4 ldi4(const char* p) { return *reinterpret_cast<const int4*>(p); }

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
    // gradient store of scan position pos, channels ch..ch+3 (same contract as wkv6_chunk_bwd12k.hip: emit)
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
    auto load_chunk = [&](int c, int p, int ch0) {             // tokens past the end load zeros
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
    auto decay_scan = [&](int c, int I, int p, int ch0, int x) {
        const bool valid = c * CHK + CHK <= ntok || c * CHK + p < ntok;      // (the first test is wave-uniform: full chunks skip the lane test)
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
        if (x == 8) *reinterpret_cast<float4*>(smem + L_MID + (I * 64 + ch0) * 4) = make_float4(loc[0], loc[1], loc[2], loc[3]);
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

    char* const img = smem + L_IMG;
    f4v Gt = {0.f, 0.f, 0.f, 0.f};                             // G[i = 16 I + x][j = 16 q + 4 g + e]
    float gu_acc[4] = {0.f, 0.f, 0.f, 0.f};
    float gw_loc[4] = {0.f, 0.f, 0.f, 0.f}, gw_lwe[4] = {0.f, 0.f, 0.f, 0.f};   // gw of the previous chunk, finished one chunk later
    float rc_run = 0.f;                                        // wave 15, lane = channel: sum of r dq - k dk over everything behind the chunk
    // gw of chunk cp needs the sums of (r dq - k dk) over the blocks behind this one in its chunk and over everything behind
    // the chunk: wave 15 (lane = channel) turns the per-block totals of phase C into those prefixes once for the workgroup ...
    auto reduce_tdl = [&](int ln) {
        const float t0 = *reinterpret_cast<const float*>(smem + L_TDL + ln * 4), t1 = *reinterpret_cast<const float*>(smem + L_TDL + (64 + ln) * 4);
        const float t2 = *reinterpret_cast<const float*>(smem + L_TDL + (128 + ln) * 4), t3 = *reinterpret_cast<const float*>(smem + L_TDL + (192 + ln) * 4);
        const float r3 = rc_run, r2 = r3 + t3, r1 = r2 + t2, r0 = r1 + t1;
        rc_run = r0 + t0;
        *reinterpret_cast<float*>(smem + L_RCX + ln * 4) = r0; *reinterpret_cast<float*>(smem + L_RCX + (64 + ln) * 4) = r1;
        *reinterpret_cast<float*>(smem + L_RCX + (128 + ln) * 4) = r2; *reinterpret_cast<float*>(smem + L_RCX + (192 + ln) * 4) = r3;
    };
    // ... and every wave finishes its tokens one barrier later
    auto finish_gw = [&](int cp, int I, int p, int ch0) {
        const float4 rc = ldf4(smem + L_RCX + (I * 64 + ch0) * 4);
        float o_gw[4] = {(rc.x + gw_loc[0]) * gw_lwe[0], (rc.y + gw_loc[1]) * gw_lwe[1], (rc.z + gw_loc[2]) * gw_lwe[2], (rc.w + gw_loc[3]) * gw_lwe[3]};
        emit(3, rs_gw, ogw, cp * CHK + p, REV_W, ch0, o_gw);
    };

    if (nC > 0) {
        request_ckpt(nC - 1);                                  // (before the loads: phase A infers the DMA's arrival from theirs)
        load_chunk(nC - 1, p, ch0);
        decay_scan(nC - 1, I, p, ch0, x);
        asm volatile("" : "+v"(pr.x), "+v"(pr.y), "+v"(pk.x), "+v"(pk.y), "+v"(pv.x), "+v"(pv.y), "+v"(pg.x), "+v"(pg.y));   // (see the end of phase C)
    }
    __syncthreads();
    // Lane- and wave-derived quantities (token, channels, LDS addresses, triangle masks, the wave's block / quarter and every
    // condition on them) are loop invariants that hipcc would hoist out of the chunk loop and then spill (the kernel lives at
    // 128 VGPRs and the SGPR file is full of buffer descriptors); each phase re-derives them from opaque copies of the lane and
    // wave ids instead: a handful of integer instructions per phase.
#define WKV6_LANE_VIEW() int ln_ = lane, wv_ = wid; asm volatile("" : "+v"(ln_), "+s"(wv_)); \
        const int wid = wv_, I = wv_ >> 2, q = wv_ & 3; (void)wid; (void)q; \
        const int x = ln_ & 15, g = ln_ >> 4, p = 16 * I + x, ch0 = 16 * q + 4 * g; \
        const int troff = (4 * g + (x >> 2)) * RSB + 8 * (x & 3);   /* transposed read of a 16-row block: + first row * RSB + 32 * column group */ \
        (void)p; (void)ch0; (void)troff
    for (int c = nC - 1; c >= 0; --c) {
        // =============================== phase A: operands of chunk c ===============================================
        uint2 rk_r, rk_k;                                      // this chunk's r, k (raw) for the epilogue
        float fR[4], fK[4], lwc[4];
        WKV6_T(ts0);
        {
            WKV6_LANE_VIEW();
            // The checkpoint DMA is invisible to the compiler's counters, but it was issued BEFORE this chunk's input loads and
            // vector-memory operations retire in issue order: once the loads have returned (the compiler waits for them ahead of
            // this statement, which reads their results) the DMA has landed.  The previous chunk's gradient stores are younger
            // and stay in flight -- a plain s_waitcnt vmcnt(0) here would wait for them at the head of every chunk.
            asm volatile("" :: "v"(pr.x), "v"(pk.y), "v"(pv.x), "v"(pg.y) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            // frame of this block for this lane's channels: N_I = rint(P_I + m_I), P_I = sum of the totals of the blocks before I
            // (always summed in block order: every wave that forms a prefix must get the same bits), m_I = the in-block prefix at
            // token 8.  Also N of the block before (0 for block 0): the chains of phase C move from frame to frame.
            float nI[4], nP[4] = {0.f, 0.f, 0.f, 0.f}, mI[4], pI[4] = {0.f, 0.f, 0.f, 0.f};
            for (int J = 0; J < I; ++J) {                      // (wave-uniform trip count)
                const float4 t = ldf4(smem + L_TOT + (J * 64 + ch0) * 4);
                if (J == I - 1) {
                    const float4 m4 = ldf4(smem + L_MID + (J * 64 + ch0) * 4);
                    nP[0] = __builtin_rintf(pI[0] + m4.x); nP[1] = __builtin_rintf(pI[1] + m4.y);
                    nP[2] = __builtin_rintf(pI[2] + m4.z); nP[3] = __builtin_rintf(pI[3] + m4.w);
                }
                pI[0] += t.x; pI[1] += t.y; pI[2] += t.z; pI[3] += t.w;
            }
            {
                const float4 m4 = ldf4(smem + L_MID + (I * 64 + ch0) * 4);
                const float mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float cmid = pI[e] + mm[e];
                    nI[e] = __builtin_rintf(cmid);
                    mI[e] = mm[e] - (cmid - nI[e]);            // N_I - P_I = m - phi, |phi| <= 0.5: the frame sits within half a bit of token 8
                    // fR = 2^{C_p - N_I} = 2^{loc - (N_I - P_I)},  fK = 2^{N_I - C_{p+1}} = 2^{(N_I - P_I) - (loc + lw2)}
                    fR[e] = exp2_fast(loc[e] - mI[e]);
                    fK[e] = exp2_fast(mI[e] - (loc[e] + lw2v[e]));
                    lwc[e] = lwe[e];
                }
            }
            if (x == 0) {
                *reinterpret_cast<int4*>(smem + L_NI + (I * 64 + ch0) * 4) = make_int4((int)nI[0], (int)nI[1], (int)nI[2], (int)nI[3]);
                *reinterpret_cast<int4*>(smem + L_DD + (I * 64 + ch0) * 4) =
                    make_int4((int)(nI[0] - nP[0]), (int)(nI[1] - nP[1]), (int)(nI[2] - nP[2]), (int)(nI[3] - nP[3]));   // frame I-1 -> I (<= 0)
                float pj[4] = {pI[0], pI[1], pI[2], pI[3]};
                for (int J = I; J < 4; ++J) {                  // the rest of the chunk: P4
                    const float4 t = ldf4(smem + L_TOT + (J * 64 + ch0) * 4);
                    pj[0] += t.x; pj[1] += t.y; pj[2] += t.z; pj[3] += t.w;
                }
                const float nE[4] = {__builtin_rintf(pj[0]), __builtin_rintf(pj[1]), __builtin_rintf(pj[2]), __builtin_rintf(pj[3])};
                // packed exponent shifts of this block's Khat into the frame of the chunk end (G term of gv)
                *reinterpret_cast<uint2*>(smem + L_SHE + I * 128 + ch0 * 2) =
                    make_uint2(pack_shift(nI[0] - nE[0], nI[1] - nE[1]), pack_shift(nI[2] - nE[2], nI[3] - nE[3]));
                if (I == 3) *reinterpret_cast<int4*>(smem + L_DD + (4 * 64 + ch0) * 4) =
                    make_int4((int)(nE[0] - nI[0]), (int)(nE[1] - nI[1]), (int)(nE[2] - nI[2]), (int)(nE[3] - nI[3]));   // frame 3 -> chunk end
                if (I == 0) *reinterpret_cast<float4*>(smem + L_PT + ch0 * 4) = make_float4(pj[0], pj[1], pj[2], pj[3]);
            }
            rk_r = pr; rk_k = pk;
            const float rv[4] = {bf_lo(pr.x), bf_hi(pr.x), bf_lo(pr.y), bf_hi(pr.y)};
            const float kv[4] = {bf_lo(pk.x), bf_hi(pk.x), bf_lo(pk.y), bf_hi(pk.y)};
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
                const float4 s4 = ldf4(smem + L_CKQ + wid * 1024 + ln_ * 16);
                const int j = 16 * (wid >> 2) + x, i0 = tile_ch(wid & 3) + 8 * g;
                const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
                split4(sv, hi, lo);
                *reinterpret_cast<uint2*>(smem + L_SOP + j * RSB + i0 * 2) = hi;
                *reinterpret_cast<uint2*>(smem + L_SOP + ARR64 + j * RSB + i0 * 2) = lo;
            }
            // Gop = 2^{P4 - rint(P4)} (.) G -> operand image [i][j]
            {
                const int i = 16 * I + x;
                // (summed in the order every other use of P4 sums it: the rounding to NE must agree bit for bit)
                const float pt = ((*reinterpret_cast<const float*>(smem + L_TOT + i * 4) + *reinterpret_cast<const float*>(smem + L_TOT + (64 + i) * 4))
                                  + *reinterpret_cast<const float*>(smem + L_TOT + (128 + i) * 4)) + *reinterpret_cast<const float*>(smem + L_TOT + (192 + i) * 4);
                const float sc = exp2_fast(pt - __builtin_rintf(pt));
                const float gs4[4] = {Gt[0] * sc, Gt[1] * sc, Gt[2] * sc, Gt[3] * sc};
                split4(gs4, hi, lo);
                *reinterpret_cast<uint2*>(smem + L_GOP + i * RSB + ch0 * 2) = hi;
                *reinterpret_cast<uint2*>(smem + L_GOP + ARR64 + i * RSB + ch0 * 2) = lo;
            }
            // the landing zone has been read back (the split consumed it): request the next checkpoint, then the next chunk's inputs
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (c > 0) {
                request_ckpt(c - 1);
                load_chunk(c - 1, p, ch0);
            }
        }
        WKV6_T(ts1);
        __syncthreads();
        WKV6_T(ts2);
        // =============================== phase B: score and dA tiles, once for the workgroup ==========================
        {
          WKV6_LANE_VIEW();
          if (wid < 10) {
            // ---- waves 0..9: score tile (tI, tJ) = tile `wid`
            const int tI = (wid >= 1) + (wid >= 3) + (wid >= 6), tJ = wid - tI * (tI + 1) / 2;
            const bool diag = tI == tJ;
            const char* const rowI = img + (16 * tI + x) * RSB + 16 * g;      // + array, + 64 s
            const char* const rowJ = img + (16 * tJ + x) * RSB + 16 * g;
            f4v sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                b8v rh = ld_b8(rowI + C_RH * ARR64 + 64 * s), rl = ld_b8(rowI + C_RL * ARR64 + 64 * s);
                const b8v kh = ld_b8(rowJ + C_KH * ARR64 + 64 * s), kl = ld_b8(rowJ + C_KL * ARR64 + 64 * s);
                if (!diag) {   // Rhat of block I into the frame of block J: channels 32 s + 8 g .. +7, shift N_J - N_I >= 0
                    const char* const nI_ = smem + L_NI + (tI * 64 + 32 * s + 8 * g) * 4;
                    const char* const nJ_ = smem + L_NI + (tJ * 64 + 32 * s + 8 * g) * 4;
                    const int4 i0 = ldi4(nI_), i1 = ldi4(nI_ + 16), j0 = ldi4(nJ_), j1 = ldi4(nJ_ + 16);
                    const unsigned sh[4] = {pack_shift_i(j0.x - i0.x, j0.y - i0.y), pack_shift_i(j0.z - i0.z, j0.w - i0.w),
                                            pack_shift_i(j1.x - i1.x, j1.y - i1.y), pack_shift_i(j1.z - i1.z, j1.w - i1.w)};
                    rh = shift_frag(rh, sh);
                    rl = shift_frag(rl, sh);
                }
                sc = mfma32(rh, kh, sc);                       // A[row a][col b]: lane col b = x, rows a = 4 g + e
                sc = mfma32(rh, kl, sc);
                sc = mfma32(rl, kh, sc);
            }
            float scm[4] = {sc[0], sc[1], sc[2], sc[3]};
            if (diag) {
                float cf[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float4 t = ldf4(smem + L_CFQ + (qq * 64 + 16 * tI + 4 * g) * 4);
                    cf[0] += t.x; cf[1] += t.y; cf[2] += t.z; cf[3] += t.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = 4 * g + e;                   // row index a; column index b = x
                    scm[e] = x < o ? sc[e] : (x == o ? cf[e] : 0.f);       // b < a, bonus on the diagonal
                }
            }
            uint2 th, tl;
            split4(scm, th, tl);
            *reinterpret_cast<uint4*>(smem + L_SCF + wid * 1024 + ln_ * 16) = make_uint4(th.x, th.y, tl.x, tl.y);
          } else {
            // ---- waves 10..15: the dA tiles, both orientations: waves 10..13 two tiles each (0..7), waves 14, 15 tiles 8, 9
            const int t0 = wid < 14 ? 2 * (wid - 10) : wid - 6, nt = wid < 14 ? 2 : 1;
            for (int k_ = 0; k_ < nt; ++k_) {
                const int t = t0 + k_;
                const int tI = (t >= 1) + (t >= 3) + (t >= 6), tJ = t - tI * (tI + 1) / 2;
                const char* const rowI = img + (16 * tI + x) * RSB + 16 * g;
                const char* const rowJ = img + (16 * tJ + x) * RSB + 16 * g;
                f4v dab = {0.f, 0.f, 0.f, 0.f}, dba = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const b8v gyr = ld_b8(rowI + C_GY * ARR64 + 64 * s), vr = ld_b8(rowJ + C_V * ARR64 + 64 * s);
                    dab = mfma32(gyr, vr, dab);                // dA[row a][col b]
                    dba = mfma32(vr, gyr, dba);                // dA^T[row b][col a]
                }
                float dabm[4] = {dab[0], dab[1], dab[2], dab[3]}, dbam[4] = {dba[0], dba[1], dba[2], dba[3]};
                if (tI == tJ) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int o = 4 * g + e;
                        dabm[e] = x < o ? dab[e] : 0.f;        // dA[a = o][b = x], strictly lower
                        dbam[e] = o < x ? dba[e] : 0.f;        // dA^T[b = o][a = x], strictly lower
                    }
                }
                uint2 th, tl;
                split4(dabm, th, tl);
                *reinterpret_cast<uint4*>(smem + L_DAF + (2 * t) * 1024 + ln_ * 16) = make_uint4(th.x, th.y, tl.x, tl.y);
                split4(dbam, th, tl);
                *reinterpret_cast<uint4*>(smem + L_DAF + (2 * t + 1) * 1024 + ln_ * 16) = make_uint4(th.x, th.y, tl.x, tl.y);
            }
            if (wid == 15 && c + 1 < nC) reduce_tdl(ln_);     // prefixes for the gw of chunk c + 1 (its totals: phase C of c + 1)
            if (wid == 14) {                                   // gy.v per token, summed over the four channel quarters (lane = token)
                const float s_ = (*reinterpret_cast<const float*>(smem + L_VGQ + ln_ * 4) + *reinterpret_cast<const float*>(smem + L_VGQ + (64 + ln_) * 4))
                               + (*reinterpret_cast<const float*>(smem + L_VGQ + (128 + ln_) * 4) + *reinterpret_cast<const float*>(smem + L_VGQ + (192 + ln_) * 4));
                *reinterpret_cast<float*>(smem + L_VG + ln_ * 4) = s_;
            }
          }
        }
        WKV6_T(ts3);
        __syncthreads();
        WKV6_T(ts4);
        // =============================== phase C: this wave's tiles, epilogue, G update ==============================
        {
            WKV6_LANE_VIEW();
            if (c + 1 < nC) finish_gw(c + 1, I, p, ch0);      // (uses the previous chunk's gw_loc / gw_lwe: before they are replaced)
            // Frame changes along the chains below are exact: v_ldexp_f32 on the running MFMA accumulator, which then goes back in
            // as the C operand of the next tile's MFMAs.  A float factor would not do: 2^{N_I - N_J} may lie far below the fp32
            // range while the term it scales (up to 2^104 |k dA|) times fR (up to 2^104) is of order one -- two adjacent tokens on
            // either side of a block boundary in a channel that decays by e^-9 per token.  All shifts are <= 0 (the frames follow
            // the decay), so nothing can overflow on the way.
            // ---- dq^T[i = ch0 + e][a = x]: state term (frame of the chunk start), then the blocks 0 .. I, each in its own frame
            f4v tq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jg = 0; jg < 4; ++jg) {                   // sum_j S[i][j] gy_a[j], four 16-wide slices of j
                const s4v shf = tr_read(smem + L_SOP + 16 * jg * RSB + troff + 32 * q);           // S[i = 16q + x][j = 16jg + 4g + e]
                const s4v slf = tr_read(smem + L_SOP + ARR64 + 16 * jg * RSB + troff + 32 * q);
                const s4v gyb = *reinterpret_cast<const s4v*>(img + C_GY * ARR64 + p * RSB + (16 * jg + 4 * g) * 2);   // gy[a = x][j]
                tq = mfma16(shf, gyb, tq);
                tq = mfma16(slf, gyb, tq);
            }
            for (int J = 0; J <= I; ++J) {
                const int4 dd = ldi4(smem + L_DD + (J * 64 + ch0) * 4);
                tq = f4v{ldexpf(tq[0], dd.x), ldexpf(tq[1], dd.y), ldexpf(tq[2], dd.z), ldexpf(tq[3], dd.w)};
                const s4v khf = tr_read(img + C_KH * ARR64 + 16 * J * RSB + troff + 32 * q);     // Khat[b = 4g+e][i = 16q + x]
                const s4v klf = tr_read(img + C_KL * ARR64 + 16 * J * RSB + troff + 32 * q);
                const uint4 f = *reinterpret_cast<const uint4*>(smem + L_DAF + (2 * (I * (I + 1) / 2 + J) + 1) * 1024 + ln_ * 16);
                const s4v d_hi = __builtin_bit_cast(s4v, make_uint2(f.x, f.y)), d_lo = __builtin_bit_cast(s4v, make_uint2(f.z, f.w));
                tq = mfma16(khf, d_hi, tq);                    // sum_b Khat[b][i] dA[a][b]
                tq = mfma16(khf, d_lo, tq);
                tq = mfma16(klf, d_hi, tq);
            }
            // ---- dk^T[i = ch0 + e][b = x]: G term (frame of the chunk end), then the blocks 3 .. I
            f4v tk;
            {
                f4v o = {0.f, 0.f, 0.f, 0.f};                  // (16x16x32 shape: its own accumulator)
#pragma unroll
                for (int s = 0; s < 2; ++s) {                  // sum_j Gop[i][j] v_b[j]
                    const b8v gh = ld_b8(smem + L_GOP + (16 * q + x) * RSB + (32 * s + 8 * g) * 2);
                    const b8v gl = ld_b8(smem + L_GOP + ARR64 + (16 * q + x) * RSB + (32 * s + 8 * g) * 2);
                    const b8v vr = ld_b8(img + C_V * ARR64 + p * RSB + (32 * s + 8 * g) * 2);
                    o = mfma32(gh, vr, o);
                    o = mfma32(gl, vr, o);
                }
                tk = o;
            }
            for (int I2 = 3; I2 >= I; --I2) {
                const int4 dd = ldi4(smem + L_DD + ((I2 + 1) * 64 + ch0) * 4);     // what sits in frame I2 + 1 (4: chunk end) seen from frame I2
                tk = f4v{ldexpf(tk[0], dd.x), ldexpf(tk[1], dd.y), ldexpf(tk[2], dd.z), ldexpf(tk[3], dd.w)};
                const s4v rhf = tr_read(img + C_RH * ARR64 + 16 * I2 * RSB + troff + 32 * q);    // Rhat[a = 4g+e][i = 16q + x]
                const s4v rlf = tr_read(img + C_RL * ARR64 + 16 * I2 * RSB + troff + 32 * q);
                const uint4 f = *reinterpret_cast<const uint4*>(smem + L_DAF + (2 * (I2 * (I2 + 1) / 2 + I)) * 1024 + ln_ * 16);
                const s4v d_hi = __builtin_bit_cast(s4v, make_uint2(f.x, f.y)), d_lo = __builtin_bit_cast(s4v, make_uint2(f.z, f.w));
                tk = mfma16(rhf, d_hi, tk);                    // sum_a Rhat[a][i] dA[a][b]
                tk = mfma16(rhf, d_lo, tk);
                tk = mfma16(rlf, d_hi, tk);
            }
            // ---- gv^T[j = ch0 + e][b = x]
            f4v ov = {0.f, 0.f, 0.f, 0.f};
            s4v gyf[4];                                        // gy[a = 16 I2 + 4g+e][j = 16q + x]: also the A operand of the G update
#pragma unroll
            for (int I2 = 0; I2 < 4; ++I2) gyf[I2] = tr_read(img + C_GY * ARR64 + 16 * I2 * RSB + troff + 32 * q);
#pragma unroll
            for (int I2 = 0; I2 < 4; ++I2) {
                if (I2 < I) continue;
                const uint4 f = *reinterpret_cast<const uint4*>(smem + L_SCF + (I2 * (I2 + 1) / 2 + I) * 1024 + ln_ * 16);
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
                const uint2 w2 = *reinterpret_cast<const uint2*>(smem + L_SHE + I * 128 + 32 * ig + 8 * g);           // shifts of channels i .. i+3
                kh2.x = shift_pair(kh2.x, w2.x); kh2.y = shift_pair(kh2.y, w2.y);
                kl2.x = shift_pair(kl2.x, w2.x); kl2.y = shift_pair(kl2.y, w2.y);
                const s4v khs = __builtin_bit_cast(s4v, kh2), kls = __builtin_bit_cast(s4v, kl2);
                ov = mfma16(gth, khs, ov);
                ov = mfma16(gth, kls, ov);
                ov = mfma16(gtl, khs, ov);
            }
            // ---- G[i = 16 I + x][j = ch0 + e] <- 2^{P4[i]} G + sum_I2 2^{N_I2[i]} sum_a gy[a][j] Rhat[a][i]
            // (the terms are of order one before scaling: a factor that underflows to 0 scales a negligible term)
            {
                const int i = 16 * I + x;
                const float e4 = exp2_fast(*reinterpret_cast<const float*>(smem + L_PT + i * 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) Gt[e] *= e4;
#pragma unroll
                for (int I2 = 0; I2 < 4; ++I2) {
                    const s4v rth = tr_read(img + C_RH * ARR64 + 16 * I2 * RSB + troff + 32 * I);    // Rhat[a = 4g+e][i = 16 I + x]
                    const s4v rtl = tr_read(img + C_RL * ARR64 + 16 * I2 * RSB + troff + 32 * I);
                    f4v o = {0.f, 0.f, 0.f, 0.f};
                    o = mfma16(gyf[I2], rth, o);
                    o = mfma16(gyf[I2], rtl, o);
                    const float s2 = ldexpf(1.0f, *reinterpret_cast<const int*>(smem + L_NI + (I2 * 64 + i) * 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) Gt[e] = fmaf(o[e], s2, Gt[e]);
                }
            }
            // ---- epilogue: token pos, channels ch0 .. ch0+3
            {
                const int pos = c * CHK + p;
                const float vg = *reinterpret_cast<const float*>(smem + L_VG + p * 4);
                const float rv[4] = {bf_lo(rk_r.x), bf_hi(rk_r.x), bf_lo(rk_r.y), bf_hi(rk_r.y)};
                const float kv[4] = {bf_lo(rk_k.x), bf_hi(rk_k.x), bf_lo(rk_k.y), bf_hi(rk_k.y)};
                float o_gr[4], o_gk[4], dl[4], sfx[4], bt[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dq = fR[e] * tq[e], dk = fK[e] * tk[e];
                    o_gr[e] = fmaf(vg * uu[e], kv[e], dq);
                    o_gk[e] = fmaf(vg * uu[e], rv[e], dk);
                    gu_acc[e] = fmaf(vg * rv[e], kv[e], gu_acc[e]);
                    bt[e] = kv[e] * dk;
                    dl[e] = fmaf(rv[e], dq, -bt[e]);
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
            if (c > 0) {
                decay_scan(c - 1, I, p, ch0, x);
                // The scan has just waited for w, the youngest of the next chunk's input loads, with the exact count of the younger
                // gradient stores (straight-line code: s_waitcnt vmcnt(3)).  Touch the other four here too: left to the head of the
                // loop, their wait is placed at a control-flow merge where the compiler's counters have collapsed to vmcnt(0),
                // which also drains the stores issued just above -- a full store latency exposed at the head of every chunk.
                asm volatile("" : "+v"(pr.x), "+v"(pr.y), "+v"(pk.x), "+v"(pk.y), "+v"(pv.x), "+v"(pv.y), "+v"(pg.x), "+v"(pg.y));
            }
        }
        WKV6_T(ts5);
        __syncthreads();
        WKV6_T(ts6);
        WKV6_ACC(0, ts1, ts0); WKV6_ACC(1, ts2, ts1); WKV6_ACC(2, ts3, ts2); WKV6_ACC(3, ts4, ts3); WKV6_ACC(4, ts5, ts4); WKV6_ACC(5, ts6, ts5);
    }
    if (nC > 0) {                                              // gw of chunk 0
        if (wid == 15) reduce_tdl(lane);
        __syncthreads();
        finish_gw(0, I, p, ch0);
    }
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
