// Chunked MFMA formulation of the WKV6 forward for gfx950 (bf16 I/O).
//
// The token-serial scan spends 3 fp32 VALU ops per state element per token and is issue-latency bound
// (DESIGN.md section 4).  Here the O(N^2)-per-token work runs on the matrix cores instead:
//
//   per 16-token block (a, b = token in block, i = key channel, j = value channel, c_a = exclusive
//   cumulative log-decay inside the block, lw = -exp(w) clamped to >= LW_MIN), with
//   Rhat_a = r_a . e^{c_a - c_8},  Khat_b = k_b . e^{c_8 - c_{b+1}}   (one reference point, token 8, per block,
//   so every exponent spans at most 8 tokens),  E8 = e^{c_8},  E16 = e^{c_16},  E16m8 = e^{c_16 - c_8}:
//     scores   A[a][b] = Rhat_a . Khat_b  (b < a),     A[a][a] = sum_i r_a u k_a
//     output   y_a = A[a][:] V  +  Rhat_a^T (E8 (.) S)                     S = state at block entry
//     state    S  <- E16 (.) S + E16m8 (.) (Khat^T V)
//   which is the recurrence of cuda/wkv6_cuda.cu:44-57 re-associated (same algebra as the reference's own
//   alternative backend, fla/ops/rwkv6/chunk.py:114-309, but with the state kept in registers for the whole
//   sequence instead of being spilled per chunk, and with split-bf16 operands instead of bf16-rounded ones).
//
//   Every fp32 operand of an MFMA is split into bf16 hi + lo parts (x = hi + lo + O(2^-16 x)) and the product
//   is formed as hi*hi + hi*lo + lo*hi with fp32 accumulation, so results carry ~2^-16 relative error instead
//   of bf16's 2^-8; r, k, v themselves are exact in bf16.
//
// Work split: one 512-thread workgroup (8 wave64) per (batch, head); tokens go through in groups of 64.
//   producer waves 4..7: wave 4+w turns block w of group g+1 into MFMA operands in the other LDS buffer
//            (decays, cumulative sums, scaling, hi/lo split; lane = 4 channels x 4 tokens) and issues the global
//            loads of group g+2;
//   consumer waves 0..3: walk the 4 blocks of group g; wave w owns value columns [16w, 16w+16) of the state as
//            the C-layout of four 16x16 MFMA tiles (16 fp32 VGPRs): the state-update MFMA result lands in that
//            layout and the same registers, scaled and converted to bf16, are the A operand of the
//            inter-block product.
//   One workgroup barrier per group; producers and consumers share each SIMD (2 waves per SIMD), so operand
//   preparation (VALU/transcendental) overlaps the MFMA chain.
//
// Decays smaller than e^LW_MIN per token are clamped (their true effect on any output is < 1.3e-4 of the
// carried state, far below bf16 output resolution); the exact scan kernels remain available (WKV6_ALGO_SCAN).
#include <type_traits>
#include "wkv6_chunk.h"

namespace wkv6 {
namespace {

using namespace chunk;

#ifndef WKV6_FWD_PAIR
#define WKV6_FWD_PAIR 0                     // role -> SIMD pairing of the forward's waves (chunk_fwd_body)
#endif
constexpr int GRP_BYTES = NBLK * BLK_BYTES;
constexpr int CKX_BYTES = 4 * 4096;         // checkpoint transposition buffers of the four consumers (row order: wkv6_scan.h)
// y leaves through LDS as whole token rows.  A consumer's result tile is 16 tokens x 16 channels = 32-byte pieces of the [B, T, C]
// rows, and stored as such (8 bytes per lane) the forward ran at exactly the speed of its own store shape: the kernel's memory
// instructions with NO arithmetic take 0.226-0.243 ms at config 2, the same with y as full 128-byte rows at 16 bytes per lane
// 0.190-0.202 (tools/microbench/head_slices.hip: fwd_shapes, profiles/r05_head_slices_microbench.txt).  So the consumers park a
// group's y (bf16) in a double-buffered [64 tokens][144 B] image and, behind the group barrier they have anyway, each stores 16
// whole rows of the group with two 16-byte-per-lane instructions.
constexpr int CONSUMER_STAGGER = 3;         // s_sleep units (64 cycles) by which consumer w trails consumer w - 1 into a group (4 until the
                                            // producers' serial path got shorter late in round 5; re-tuned: profiles/r05_stagger.txt)
constexpr int YRS = 144;                    // staged y row: 128 B + 16 B pad (16-byte row reads stay aligned; writes are 2-way at most)
constexpr int YS_BYTES = GRP * YRS;

// STATE_ONLY: no outputs, only the state recurrence (first half of the self-contained backward).
// With a.ckpt the state is dumped every CKPT_TOK = 64 tokens (fp32, in the register order of the backward's row waves:
// wkv6_scan.h) for the backward kernel.
// ACC: add into y (from a.y_f32 when given) instead of overwriting -- the reverse half of wkv6_bi.
// GN: the per-head GroupNorm and the gate multiply behind the operator (src/model.py:462-468) happen in the
// store epilogue: a token's statistics span the head's 64 channels = the four consumer waves, which exchange their 16-channel
// sums (of the bf16-rounded y, what nn.GroupNorm would see) through LDS at the group barrier the kernel has anyway and write
// GroupNorm_H(y) * gate one group later from the y they kept in registers; y makes no round trip through HBM.
// The kernel proper is a device function of (arguments, workgroup slot): chunk_fwd_kernel runs it on its one argument block,
// chunk_fwd_pair_kernel (the two WKV problems of a bidirectional composition in ONE launch) on one of two.
// CLK: the in-run clock probe (wkv6_set_clock_buffer) is compiled into the plain kernel only.
// The producers' raw input registers of a call.  In the persistent wkv6_bi launch (CHAIN) they outlive the call: a call's producers, idle
// while the consumers work through its last group, prepare group 0 of the NEXT call (the row's reversed half, or the slot's next row) and
// leave that call's group 1 in flight into these registers.
struct FwdRaw {
    uint2 pr[4], pk[4], pv[4], pw[4];
    float4 pe[4];
};
// group `grp` of problem (a, slot), block wv, requested into raw -- the one definition of "which bytes": the body's own requests and the
// previous call's early one must agree
template <bool W_RAW, bool STATE_ONLY, bool AFF>
__device__ __forceinline__ void fwd_request_group(const ScanArgs& a, const int b, const int h, const int ntok, const int rev, const int grp, const int wv,
                                                  const int lane, FwdRaw& raw)   // rev: the row reversed (a's own reversal fields are not read)
{
    const int c4 = lane & 15, tq = lane >> 4;
    const long base = (long)b * a.T * a.C + (long)h * HEAD;
    const TokAddr<AFF> tok(ntok, rev != 0);
    const int C_ = a.C;
    const unsigned span = ntok > 0 ? (unsigned)(ntok - 1) * a.C : 0u;
    const unsigned nb2 = ntok > 0 ? span * 2 + 128 : 0;
    const rsrc_t rs_r = make_rsrc(reinterpret_cast<const bf16_t*>(a.r) + base, nb2), rs_k = make_rsrc(reinterpret_cast<const bf16_t*>(a.k) + base, nb2);
    const rsrc_t rs_v = make_rsrc(reinterpret_cast<const bf16_t*>(a.v) + base, nb2);
    const rsrc_t rs_w = W_RAW ? make_rsrc(reinterpret_cast<const bf16_t*>(a.w) + base, nb2)
                              : make_rsrc(reinterpret_cast<const float*>(a.w) + base, ntok > 0 ? span * 4 + 256 : 0);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        const int pl = wv * BLK + 4 * tq + tt, lp = tok.lane(pl, 4 * c4, C_);
        if constexpr (!STATE_ONLY) raw.pr[tt] = buf_load8(rs_r, tok.off(grp * GRP, pl, 4 * c4, C_, REV_R, lp) * 2);
        else raw.pr[tt] = make_uint2(0u, 0u);
        raw.pk[tt] = buf_load8(rs_k, tok.off(grp * GRP, pl, 4 * c4, C_, REV_K, lp) * 2);
        raw.pv[tt] = buf_load8(rs_v, tok.off(grp * GRP, pl, 4 * c4, C_, REV_V, lp) * 2);
        if constexpr (W_RAW) raw.pw[tt] = buf_load8(rs_w, tok.off(grp * GRP, pl, 4 * c4, C_, REV_W, lp) * 2);
        else raw.pe[tt] = buf_load16f(rs_w, tok.off(grp * GRP, pl, 4 * c4, C_, REV_W, lp) * 4);
    }
}
// What a call of the persistent wkv6_bi launch (CHAIN) knows beyond its argument block.  Its producers, idle while the consumers work through
// the call's last group, PREPARE group 0 of the call that follows it in the workgroup slot (the pipeline fill: one group's time per call, 12
// calls per slot at BASELINE configs[2]) into the operand buffer the last group does not use: group g of a call lives in buffer (g + pb) & 1,
// and a call that was prepared for (chained_in) has neither prologue nor opening barrier.  The call that follows is row nx_bh of the SAME
// argument block (a pointer to another block would put both into scratch memory), reversed or not, with the bonus term or not.
struct FwdChain {
    int b, ntok;                   // this call's batch index and row length (looked up a row ahead by the launch: order[] -> lens[])
    int pb;                        // operand buffer of this call's group 0
    bool chained_in;               // the call before this one has prepared this call's group 0 (and requested group 1)
    bool nx_valid;                 // there is a call behind this one ...
    unsigned nx_bh;                // ... on row nx_bh (batch index nx_b, length nx_ntok),
    int nx_b, nx_ntok;
    bool nx_rev, nx_use_u;         // reversed / with the bonus vector
};
// (unsplit launches) does hardware wave `hw` of a workgroup play a producer?  One definition: chunk_fwd_body's role map below
__device__ __forceinline__ bool fwd_hw_wave_produces(int hw) { return WKV6_FWD_PAIR == 1 ? (hw & 3) < 2 : hw >= 4; }
template <bool W_RAW, bool STATE_ONLY, bool ACC, bool GN, bool AFF, bool CLK = false, bool CHAIN = false>
__device__ __forceinline__ void chunk_fwd_body(const ScanArgs& a, const unsigned slot, const unsigned sslot, FwdRaw& raw, const FwdChain& ch = FwdChain{})
{
    [[maybe_unused]] const bool nxvalid = ch.nx_valid, chained_in = ch.chained_in, nx_use_u = ch.nx_use_u;
    [[maybe_unused]] const unsigned nxbh = ch.nx_bh;
    [[maybe_unused]] const int nxrev = ch.nx_rev, b_known = ch.b, ntok_known = ch.ntok, nxb = ch.nx_b, nxntok = ch.nx_ntok, pb = CHAIN ? ch.pb : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][NBLK][BLK_BYTES] | GN: float [2][4 waves][NBLK][16][2] | float [4 consumers][1024] | y rows [2][64][YRS]
    const int tid = threadIdx.x, lane = tid & 63;
    const SplitConst spc = split_const();
    // a.split (B*H <= half the CUs): two 6-wave workgroups per (batch, head) on two CUs, each with all four producers and two
    // of the four consumers (value columns [32 part, 32 part + 32)): hardware wave w plays consumer 2 part + w for w < 2 and
    // producer w - 2 (wave id 4 + w - 2) otherwise.  The preparation is duplicated, on CUs that would otherwise idle.
    const int hwid = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform, provably so
    const int part = a.split ? (int)(slot & 1) : 0;
    const int bh = a.split ? (int)(slot >> 1) : (CLK ? (int)xcd_row_of_slot(slot, (unsigned)(a.B * a.H)) : (int)slot);   // (CLK: the plain kernel)
    // Role -> hardware wave.  The waves of a workgroup go to the CU's four SIMDs round-robin (hardware waves w and w + 4 share one:
    // profiles/r06_fwd_pairing.txt has the HW_ID read-back).  WKV6_FWD_PAIR = 0: waves 0..3 consume, 4..7 produce -- one producer and one
    // consumer per SIMD.  1: the roles are paired with themselves -- two SIMDs host two producers each, two host two consumers each (split
    // launches, 6 waves: the two consumers share a SIMD, two producers share one, two have a SIMD to themselves).  cidx: the consumer's
    // index within its workgroup (its LDS regions, its rows of a y flush, its stagger).
    int wid, cidx;
#if WKV6_FWD_PAIR == 1
    {
        const int simd = hwid & 3, second = hwid >> 2;
        if (a.split) {
            cidx = second;                                               // (consumers: hardware waves 0 and 4)
            wid = simd == 0 ? 2 * part + second : (simd == 1 ? 4 + second : 4 + simd);   // producers: waves 1, 5 -> blocks 0, 1; 2 -> 2; 3 -> 3
        } else {
            cidx = 2 * (simd - 2) + second;                              // (consumers: hardware waves 2, 6, 3, 7)
            wid = simd < 2 ? 4 + 2 * simd + second : cidx;               // producers: waves 0, 4, 1, 5 -> blocks 0, 1, 2, 3
        }
    }
#else
    wid = a.split ? (hwid < 2 ? 2 * part + hwid : hwid + 2) : hwid;
    cidx = hwid;
#endif
    const bool producer = wid >= 4;
    const int wv = wid & 3;                                          // block (producer) / column tile (consumer)
    // (CHAIN: the persistent launch has looked the row's batch index and length up -- two dependent memory round trips -- a call ahead)
    const int b = CHAIN ? b_known : (a.order ? a.order[bh / a.H] : bh / a.H), h = bh % a.H;
    const long base = (long)b * a.T * a.C + (long)h * HEAD;   // (batch, head) origin: uniform, folded into the pointers;
                                                              // per-lane offsets below stay 32-bit (T*C < 2^31, checked by the API)
    const bf16_t* const gr_ = reinterpret_cast<const bf16_t*>(a.r) + base;
    const bf16_t* const gk_ = reinterpret_cast<const bf16_t*>(a.k) + base;
    const bf16_t* const gv_ = reinterpret_cast<const bf16_t*>(a.v) + base;
    bf16_t* const gy_ = reinterpret_cast<bf16_t*>(a.y) + base;
    int ntok = a.T;
    if constexpr (CHAIN) ntok = ntok_known;
    else if (a.lens) ntok = min(max(a.lens[b], 0), a.T);
    const int ngrp = (ntok + GRP - 1) / GRP;
    const TokAddr<AFF> tok(a, b, ntok);                           // token addressing (wkv6_scan.h): AFF = no per-tensor reversal map
    const int C_ = a.C;
#ifdef WKV6_STAMP
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
#endif
#ifdef WKV6_DEBUGBUF
    unsigned long long clk0 = 0, rtc0 = 0, clk1 = 0, rtc1 = 0;
    WKV6_CLK(clk0, rtc0);
#endif
    if constexpr (CLK) { if (hwid == 0) clock_stamp(a, slot, 0); }

    if (producer) {
        // ================================ producer: operands of block wv =================================
        const int c4 = lane & 15, tq = lane >> 4;                    // 4 channels x 4 tokens per lane
        float uu[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.use_u) io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.u) + h * HEAD + 4 * c4, uu);
        uint2 (&pr)[4] = raw.pr, (&pk)[4] = raw.pk, (&pv)[4] = raw.pv, (&pw)[4] = raw.pw;
        float4 (&pe)[4] = raw.pe;
        float dtot[4] = {0.f, 0.f, 0.f, 0.f};                            // log2-decay summed over this wave's blocks (a.dsum)
        // buffer resources over this row's first ntok tokens: loads of tokens past the end return 0 without a branch
        const unsigned span = ntok > 0 ? (unsigned)(ntok - 1) * a.C : 0u;     // elements up to the last token's head slice
        const rsrc_t rs_r = make_rsrc(gr_, ntok > 0 ? span * 2 + 128 : 0), rs_k = make_rsrc(gk_, ntok > 0 ? span * 2 + 128 : 0);
        const rsrc_t rs_v = make_rsrc(gv_, ntok > 0 ? span * 2 + 128 : 0);
        const rsrc_t rs_w = W_RAW ? make_rsrc(reinterpret_cast<const bf16_t*>(a.w) + base, ntok > 0 ? span * 2 + 128 : 0)
                                  : make_rsrc(reinterpret_cast<const float*>(a.w) + base, ntok > 0 ? span * 4 + 256 : 0);
        const int lp_in[4] = {tok.lane(wv * BLK + 4 * tq, 4 * c4, C_), tok.lane(wv * BLK + 4 * tq + 1, 4 * c4, C_),
                              tok.lane(wv * BLK + 4 * tq + 2, 4 * c4, C_), tok.lane(wv * BLK + 4 * tq + 3, 4 * c4, C_)};
        auto load_quad = [&](int grp, int tt) {                       // token quad tt of the wave's block: one instruction per tensor
            {
                const int pl = wv * BLK + 4 * tq + tt;
                const unsigned ir = tok.off(grp * GRP, pl, 4 * c4, C_, REV_R, lp_in[tt]), ik = tok.off(grp * GRP, pl, 4 * c4, C_, REV_K, lp_in[tt]);
                const unsigned iv = tok.off(grp * GRP, pl, 4 * c4, C_, REV_V, lp_in[tt]), iw = tok.off(grp * GRP, pl, 4 * c4, C_, REV_W, lp_in[tt]);
                if constexpr (!STATE_ONLY) pr[tt] = buf_load8(rs_r, ir * 2);
                else pr[tt] = make_uint2(0u, 0u);
                pk[tt] = buf_load8(rs_k, ik * 2);
                pv[tt] = buf_load8(rs_v, iv * 2);
                if constexpr (W_RAW) pw[tt] = buf_load8(rs_w, iw * 2);
                else pe[tt] = buf_load16f(rs_w, iw * 4);
            }
        };
        auto load_group = [&](int grp) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) load_quad(grp, tt);
        };
        // `next`: group whose loads are requested as soon as the raw registers are consumed, -1: none.  (Requested behind the whole
        // preparation they were in flight only for the barrier wait, ~1000 cycles of every group exposed; on its own the earlier request
        // gained nothing -- the consumers' exposed LDS round trips took the time over -- together with their up-front operand requests
        // 2-3 %: profiles/r04_fwd_prefetch.txt.)
        // -DWKV6_STAMP -DWKV6_STAMP_PREP: the producers' record holds the cycles of the preparation's phases instead of the loop's:
        // [unpack / lw / in-lane sums / r.u.k / V copy, next loads' issue, prefix butterfly + block factors, scale + split + stores,
        //  score tile (LDS round trip, 6 MFMAs, mask, split, store)]
#if defined(WKV6_STAMP) && defined(WKV6_STAMP_PREP)
#define WKV6_TP(n) do { unsigned long long t_; WKV6_T(t_); stamp_acc[n] += t_ - tprev; tprev = t_; } while (0)
        unsigned long long tprev = 0;
#else
#define WKV6_TP(n) do { } while (0)
#endif
        auto prep_group = [&](int grp, int buf, const int ntok, const float (&uu)[4], auto&& request_next) {
            char* const bb = smem + buf * GRP_BYTES + wv * BLK_BYTES;
            float r[4][4], k[4][4], cs[4][4];
#if defined(WKV6_STAMP) && defined(WKV6_STAMP_PREP)
            WKV6_T(tprev);
#endif
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const bool valid = grp * GRP + wv * BLK + 4 * tq + tt < ntok;
                r[tt][0] = bf_lo(pr[tt].x); r[tt][1] = bf_hi(pr[tt].x); r[tt][2] = bf_lo(pr[tt].y); r[tt][3] = bf_hi(pr[tt].y);
                k[tt][0] = bf_lo(pk[tt].x); k[tt][1] = bf_hi(pk[tt].x); k[tt][2] = bf_lo(pk[tt].y); k[tt][3] = bf_hi(pk[tt].y);
                float lw[4];
                if constexpr (W_RAW) {
                    // lw = -exp(w), in log2 units: -(log2 e) 2^{w log2 e}
                    lw[0] = -LOG2E * exp2_fast(LOG2E * bf_lo(pw[tt].x)); lw[1] = -LOG2E * exp2_fast(LOG2E * bf_hi(pw[tt].x));
                    lw[2] = -LOG2E * exp2_fast(LOG2E * bf_lo(pw[tt].y)); lw[3] = -LOG2E * exp2_fast(LOG2E * bf_hi(pw[tt].y));
                } else {
                    lw[0] = pe[tt].x; lw[1] = pe[tt].y; lw[2] = pe[tt].z; lw[3] = pe[tt].w;
                    if (a.wkind == 2) {   // the inference entry points pass the decay d = exp(-exp(w)) itself (cuda/rwkv6.cu:38)
#pragma unroll
                        for (int c = 0; c < 4; ++c) lw[c] = __builtin_amdgcn_logf(lw[c]);   // log2 d; d = 0 -> -inf -> clamped below
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) lw[c] *= LOG2E;
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float l = valid ? fmaxf(lw[c], LW_MIN2) : 0.f;      // log2 units from here on
                    cs[tt][c] = tt ? cs[tt - 1][c] + l : l;              // inclusive, within this lane's 4 tokens
                }
                if constexpr (!STATE_ONLY) {
                    // bonus coefficient of the diagonal: sum over the 64 channels = 4 in-lane x 16 lanes of the row
                    float part = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) part = fmaf(r[tt][c] * uu[c], k[tt][c], part);
                    part = row_sum16(part);
                    if (c4 == 0) *reinterpret_cast<float*>(bb + OFF_COEF + (4 * tq + tt) * 4) = part;
                }
                *reinterpret_cast<uint2*>(bb + A_V * ARR + (4 * tq + tt) * RSB + 8 * c4) = pv[tt];
            }
            WKV6_TP(0);
            // (the sixteen loads of the next group: ~2000 cycles of issue for the wave wherever they are placed -- behind the whole
            // preparation, here, or four at a time inside the loop above (+6 %): the four producers' 32 KB per group are a third of
            // what the CU's vector-memory pipe moves in a group at ~10 B per cycle, profiles/r05_fwd_prep_stamps.txt; v moved by the
            // consumer waves instead -- which wait ~1800 cycles at the group barrier -- as two 16-byte-per-lane loads a group ahead: +3 %,
            // profiles/r05_fwd_v_by_consumers.txt)
            // (the raw registers are dead here -- kept so: left to itself hipcc sinks the decay arithmetic below the requests, whose
            // destinations are the raw registers, and copies all eight of them out of the way first)
            __builtin_amdgcn_sched_barrier(0);
            request_next();
            WKV6_TP(1);
            float pre[4], c8[4], c16[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // exclusive prefix over the four token quads (= the four DPP rows) by butterfly on the vector ALU
                // (v_permlane16_swap / v_permlane32_swap, wkv6_chunk.h) instead of four dependent ds_bpermute round trips
                float a_, b_;
                rows16(cs[3][c], a_, b_);
                float pfx = (tq & 1) ? a_ : 0.f;
                halves32(a_ + b_, a_, b_);
                pfx += (tq & 2) ? a_ : 0.f;
                pre[c] = pfx;                                            // decay accumulated before this lane's tokens
                c8[c] = a_;                                              // ... before token 8 (rows 0, 1)
                c16[c] = a_ + b_;                                        // whole block
                dtot[c] += c16[c];                                       // ... and over all the blocks this wave prepares
            }
            if (tq == 0) {
                *reinterpret_cast<float4*>(bb + OFF_E8 + 16 * c4) =
                    make_float4(exp2_fast(c8[0]), exp2_fast(c8[1]), exp2_fast(c8[2]), exp2_fast(c8[3]));
                *reinterpret_cast<float4*>(bb + OFF_E16M8 + 16 * c4) =
                    make_float4(exp2_fast(c16[0] - c8[0]), exp2_fast(c16[1] - c8[1]), exp2_fast(c16[2] - c8[2]), exp2_fast(c16[3] - c8[3]));
            }
            WKV6_TP(2);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                float rh[4], kh[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float cex = tt ? pre[c] + cs[tt - 1][c] : pre[c];
                    const float cin = pre[c] + cs[tt][c];
                    rh[c] = r[tt][c] * exp2_fast(cex - c8[c]);
                    kh[c] = k[tt][c] * exp2_fast(c8[c] - cin);
                }
                char* const row = bb + (4 * tq + tt) * RSB + 8 * c4;
                uint2 hi, lo;
                if constexpr (!STATE_ONLY) {
                    split4(rh, hi, lo, spc);
                    *reinterpret_cast<uint2*>(row + A_RH * ARR) = hi; *reinterpret_cast<uint2*>(row + A_RL * ARR) = lo;
                }
                split4(kh, hi, lo, spc);
                *reinterpret_cast<uint2*>(row + A_KH * ARR) = hi; *reinterpret_cast<uint2*>(row + A_KL * ARR) = lo;
            }
            WKV6_TP(3);
            if constexpr (!STATE_ONLY) {
                // Scores of this block, once for all four consumers: sc[b][a] = sum_i Khat[b][i] Rhat[a][i] from the rows this
                // wave has just written (LDS operations of one wave execute in order), masked to b < a with the bonus
                // coefficient on the diagonal, split, and stored as the B fragment each consumer lane needs
                // (lane: column a = x, k rows b = 4g+q).
                const int x = lane & 15, g = lane >> 4;
                // (the stores above are uint2 / float typed, the fragment loads bf16x8 typed: the library is built with
                // -fno-strict-aliasing so that the loads stay below the stores; LDS operations of one wave execute in order)
                f4v sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int off = x * RSB + (32 * s + 8 * g) * 2;
                    const b8v kh = ld_b8(bb + A_KH * ARR + off), kl = ld_b8(bb + A_KL * ARR + off);
                    const b8v rh = ld_b8(bb + A_RH * ARR + off), rl = ld_b8(bb + A_RL * ARR + off);
                    sc = mfma32(kh, rh, sc);
                    sc = mfma32(kh, rl, sc);
                    sc = mfma32(kl, rh, sc);
                }
                const float cf = *reinterpret_cast<const float*>(bb + OFF_COEF + x * 4);
                float scm[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int bt = 4 * g + q;                            // key token; query token = x
                    scm[q] = bt < x ? sc[q] : (bt == x ? cf : 0.f);
                }
                uint2 sh, sl;
                split4(scm, sh, sl, spc);
                *reinterpret_cast<uint4*>(bb + OFF_SC + lane * 16) = make_uint4(sh.x, sh.y, sl.x, sl.y);   // hi (4 key tokens) | lo: the lane's 8 k-slots
            }
            WKV6_TP(4);
        };

        // (the next group's requests go out unconditionally: past the last group they lie past the end of the buffer resources and cost
        // nothing -- as a conditional they made the raw registers a merge of "loaded" and "kept", which hipcc resolved with sixteen register
        // copies and a full s_waitcnt vmcnt(0) per group; the last group's barrier is peeled off the loop for the same reason)
        // (CHAIN: the bonus vector of the call that follows, requested here, used at this call's end)
        float uu_nx[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (CHAIN) { if (nxvalid && nx_use_u) io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.u) + (nxbh % (unsigned)a.H) * HEAD + 4 * c4, uu_nx); }
        if (!(CHAIN && chained_in)) {
            if (ngrp > 0) {
                load_group(0);
                prep_group(0, pb & 1, ntok, uu, [&]() { load_group(1); });
            }
            __syncthreads();
        }
        for (int grp = 0; grp + 1 < ngrp; ++grp) {
            WKV6_T(ts0);
#ifdef WKV6_STAMP
            asm volatile("" :: "v"(pr[0].x), "v"(pk[0].x), "v"(pv[0].x), "v"(pw[0].x), "v"(pr[3].x), "v"(pk[3].x), "v"(pv[3].x),
                         "v"(pw[3].x));                                  // wait for the loads here
#endif
            WKV6_T(ts1);
            prep_group(grp + 1, (grp + 1 + pb) & 1, ntok, uu, [&]() { load_group(grp + 2); });
            WKV6_T(ts2);
            WKV6_T(ts3);
            __syncthreads();
            WKV6_T(ts4);
#ifndef WKV6_STAMP_PREP
            WKV6_ACC(0, ts1, ts0); WKV6_ACC(1, ts2, ts1); WKV6_ACC(2, ts3, ts2); WKV6_ACC(3, ts4, ts3);
#else
            WKV6_ACC(5, ts4, ts3);                            // (barrier wait)
#endif
        }
        if constexpr (CHAIN) {
            // the consumers are busy with this call's last group: the raw registers are dead, the next call's first inputs can fly
            if (nxvalid) {
                const int nxh = (int)(nxbh % (unsigned)a.H);
                fwd_request_group<W_RAW, STATE_ONLY, AFF>(a, nxb, nxh, nxntok, nxrev, 0, wv, lane, raw);
                prep_group(0, (ngrp + pb) & 1, nxntok, uu_nx, [&]() { fwd_request_group<W_RAW, STATE_ONLY, AFF>(a, nxb, nxh, nxntok, nxrev, 1, wv, lane, raw); });
            }
        }
        if (ngrp > 0) __syncthreads();                        // the last group is consumed
        if (a.dsum && tq == 0 && part == 0)
            *reinterpret_cast<float4*>(a.dsum + ((long)(b * a.H + h) * 4 + wv) * HEAD + 4 * c4) = make_float4(dtot[0], dtot[1], dtot[2], dtot[3]);
    } else {
        // ============================== consumer: value columns [16wv, 16wv+16) =========================
        // lane (x = lane&15, g = lane>>4) holds S[i = tile_ch(it) + 8g + q][j = 16wv + x] in St[it][q]
        const int x = lane & 15, g = lane >> 4;
        f4v St[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            float t4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.s0) {
                const long so_ = (long)b * a.s0_bstride + ((long)h * HEAD + 16 * wv + x) * HEAD + tile_ch(it) + 8 * g;
                if (a.state_f32) io4<float>::load(reinterpret_cast<const float*>(a.s0) + so_, t4);
                else io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.s0) + so_, t4);
            }
            St[it] = f4v{t4[0], t4[1], t4[2], t4[3]};
            asm volatile("" : "+v"(St[it]));                       // (an entry state still in flight is waited for here, not inside the loop)
        }
        // Token contractions (K = the block's 16 tokens) run as ONE 16x16x32 MFMA per product pair: lane group g supplies k-slots 8g .. 8g+7
        // = (hi part, tokens 4g .. 4g+3 | lo part, the same tokens) of the split operand against the exact operand's four tokens twice
        // (V | V) -- the very fragments the two 16x16x16 MFMAs took, concatenated in registers: no LDS read changes.  Both MFMA shapes
        // cost 16 cycles (profiles/r05_issue_floor.md), so this halves the MFMAs of the state update and of the score product.
        int troff = (4 * g + (x >> 2)) * RSB + 8 * (x & 3);      // transposed read, natural columns (this wave's V tile)
        int trow = (4 * g + (x >> 2)) * RSB + 16 * (x & 3);      // transposed read, tile-labelled columns: + tile_tr(t)
        // (a null y -- the fused-epilogue forward of an inference call -- gets a zero-sized resource: its stores are dropped)
        const rsrc_t rs_y = make_rsrc(a.y ? gy_ : nullptr, (!STATE_ONLY && a.y && ntok > 0) ? (unsigned)(ntok - 1) * a.C * 2u + 128u : 0u);
        const unsigned nst = ((unsigned)a.T + CKPT_TOK - 1) / CKPT_TOK;         // checkpoint slots of this (batch, head): 16 KB each
        // (two-level scan: this batch row is segment b % S of sequence b / S; the S segments' slots are consecutive, which is the
        // whole sequence's ordinary checkpoint layout)
        const int segs = a.ckpt_segs > 1 ? a.ckpt_segs : 1;
        const long ck_slot0 = ((long)((b / segs) * a.H + h) * segs + b % segs) * nst;
        const rsrc_t rs_ck = make_rsrc(a.ckpt ? a.ckpt + ck_slot0 * (HEAD * HEAD) : nullptr, a.ckpt ? nst * 16384u : 0u);
        // ACC (second half of wkv6_bi): the addends of a whole group are requested a group ahead, as raw bits, into a second register
        // set (handed over at the end of the group).  They used to be requested one block ahead through a two-way branch (fp32 side
        // buffer / bf16 output) whose conversion made hipcc wait for each load right behind its issue: a full HBM latency per block
        // (profiles/r04_bi_acc_prefetch.txt).  The block loop is unrolled for this instantiation: static register indices.
        // (wkv6_bi's halves go through buffer resources over the row's first ntok tokens like the plain path: tokens past the end
        // read zero / are dropped by the hardware -- no per-lane predicates, no 64-bit address arithmetic)
        // (one launch for both halves, chunk_fwd_bi_kernel: the fp32 side buffer is this workgroup slot's own scratch, [T][64] with a
        // token stride of 64)
        const unsigned ystr = a.side_compact ? (unsigned)HEAD : (unsigned)a.C;
        const unsigned ych = a.side_compact ? 16u * (unsigned)cidx : 16u * (unsigned)wv;             // this consumer's channels in a side row
        const rsrc_t rs_yf = make_rsrc(a.y_f32 ? a.y_f32 + (a.side_compact ? (long)sslot * a.T * HEAD : base) : nullptr,
                                       (a.y_f32 && !STATE_ONLY && ntok > 0) ? (unsigned)(ntok - 1) * ystr * 4u + 256u : 0u);
        // lane parts of this consumer's result tile (token x of a block, channels 16 wv + 4 g ..) in y and in the fp32 side buffer
        const int lp_y = tok.lane(x, 16 * wv + 4 * g, C_), lp_ys = tok.lane(x, (int)ych + 4 * g, (int)ystr);
        [[maybe_unused]] uint4 acc_cur[NBLK] = {}, acc_nxt[NBLK] = {};
        auto acc_request = [&](int grp_, uint4 (&dst)[NBLK]) {
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
                const unsigned idx = tok.off(grp_ * GRP + blk * BLK, x, 16 * wv + 4 * g, C_, REV_Y, lp_y);
                if (a.y_f32) {
                    const float4 t = buf_load16f(rs_yf, tok.off(grp_ * GRP + blk * BLK, x, (int)ych + 4 * g, (int)ystr, REV_Y, lp_ys) * 4u);
                    dst[blk] = make_uint4(__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z), __float_as_uint(t.w));
                } else {
                    const uint2 t = buf_load8(rs_y, idx * 2u);
                    dst[blk] = make_uint4(t.x, t.y, 0u, 0u);
                }
            }
        };
        auto acc_add = [&](const uint4& r, float (&o)[4]) {
            if (a.y_f32) { o[0] += __uint_as_float(r.x); o[1] += __uint_as_float(r.y); o[2] += __uint_as_float(r.z); o[3] += __uint_as_float(r.w); }
            else { o[0] += bf_lo(r.x); o[1] += bf_hi(r.x); o[2] += bf_lo(r.y); o[3] += bf_hi(r.y); }
        };
        if constexpr (ACC) {
            acc_request(0, acc_cur);
            // (waited for here, not at the first use inside the loop -- where the wait, sized for the loop's first entry, would also
            //  cover the next group's requests every time round: wkv6_chunk_bwd12k.hip has the same note)
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) asm volatile("" : "+v"(acc_cur[blk].x), "+v"(acc_cur[blk].y), "+v"(acc_cur[blk].z), "+v"(acc_cur[blk].w));
        }
        // GN epilogue state: this group's y (bf16-rounded) and gate per block, the channel's affine parameters
        float gn_y[NBLK][4], gn_ga[4] = {1.f, 1.f, 1.f, 1.f}, gn_be[4] = {0.f, 0.f, 0.f, 0.f};
        uint2 gn_g[NBLK];
        unsigned gn_off[NBLK];
        char* const gn_stat = smem + 2 * GRP_BYTES;
        char* const ckx = smem + 2 * GRP_BYTES + (GN ? 4096 : 0) + cidx * 4096;   // this consumer's checkpoint transposition buffer
        // staged y rows (see YRS): consumer hardware wave c = hwid of the workgroup's ncw = 4 (2 in split mode) owns bytes 32 c .. + 31
        // of a row piece of 32 ncw bytes and, at the flush, rows (64 / ncw) c .. of the group
        char* const ys = smem + 2 * GRP_BYTES + (GN ? 4096 : 0) + CKX_BYTES;
        const int ylsh = a.split ? 2 : 3;                          // log2(lanes per staged row piece: 16-byte chunks)
        auto stage_y = [&](int grp_, int blk, uint2 yb) {
            *reinterpret_cast<uint2*>(ys + (grp_ & 1) * YS_BYTES + (BLK * blk + x) * YRS + 32 * cidx + 8 * g) = yb;
        };
        const int fl_chunk = lane & ((1 << ylsh) - 1);
        const int fl_row[2] = {(GRP >> (ylsh - 1)) * cidx + (lane >> ylsh), (GRP >> (ylsh - 1)) * cidx + (64 >> ylsh) + (lane >> ylsh)};
        const int lp_fl[2] = {tok.lane(fl_row[0], 32 * part + 8 * fl_chunk, C_), tok.lane(fl_row[1], 32 * part + 8 * fl_chunk, C_)};
        auto flush_y = [&](int grp_) {     // after the barrier that closed group grp_: every consumer's pieces of it are in the image
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = fl_row[i], chunk = fl_chunk;
                const uint4 v = *reinterpret_cast<const uint4*>(ys + (grp_ & 1) * YS_BYTES + row * YRS + 16 * chunk);
                const unsigned off = tok.off(grp_ * GRP, row, 32 * part + 8 * chunk, C_, REV_Y, lp_fl[i]) * 2u;
                buf_store16(rs_y, off, v);                         // (tokens past the end: dropped by the bounds check)
            }
        };
        const unsigned gn_bytes = (GN && ntok > 0) ? (unsigned)(ntok - 1) * a.C * 2u + 128u : 0u;
        const rsrc_t rs_gate = make_rsrc(GN ? reinterpret_cast<const bf16_t*>(a.gn_gate) + base : nullptr, gn_bytes);
        const rsrc_t rs_out = make_rsrc(GN ? reinterpret_cast<bf16_t*>(a.gn_out) + base : nullptr, gn_bytes);
        if constexpr (GN) {
            io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.gn_gamma) + h * HEAD + 16 * wv + 4 * g, gn_ga);
            io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.gn_beta) + h * HEAD + 16 * wv + 4 * g, gn_be);
        }
        // normalise, gate and store the blocks of group `grp` (after the barrier that published every wave's partial sums)
        auto gn_finish = [&](int grp) {
            const char* const st = gn_stat + (grp & 1) * 2048;
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const float2 t = *reinterpret_cast<const float2*>(st + ((ww * NBLK + blk) * 16 + x) * 8);
                    s1 += t.x; s2 += t.y;
                }
                const float mean = s1 * (1.f / 64.f);
                const float var = fmaxf(fmaf(-mean, mean, s2 * (1.f / 64.f)), 0.f);      // biased variance, as nn.GroupNorm
                const float rstd = rsqrtf(var + a.gn_eps);
                const float gt[4] = {bf_lo(gn_g[blk].x), bf_hi(gn_g[blk].x), bf_lo(gn_g[blk].y), bf_hi(gn_g[blk].y)};
                float o[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) o[qq] = fmaf((gn_y[blk][qq] - mean) * rstd, gn_ga[qq], gn_be[qq]) * gt[qq];
                buf_store8(rs_out, gn_off[blk], make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])));
                const int p = grp * GRP + blk * BLK + x;
                if (a.gn_stats && wv == 0 && g == 0 && p < ntok) {
                    float* const sp = a.gn_stats + (((long)b * a.T + tok.token(p, REV_Y)) * a.H + h) * 2;
                    sp[0] = mean; sp[1] = rstd;
                }
            }
        };
        if (!(CHAIN && chained_in)) __syncthreads();            // (a call that was prepared for: the barriers that closed the call before it)
        constexpr bool STAGE_Y = !STATE_ONLY && !GN;              // (bf16 y of this launch goes through the staged rows; a y_f32 first half stores directly)
        const bool staged = STAGE_Y && (ACC || !a.y_f32);
        for (int grp = 0; grp < ngrp; ++grp) {
            WKV6_T(ts0);
            // The four consumers leave the group barrier together, burst their block operands' LDS reads at the same moments -- and then wait
            // ~1300 cycles for the producers at the next barrier.  Consumer w starts a group w x 192 cycles late (256 when this was measured): the bursts no longer
            // collide (nor with the producers' stores), the delay comes out of the barrier wait.  Same box: forward -2.7 % (0.2198 ->
            // 0.2137 ms; 128 or 448 cycles per wave -0.8 / -1.5 %; staggering the producers too +1.2 ... 2 %: profiles/r05_stagger.txt);
            // 192 cycles per wave since the end of round 5 (another -3.4 % after the producers' path had been shortened).
            for (int i_ = 0; i_ < cidx; ++i_) __builtin_amdgcn_s_sleep(CONSUMER_STAGGER);   // (cidx: the consumer's index within its workgroup)
            if (staged && grp > 0) flush_y(grp - 1);
            if constexpr (ACC) acc_request(grp + 1, acc_nxt);     // (past the last group: past the end of the resource, reads zero)

            // Rolled (runtime trip count): a fully unrolled 4-block body is no faster.  -DWKV6_FWD_UNROLL builds the unrolled
            // body for tools/check_unrolled_fwd.sh (the wrong y that build once produced was the mixed-shape
            // MFMA accumulation hazard, not a reordered LDS read).
#ifdef WKV6_FWD_UNROLL
            constexpr bool unrolled = true;
#else
            constexpr bool unrolled = GN || ACC;                  // the GN epilogue and the ACC addends keep per-block values in registers: static indices
#endif
            // blocks past the end are neutral (zero-filled operands) in the unrolled form
            const int nb = unrolled ? NBLK : min(NBLK, (ntok - grp * GRP + BLK - 1) / BLK);
            constexpr int unroll_by = unrolled ? NBLK : 1;
#pragma unroll unroll_by
            for (int blk = 0; blk < nb; ++blk) {
                const char* const bb = smem + ((grp + pb) & 1) * GRP_BYTES + blk * BLK_BYTES;
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                v4u ckd[4];
                int ck_off = -1;                                  // >= 0: a row-order checkpoint waits in ckd for its stores
                if (a.ckpt && blk == 0) {   // state at every group (= CKPT_TOK-token) boundary, for the backward kernel (slots past T: dropped)
                    const unsigned st = (unsigned)grp;
                    // The backward's row waves want key row i on the lane and four consecutive value columns per register quad;
                    // here a lane holds one value column j and four consecutive key rows per quad.  The 64 x 16 slice of this
                    // wave goes through 4 KB of LDS: 16 scalar writes, 4 float4 reads, both conflict-free with the float index
                    //     jl0 | jl1 | jl2^i0 | jl3^i1 | i3 | i2 | i0 | i1 | i4 | i5        (i = key row, jl = j - 16 wv)
                    // (the 32 lanes of a write group differ in jl and i3, the 16 lanes of a float4 read group in i0..i3 with
                    // jl3 = i2 ^ i3 ^ const: both land on distinct banks), then leaves as four coalesced 16-byte stores -- per
                    // (row wave, column tile, g) 16 lanes cover 256 contiguous bytes of the backward's register image.  The reads are
                    // issued here, the stores at the end of the block: their LDS latency runs under the block's MFMAs.
#pragma unroll
                    for (int it = 0; it < 4; ++it)
#pragma unroll
                        for (int q = 0; q < 4; ++q)     // S[i = tile_ch(it) + 8g + q][jl = x]
                            *reinterpret_cast<float*>(ckx + ((x ^ (q << 2)) + 16 * (g & 1) + 256 * (g >> 1) + 64 * (q & 1) + 128 * (q >> 1)
                                                             + 32 * (it & 1) + 512 * (it >> 1)) * 4) = St[it][q];
#pragma unroll
                    for (int wb = 0; wb < 4; ++wb)      // lane (x, g) -> S[i = 16 wb + x][j = 16 wv + 4 (g >> 1) + 8 (g & 1) + 0..3]
                        ckd[wb] = *reinterpret_cast<const v4u*>(ckx + (4 * ((g >> 1) ^ (x & 1)) + 8 * ((g & 1) ^ ((x >> 1) & 1)) + 16 * ((x >> 3) & 1)
                                                                       + 32 * ((x >> 2) & 1) + 64 * (x & 1) + 128 * ((x >> 1) & 1) + 256 * wb) * 4);
                    ck_off = (int)(st * 16384u + (((2 * (wv >> 1) + (g >> 1)) * 64 + 16 * (2 * (wv & 1) + (g & 1)) + x) * 16));
                }
                // value fragment: lane holds V[4g + e][16wv + x], e = 0..3  (A operand of (2), B operand of (4))
                // ALL of the block's LDS operands are requested here, in one go, and the scheduler may not sink them: left to itself hipcc
                // issues each read right in front of its use, and with two waves on a SIMD every one of the ~8 round trips of a block
                // was exposed (the consumers' 5.7 k cycles per group were LDS latency, not issue: profiles/r04_fwd_prefetch.txt).
                const s4v vf4 = tr_read(bb + A_V * ARR + troff + 32 * wv);
                const b8v vf = __builtin_bit_cast(b8v, __builtin_shufflevector(vf4, vf4, 0, 1, 2, 3, 4, 5, 6, 7));     // (V | V)
                [[maybe_unused]] uint4 scp = {};
                [[maybe_unused]] float4 pm0[2] = {}, pm1[2] = {};
                [[maybe_unused]] b8v pzh[2] = {}, pzl[2] = {};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    pm0[s] = *reinterpret_cast<const float4*>(bb + OFF_E8 + (32 * s + 8 * g) * 4);
                    pm1[s] = *reinterpret_cast<const float4*>(bb + OFF_E8 + (32 * s + 8 * g + 4) * 4);
                }
                if constexpr (!STATE_ONLY) {
                    scp = *reinterpret_cast<const uint4*>(bb + OFF_SC + lane * 16);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int off = x * RSB + (32 * s + 8 * g) * 2;
                        pzh[s] = ld_b8(bb + A_RH * ARR + off);
                        pzl[s] = ld_b8(bb + A_RL * ARR + off);
                    }
                }
                b8v pk8[4];                                         // Khat (hi | lo along K) of tile `it`
                float4 pdm[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    pk8[it] = __builtin_bit_cast(b8v, __builtin_shufflevector(tr_read(bb + A_KH * ARR + trow + tile_tr(it)),
                                                                              tr_read(bb + A_KL * ARR + trow + tile_tr(it)), 0, 1, 2, 3, 4, 5, 6, 7));
                    pdm[it] = *reinterpret_cast<const float4*>(bb + OFF_E16M8 + (tile_ch(it) + 8 * g) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                // T = E8 (.) S, tile by tile (tiles 2s, 2s + 1 = k-slots e = 0..3, 4..7 of k-step s): the operand of (3) -- and, since
                // E16 = E16m8 E8, the state one block on is E16m8 (.) (T + Khat^T V): T goes into the MFMA of (4) as its accumulator and
                // the update is ONE multiplication per element (round 5; it was E16 (.) S + E16m8 (.) (Khat^T V): a multiplication and an fma,
                // and a second decay vector to read)
                f4v Tt[4];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float4 m0 = pm0[s], m1 = pm1[s];
                    Tt[2 * s] = f4v{St[2 * s][0] * m0.x, St[2 * s][1] * m0.y, St[2 * s][2] * m0.z, St[2 * s][3] * m0.w};
                    Tt[2 * s + 1] = f4v{St[2 * s + 1][0] * m1.x, St[2 * s + 1][1] * m1.y, St[2 * s + 1][2] * m1.z, St[2 * s + 1][3] * m1.w};
                }
                if constexpr (!STATE_ONLY) {
                    // (1) masked transposed scores, prepared by the producer of this block: hi | lo along K
                    // (2) y^T[j][a] = sum_b V[b][j] sc[b][a]: (V | V) x (sc_hi ; sc_lo), one MFMA.  Every MFMA of this wave is a
                    // 16x16x32 now, so one accumulator chain serves (2) and (3) (a 16x16x16 MFMA taking a 16x16x32 result as SrcC
                    // fewer than 5 wait states later reads stale registers on gfx950: DESIGN.md section 4, "mixed-shape accumulation").
                    f4v yt = {0.f, 0.f, 0.f, 0.f};
                    yt = mfma32(vf, __builtin_bit_cast(b8v, scp), yt);
                    // (3) y^T[j][a] += sum_i (E8 S)[i][j] Rhat[a][i]; k-slot (s, g, e) <-> channel 32s + 8g + e
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float t0[4] = {Tt[2 * s][0], Tt[2 * s][1], Tt[2 * s][2], Tt[2 * s][3]};
                        const float t1[4] = {Tt[2 * s + 1][0], Tt[2 * s + 1][1], Tt[2 * s + 1][2], Tt[2 * s + 1][3]};
                        uint2 h0, l0, h1, l1;
                        split4(t0, h0, l0, spc);
                        split4(t1, h1, l1, spc);
                        const b8v s_hi = __builtin_bit_cast(b8v, make_uint4(h0.x, h0.y, h1.x, h1.y));
                        const b8v s_lo = __builtin_bit_cast(b8v, make_uint4(l0.x, l0.y, l1.x, l1.y));
                        const b8v zh = pzh[s], zl = pzl[s];
                        yt = mfma32(s_hi, zh, yt);
                        yt = mfma32(s_hi, zl, yt);
                        yt = mfma32(s_lo, zh, yt);
                    }
                    {   // store: lane holds y[token x][j = 16wv + 4g + q]
                        const int p = grp * GRP + blk * BLK + x;
                        float o[4] = {yt[0], yt[1], yt[2], yt[3]};
                        if (!ACC && !a.y_f32) {                          // plain store: tokens past the end are dropped by the hardware
                            const unsigned off = tok.off(grp * GRP + blk * BLK, x, 16 * wv + 4 * g, C_, REV_Y, lp_y) * 2u;
                            const uint2 yb = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
                            if constexpr (GN) buf_store8(rs_y, off, yb);   // (a null y: zero-sized resource, the store is dropped)
                            else stage_y(grp, blk, yb);
                            if constexpr (GN) {
                                const float yr[4] = {bf_lo(yb.x), bf_hi(yb.x), bf_lo(yb.y), bf_hi(yb.y)};
                                gn_off[blk] = off;
                                gn_g[blk] = buf_load8(rs_gate, off);
                                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                                for (int qq = 0; qq < 4; ++qq) { gn_y[blk][qq] = yr[qq]; s1 += yr[qq]; s2 = fmaf(yr[qq], yr[qq], s2); }
                                const float two[4] = {s1, s2, 0.f, 0.f};
                                const float red = col_reduce(two);       // row 0: sum over this wave's 16 channels, row 2: sum of squares
                                if ((g & 1) == 0)
                                    *reinterpret_cast<float*>(gn_stat + (grp & 1) * 2048 + ((wv * NBLK + blk) * 16 + x) * 8 + (g >> 1) * 4) = red;
                            }
                        } else {
                            if constexpr (ACC) acc_add(acc_cur[blk], o);          // requested a group ago
                            if (!ACC && a.y_f32) buf_store16f(rs_yf, tok.off(grp * GRP + blk * BLK, x, (int)ych + 4 * g, (int)ystr, REV_Y, lp_ys) * 4u, o);
                            else stage_y(grp, blk, make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])));
                        }
                    }
                }
                // (4) S[it] <- E16m8 (.) (T[it] + Khat^T V)   (= E16 (.) S[it] + E16m8 (.) (Khat^T V))
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const f4v o = mfma32(pk8[it], vf, Tt[it]);       // T + (Khat_hi | Khat_lo)^T (V ; V)
                    const float4 dm = pdm[it];
                    St[it][0] = dm.x * o[0];
                    St[it][1] = dm.y * o[1];
                    St[it][2] = dm.z * o[2];
                    St[it][3] = dm.w * o[3];
                }
                if (ck_off >= 0) {
#pragma unroll
                    for (int wb = 0; wb < 4; ++wb)
                        __builtin_amdgcn_raw_buffer_store_b128(ckd[wb], rs_ck, ck_off + wb * 4096, 0, 2 /* slc: streaming */);
                }
            }
            if constexpr (ACC) {
#pragma unroll
                for (int blk = 0; blk < NBLK; ++blk) acc_cur[blk] = acc_nxt[blk];
            }
            WKV6_T(ts1);
            __syncthreads();
            WKV6_T(ts2);
            WKV6_ACC(0, ts1, ts0); WKV6_ACC(1, ts2, ts1);
            if constexpr (GN) gn_finish(grp);
        }
        if (staged && ngrp > 0) flush_y(ngrp - 1);                // (the loop's last barrier closed the last group)
        if (a.s_out) {
            const long so_ = ((long)b * a.H + h) * HEAD * HEAD + (long)(16 * wv + x) * HEAD + 8 * g;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float t4[4] = {St[it][0], St[it][1], St[it][2], St[it][3]};
                if (a.state_f32) io4<float>::store(reinterpret_cast<float*>(a.s_out) + so_ + tile_ch(it), t4);
                else io4<bf16_t>::store(reinterpret_cast<bf16_t*>(a.s_out) + so_ + tile_ch(it), t4);
            }
        }
        if constexpr (CHAIN) {
            // a consumer wave carries nothing from call to call: "define" the carried registers here (no instruction), or the compiler keeps
            // them alive -- 32 to 48 registers -- through all of the consumer path for the producer waves' sake
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                asm volatile("" : "=v"(raw.pr[tt].x), "=v"(raw.pr[tt].y), "=v"(raw.pk[tt].x), "=v"(raw.pk[tt].y),
                                  "=v"(raw.pv[tt].x), "=v"(raw.pv[tt].y), "=v"(raw.pw[tt].x), "=v"(raw.pw[tt].y));
                asm volatile("" : "=v"(raw.pe[tt].x), "=v"(raw.pe[tt].y), "=v"(raw.pe[tt].z), "=v"(raw.pe[tt].w));
            }
        }
    }
#ifdef WKV6_DEBUGBUF
    WKV6_CLK(clk1, rtc1);
    if (a.aux && lane == 0) {
        unsigned long long* const d = reinterpret_cast<unsigned long long*>(a.aux) + ((long)bh * 16 + wid) * 8;
#ifdef WKV6_STAMP
        for (int i = 0; i < 6; ++i) d[i] = stamp_acc[i];
#else
        // which SIMD the wave ran on: HW_REG_HW_ID (wave_id 3:0, simd_id 5:4, pipe_id 7:6, cu_id 11:8, sh_id 12, se_id 15:13) | hardware wave << 32
        d[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)hwid << 32);
#endif
        d[6] = clk1 - clk0;
        d[7] = rtc1 - rtc0;
    }
#endif
    if constexpr (CLK) { if (hwid == 0) clock_stamp(a, slot, 1); }
    if (!STATE_ONLY && !ACC && a.zero_tail) {
        const float z[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = ntok + (tid >> 4); part == 0 && t < a.T; t += (int)(blockDim.x >> 4))
            io4<bf16_t>::store(gy_ + (unsigned)(t * a.C + 4 * (tid & 15)), z);
    }
}

template <bool W_RAW, bool STATE_ONLY, bool ACC, bool GN, bool AFF>
__global__ __launch_bounds__(512) void chunk_fwd_kernel(const ScanArgs a)
{
    FwdRaw raw;
    chunk_fwd_body<W_RAW, STATE_ONLY, ACC, GN, AFF, !STATE_ONLY && !ACC && !GN>(a, blockIdx.x, 0u, raw);
}

// Two problems of the same shape in one grid of 2 B H workgroups: slots [0, B H) serve a0, the rest a1 (src/model_bi.py:331-348,
// src/model_ext.py:421-437: the forward-direction and the reversed-direction operator calls of a bidirectional time-mix layer).
template <bool W_RAW>
__global__ __launch_bounds__(512) void chunk_fwd_pair_kernel(const ScanArgs a0, const ScanArgs a1)
{
    const unsigned n = (unsigned)(a0.B * a0.H);
    const bool second = blockIdx.x >= n;                      // workgroup-uniform: the argument block is read through one of two
    FwdRaw raw;
    chunk_fwd_body<W_RAW, false, false, false, false>(second ? a1 : a0, second ? blockIdx.x - n : blockIdx.x, 0u, raw);   // kernarg addresses; per-tensor reversal maps: general addressing
}

// Both halves of wkv6_bi in one persistent launch (cuda/wkv6_bi_cuda.cu:363-368 is one launch too): workgroup slot s walks the rows
// s, s + slots, ... of the length-ordered batch x head list; per row the forward-direction scan leaves y in the slot's fp32 scratch
// ([T][64], 128 KB at T = 512), the reversed-direction scan adds it and rounds once.
// (one argument block: the reversed-direction problem differs from the forward-direction one in five fields)
template <bool W_RAW>
__global__ __launch_bounds__(512) void chunk_fwd_bi_kernel(const ScanArgs a1_, float* const ckpt2)
{
    // (a1_ is read here for the row walk only; the bodies get per-call copies of the argument block: load_kernargs, wkv6_chunk.h)
    const unsigned n = (unsigned)(a1_.B * a1_.H);
    const int H_ = a1_.H, T_ = a1_.T, use_u_ = a1_.use_u;
    const int* const order_ = a1_.order;
    const int* const lens_ = a1_.lens;
    // the rows are ordered by decreasing length (a.order): slot j takes row j of the first round of gridDim.x rows, row gridDim.x - 1 - j
    // of the second, ... (boustrophedon), so that every slot gets long and short rows alike -- in plain round-robin order slot 0 would
    // take the longest row of every round and the last slot the shortest (+-12 % of the mean at BASELINE configs[2])
    // (the producers' raw input registers outlive a call: each call's producers request the first inputs of the call that follows --
    // FwdRaw above.  The producer waves write no global memory and take no part in the fences between the halves, which would make them
    // wait for those requests.)
    FwdRaw raw;
    const bool producer_wave = fwd_hw_wave_produces((int)(threadIdx.x >> 6));
    // a row's batch index and length: a.order[row / H] -> a.lens[b], two dependent memory round trips that used to open every call; looked up
    // one row ahead here
    const auto row_of = [&](unsigned it) { return it * gridDim.x + ((it & 1) ? gridDim.x - 1 - blockIdx.x : blockIdx.x); };
    const auto lookup = [&](unsigned row, int& b, int& ntok) {
        b = 0; ntok = 0;
        if (row < n) {
            b = order_ ? order_[row / H_] : (int)(row / H_);
            ntok = lens_ ? min(max(lens_[b], 0), T_) : T_;
        }
    };
    int b_cur, ntok_cur;
    lookup(row_of(0), b_cur, ntok_cur);
    int pb = 0;                                                     // operand buffer of a call's group 0 (see chunk_fwd_body, CHAIN)
    bool chained = false;                                           // the call that starts has been prepared for by the one before it
    for (unsigned it = 0; it * gridDim.x < n; ++it) {
        const unsigned row = row_of(it);
        if (row >= n) continue;                                     // (the last round may be short; workgroup-uniform)
        const unsigned row_nx = row_of(it + 1);
        int b_nx, ntok_nx;
        lookup(row_nx, b_nx, ntok_nx);
        const int ngrp_cur = (ntok_cur + GRP - 1) / GRP;
        {
            ScanArgs a1;
            load_kernargs(a1);
            chunk_fwd_body<W_RAW, false, false, false, true, false, true>(a1, row, blockIdx.x, raw,
                                                                          FwdChain{b_cur, ntok_cur, pb, chained, true, row, b_cur, ntok_cur, true, false});
        }
        pb = (pb + ngrp_cur) & 1;
        if (!producer_wave) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (!producer_wave) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        {
            ScanArgs a2;
            load_kernargs(a2);
            a2.reverse = 1; a2.use_u = 0; a2.accumulate = 1; a2.zero_tail = 0; a2.ckpt = ckpt2;
            chunk_fwd_body<W_RAW, false, true, false, true, false, true>(a2, row, blockIdx.x, raw,
                                                                         FwdChain{b_cur, ntok_cur, pb, true, row_nx < n, row_nx, b_nx, ntok_nx, false, use_u_ != 0});
        }
        pb = (pb + ngrp_cur) & 1;
        chained = true;
        __syncthreads();
        b_cur = b_nx; ntok_cur = ntok_nx;
    }
}

template <bool W_RAW, bool STATE_ONLY, bool ACC, bool GN, bool AFF> hipError_t launch_fwd_variant2(const ScanArgs& a, hipStream_t st)
{
    constexpr size_t lds = 2 * (size_t)GRP_BYTES + (GN ? 4096 : 0) + CKX_BYTES + 2 * YS_BYTES;
    static LdsAttrOnce attr;                   // per instantiation and device
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(chunk_fwd_kernel<W_RAW, STATE_ONLY, ACC, GN, AFF>), lds)) return e;
    if (a.split) hipLaunchKernelGGL((chunk_fwd_kernel<W_RAW, STATE_ONLY, ACC, GN, AFF>), dim3(2 * a.B * a.H), dim3(384), lds, st, a);
    else hipLaunchKernelGGL((chunk_fwd_kernel<W_RAW, STATE_ONLY, ACC, GN, AFF>), dim3(a.B * a.H), dim3(512), lds, st, a);
    return hipGetLastError();
}
// per-tensor reversal maps (a.rev_n: the compositions' *_rev_ex calls and their state passes) take the general token addressing; the
// accumulating half of wkv6_bi and the GroupNorm epilogue never carry one
template <bool W_RAW, bool STATE_ONLY, bool ACC, bool GN = false> hipError_t launch_fwd_variant(const ScanArgs& a, hipStream_t st)
{
    if constexpr (!ACC && !GN) { if (a.rev_n) return launch_fwd_variant2<W_RAW, STATE_ONLY, false, false, false>(a, st); }
    else if (a.rev_n) return hipErrorNotSupported;
    return launch_fwd_variant2<W_RAW, STATE_ONLY, ACC, GN, true>(a, st);
}

}  // namespace

// per-lane byte offsets are 32-bit: bf16 tensors need (T + 64) C < 2^31 (checked by the API), the fp32 decay input half of that
static bool offsets_fit(const ScanArgs& a) { return a.wkind == 1 || ((long)a.T + 64) * a.C < (1L << 30); }

int want_split(int BH);     // wkv6_chunk_bwd12k.hip

hipError_t launch_chunk_fwd(const ScanArgs& a_, hipStream_t st)
{
    if (!offsets_fit(a_)) return hipErrorInvalidValue;
    ScanArgs a = a_;
    a.split = want_split(a.B * a.H);
    a.clk = clock_claim(0, &a.clk_slots);
#ifdef WKV6_DEBUGBUF
    a.aux = reinterpret_cast<float*>(g_stamp_buffer);
#endif
    const bool raw = a.wkind == 1;          // 0: fp32 ew = -exp(w), 1: raw w in bf16, 2: fp32 decay exp(-exp(w))
    if (a.gn_out) {                         // fused GroupNorm * gate epilogue: the four consumer waves of a head in one workgroup
        if (a.split || a.accumulate || a.y_f32 || a.zero_tail) return hipErrorNotSupported;
        return raw ? launch_fwd_variant<true, false, false, true>(a, st) : launch_fwd_variant<false, false, false, true>(a, st);
    }
    if (a.accumulate) return raw ? launch_fwd_variant<true, false, true>(a, st) : launch_fwd_variant<false, false, true>(a, st);
    return raw ? launch_fwd_variant<true, false, false>(a, st) : launch_fwd_variant<false, false, false>(a, st);
}

// Both problems of a bidirectional composition in one launch (same shape and decay kind; checkpoints optional).  Falls back to
// two launches where one (batch, head) is split over two workgroups.
hipError_t launch_chunk_fwd_pair(const ScanArgs& a0_, const ScanArgs& a1_, hipStream_t st)
{
    if (!offsets_fit(a0_) || !offsets_fit(a1_)) return hipErrorInvalidValue;
    if (a0_.B != a1_.B || a0_.T != a1_.T || a0_.C != a1_.C || a0_.H != a1_.H || a0_.wkind != a1_.wkind) return hipErrorInvalidValue;
    const auto plain = [](const ScanArgs& a) { return !a.accumulate && !a.y_f32 && !a.zero_tail && !a.gn_out && !a.dsum && !a.ckpt_segs; };
    if (!plain(a0_) || !plain(a1_)) return hipErrorNotSupported;
    if (want_split(a0_.B * a0_.H)) {
        if (hipError_t e = launch_chunk_fwd(a0_, st)) return e;
        return launch_chunk_fwd(a1_, st);
    }
    ScanArgs a0 = a0_, a1 = a1_;
    a0.split = a1.split = 0;
    constexpr size_t lds = 2 * (size_t)GRP_BYTES + CKX_BYTES + 2 * YS_BYTES;
    static LdsAttrOnce attr_raw, attr_ew;
    if (a0.wkind == 1) {
        if (hipError_t e = attr_raw.ensure(reinterpret_cast<const void*>(chunk_fwd_pair_kernel<true>), lds)) return e;
        hipLaunchKernelGGL((chunk_fwd_pair_kernel<true>), dim3(2 * a0.B * a0.H), dim3(512), lds, st, a0, a1);
    } else {
        if (hipError_t e = attr_ew.ensure(reinterpret_cast<const void*>(chunk_fwd_pair_kernel<false>), lds)) return e;
        hipLaunchKernelGGL((chunk_fwd_pair_kernel<false>), dim3(2 * a0.B * a0.H), dim3(512), lds, st, a0, a1);
    }
    return hipGetLastError();
}

int bi_slots(int BH);       // wkv6_chunk_bwd12k.hip
hipError_t launch_chunk_fwd_bi(const ScanArgs& a1_, const ScanArgs& a2_, int* slots, hipStream_t st)
{
    const int n = bi_slots(a1_.B * a1_.H);
    if (slots) *slots = n;
    if (!n || !a1_.y_f32 || a1_.gn_out || a1_.dsum || a1_.ckpt_segs) return hipErrorNotSupported;
    if (!offsets_fit(a1_)) return hipErrorInvalidValue;
    ScanArgs a1 = a1_, a2 = a2_;
    a1.split = a2.split = 0;
    a1.side_compact = a2.side_compact = 1;
#ifdef WKV6_DEBUGBUF
    a1.aux = a2.aux = reinterpret_cast<float*>(g_stamp_buffer);
#endif
    constexpr size_t lds = 2 * (size_t)GRP_BYTES + CKX_BYTES + 2 * YS_BYTES;
    static LdsAttrOnce attr_raw, attr_ew;
    if (a1.wkind == 1) {
        if (hipError_t e = attr_raw.ensure(reinterpret_cast<const void*>(chunk_fwd_bi_kernel<true>), lds)) return e;
        hipLaunchKernelGGL((chunk_fwd_bi_kernel<true>), dim3(n), dim3(512), lds, st, a1, a2.ckpt);
    } else {
        if (hipError_t e = attr_ew.ensure(reinterpret_cast<const void*>(chunk_fwd_bi_kernel<false>), lds)) return e;
        hipLaunchKernelGGL((chunk_fwd_bi_kernel<false>), dim3(n), dim3(512), lds, st, a1, a2.ckpt);
    }
    return hipGetLastError();
}

// state recurrence only, dumping the stage-entry states into a.ckpt (first half of the self-contained backward)
hipError_t launch_chunk_state_pass(const ScanArgs& a_, hipStream_t st)
{
    if (!offsets_fit(a_)) return hipErrorInvalidValue;
    ScanArgs a = a_;
    a.split = want_split(a.B * a.H);
    return a.wkind == 1 ? launch_fwd_variant<true, true, false>(a, st) : launch_fwd_variant<false, true, false>(a, st);
}

size_t chunk_ckpt_floats(int B, int T, int H)
{
    return (size_t)B * H * ((T + CKPT_TOK - 1) / CKPT_TOK) * HEAD * HEAD;
}

}  // namespace wkv6
