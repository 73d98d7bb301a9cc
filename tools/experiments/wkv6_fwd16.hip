// Chunked MFMA forward of WKV6 for gfx950 (bf16 I/O), 16-wave version.
//
// Block algebra (16-token blocks, a, b = token in block, i = key channel, j = value channel, c_a = exclusive cumulative
// log-decay inside the block, lw = -exp(w) clamped to >= LW_MIN, reference point rho = c_8 of the block):
//     Rhat_a = r_a . e^{c_a - rho},   Khat_b = k_b . e^{rho - c_{b+1}}       (every exponent spans at most 8 tokens)
//     scores   A[a][b] = Rhat_a . Khat_b  (b < a),     A[a][a] = sum_i r_a u k_a
//     output   y_a = A[a][:] V + Rhat_a^T Sref_n
//     state    Sref_{n+1} = alpha_n (.) (Sref_n + Khat^T V),      alpha_n = e^{(c_16 - rho)_n + rho_{n+1}}
// where Sref_n = e^{rho_n} (.) S_n is the state of cuda/wkv6_cuda.cu:44-57 "as seen from the reference point" of block
// n.  Carrying the state at the reference points makes the update a single multiply per element (the decay between two
// consecutive reference points) with the Khat^T V product accumulated by the MFMA straight into the state registers,
// and lets the output product use the state as it stands.  alpha_n needs the decay of the first 8 tokens of block n+1:
// the wave that prepares block n loads those 8 extra rows of w itself, so blocks stay independent.
//
// Every fp32 MFMA operand is split into bf16 hi + lo (x = hi + lo + O(2^-16 x)); products are hi*hi + hi*lo + lo*hi
// with fp32 accumulation; r, k, v are exact in bf16.
//
// Work split: one 1024-thread workgroup (16 wave64, four per SIMD) per (batch, head); 64-token groups, LDS image
// double-buffered, one barrier per group.
//   producer wave (block pb, channel half ph), waves 8..15: decays, cumulative sums, scaling, hi/lo split of its half
//            of block pb of group g+1 (lane = 4 channels x 2 tokens), the block's masked scores restricted to its
//            channel half (handed over as the consumers' MFMA fragment), alpha; global loads of group g+2 in flight.
//   consumer wave (value tile w, key half kh), waves 0..7: owns S[i in half kh][j in 16w..16w+15] as two 16x16 C-layout
//            tiles (8 VGPRs).  Per block: partial output over its key half (3 + 2 MFMAs: state term and the scores of
//            its half), state update (4 MFMAs + 8 multiplies).  The two halves of an output tile are added through
//            LDS one group later by the wave that owns the block (even blocks: kh = 0, odd: kh = 1), which also
//            rounds and stores y.
// Checkpoints for the backward (a.ckpt): Sref at every 32-token stage entry, fp32, [wave w][tile][lane][4].
#include <type_traits>
#include "wkv6_chunk.h"

namespace wkv6 {
namespace {

using namespace chunk;

enum { F_RH = 0, F_RL, F_KH, F_KL, F_V, F_NARR };              // bf16 [16][RSB/2] each
constexpr int FOFF_ALPHA = F_NARR * ARR;                       // float[64]  alpha_n
constexpr int FOFF_E8 = FOFF_ALPHA + 256;                      // float[64]  e^{rho_n}   (initial state -> Sref_0)
constexpr int FOFF_SC = FOFF_E8 + 256;                         // uint4 [2][64]  masked scores^T of each channel half, bf16x4 hi | lo
constexpr int FOFF_COEF = FOFF_SC + 2048;                      // float [2][16]  per-half sum_i r u k (producer-internal)
constexpr int FOFF_BETA = FOFF_COEF + 128;                     // float[64]  second factor of alpha (only when FLAG is set)
constexpr int FOFF_FLAG = FOFF_BETA + 256;                     // int [2]    per channel half: alpha is stored as two factors
constexpr int FBLK_BYTES = FOFF_FLAG + 16;
constexpr int FGRP_BYTES = NBLK * FBLK_BYTES;
constexpr int PART_OFF = 2 * FGRP_BYTES;                       // float4 [2 buffers][4 blocks][4 tiles][64 lanes]
constexpr int PART_BYTES = NBLK * 4 * 64 * 16;
constexpr int FWD16_LDS = PART_OFF + 2 * PART_BYTES;
static_assert(FWD16_LDS <= 160 * 1024, "LDS budget");
constexpr int DPP_ROW_HALF_MIRROR = 0x141;

// Diagnostic build (-DWKV6_STAMP, tools/ablate.sh): every wave accumulates s_memtime cycles per phase into a debug buffer
// set through wkv6_set_debug_buffer(); no stamp executes in the normal build.
#ifdef WKV6_STAMP
#define WKV6_T(var) do { __builtin_amdgcn_sched_barrier(0); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WKV6_ACC(k, t1, t0) stamp_acc[k] += (t1) - (t0)
#else
#define WKV6_T(var) do { } while (0)
#define WKV6_ACC(k, t1, t0) do { } while (0)
#endif

// Timing-only ablation switches (tools/ablate.sh builds one library per switch; results are wrong by construction).
#ifdef WKV6_ABL_NOEXP
#define WKV6_EXPF(x) ((x) * 0.25f + 1.0f)
#else
#define WKV6_EXPF(x) __expf(x)
#endif
#ifdef WKV6_ABL_NOSHFL
#define WKV6_SHFL_UP(x, n) (x)
#define WKV6_SHFL(x, n) (x)
#define WKV6_SHFL_XOR(x, n) (x)
#else
#define WKV6_SHFL_UP(x, n) __shfl_up(x, n)
#define WKV6_SHFL(x, n) __shfl(x, n)
#define WKV6_SHFL_XOR(x, n) __shfl_xor(x, n)
#endif

template <bool W_RAW, bool STATE_ONLY, bool ACC>
__global__ __launch_bounds__(1024) void fwd16_kernel(const ScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wid >= 8;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const long base = (long)b * a.T * a.C + (long)h * HEAD;   // per-lane offsets stay 32-bit (T*C < 2^31, checked by the API)
    const bf16_t* const gr_ = reinterpret_cast<const bf16_t*>(a.r) + base;
    const bf16_t* const gk_ = reinterpret_cast<const bf16_t*>(a.k) + base;
    const bf16_t* const gv_ = reinterpret_cast<const bf16_t*>(a.v) + base;
    bf16_t* const gy_ = reinterpret_cast<bf16_t*>(a.y) + base;
    int ntok = a.T;
    if (a.lens) ntok = min(max(a.lens[b], 0), a.T);
    const int ngrp = (ntok + GRP - 1) / GRP;
#ifdef WKV6_STAMP
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
#endif

    if (producer) {
        // ================= producer: channels ch0..ch0+3, tokens 2tq, 2tq+1 of block pb ==================
        const int pb = (wid - 8) & 3, ph = (wid - 8) >> 2;
#ifndef WKV6_ABL_NOPRIO
        // The producers carry the longest instruction stream of a group and, being the youngest waves of their SIMD, lose
        // the issue arbitration (priority, then age) to the consumers, which then wait for them at the barrier.
        __builtin_amdgcn_s_setprio(3);
#endif
        const int c8i = lane & 7, tq = lane >> 3;
        const int ch0 = 32 * ph + 4 * c8i;
        float uu[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.use_u) io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.u) + h * HEAD + ch0, uu);
        // Two sets of load registers: the rows of group g+2 are requested right after group g has been turned into
        // operands, so a request has the preparation of group g+1 plus a barrier interval to complete.
        using WT = std::conditional_t<W_RAW, uint2, float4>;      // one token's 4 decay inputs: raw bf16 or fp32
        struct Rows { uint2 pr[2], pk[2], pv[2]; WT pw[2], pwn; };
        Rows LA;
        [[maybe_unused]] Rows LB;
        // Requests are unconditional (rows past the end are clamped to the last row and zeroed when they are consumed): with
        // a fixed number of loads per call the compiler's counted s_waitcnt lets a request stay in flight across the
        // preparation of the other register set; a conditional load makes it wait for everything outstanding.
        auto row_index = [&](int p) {
            const int pc = min(p, ntok - 1);
            return (unsigned)((a.reverse ? ntok - 1 - pc : pc) * a.C + ch0);
        };
        auto load_w = [&](int p, WT& wv) {
            const unsigned idx = row_index(p);
#ifdef WKV6_ABL_NOLOAD
            if constexpr (W_RAW) wv = make_uint2(idx, 0u); else wv = make_float4(0.f, 0.f, 0.f, 0.f);
#else
            if constexpr (W_RAW) wv = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a.w) + base + idx);
            else wv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.w) + base + idx);
#endif
        };
        auto load_group = [&](int grp, Rows& L) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int p = grp * GRP + pb * BLK + 2 * tq + tt;
                const unsigned idx = row_index(p);
#ifdef WKV6_ABL_NOLOAD
                L.pr[tt] = L.pk[tt] = L.pv[tt] = make_uint2(idx, 0u);
#else
                if constexpr (!STATE_ONLY) L.pr[tt] = *reinterpret_cast<const uint2*>(gr_ + idx);
                L.pk[tt] = *reinterpret_cast<const uint2*>(gk_ + idx);
                L.pv[tt] = *reinterpret_cast<const uint2*>(gv_ + idx);
#endif
                load_w(p, L.pw[tt]);
            }
            load_w(grp * GRP + (pb + 1) * BLK + tq, L.pwn);    // token tq (0..7) of the next block: for alpha
        };
        // log-decay of 4 channels of one token (clamped; 0 for a token past the end)
        auto log_decay = [&](const WT& wv, bool valid, float (&l)[4]) {
            float lw[4];
            if constexpr (W_RAW) {
                lw[0] = -WKV6_EXPF(bf_lo(wv.x)); lw[1] = -WKV6_EXPF(bf_hi(wv.x));
                lw[2] = -WKV6_EXPF(bf_lo(wv.y)); lw[3] = -WKV6_EXPF(bf_hi(wv.y));
            } else {
                lw[0] = wv.x; lw[1] = wv.y; lw[2] = wv.z; lw[3] = wv.w;
                if (a.wkind == 2) {   // the inference entry points pass the decay d = exp(-exp(w)) itself (cuda/rwkv6.cu:38)
#pragma unroll
                    for (int c = 0; c < 4; ++c) lw[c] = __logf(lw[c]);   // d = 0 -> -inf -> clamped below
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) l[c] = valid ? fmaxf(lw[c], LW_MIN) : 0.f;
        };
        auto prep_group = [&](int grp, int buf, const Rows& L) {
            char* const bb = smem + buf * FGRP_BYTES + pb * FBLK_BYTES;
            float r[2][4], k[2][4], cs[2][4];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const bool valid = grp * GRP + pb * BLK + 2 * tq + tt < ntok;
                const uint2 zero2 = make_uint2(0u, 0u);
                const uint2 rr = valid && !STATE_ONLY ? L.pr[tt] : zero2, kk = valid ? L.pk[tt] : zero2, vv = valid ? L.pv[tt] : zero2;
                r[tt][0] = bf_lo(rr.x); r[tt][1] = bf_hi(rr.x); r[tt][2] = bf_lo(rr.y); r[tt][3] = bf_hi(rr.y);
                k[tt][0] = bf_lo(kk.x); k[tt][1] = bf_hi(kk.x); k[tt][2] = bf_lo(kk.y); k[tt][3] = bf_hi(kk.y);
                float l[4];
                log_decay(L.pw[tt], valid, l);
#pragma unroll
                for (int c = 0; c < 4; ++c) cs[tt][c] = (tt ? cs[tt - 1][c] : 0.f) + l[c];
                const int tok = 2 * tq + tt;
                if constexpr (!STATE_ONLY) {
                    // bonus coefficient of the diagonal, restricted to this channel half: 4 in-lane x the 8 lanes of the token
                    float part = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) part = fmaf(r[tt][c] * uu[c], k[tt][c], part);
                    part += dpp_mov<DPP_XOR1>(part);
                    part += dpp_mov<DPP_XOR2>(part);
                    part += dpp_mov<DPP_ROW_HALF_MIRROR>(part);
                    if (c8i == 0) *reinterpret_cast<float*>(bb + FOFF_COEF + (ph * 16 + tok) * 4) = part;
                }
                *reinterpret_cast<uint2*>(bb + F_V * ARR + tok * RSB + ch0 * 2) = vv;
            }
            float ln[4];
            log_decay(L.pwn, grp * GRP + (pb + 1) * BLK + tq < ntok, ln);
            float pre[4], c8[4], c16[4], h1n[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float inc = cs[1][c];                                        // inclusive prefix over the 8 token-pair lanes
                float t = WKV6_SHFL_UP(inc, 8);
                if (tq >= 1) inc += t;
                t = WKV6_SHFL_UP(inc, 16);
                if (tq >= 2) inc += t;
                t = WKV6_SHFL_UP(inc, 32);
                if (tq >= 4) inc += t;
                pre[c] = inc - cs[1][c];
                c8[c] = WKV6_SHFL(pre[c], 32 + c8i);                            // before token 8  (lane tq = 4)
                c16[c] = WKV6_SHFL(inc, 56 + c8i);                              // whole block     (lane tq = 7)
                float s = ln[c];                                             // first 8 tokens of the next block
                s += WKV6_SHFL_XOR(s, 8);
                s += WKV6_SHFL_XOR(s, 16);
                s += WKV6_SHFL_XOR(s, 32);
                h1n[c] = s;
            }
            // alpha = e^{(c16 - rho) + rho_next}: down to e^{-144} with the per-token clamp at LW_MIN, below the fp32 range,
            // while (Sref + Khat^T V) holds terms scaled up by as much as e^{+72} that alpha must bring back.  When any
            // channel of the half comes close, the block stores the two factors separately and its consumers multiply twice.
            const bool two = __builtin_amdgcn_ballot_w64(fminf(fminf(c16[0] - c8[0] + h1n[0], c16[1] - c8[1] + h1n[1]),
                                                                fminf(c16[2] - c8[2] + h1n[2], c16[3] - c8[3] + h1n[3])) < -80.f) != 0;
            if (lane == 0) *reinterpret_cast<int*>(bb + FOFF_FLAG + ph * 4) = two ? 1 : 0;
            if (tq == 0) {
                if (two) {
                    *reinterpret_cast<float4*>(bb + FOFF_ALPHA + ch0 * 4) =
                        make_float4(WKV6_EXPF(c16[0] - c8[0]), WKV6_EXPF(c16[1] - c8[1]), WKV6_EXPF(c16[2] - c8[2]), WKV6_EXPF(c16[3] - c8[3]));
                    *reinterpret_cast<float4*>(bb + FOFF_BETA + ch0 * 4) =
                        make_float4(WKV6_EXPF(h1n[0]), WKV6_EXPF(h1n[1]), WKV6_EXPF(h1n[2]), WKV6_EXPF(h1n[3]));
                } else {
                    *reinterpret_cast<float4*>(bb + FOFF_ALPHA + ch0 * 4) =
                        make_float4(WKV6_EXPF(c16[0] - c8[0] + h1n[0]), WKV6_EXPF(c16[1] - c8[1] + h1n[1]),
                                    WKV6_EXPF(c16[2] - c8[2] + h1n[2]), WKV6_EXPF(c16[3] - c8[3] + h1n[3]));
                    *reinterpret_cast<float4*>(bb + FOFF_BETA + ch0 * 4) = make_float4(1.f, 1.f, 1.f, 1.f);
                }
                *reinterpret_cast<float4*>(bb + FOFF_E8 + ch0 * 4) =
                    make_float4(WKV6_EXPF(c8[0]), WKV6_EXPF(c8[1]), WKV6_EXPF(c8[2]), WKV6_EXPF(c8[3]));
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                float rh[4], kh[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float cex = pre[c] + (tt ? cs[tt - 1][c] : 0.f);
                    const float cin = pre[c] + cs[tt][c];
                    rh[c] = r[tt][c] * WKV6_EXPF(cex - c8[c]);
                    kh[c] = k[tt][c] * WKV6_EXPF(c8[c] - cin);
                }
                char* const row = bb + (2 * tq + tt) * RSB + ch0 * 2;
                uint2 hi, lo;
#ifdef WKV6_ABL_NOSTORE
                split4(rh, hi, lo);
                asm volatile("" :: "v"(hi.x), "v"(hi.y), "v"(lo.x), "v"(lo.y), "v"(row));
                split4(kh, hi, lo);
                asm volatile("" :: "v"(hi.x), "v"(hi.y), "v"(lo.x), "v"(lo.y));
#else
                if constexpr (!STATE_ONLY) {
                    split4(rh, hi, lo);
                    *reinterpret_cast<uint2*>(row + F_RH * ARR) = hi; *reinterpret_cast<uint2*>(row + F_RL * ARR) = lo;
                }
                split4(kh, hi, lo);
                *reinterpret_cast<uint2*>(row + F_KH * ARR) = hi; *reinterpret_cast<uint2*>(row + F_KL * ARR) = lo;
#endif
            }
#ifndef WKV6_ABL_NOSCORE
            if constexpr (!STATE_ONLY) {
                // Scores of this block over this wave's channel half, once for the four consumers of the half:
                // sc[b][a] = sum_{i in half} Khat[b][i] Rhat[a][i] from the rows this wave has just written (LDS operations
                // of one wave execute in order; the library is built with -fno-strict-aliasing so the differently typed
                // loads stay below the stores), masked to b < a with the half's bonus coefficient on the diagonal, split,
                // and stored as the B fragment each consumer lane needs (lane: column a = x, k rows b = 4g+q).
                const int x = lane & 15, g = lane >> 4;
                const int off = x * RSB + (32 * ph + 8 * g) * 2;
                const b8v kh = ld_b8(bb + F_KH * ARR + off), kl = ld_b8(bb + F_KL * ARR + off);
                const b8v rh = ld_b8(bb + F_RH * ARR + off), rl = ld_b8(bb + F_RL * ARR + off);
                f4v sc = {0.f, 0.f, 0.f, 0.f};
                sc = mfma32(kh, rh, sc);
                sc = mfma32(kh, rl, sc);
                sc = mfma32(kl, rh, sc);
                const float cf = *reinterpret_cast<const float*>(bb + FOFF_COEF + (ph * 16 + x) * 4);
                float scm[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int bt = 4 * g + q;                            // key token; query token = x
                    scm[q] = bt < x ? sc[q] : (bt == x ? cf : 0.f);
                }
                uint2 sh, sl;
                split4(scm, sh, sl);
                *reinterpret_cast<uint4*>(bb + FOFF_SC + ph * 1024 + lane * 16) = make_uint4(sh.x, sh.y, sl.x, sl.y);
            }
#endif
        };

        if (ngrp > 0) {
            if constexpr (W_RAW) {
                load_group(0, LA);
                load_group(1, LB);
                prep_group(0, 0, LA);
                load_group(2, LA);
                __syncthreads();
                for (int grp = 0; grp < ngrp; grp += 2) {
                    WKV6_T(ts0);
#ifdef WKV6_STAMP
                    asm volatile("" :: "v"(LB.pr[0].x), "v"(LB.pk[0].x), "v"(LB.pv[0].x), "v"(LB.pw[0].x), "v"(LB.pr[1].x),
                                 "v"(LB.pk[1].x), "v"(LB.pv[1].x), "v"(LB.pw[1].x), "v"(LB.pwn.x));   // wait for the whole set here
#endif
                    WKV6_T(ts1);
#ifndef WKV6_ABL_NOPROD
                    if (grp + 1 < ngrp) prep_group(grp + 1, 1, LB);
#ifdef WKV6_STAMP
                    { unsigned long long tm; WKV6_T(tm); WKV6_ACC(3, tm, ts1); }
#endif
                    load_group(grp + 3, LB);
#endif
                    WKV6_T(ts2);
                    __syncthreads();                              // consumers have finished group grp
                    WKV6_T(ts3);
                    WKV6_ACC(0, ts1, ts0); WKV6_ACC(1, ts2, ts1); WKV6_ACC(2, ts3, ts2);
                    if (grp + 1 >= ngrp) break;
                    WKV6_T(ts0);
#ifdef WKV6_STAMP
                    asm volatile("" :: "v"(LA.pr[0].x), "v"(LA.pk[0].x), "v"(LA.pv[0].x), "v"(LA.pw[0].x), "v"(LA.pr[1].x),
                                 "v"(LA.pk[1].x), "v"(LA.pv[1].x), "v"(LA.pw[1].x), "v"(LA.pwn.x));
#endif
                    WKV6_T(ts1);
#ifndef WKV6_ABL_NOPROD
                    if (grp + 2 < ngrp) prep_group(grp + 2, 0, LA);
#ifdef WKV6_STAMP
                    { unsigned long long tm; WKV6_T(tm); WKV6_ACC(3, tm, ts1); }
#endif
                    load_group(grp + 4, LA);
#endif
                    WKV6_T(ts2);
                    __syncthreads();                              // ... group grp + 1
                    WKV6_T(ts3);
                    WKV6_ACC(0, ts1, ts0); WKV6_ACC(1, ts2, ts1); WKV6_ACC(2, ts3, ts2);
                }
            } else {
                // fp32 decay inputs (the reference-signature entry points): one register set, requests one barrier interval ahead
                load_group(0, LA);
                prep_group(0, 0, LA);
                load_group(1, LA);
                __syncthreads();
                for (int grp = 0; grp < ngrp; ++grp) {
                    if (grp + 1 < ngrp) prep_group(grp + 1, (grp + 1) & 1, LA);
                    load_group(grp + 2, LA);
                    __syncthreads();
                }
            }
        } else {
            __syncthreads();
        }
    } else {
        // ============ consumer: value columns [16w, 16w+16), key channels [32kh, 32kh+32) ================
        // lane (x = lane&15, g = lane>>4) holds Sref[i = tile_ch(2kh+t) + 8g + q][j = 16w + x] in St[t][q]:
        // together the 8 contiguous channels 32kh + 8g .. +7, i.e. k-slot (g, e) of the half's 16x16x32 MFMA
        const int w = wid & 3, kh = wid >> 2;
        const int x = lane & 15, g = lane >> 4;
        f4v St[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float t4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.s0) {
                const long so_ = (long)b * a.s0_bstride + ((long)h * HEAD + 16 * w + x) * HEAD + tile_ch(2 * kh + t) + 8 * g;
                if (a.state_f32) io4<float>::load(reinterpret_cast<const float*>(a.s0) + so_, t4);
                else io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.s0) + so_, t4);
            }
            St[t] = f4v{t4[0], t4[1], t4[2], t4[3]};
        }
        const int troff = (4 * g + (x >> 2)) * RSB + 8 * (x & 3);      // transposed read, natural columns (this wave's V tile)
        const int trow = (4 * g + (x >> 2)) * RSB + 16 * (x & 3);      // transposed read, tile-labelled columns: + tile_tr(t)
        f4v own[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // partial outputs of the two blocks of a group this wave owns
        // finish the two owned blocks of group gp: add the other key half's partial, round, store
        auto finish = [&](int gp) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int blk = 2 * pr + kh;
                const f4v oth = *reinterpret_cast<const f4v*>(smem + PART_OFF + (gp & 1) * PART_BYTES + ((blk * 4 + w) * 64 + lane) * 16);
                const int p = gp * GRP + blk * BLK + x;
                const bool valid = p < ntok;
                const int pc = valid ? p : 0;                            // padding lanes still form a legal address
                const int t = a.reverse ? ntok - 1 - pc : pc;
                const unsigned idx = (unsigned)(t * a.C + 16 * w + 4 * g);
                float o[4] = {own[pr][0] + oth[0], own[pr][1] + oth[1], own[pr][2] + oth[2], own[pr][3] + oth[3]};
                if constexpr (ACC) {
                    float old[4];
                    if (a.y_f32) io4<float>::load(a.y_f32 + base + idx, old);
                    else io4<bf16_t>::load(gy_ + idx, old);
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] += old[q];
                }
                if (valid) {
                    if (!ACC && a.y_f32) io4<float>::store(a.y_f32 + base + idx, o);
                    else io4<bf16_t>::store(gy_ + idx, o);
                }
            }
        };
        __syncthreads();                                          // first group image is ready
        if (a.s0 && ngrp > 0) {                                   // S_0 -> Sref_0 = e^{rho_0} (.) S_0
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float4 e8 = *reinterpret_cast<const float4*>(smem + FOFF_E8 + (tile_ch(2 * kh + t) + 8 * g) * 4);
                St[t][0] *= e8.x; St[t][1] *= e8.y; St[t][2] *= e8.z; St[t][3] *= e8.w;
            }
        }
        // One 64-token group: four blocks with no control flow in between (blocks past the end of the sequence are neutral:
        // zero operands, alpha = 1, nothing stored).  All LDS operands of block n+1 are requested before block n is computed
        // (software pipeline, one block deep): a lone wave otherwise spends most of a block waiting for one LDS round trip
        // after the other.  TWO: some block of the group stores alpha as two factors (extreme decays).
        struct Ops {           // what a block needs first: requested one block ahead
            s4v vf;            // V[4g + e][16w + x]
            b8v zh, zl;        // Rhat rows, this half's k-step
        };
        auto load_ops = [&](const char* bb) {
            Ops o;
            o.vf = tr_read(bb + F_V * ARR + troff + 32 * w);
            if constexpr (!STATE_ONLY) {
                const int off = x * RSB + (32 * kh + 8 * g) * 2;
                o.zh = ld_b8(bb + F_RH * ARR + off);
                o.zl = ld_b8(bb + F_RL * ARR + off);
            }
            return o;
        };
        auto group_body = [&](int grp, auto two_tag) {
            constexpr bool TWO = decltype(two_tag)::value;
            const char* const gb = smem + (grp & 1) * FGRP_BYTES;
            char* const pbuf = smem + PART_OFF + (grp & 1) * PART_BYTES;
            Ops cur = load_ops(gb);
            f4v part[2];
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
                const char* const bb = gb + blk * FBLK_BYTES;
                // requests of this step: what the state update at the end of this block needs, and the next block's head
                s4v kf[2][2];      // Khat^T fragments of the two state tiles, hi / lo
                float4 al[2];
                uint4 scp = make_uint4(0u, 0u, 0u, 0u);            // masked scores^T of this key half, hi | lo
                if constexpr (!STATE_ONLY) scp = *reinterpret_cast<const uint4*>(bb + FOFF_SC + kh * 1024 + lane * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int it = 2 * kh + t;
                    kf[t][0] = tr_read(bb + F_KH * ARR + trow + tile_tr(it));
                    kf[t][1] = tr_read(bb + F_KL * ARR + trow + tile_tr(it));
                    al[t] = *reinterpret_cast<const float4*>(bb + FOFF_ALPHA + (tile_ch(it) + 8 * g) * 4);
                }
                Ops nxt = cur;
                if (blk + 1 < NBLK) nxt = load_ops(bb + FBLK_BYTES);
                __builtin_amdgcn_sched_barrier(0);                // keep the requests above, the arithmetic below
                if ((blk & 1) == 0 && a.ckpt && grp * GRP + blk * BLK < a.T) {   // Sref at every 32-token stage entry, for the backward
                    float* const ck = a.ckpt + ((long)blockIdx.x * ((a.T + CKPT_TOK - 1) / CKPT_TOK) +
                                                (grp * GRP + blk * BLK) / CKPT_TOK) * (HEAD * HEAD);
#pragma unroll
                    for (int t = 0; t < 2; ++t)   // streamed: written once, read once by the backward
                        __builtin_nontemporal_store(St[t], reinterpret_cast<f4v*>(ck + ((w * 4 + 2 * kh + t) * 64 + lane) * 4));
                }
                if constexpr (!STATE_ONLY) {
                    const float t0[4] = {St[0][0], St[0][1], St[0][2], St[0][3]};
                    const float t1[4] = {St[1][0], St[1][1], St[1][2], St[1][3]};
                    uint2 h0, l0, h1, l1;
                    split4(t0, h0, l0);
                    split4(t1, h1, l1);
                    const b8v s_hi = __builtin_bit_cast(b8v, make_uint4(h0.x, h0.y, h1.x, h1.y));
                    const b8v s_lo = __builtin_bit_cast(b8v, make_uint4(l0.x, l0.y, l1.x, l1.y));
                    // one accumulator per MFMA shape (mixed-shape accumulation hazard)
                    f4v yi = {0.f, 0.f, 0.f, 0.f}, yt = {0.f, 0.f, 0.f, 0.f};
                    yt = mfma32(s_hi, cur.zh, yt);                 // y^T[j][a] = sum_{i in half} Sref[i][j] Rhat[a][i]
                    yt = mfma32(s_hi, cur.zl, yt);
                    yt = mfma32(s_lo, cur.zh, yt);
                    const s4v sc_hi = __builtin_bit_cast(s4v, make_uint2(scp.x, scp.y));
                    const s4v sc_lo = __builtin_bit_cast(s4v, make_uint2(scp.z, scp.w));
                    yi = mfma16(cur.vf, sc_hi, yi);                // y^T[j][a] += sum_b V[b][j] sc[b][a]
                    yi = mfma16(cur.vf, sc_lo, yi);
                    part[blk & 1] = yt + yi;
                }
                // Sref <- alpha (.) (Sref + Khat^T V), accumulated by the MFMA into the state registers
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    St[t] = mfma16(kf[t][0], cur.vf, St[t]);
                    St[t] = mfma16(kf[t][1], cur.vf, St[t]);
                    St[t][0] *= al[t].x; St[t][1] *= al[t].y; St[t][2] *= al[t].z; St[t][3] *= al[t].w;
                    if constexpr (TWO) {   // rare path: second factor read just in time
                        const float4 be = *reinterpret_cast<const float4*>(bb + FOFF_BETA + (tile_ch(2 * kh + t) + 8 * g) * 4);
                        St[t][0] *= be.x; St[t][1] *= be.y; St[t][2] *= be.z; St[t][3] *= be.w;
                    }
                }
                if constexpr (!STATE_ONLY) {
                    if (blk & 1) {   // the block of the pair this wave owns (2pr + kh) stays in registers, the other goes to its owner
                        const int pr = blk >> 1;
                        f4v give;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            own[pr][q] = kh ? part[1][q] : part[0][q];
                            give[q] = kh ? part[0][q] : part[1][q];
                        }
                        *reinterpret_cast<f4v*>(pbuf + (((2 * pr + 1 - kh) * 4 + w) * 64 + lane) * 16) = give;
                    }
                }
                cur = nxt;
            }
        };
        for (int grp = 0; grp < ngrp; ++grp) {
            WKV6_T(ts0);
            if constexpr (!STATE_ONLY) {
                if (grp > 0) finish(grp - 1);
            }
            WKV6_T(ts1);
#ifndef WKV6_ABL_NOCONS
            const char* const gb = smem + (grp & 1) * FGRP_BYTES + FOFF_FLAG + kh * 4;
            const int any2 = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(gb) |
                                                            *reinterpret_cast<const int*>(gb + FBLK_BYTES) |
                                                            *reinterpret_cast<const int*>(gb + 2 * FBLK_BYTES) |
                                                            *reinterpret_cast<const int*>(gb + 3 * FBLK_BYTES));
            if (any2) group_body(grp, std::true_type{});      // rare: extreme decays (see the producer)
            else group_body(grp, std::false_type{});
#endif
            WKV6_T(ts2);
            __syncthreads();
            WKV6_T(ts3);
            WKV6_ACC(0, ts1, ts0); WKV6_ACC(1, ts2, ts1); WKV6_ACC(2, ts3, ts2);
        }
        if constexpr (!STATE_ONLY) {
            if (ngrp > 0) finish(ngrp - 1);
        }
        if (a.s_out) {   // after the last block alpha = e^{c_16 - rho} (no next block): St is the plain final state
            const long so_ = ((long)b * a.H + h) * HEAD * HEAD + (long)(16 * w + x) * HEAD + 8 * g;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float t4[4] = {St[t][0], St[t][1], St[t][2], St[t][3]};
                if (a.state_f32) io4<float>::store(reinterpret_cast<float*>(a.s_out) + so_ + tile_ch(2 * kh + t), t4);
                else io4<bf16_t>::store(reinterpret_cast<bf16_t*>(a.s_out) + so_ + tile_ch(2 * kh + t), t4);
            }
        }
    }
#ifdef WKV6_STAMP
    if (a.aux && lane == 0) {
        unsigned long long* const d = reinterpret_cast<unsigned long long*>(a.aux) + ((long)blockIdx.x * 16 + wid) * 4;
        d[0] = stamp_acc[0]; d[1] = stamp_acc[1]; d[2] = stamp_acc[2]; d[3] = stamp_acc[3];
    }
#endif
    if (!STATE_ONLY && !ACC && a.zero_tail) {
        const float z[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = ntok + (tid >> 4); t < a.T; t += 64)
            io4<bf16_t>::store(gy_ + (unsigned)(t * a.C + 4 * (tid & 15)), z);
    }
}

#ifdef WKV6_STAMP
unsigned long long* g_stamp_buffer = nullptr;
#endif

template <bool W_RAW, bool STATE_ONLY, bool ACC> hipError_t launch_fwd16_variant(const ScanArgs& a_in, hipStream_t st)
{
    ScanArgs a = a_in;
#ifdef WKV6_STAMP
    a.aux = reinterpret_cast<float*>(g_stamp_buffer);
#else
    a.aux = nullptr;
#endif
    static LdsAttrOnce attr;                   // per instantiation and device
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(fwd16_kernel<W_RAW, STATE_ONLY, ACC>), FWD16_LDS)) return e;
    hipLaunchKernelGGL((fwd16_kernel<W_RAW, STATE_ONLY, ACC>), dim3(a.B * a.H), dim3(1024), FWD16_LDS, st, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_chunk_fwd(const ScanArgs& a, hipStream_t st)
{
    const bool raw = a.wkind == 1;          // 0: fp32 ew = -exp(w), 1: raw w in bf16, 2: fp32 decay exp(-exp(w))
    if (a.accumulate) return raw ? launch_fwd16_variant<true, false, true>(a, st) : launch_fwd16_variant<false, false, true>(a, st);
    return raw ? launch_fwd16_variant<true, false, false>(a, st) : launch_fwd16_variant<false, false, false>(a, st);
}

// state recurrence only, dumping the stage-entry states into a.ckpt (first half of the self-contained backward)
hipError_t launch_chunk_state_pass(const ScanArgs& a, hipStream_t st)
{
    return a.wkind == 1 ? launch_fwd16_variant<true, true, false>(a, st) : launch_fwd16_variant<false, true, false>(a, st);
}

size_t chunk_ckpt_floats(int B, int T, int H)
{
    return (size_t)B * H * ((T + CKPT_TOK - 1) / CKPT_TOK) * HEAD * HEAD;
}

}  // namespace wkv6

#ifdef WKV6_STAMP
extern "C" void wkv6_set_debug_buffer(void* p) { wkv6::g_stamp_buffer = reinterpret_cast<unsigned long long*>(p); }
#endif
