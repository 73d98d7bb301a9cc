// Chunked MFMA backward of WKV6 for gfx950 (bf16 I/O): 12-wave staged kernel over 64-token checkpoints (the default backward).
// Companion of wkv6_chunk.hip.
//
// Per 16-token block (a = query token, b = key token, c_a exclusive cumulative log decay, S = forward state at
// block entry, G = dL/d(state after the block)), with Rhat_a = r_a e^{c_a - c_8}, Khat_b = k_b e^{c_8 - c_{b+1}},
// fR_a = e^{c_a - c_8}, fK_b = e^{c_8 - c_{b+1}}, E8 = e^{c_8}, E16 = e^{c_16}, E16m8 = e^{c_16 - c_8}:
//   dA[a][b]  = gy_a . v_b                         (b < a),      vg_a = gy_a . v_a
//   gv_b      = sum_{a>b} A[a][b] gy_a + (sum_i r_b u k_b) gy_b + sum_i Khat_b[i] (E16m8 (.) G)[i][:]
//   dq_a      = fR_a (.) ( sum_{b<a} dA[a][b] Khat_b + (E8 (.) S) gy_a )        gr_a = dq_a + vg_a u (.) k_a
//   dk_b      = fK_b (.) ( sum_{a>b} dA[a][b] Rhat_a + (E16m8 (.) G) v_b )      gk_b = dk_b + vg_b u (.) r_b
//   G_entry   = E16 (.) G + E8 (.) sum_a Rhat_a gy_a^T
//   gw_t      = lw_t (.) ( sum_{s>t} (r_s (.) dq_s - k_s (.) dk_s) - k_t (.) dk_t )   (suffix sum over the whole
//               sequence; identity of fla/ops/rwkv6/recurrent_fuse.py:394-396, same as the scan kernels)
//   gu       += vg_a r_a (.) k_a
// i.e. the adjoint of cuda/wkv6_cuda.cu:44-57 (reference backward: cuda/wkv6_cuda.cu:63-227), re-associated.
//
// The forward states are needed in reverse order.  The forward kernel (or, for a self-contained backward, its state-only variant,
// launch_chunk_state_pass) dumps the state at every 64-token boundary (fp32, 4 B per token-channel) in the register order of this
// kernel's row waves (wkv6_scan.h: CKPT_TOK); this kernel walks 32-token stages backwards, takes the checkpoint of the
// enclosing 64-token pair straight into registers (four coalesced 16-byte loads per lane, requested a stage's chain ahead) and
// rebuilds what lies between: one block state inside every stage as before, and on the ODD stage of a pair the two block states
// of the even stage in front of it.  That rebuild needs Khat, V and the block decays of stage s-1 while stage s is consumed, one
// stage before the pipeline would otherwise have them: so the stage image is made in two parts,
//   * the "K part" (Khat hi | lo, V, E16, E16m8: 8 KB per block) TWO stages ahead, into a ring of three slots,
//   * the "R part" (Rhat hi | lo, gy, raw r, k, fR, lw, E8, r.u.k, gy.v: 22 KB per block) one stage ahead, double-buffered,
// with the decay prefix of a stage carried in the producers' registers from its K part to its R part.  The ring takes exactly the
// LDS the checkpoint queue of the 32-token scheme (wkv6_chunk_bwd12.hip, LDS-DMA) used to occupy: 160 KB in all.
//
// One 768-thread workgroup (12 wave64) per (batch, head), one barrier per stage, three roles with one wave of each per SIMD (<= 168 VGPRs):
//   * "row" waves 0..3 own key rows [16w,16w+16): forward states with lane = key row; they produce gr, gk, gw, gu;
//   * "column" waves 4..7 own value columns [16w,16w+16) and the adjoint state G (lane = value column); they produce gv, gs, run the
//     stage's G recurrence early and publish the operand (E16m8 (.) G) hi | lo through LDS for the row waves' gk (GOP_OFF below),
//     and they copy v (of stage s-2, K part) and gy (of stage s-1, R part) from global memory into the images;
//   * four producer waves (wave = block x channel half, lane = 4 channels x 2 tokens) prepare the R part of stage s-1 and the K part of
//     stage s-2 while stage s is consumed, having requested the inputs of the stages behind those before they start;
//   * tiles that every wave of a role needs alike -- dA in both orientations, the masked scores -- are made once per workgroup (row
//     waves 2, 3 and 0, 1) and handed over as MFMA fragments; the waves of a workgroup are not synchronised inside a stage, so every
//     hand-over carries a tag (stage index + 1) its readers poll (XT_OFF, xflag below).
// Launches with few (batch, head) pairs put the row role and the column role of a pair on two CUs (template parameter SPLIT below).
#include <type_traits>
#include "wkv6_chunk.h"

namespace wkv6 {

namespace {

using namespace chunk;

enum { R_RH = 0, R_RL, R_GY, R_R, R_K, NR_ARR };                       // R part, bf16 [16][RSB/2] each
enum { K_KH = 0, K_KL, K_V, NK_ARR };                                  // K part
constexpr int FRS = 72 * 4;                                            // bytes per fp32 token row (conflict-free float4 row reads)
constexpr int ROFF_FR = NR_ARR * ARR;                                  // float [16][72]  fR_a = e^{c_a - c_8}
// (no fK array: fK_b = e^{c_8 - c_{b+1}} = 1 / fR_{b+1}, with fR_16 = E16m8 -- the row waves read fR one token further and take the reciprocal)
constexpr int ROFF_LW = ROFF_FR + BLK * FRS;                           // float [16][72]  lw_a e^{lw_a - max(lw_a, LW_MIN)}
constexpr int ROFF_E8 = ROFF_LW + BLK * FRS;                           // float [64]
constexpr int ROFF_COEF = ROFF_E8 + 256;                               // float [2][16]  per-half sum_i r u k
constexpr int ROFF_VG = ROFF_COEF + 128;                               // float [2][16]  per-half gy_a . v_a
constexpr int RBLK_BYTES = ROFF_VG + 128;
constexpr int KOFF_E16 = NK_ARR * ARR;                                 // float [64]
constexpr int KOFF_E16M8 = KOFF_E16 + 256;
constexpr int KBLK_BYTES = KOFF_E16M8 + 256;
constexpr int STG = 32, SBLK = STG / BLK;                             // tokens / blocks per stage
constexpr int RBUF_BYTES = SBLK * RBLK_BYTES;                          // R part of one stage; two of them
constexpr int KBUF_BYTES = SBLK * KBLK_BYTES;                          // K part of one stage; ring of three
constexpr int KRING = 3;
constexpr int KP_OFF = 2 * RBUF_BYTES;
constexpr int CKT = CKPT_TOK;                                          // tokens between checkpoints (wkv6_scan.h)
// Tiles that every row wave (dA, both orientations, per block) and every column wave (masked scores per block) needs are computed
// ONCE per workgroup and handed over as MFMA fragments: row waves 2 and 3 make the dA tiles of blocks 0 and 1 (both orientations), row
// waves 0 and 1 the score tiles of blocks 0 and 1.  The waves of a role are not synchronised inside a stage, so each tile carries a tag
// (stage index + 1) that its readers poll; the tags live in the unused padding of the first image rows.
constexpr int XT_OFF = KP_OFF + KRING * KBUF_BYTES;                    // uint4 [4 tiles][64 lanes]  dA fragments: bf16x4 hi | bf16x4 lo
constexpr int XS_OFF = XT_OFF + 4 * 1024;                              // uint4 [2 tiles][64 lanes]  score fragments
constexpr int XFLAG_OFF = 128;                                         // int [6] in the padding of row 0 of the first array of buffer 0
// G once per workgroup: the column waves own the adjoint state and publish the operand the row waves need, (E16m8 (.) G) split into
// bf16 hi | lo, stored [value column j][key row i] with 144-byte rows, twice per stage (for block 1, then for block 0: one buffer, tags
// GA "block 1's version is there" / GB "taken" / GC "block 0's version is there" / GD "taken"); the row waves take it with transposing
// reads instead of carrying, updating, scaling and splitting a second copy of G by rows.  A tag needs ~1000 cycles from one wave's
// LDS queue into another wave's poll while the LDS is busy (profiles/r04_handover_timeline.txt), so the chain GA -> GB -> GC is kept
// off the row waves' path: block 1's version of a stage is written in the TAIL of the stage before it (behind GD; the stage barrier
// orders it), the row waves take it first thing and release it (GB) some 1500 cycles before they need block 0's.
constexpr int GRS = 144;                                               // bytes per row of the published operand
constexpr int GOP_OFF = XS_OFF + 2 * 1024;                             // bf16 [2 (hi | lo)][64][GRS / 2]
constexpr int TAG_DA = 0, TAG_SC = 4, TAG_GA = 8, TAG_GB = 12, TAG_GC = 16, TAG_GD = 20, N_TAGS = 24;   // tag slots (see xflag)
constexpr int BWD12K_LDS = GOP_OFF + 2 * HEAD * GRS;
static_assert(BWD12K_LDS <= 160 * 1024, "LDS budget");

// One step of four independent in-row suffix sums: x[q] += x[q] of the lane `ctrl` names (lanes without a source keep their
// value: DPP disables them).  Single v_add_f32_dpp instructions (the builtin gives v_mov_dpp + v_add).  A VALU write of a VGPR
// must be 2 wait states ahead of a DPP read of it and nothing inside an asm string is padded by the compiler: the four
// chains are interleaved so that they keep that distance among themselves, and each step opens with s_nop 1 for whatever
// the compiler may have placed in front of it (a register copy, the subtraction that produced x).
#define WKV6_DPP_ACC4(x, ctrl) asm("s_nop 1\n\t" \
    "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]))

// split a C-layout tile pair (8 floats) into the hi / lo bf16x8 fragments of one k-step
__device__ __forceinline__ void split8(const float (&t0)[4], const float (&t1)[4], b8v& hi, b8v& lo, const SplitConst& spc)
{
    uint2 h0, l0, h1, l1;
    split4(t0, h0, l0, spc);
    split4(t1, h1, l1, spc);
    hi = __builtin_bit_cast(b8v, make_uint4(h0.x, h0.y, h1.x, h1.y));
    lo = __builtin_bit_cast(b8v, make_uint4(l0.x, l0.y, l1.x, l1.y));
}

// GEN = 0: gradients are plain bf16 stores (buffer stores, tokens past the end dropped by the hardware): no store-mode
// branches at all.  GEN = 1: the first half of wkv6_bi (fp32 side buffers instead of the outputs, tail zeroing) -- stores only.
// GEN = 2: the second half (adds the first half and rounds once); the addends are requested ahead of the work whose result
// they meet.  One instantiation each: with the modes as run-time branches of one kernel hipcc merges the "a load may be
// pending" state of the accumulate path into the other paths and drains the vector-memory queue -- s_waitcnt vmcnt(0) -- around
// every store.
// (The kernel proper is a device function of (arguments, workgroup slot) so that chunk_bwd12k_pair_kernel can run it on one of two
// argument blocks: SURVEY.md row n2.)
// SPLIT (ScanArgs::split: few (batch, head) pairs, B*H <= half the CUs): the row role and the column role run as two 8-wave workgroups
// on two CUs, each with its own four producers (duplicated preparation on CUs that would otherwise idle); hardware wave w of
// workgroup part p plays wave w (p = 0: row role) / 4 + w (p = 1: column role) for w < 4 and producer 8 + (w - 4) for w >= 4.  The
// roles cannot share LDS there: G is kept in both orientations (by rows in the row workgroup for gk, by columns in the column
// workgroup for gv: no published operand), the producers copy v and gy themselves, and the column waves make their own score
// tiles.  Every wave does the same arithmetic as in the one-workgroup launch: the outputs are bit-identical.
// CLK: the in-run clock probe (wkv6_set_clock_buffer) is compiled into the plain kernel only (it costs ~6 SGPRs, which the wkv6_bi
// and pair instantiations do not have).
// What a call of the persistent wkv6_bi launch (CHAIN) knows beyond its argument block: the row's batch index and length, looked up a row ahead
// by the launch (a.order[row / H] -> a.lens[b]: two dependent memory round trips, ~3000 cycles that used to open every call).
// Round 6: the calls of a workgroup slot are CHAINED, as the forward's have been since round 5.  A call opens with two serial preparations
// by its producers (K part of its last stage; then that stage's R part + the K part of the stage in front) and two rounds of v / gy copies
// by its column waves before its first stage can start: ~4.7 of the ~7 us a call costs on top of its stages, 12 calls per slot at BASELINE
// configs[2].  The producers are idle while a call's last two stages are consumed (only the R part of stage 0 is left to make): there they
// now prepare the call that FOLLOWS (the row's reversed half, or the slot's next row) -- its first K part beside this call's last R part
// during stage 1, its first R part + second K part during stage 0 -- and the column waves move that call's v / gy with their ordinary
// requests a stage ahead.  The stage -> buffer maps carry a per-call offset (ko, ro) so that the slots simply keep rotating across the
// call boundary: a call's stage s lives in ring slot (s + ko) mod 3 and R buffer (s + ro) & 1, the next call's first stage takes the slot
// below stage 0's -- which is the one that is free.  Calls of fewer than two stages are not chained (nor chained into).
struct BwdChain {
    int b, ntok;
    int ko, ro;                    // this call's slot offsets
    bool chained_in;               // the call before this one has made this call's first images and left its next requests in flight
    bool nx_chain;                 // this call prepares the one behind it: row nx_bh (batch index nx_b, length nx_ntok) of the same argument block,
    unsigned nx_bh;
    int nx_b, nx_ntok;
    bool nx_rev, nx_use_u;         // ... reversed / with the bonus vector
    const float* nx_ckpt;          // ... and its checkpoints (the row waves request its first one during this call's last stage)
};
// The registers of a call that outlive it in the persistent launch: the producers' inputs in flight and what a stage's K part hands to its
// R part (see prep), the column waves' v kept for the vg of the stage whose gy arrives an iteration later.
struct BwdCarry {
    uint2 pr[2], pk[2], pw[2];
    float4 pe[2];
    uint2 ck_[2];
    float cfr[2][4], clw[2][4], cc8[4];
    uint2 cvp[2];
    f4v CK[4];                     // row waves: the checkpoint of the call's first stage pair, requested by the call before it
};
// ROLE: 0 = the wave finds its role from its index (every kernel but the persistent wkv6_bi launch); 1 / 2 / 3 = this instantiation holds the
// row / column / producer role only -- chunk_bwd12k_bi_kernel branches on the role ONCE and runs each role's row loop as its own code, so
// that what one role carries from call to call (BwdCarry) is not live through the other roles' paths.
template <bool W_RAW, int GEN, bool SPLIT, bool AFF, bool CLK = false, bool CHAIN = false, int ROLE = 0>
__device__ __forceinline__ void chunk_bwd12k_body(const ScanArgs& a, const unsigned slot, BwdCarry& cy, const unsigned sslot = 0, const BwdChain& ch = BwdChain{})
{
    [[maybe_unused]] const bool chained_in = CHAIN && ch.chained_in, nx_chain = CHAIN && ch.nx_chain;
    [[maybe_unused]] const int ko = CHAIN ? ch.ko : 0, ro = CHAIN ? ch.ro : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2][SBLK][RBLK_BYTES] | [3][SBLK][KBLK_BYTES] | tiles | G operand
    int tid_ = threadIdx.x;
    // (persistent wkv6_bi launch: everything derived from the lane index is re-derived per call.  Left visible, hipcc hoists those ~25
    // lane-invariant address parts of BOTH bodies out of the row loop and keeps them alive across the calls -- vector-register spills)
    if constexpr (CHAIN) asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    const SplitConst spc = split_const();
    const int hwid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = SPLIT ? (int)(slot & 1) : 0;
    const int bh = SPLIT ? (int)(slot >> 1) : (CLK ? (int)xcd_row_of_slot(slot, (unsigned)(a.B * a.H)) : (int)slot);   // (CLK: the plain kernel)
    const int wid = SPLIT ? (hwid < 4 ? hwid + 4 * part : hwid + 4) : hwid;
    const bool rowrole = ROLE ? ROLE == 1 : wid < 4, producer = ROLE ? ROLE == 3 : wid >= 8;
    const int wv = wid & 3;                                              // tile owned by a row / column wave
    const int pb = wid & 1, half = (wid >> 1) & 1;                       // producer: block of the stage, channel half
    const int b = CHAIN ? ch.b : (a.order ? a.order[bh / a.H] : bh / a.H), h = bh % a.H;
    const long base = (long)b * a.T * a.C + (long)h * HEAD;   // (batch, head) origin: uniform, folded into the pointers;
                                                              // per-lane offsets below stay 32-bit (T*C < 2^31, checked by the API)
    const bf16_t* const gr_ = reinterpret_cast<const bf16_t*>(a.r) + base;
    const bf16_t* const gk_ = reinterpret_cast<const bf16_t*>(a.k) + base;
    const bf16_t* const gv_ = reinterpret_cast<const bf16_t*>(a.v) + base;
    const bf16_t* const ggy = reinterpret_cast<const bf16_t*>(a.gy) + base;
    bf16_t* const ogr = reinterpret_cast<bf16_t*>(a.gr) + base;
    bf16_t* const ogk = reinterpret_cast<bf16_t*>(a.gk) + base;
    bf16_t* const ogv = reinterpret_cast<bf16_t*>(a.gv) + base;
    bf16_t* const ogw = reinterpret_cast<bf16_t*>(a.gw) + base;
    int ntok = a.T;
    if constexpr (CHAIN) ntok = ch.ntok;
    else if (a.lens) ntok = min(max(a.lens[b], 0), a.T);
    const TokAddr<AFF> tok(a, b, ntok);                       // token addressing (wkv6_scan.h): AFF = no per-tensor reversal map
    const int C_ = a.C;
    // buffer resources over this row's first ntok tokens (wkv6_common.h): loads past the end return 0, stores are dropped
    const unsigned nbytes = ntok > 0 ? (unsigned)(ntok - 1) * a.C * 2u + 128u : 0u;
    const rsrc_t rs_gr = make_rsrc(ogr, nbytes), rs_gk = make_rsrc(ogk, nbytes), rs_gv = make_rsrc(ogv, nbytes), rs_gw = make_rsrc(ogw, nbytes);
    // wkv6_bi's halves: the first (GEN == 1) writes its four gradients into fp32 side buffers (always there: checked at launch), the
    // second (GEN == 2) requests them ahead of the work whose result they meet, adds and rounds once.  Side buffers and outputs alike
    // go through buffer resources over the row's first ntok tokens: tokens past the end read zero / are dropped by the hardware.
    // (one launch for both halves, chunk_bwd12k_bi_kernel: the side buffers are this workgroup slot's own scratch, [T][64] fp32 with
    // a token stride of 64 -- written by the first half, read back by the second from the L2 / Infinity Cache, rewritten for the next row)
    const long sbase = (GEN && a.side_compact) ? (long)sslot * a.T * HEAD : base;
    const unsigned sstr = (GEN && a.side_compact) ? (unsigned)HEAD : (unsigned)a.C;      // elements between tokens in the side buffers
    const unsigned nbytes4 = ntok > 0 ? (unsigned)(ntok - 1) * sstr * 4u + 256u : 0u;
    const rsrc_t rs_side[4] = {make_rsrc(GEN && a.g_f32[0] ? a.g_f32[0] + sbase : nullptr, GEN && a.g_f32[0] ? nbytes4 : 0u),
                               make_rsrc(GEN && a.g_f32[1] ? a.g_f32[1] + sbase : nullptr, GEN && a.g_f32[1] ? nbytes4 : 0u),
                               make_rsrc(GEN && a.g_f32[2] ? a.g_f32[2] + sbase : nullptr, GEN && a.g_f32[2] ? nbytes4 : 0u),
                               make_rsrc(GEN && a.g_f32[3] ? a.g_f32[3] + sbase : nullptr, GEN && a.g_f32[3] ? nbytes4 : 0u)};
    // (lane parts of the offsets of this wave's result tile: token x_ of a block, channels 16 wv + 4 g_ .. in the side buffers; the merged
    // two-block rows of `put` in the outputs)
    const int lp_side = tok.lane(lane & 15, 16 * wv + 4 * (lane >> 4), (int)sstr);
    const int lp_put = tok.lane((((lane >> 4) & 1) ? 0 : BLK) + (lane & 15), 16 * wv + 8 * (lane >> 5), C_);
    auto fetch_old = [&](int which, const rsrc_t& rs, int pu, unsigned bit, float (&old)[4]) {     // pu: first position of the block
        if constexpr (GEN == 2) {
            const unsigned idx = tok.off(pu, lane & 15, 16 * wv + 4 * (lane >> 4), (int)sstr, bit, lp_side);
            const float4 t = buf_load16f(rs_side[which], idx * 4u);   // (the halves of wkv6_bi always meet in the fp32 side buffers: checked at launch)
            old[0] = t.x; old[1] = t.y; old[2] = t.z; old[3] = t.w;
        }
    };
#ifdef WKV6_STAMP
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0;
    [[maybe_unused]] unsigned long long tp0 = 0, tp1 = 0;
// slot 5 of a wave's record, chosen at build time (-DWKV6_STAMP5=1: cycles inside tag polls; =2: around the issue of vector-memory instructions)
#ifndef WKV6_STAMP5
#define WKV6_STAMP5 1
#endif
#define WKV6_T5(kind, var) do { if (WKV6_STAMP5 == (kind)) WKV6_T(var); } while (0)
#define WKV6_ACC5(kind, t1, t0) do { if (WKV6_STAMP5 == (kind)) stamp_acc[5] += (t1) - (t0); } while (0)
    unsigned long long poll_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // cycles inside the polls of each tag group (DA SC GA GB GC GD GE GF)
#define WKV6_ACCP(idx, t1, t0) do { if (WKV6_STAMP5 == 1) poll_acc[(idx) >> 2] += (t1) - (t0); } while (0)
// -DWKV6_STAMP5=3: times of the hand-over events since the start of the stage instead (records 12 / 13, see tools/time_ops.py)
#define WKV6_EV(k) do { if (WKV6_STAMP5 == 3) { unsigned long long te_; WKV6_T(te_); poll_acc[k] += te_ - ts0; } } while (0)
#else
#define WKV6_ACCP(idx, t1, t0) do { } while (0)
#define WKV6_EV(k) do { } while (0)
#define WKV6_T5(kind, var) do { } while (0)
#define WKV6_ACC5(kind, t1, t0) do { } while (0)
#endif
    // Gradient stores.  A wave's result tile is 16 tokens x 16 channels: 32 bytes per token row of the [B, T, C] tensors.  Block 1's packed
    // bf16 result waits for block 0's; one v_permlane16_swap per dword then gives the even lane rows block 1 (own four channels + the odd
    // partner row's four) and the odd rows block 0: ONE 16-byte-per-lane instruction stores both blocks of a stage.  (Measured, same box,
    // profiles/r04_store_modes.txt: 8-byte stores per block 0.4246 ms, this 0.4036-0.4064, full 128-byte rows through an LDS staging
    // area 0.4135-0.4185 -- its two extra hand-overs cost more than the wider rows gain; tools/microbench/head_slices.hip has the shapes
    // without arithmetic.)
    const int x_ = lane & 15, g_ = lane >> 4;
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    typedef volatile v4u_t __attribute__((address_space(3))) lds_vv4u;
    [[maybe_unused]] uint2 held_st[4] = {};
    auto put = [&](int which, const rsrc_t& rs, int stg, int blk, unsigned bit, uint2 v) {   // this lane: token x_ of the block, channels 16 wv + 4 g_ .. + 3
        if constexpr (GEN == 1) return;                            // (went to the fp32 side buffer in emit)
        if (blk == SBLK - 1) { held_st[which] = v; return; }      // (the stage's blocks are walked 1, 0)
        const auto sx = __builtin_amdgcn_permlane16_swap(held_st[which].x, v.x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(held_st[which].y, v.y, false, false);
        // even lane rows: block 1's token, odd rows: block 0's
        buf_store16(rs, tok.off(stg * STG, ((g_ & 1) ? 0 : BLK) + x_, 16 * wv + 8 * (g_ >> 1), C_, bit, lp_put) * 2u,
                    make_uint4(sx[0], sy[0], sx[1], sy[1]));
    };
    // the result of one block: fp32 side buffer (first half of wkv6_bi: 64-byte pieces, 16 bytes per lane as they are) or the packed
    // bf16 quad for `put`
    auto emit = [&](int which, int stg, int blk, unsigned bit, float (&o)[4], const float (&old)[4]) -> uint2 {
        if constexpr (GEN == 1) {
            const unsigned idx = tok.off(stg * STG + blk * BLK, x_, 16 * wv + 4 * g_, (int)sstr, bit, lp_side);
            buf_store16f(rs_side[which], idx * 4u, o);
            return make_uint2(0u, 0u);
        }
        if constexpr (GEN == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] += old[q];
        }
        return make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
    };

#ifdef WKV6_DEBUGBUF
    unsigned long long clk0 = 0, rtc0 = 0, clk1 = 0, rtc1 = 0;
    WKV6_CLK(clk0, rtc0);
#endif
    if constexpr (CLK) { if (hwid == 0) clock_stamp(a, slot, 0); }
    const int ngrp = (ntok + STG - 1) / STG;                      // stages
    // images of stage s: R part in buffer s & 1, K part in ring slot s mod 3
    // (CHAIN: + the call's slot offsets; stage n' - 1 of the call that follows takes the ring slot below this call's stage 0 and the other
    // R buffer: rpart_nx / kpart_nx address that call's stage counted from its END, e = n' - 1 - stage)
    auto rpart = [&](int stg, int blk) { return smem + ((stg + ro) & 1) * RBUF_BYTES + blk * RBLK_BYTES; };
    auto kpart = [&](int stg, int blk) { return smem + KP_OFF + (int)((unsigned)(stg + ko) % (unsigned)KRING) * KBUF_BYTES + blk * KBLK_BYTES; };
    [[maybe_unused]] auto rpart_nx = [&](int e, int blk) { return smem + ((ro + 1 + e) & 1) * RBUF_BYTES + blk * RBLK_BYTES; };
    [[maybe_unused]] auto kpart_nx = [&](int e, int blk) { return smem + KP_OFF + (int)((unsigned)(ko + 2 * KRING - 1 - e) % (unsigned)KRING) * KBUF_BYTES + blk * KBLK_BYTES; };

    // ---- producer role: channels ch0..ch0+3, tokens 2tq, 2tq+1 of block pb
    const int c8i = lane & 7, tq = lane >> 3;
    const int ch0 = 32 * half + 4 * c8i;
    float uu[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.use_u) io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.u) + h * HEAD + ch0, uu);
    // (every role loads it: moving the load into the producer branch removes the one two-register spill of the GEN = 1 unsplit
    // instantiation -- a store in front of the row waves' loop and a reload behind it, nothing inside -- but sends the register
    // allocation of the default instantiation from 150 to 167 VGPRs; tests/test_isa_cpu.py keeps scratch accesses out of every loop)

    // register sets: n* = loads in flight (r of stage s-2; k, w of stage s-3), p* = the sets being worked on, c* = what a stage's K part
    // hands to its R part one iteration later (raw k, exponents of fR, the gw multipliers, c_8)
    uint2 (&pr)[2] = cy.pr, (&pk)[2] = cy.pk, (&pw)[2] = cy.pw;      // (BwdCarry: in the persistent wkv6_bi launch these outlive the call)
    float4 (&pe)[2] = cy.pe;
    uint2 (&ck_)[2] = cy.ck_;
    uint2 pv[2], pg[2], cv_[2];                         // SPLIT: the producers also move v (K part) and gy (R part)
    float (&cfr)[2][4] = cy.cfr, (&clw)[2][4] = cy.clw, (&cc8)[4] = cy.cc8;
    if (!chained_in) { cc8[0] = 0.f; cc8[1] = 0.f; cc8[2] = 0.f; cc8[3] = 0.f; }
    const rsrc_t rs_r = make_rsrc(gr_, nbytes), rs_k = make_rsrc(gk_, nbytes), rs_v = make_rsrc(gv_, nbytes), rs_g = make_rsrc(ggy, nbytes);
    const rsrc_t rs_w = W_RAW ? make_rsrc(reinterpret_cast<const bf16_t*>(a.w) + base, nbytes)
                              : make_rsrc(reinterpret_cast<const float*>(a.w) + base, ntok > 0 ? (unsigned)(ntok - 1) * a.C * 4u + 256u : 0u);
    // where a row's inputs live: buffer resources over its first tokens + its token addressing.  `cur` is this call's row; the chained
    // preparation of the call that follows addresses that call's row (row_in_of)
    struct RowIn {
        rsrc_t r, k, v, g, w;
        TokAddr<AFF> tok;
    };
    const RowIn cur{rs_r, rs_k, rs_v, rs_g, rs_w, tok};
    [[maybe_unused]] auto row_in_of = [&](int b_, int h_, int ntok_, bool rev_) {
        const long base_ = (long)b_ * a.T * a.C + (long)h_ * HEAD;
        const unsigned nb_ = ntok_ > 0 ? (unsigned)(ntok_ - 1) * a.C * 2u + 128u : 0u;
        return RowIn{make_rsrc(reinterpret_cast<const bf16_t*>(a.r) + base_, nb_), make_rsrc(reinterpret_cast<const bf16_t*>(a.k) + base_, nb_),
                     make_rsrc(reinterpret_cast<const bf16_t*>(a.v) + base_, nb_), make_rsrc(reinterpret_cast<const bf16_t*>(a.gy) + base_, nb_),
                     W_RAW ? make_rsrc(reinterpret_cast<const bf16_t*>(a.w) + base_, nb_)
                           : make_rsrc(reinterpret_cast<const float*>(a.w) + base_, ntok_ > 0 ? (unsigned)(ntok_ - 1) * a.C * 4u + 256u : 0u),
                     TokAddr<AFF>(ntok_, rev_)};
    };
    const int lp_in[2] = {tok.lane(pb * BLK + 2 * tq, ch0, C_), tok.lane(pb * BLK + 2 * tq + 1, ch0, C_)};
    auto load_r = [&](const RowIn& in, const int (&lp)[2], int stg) {       // tokens past the end load zeros
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            pr[tt] = buf_load8(in.r, in.tok.off(stg * STG, pb * BLK + 2 * tq + tt, ch0, C_, REV_R, lp[tt]) * 2u);
            if constexpr (SPLIT) pg[tt] = buf_load8(in.g, in.tok.off(stg * STG, pb * BLK + 2 * tq + tt, ch0, C_, REV_Y, lp[tt]) * 2u);
        }
    };
    auto load_kw = [&](const RowIn& in, const int (&lp)[2], int stg) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int pl = pb * BLK + 2 * tq + tt;
            const unsigned ik = in.tok.off(stg * STG, pl, ch0, C_, REV_K, lp[tt]), iw = in.tok.off(stg * STG, pl, ch0, C_, REV_W, lp[tt]);
            pk[tt] = buf_load8(in.k, ik * 2u);
            if constexpr (SPLIT) pv[tt] = buf_load8(in.v, in.tok.off(stg * STG, pl, ch0, C_, REV_V, lp[tt]) * 2u);
            if constexpr (W_RAW) pw[tt] = buf_load8(in.w, iw * 2u);
            else pe[tt] = buf_load16f(in.w, iw * 4u);
        }
    };
    // (round 5: no second register set.  The requests for the stages behind go out from INSIDE the preparation, right behind the last read of
    // the raw registers they refill -- a few hundred cycles later than in front of it, of the 5000+ they have -- which saves the twelve
    // register copies per stage that handed a loaded set over to the working set, and they are unconditional: past the first stage they lie
    // below the buffer resources' range and cost nothing, where a conditional request makes its destination a merge the compiler resolves
    // with more copies)
    // One iteration of the producers: the R part of stage sr (Rhat hi | lo, fR, lw, raw r / k, E8, r.u.k -- from r and from what the
    // K part of that stage left in ck_, cfr, clw, cc8 an iteration ago) and the K part of stage sk = sr - 1 (decays, prefix sums,
    // Khat hi | lo, E16, E16m8; refills ck_, cfr, clw, cc8).  Either may be absent (sr / sk < 0: pipeline head and tail).  The two
    // are written as ONE body on purpose: their dependent chains (exp -> scale -> split -> store here, exp -> butterfly -> exp ->
    // scale -> split -> store there) are what a producer's time is made of, and only inside one basic block does the scheduler
    // run them side by side -- as two consecutive functions the preparation took 4.7 k instead of 4.1 k cycles per stage
    // (profiles/r04_stamps_bwd12k_v1.txt).
    // (HASR / HASK are compile-time: a run-time `if (sk >= 0)` would cut the body into the very basic blocks it is meant to avoid;
    // the lane-predicated stores -- r.u.k, E8, E16, E16m8 -- come last for the same reason)
    // (round 6: the images' addresses, the K stage's row length, the R stage's bonus vector and the row the requests go to are parameters --
    // the chained preparation of the persistent launch makes the R part of one call beside the K part of the next)
    auto prep = [&](auto HASR, auto HASK, char* const rb, char* const kb, int sk, int ntok_k, const float (&uu)[4],
                    const RowIn& rq, const int (&rq_lp)[2], int next_r, int next_k) {   // next_*: stages of row rq whose r / k, w are requested (< 0: none)
        float r[2][4], ko[2][4], k[2][4], cs[2][4], lwn[2][4], coef[2] = {0.f, 0.f};
        // ---- R, first half: everything that reads the carried raw k and gw multipliers (the K part below replaces them)
        if constexpr (HASR) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                r[tt][0] = bf_lo(pr[tt].x); r[tt][1] = bf_hi(pr[tt].x); r[tt][2] = bf_lo(pr[tt].y); r[tt][3] = bf_hi(pr[tt].y);
                ko[tt][0] = bf_lo(ck_[tt].x); ko[tt][1] = bf_hi(ck_[tt].x); ko[tt][2] = bf_lo(ck_[tt].y); ko[tt][3] = bf_hi(ck_[tt].y);
                float part = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) part = fmaf(r[tt][c] * uu[c], ko[tt][c], part);
                part += dpp_mov<DPP_XOR1>(part);
                part += dpp_mov<DPP_XOR2>(part);
                part += dpp_mov<DPP_SHL4>(part);
                coef[tt] = part;
                const int tok = 2 * tq + tt;
                // raw r, k: only read as (token x, 4 channels) uint2 by the row waves; tokens >= 8 keep their 16-byte units swapped
                // in pairs so that rows x and x + 8 (same bank at the 160-B stride) do not collide in that read
                char* const rowz = rb + tok * RSB + ((ch0 * 2) ^ ((tok & 8) << 1));
                *reinterpret_cast<uint2*>(rowz + R_R * ARR) = pr[tt];
                *reinterpret_cast<uint2*>(rowz + R_K * ARR) = ck_[tt];
                *reinterpret_cast<float4*>(rb + ROFF_LW + tok * FRS + ch0 * 4) = make_float4(clw[tt][0], clw[tt][1], clw[tt][2], clw[tt][3]);
                if constexpr (SPLIT) {
                    // gy of the stage into its R part, and vg_a = gy_a . v_a over this half's 32 channels with the v the stage's K part
                    // left in cv_ (exact bf16 products, fp32 sums: the same half sums the column waves form in the one-workgroup launch)
                    *reinterpret_cast<uint2*>(rb + tok * RSB + ch0 * 2 + R_GY * ARR) = pg[tt];
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                    float pvg = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pg[tt].x), __builtin_bit_cast(bf2, cv_[tt].x), 0.f, false);
                    pvg = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pg[tt].y), __builtin_bit_cast(bf2, cv_[tt].y), pvg, false);
                    pvg += dpp_mov<DPP_XOR1>(pvg);
                    pvg += dpp_mov<DPP_XOR2>(pvg);
                    pvg += dpp_mov<DPP_SHL4>(pvg);
                    if (c8i == 0) *reinterpret_cast<float*>(rb + ROFF_VG + (half * 16 + tok) * 4) = pvg;
                }
            }
        }
        if constexpr (HASR) load_r(rq, rq_lp, next_r);           // (pr, pg are dead)
        // ---- K, first half: log-decays of the stage's tokens, their sums inside the lane, the gw multipliers
        if constexpr (HASK) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const bool valid = sk * STG + pb * BLK + 2 * tq + tt < ntok_k;
                k[tt][0] = bf_lo(pk[tt].x); k[tt][1] = bf_hi(pk[tt].x); k[tt][2] = bf_lo(pk[tt].y); k[tt][3] = bf_hi(pk[tt].y);
                ck_[tt] = pk[tt];
                if constexpr (SPLIT) {
                    *reinterpret_cast<uint2*>(kb + (2 * tq + tt) * RSB + ch0 * 2 + K_V * ARR) = pv[tt];
                    cv_[tt] = pv[tt];
                }
                float lw[4];
                if constexpr (W_RAW) {
                    lw[0] = -exp2_fast(LOG2E * bf_lo(pw[tt].x)); lw[1] = -exp2_fast(LOG2E * bf_hi(pw[tt].x));
                    lw[2] = -exp2_fast(LOG2E * bf_lo(pw[tt].y)); lw[3] = -exp2_fast(LOG2E * bf_hi(pw[tt].y));
                } else {
                    lw[0] = pe[tt].x; lw[1] = pe[tt].y; lw[2] = pe[tt].z; lw[3] = pe[tt].w;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float l = valid ? fmaxf(lw[c] * LOG2E, LW_MIN2) : 0.f;   // the decay the block algebra uses, in log2 units
                    cs[tt][c] = tt ? cs[tt - 1][c] + l : l;
                    // gw multiplier: the true lw, times d_true / d_clamped where the clamp is active (the algebra yields the
                    // exact gradient of the clamped model, d_clamped * X; the true one is d_true * X)
                    lwn[tt][c] = valid ? lw[c] : 0.f;
                }
            }
            if (__builtin_amdgcn_ballot_w64(lwn[0][0] < LW_MIN || lwn[0][1] < LW_MIN || lwn[0][2] < LW_MIN || lwn[0][3] < LW_MIN ||
                                            lwn[1][0] < LW_MIN || lwn[1][1] < LW_MIN || lwn[1][2] < LW_MIN || lwn[1][3] < LW_MIN)) {   // rare
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) lwn[tt][c] *= exp2_fast(LOG2E * fminf(lwn[tt][c] - LW_MIN, 0.f));
            }
            load_kw(rq, rq_lp, next_k);                           // (pk, pw / pe, pv are dead)
        } else {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int c = 0; c < 4; ++c) { k[tt][c] = 0.f; cs[tt][c] = 0.f; lwn[tt][c] = 0.f; }
        }
        // ---- one block of straight-line code: K's prefix butterfly and scaling beside R's scaling and split
        float pre[4], c8[4], c16[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // exclusive prefix over the 8 token-pair lanes by butterfly: after step k, tot = sum over the lane's aligned group of
            // 2^k pairs, pfx = sum over the pairs before it inside the group
            float tot = cs[1][c], a_, b_;
            const float t = dpp_mov<DPP_ROR8>(tot);
            float pfx = (tq & 1) ? t : 0.f;
            tot += t;
            rows16(tot, a_, b_);
            pfx += (tq & 2) ? a_ : 0.f;
            tot = a_ + b_;
            halves32(tot, a_, b_);
            pfx += (tq & 4) ? a_ : 0.f;
            pre[c] = pfx;
            c8[c] = a_;                                                  // tokens 0..7 = pairs 0..3 = the lower half's total
            c16[c] = a_ + b_;                                            // whole block
        }
        if constexpr (HASR) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                float rh[4], fr[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    fr[c] = exp2_fast(cfr[tt][c]);
                    rh[c] = r[tt][c] * fr[c];
                }
                const int tok = 2 * tq + tt;
                char* const row = rb + tok * RSB + ch0 * 2;
                uint2 hi, lo;
                split4(rh, hi, lo, spc);
                *reinterpret_cast<uint2*>(row + R_RH * ARR) = hi; *reinterpret_cast<uint2*>(row + R_RL * ARR) = lo;
                *reinterpret_cast<float4*>(rb + ROFF_FR + tok * FRS + ch0 * 4) = make_float4(fr[0], fr[1], fr[2], fr[3]);
            }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float kh[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float cex = tt ? pre[c] + cs[tt - 1][c] : pre[c];
                const float cin = pre[c] + cs[tt][c];
                cfr[tt][c] = cex - c8[c];                                // fR = 2^this, formed with the stage's R part
                clw[tt][c] = lwn[tt][c];
                kh[c] = k[tt][c] * exp2_fast(c8[c] - cin);
            }
            if constexpr (HASK) {
                char* const row = kb + (2 * tq + tt) * RSB + ch0 * 2;
                uint2 hi, lo;
                split4(kh, hi, lo, spc);
                *reinterpret_cast<uint2*>(row + K_KH * ARR) = hi; *reinterpret_cast<uint2*>(row + K_KL * ARR) = lo;
            }
        }
        if constexpr (HASR) {
            if (c8i == 0) {
                *reinterpret_cast<float*>(rb + ROFF_COEF + (half * 16 + 2 * tq) * 4) = coef[0];
                *reinterpret_cast<float*>(rb + ROFF_COEF + (half * 16 + 2 * tq + 1) * 4) = coef[1];
            }
        }
        if (tq == 0) {
            if constexpr (HASR)
                *reinterpret_cast<float4*>(rb + ROFF_E8 + ch0 * 4) =
                    make_float4(exp2_fast(cc8[0]), exp2_fast(cc8[1]), exp2_fast(cc8[2]), exp2_fast(cc8[3]));
            if constexpr (HASK) {
                *reinterpret_cast<float4*>(kb + KOFF_E16 + ch0 * 4) =
                    make_float4(exp2_fast(c16[0]), exp2_fast(c16[1]), exp2_fast(c16[2]), exp2_fast(c16[3]));
                *reinterpret_cast<float4*>(kb + KOFF_E16M8 + ch0 * 4) =
                    make_float4(exp2_fast(c16[0] - c8[0]), exp2_fast(c16[1] - c8[1]), exp2_fast(c16[2] - c8[2]), exp2_fast(c16[3] - c8[3]));
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) cc8[c] = c8[c];
    };

    // ---- phase-C role
    const int x = lane & 15, g = lane >> 4;
    int troff = (4 * g + (x >> 2)) * RSB + 8 * (x & 3);          // transposed read, natural columns (own tile)
    int trow = (4 * g + (x >> 2)) * RSB + 16 * (x & 3);          // transposed read, tile-labelled columns: + tile_tr(t)
    // Products of an exact operand (v, gy) with a split one (Khat, Rhat, the masked scores) over the block's 16 tokens run as ONE 16x16x32
    // MFMA: lane group g supplies k-slots 8g .. 8g+7 = (hi part, tokens 4g .. 4g+3 | lo part, the same tokens) against the exact operand's
    // four tokens twice -- the very fragments the two 16x16x16 MFMAs took, concatenated in registers (cat2), no LDS read changes.  Both
    // shapes cost 16 cycles (profiles/r05_issue_floor.md): 18 of the 94 MFMAs per SIMD and stage fewer.
    auto cat2 = [](s4v a_, s4v b_) { return __builtin_bit_cast(b8v, __builtin_shufflevector(a_, b_, 0, 1, 2, 3, 4, 5, 6, 7)); };
    // (explicit LDS address space: a volatile access through a generic pointer becomes a FLAT instruction whose wait drains the whole
    // vector-memory queue)
    typedef volatile int __attribute__((address_space(3))) lds_vint;
    // tag slot idx: the padding (bytes 128..159) of image rows 0, 1, 2 of the first array, eight per row
    auto xflag_at = [&](int idx) { return (lds_vint*)(smem + (idx >> 3) * RSB + XFLAG_OFF + (idx & 7) * 4); };
    if (tid < N_TAGS) *xflag_at(tid) = 0;                     // (ordered before any use by the barrier that opens the first stage)
    // Hand-over protocol.  Tile stores, tag stores, tag polls and tile loads are all volatile accesses: the compiler keeps their program
    // order among themselves (and the LDS executes one wave's operations in order) without a full memory barrier, which would stop it
    // from keeping the stage's many independent operand reads in flight around them.  The non-volatile operand reads that must stay
    // behind a poll (take_gop) get their address through an `asm volatile` placed after the poll, and the publication that must stay
    // behind them is preceded by an `asm volatile` that consumes their result: volatile asm statements and volatile accesses are not
    // reordered among each other.  A tag is the stage index + 1 and every role executes every stage: a reader can only ever wait for
    // a tag that some wave is about to write.
    // (tag stores, round 5: the LDS address of a tag is wave-uniform, and so is its value -- left to the compiler every store moved both
    // from scalar registers into fresh vector registers first, 2 of the ~10 vector issue slots a role spends on its 4-5 publications per
    // stage.  The slot address of tag group 0 for this wave (tag_wv: + 4 wv; tag_tb: + 8 (wv & 1), the dA tiles' pairs) is kept in a vector
    // register for the kernel's life, a tag group's distance is an immediate, the value is moved once per stage.)
    unsigned tag_wv = (unsigned)(unsigned long)(lds_vint*)(smem + XFLAG_OFF + 4 * wv);
    unsigned tag_tb = (unsigned)(unsigned long)(lds_vint*)(smem + XFLAG_OFF + 8 * (wv & 1));
    asm volatile("" : "+v"(tag_wv), "+v"(tag_tb));
    auto tag_off = [](int idx0) { return (idx0 >> 3) * RSB + (idx0 & 7) * 4; };       // idx0: a tag group's first slot (a multiple of 4)
    auto publish_at = [&](unsigned base, int idx0, int extra, int tagv) {             // after this wave's (volatile) tile store
        if (lane == 0) *(lds_vint*)(unsigned long)(base + (unsigned)(tag_off(idx0) + extra)) = tagv;
    };
    auto publish = [&](int idx0, int tagv) { publish_at(tag_wv, idx0, 0, tagv); };       // slot idx0 + wv
#ifdef WKV6_DEBUG
    // debug build: a poll that does not see its tag within ~2^22 tries writes (slot, tag index, expected, seen) to the debug buffer and traps
    auto spin_fail = [&](int idx, int tag, unsigned seen) {
        if (a.aux && lane == 0) {
            unsigned* const d = reinterpret_cast<unsigned*>(a.aux);
            d[0] = 0xdeadbeefu; d[1] = slot; d[2] = (unsigned)((wid << 8) | idx); d[3] = (unsigned)tag; d[4] = seen;
        }
        __builtin_trap();
    };
#define WKV6_SPIN_GUARD(idx, tag, seen) if (++spins > (1u << 22)) spin_fail(idx, tag, seen)
#else
#define WKV6_SPIN_GUARD(idx, tag, seen) do { } while (0)
#endif
    // readers poll two / four adjacent tags with one LDS round trip (a poll costs an LDS latency plus a drain of the wave's LDS queue):
    // lane l reads tag idx + (l & 3) (or & 1) -- one dword per lane, the same four addresses wave-wide, a broadcast -- and the wave votes:
    // one compare into a lane mask and a scalar test per poll (round 5; a 16-byte read per lane needed 4 xor + 3 or + a
    // v_readfirstlane: 8 vector issue slots per poll, ~56 of the 1066 per SIMD and stage)
    const int lane_tag4 = (lane & 3) * 4, lane_tag2 = (lane & 1) * 4;
    auto tags_differ = [&](unsigned f, int tag) { return __builtin_amdgcn_ballot_w64(f != (unsigned)tag) != 0; };
    auto await_n = [&](int idx, int tag, int lane_off) {
        [[maybe_unused]] unsigned spins = 0;
        WKV6_T5(1, tp0);
        for (;;) {
            const unsigned f = (unsigned)*(lds_vint*)((unsigned char __attribute__((address_space(3)))*)xflag_at(idx) + lane_off);
            if (!tags_differ(f, tag)) break;
            WKV6_SPIN_GUARD(idx, tag, f);
            __builtin_amdgcn_s_sleep(1);
        }
        WKV6_T5(1, tp1); WKV6_ACC5(1, tp1, tp0); WKV6_ACCP(idx, tp1, tp0);
    };
    auto await2 = [&](int idx, int tag) { await_n(idx, tag, lane_tag2); };
    auto await4 = [&](int idx, int tag) { await_n(idx, tag, lane_tag4); };
    // A poll whose tag is already there still costs an LDS round trip with the wave's LDS queue drained.  Where the schedule allows, the
    // tag read is issued ahead (peek4) with independent work behind it and only examined later (settle4): satisfied -- the usual case --
    // it costs a compare; otherwise the ordinary poll takes over.  Readers of the handed-over data are volatile accesses or get
    // their address through an asm volatile behind settle4, as after await4: the LDS serves a wave's requests in order.
    auto peek4 = [&](int idx) -> unsigned {
        return (unsigned)*(lds_vint*)((unsigned char __attribute__((address_space(3)))*)xflag_at(idx) + lane_tag4);
    };
    auto settle4 = [&](int idx, int tag, const unsigned f) { if (tags_differ(f, tag)) await4(idx, tag); };
    auto tile_store = [&](int off, uint2 hi, uint2 lo) {
        *(lds_vv4u*)(smem + off + lane * 16) = v4u_t{hi.x, hi.y, lo.x, lo.y};
    };
    auto tile_load = [&](int off) {
        const v4u_t f = *(lds_vv4u*)(smem + off + lane * 16);
        return make_uint4(f.x, f.y, f.z, f.w);
    };
    constexpr std::true_type yes_c{};
    constexpr std::false_type no_c{};
    [[maybe_unused]] const int nxn = CHAIN ? (ch.nx_ntok + STG - 1) / STG : 0;        // stages of the call that follows
    if (producer) {
        // =============== producers: R part of stage s-1 and K part of stage s-2 while stage s is consumed =====
        if (!chained_in && ngrp > 0) {
            load_kw(cur, lp_in, ngrp - 1);
            load_r(cur, lp_in, ngrp - 1);
            prep(no_c, yes_c, rpart(0, pb), kpart(ngrp - 1, pb), ngrp - 1, ntok, uu, cur, lp_in, -1, ngrp - 2);   // K part of the last stage (requests k, w of the stage in front)
            if (ngrp > 1) prep(yes_c, yes_c, rpart(ngrp - 1, pb), kpart(ngrp - 2, pb), ngrp - 2, ntok, uu, cur, lp_in, ngrp - 2, ngrp - 3);  // its R part, and the K part of the stage in front
            else prep(yes_c, no_c, rpart(0, pb), kpart(0, pb), -1, ntok, uu, cur, lp_in, -1, -1);
        }
        __syncthreads();
        // steady state: stages ngrp-1 .. 2 (a full R + K preparation each); the last two stages are peeled so that the loop body
        // is one straight preparation without the pipeline-tail cases
        for (int grp = ngrp - 1; grp >= 2; --grp) {
            WKV6_T(ts0);
#ifdef WKV6_STAMP
            asm volatile("" :: "v"(pr[0].x), "v"(pk[0].x), "v"(pr[1].x), "v"(pk[1].x));      // wait for the loads here
#endif
            WKV6_T(ts1);
            WKV6_T(ts2);
            prep(yes_c, yes_c, rpart(grp - 1, pb), kpart(grp - 2, pb), grp - 2, ntok, uu, cur, lp_in, grp - 2, grp - 3);   // r of stage s-1, k / w of stage s-2; requests r of s-2, k / w of s-3
            WKV6_T(ts3);
            __syncthreads();
            WKV6_T(ts4);
            WKV6_ACC(0, ts1, ts0); WKV6_ACC(1, ts2, ts1); WKV6_ACC(2, ts3, ts2); WKV6_ACC(3, ts4, ts3);
        }
        bool chained_out = false;
        if constexpr (CHAIN) {
            if (nx_chain) {   // (implies ngrp >= 2 and nxn >= 2: chunk_bwd12k_bi_kernel)
                // the call that follows: its row's inputs, its bonus vector
                const int nxh = (int)(ch.nx_bh % (unsigned)a.H);
                const RowIn nx = row_in_of(ch.nx_b, nxh, ch.nx_ntok, ch.nx_rev);
                const int lp_nx[2] = {nx.tok.lane(pb * BLK + 2 * tq, ch0, C_), nx.tok.lane(pb * BLK + 2 * tq + 1, ch0, C_)};
                float uu_nx[4] = {0.f, 0.f, 0.f, 0.f};
                if (ch.nx_use_u) io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.u) + nxh * HEAD + ch0, uu_nx);
                // stage 1 is consumed: the R part of stage 0 beside the K part of that call's LAST stage (into the ring slot below stage 0's)
                load_kw(nx, lp_nx, nxn - 1);
                prep(yes_c, yes_c, rpart(0, pb), kpart_nx(0, pb), nxn - 1, ch.nx_ntok, uu, nx, lp_nx, nxn - 1, nxn - 2);
                __syncthreads();
                // stage 0 is consumed: that call's first R part (into the R buffer stage 0 does not use) and its second K part; its
                // requests for the stages behind stay in flight across the call boundary
                prep(yes_c, yes_c, rpart_nx(0, pb), kpart_nx(1, pb), nxn - 2, ch.nx_ntok, uu_nx, nx, lp_nx, nxn - 2, nxn - 3);
                __syncthreads();
                chained_out = true;
            }
        }
        if (!chained_out) {
            if (ngrp >= 2) {                                           // stage 1 is consumed: only the R part of stage 0 is left to make
                prep(yes_c, no_c, rpart(0, pb), kpart(0, pb), -1, ntok, uu, cur, lp_in, -1, -1);
                __syncthreads();
            }
            if (ngrp >= 1) __syncthreads();                            // stage 0 is consumed
        }
    } else
    if (rowrole) {
        // =============== key rows [16wv, 16wv+16): gr, gk, gw, gu ==========================================
        // ST[blk][jt][q] = S_entry(blk)[i = 16wv + x][j = tile_ch(jt) + 8g + q]   (transposed tiles: lane = key row)
        float ue[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.use_u) io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.u) + h * HEAD + 16 * wv + 4 * g, ue);
        // GI[jt][q]      = G[i = 16wv + x][j = tile_ch(jt) + 8g + q]                  (SPLIT only: the adjoint state by rows)
        f4v ST[SBLK][4], GI[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) GI[t] = f4v{0.f, 0.f, 0.f, 0.f};
        float Rc[4] = {0.f, 0.f, 0.f, 0.f}, gu_acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.rc_in) io4<float>::load(a.rc_in + (long)b * a.C + h * HEAD + 16 * wv + 4 * g, Rc);   // the suffix sum beyond this segment
        // Checkpoint of the 64-token pair that holds stage `stg`: the forward wrote it in this wave's register order
        // ([row wave][jt][lane][4], wkv6_scan.h), so the slice is four coalesced 16-byte loads per lane straight into
        // registers.  Requested behind a stage's pre-phase (the registers of the stage's own states are dead by then) for the NEXT
        // stage: the latency runs under the chain, the barrier and the next stage's tile work.  Both stages of a pair use the
        // same checkpoint: requested once, ahead of the odd stage, and kept in CK for the even one.
        const unsigned nslots = ((unsigned)a.T + CKT - 1) / CKT;
        // (two-level scan over T: this batch row is segment b % S of sequence b / S; the S segments' slots are consecutive -- the
        // whole sequence's ordinary checkpoint layout, exactly as the forward wrote them: wkv6_chunk.hip)
        const int segs = a.ckpt_segs > 1 ? a.ckpt_segs : 1;
        const long ck_slot0 = ((long)((b / segs) * a.H + h) * segs + b % segs) * nslots;
        const rsrc_t rs_ck = make_rsrc(a.ckpt + ck_slot0 * (HEAD * HEAD), nslots * 16384u);
        f4v (&CK)[4] = cy.CK;                                     // (BwdCarry: a chained call finds its first checkpoint requested)
        auto request_ckpt = [&](int stg) {
            const unsigned off = (unsigned)(stg >> 1) * 16384u + (unsigned)wv * 4096u + (unsigned)lane * 16u;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const float4 t = buf_load16f(rs_ck, off + jt * 1024u);
                CK[jt] = f4v{t.x, t.y, t.z, t.w};
            }
        };
        // S <- E16 (.) S + E16m8 (.) (Khat^T V): the state one block further, from that block's K part
        auto advance = [&](const char* kb, const f4v (&Sin)[4], f4v (&Sout)[4]) {
            const s4v khf = tr_read(kb + K_KH * ARR + troff + 32 * wv);
            const s4v klf = tr_read(kb + K_KL * ARR + troff + 32 * wv);
            const float e16 = *reinterpret_cast<const float*>(kb + KOFF_E16 + (16 * wv + x) * 4);
            const float e16m8 = *reinterpret_cast<const float*>(kb + KOFF_E16M8 + (16 * wv + x) * 4);
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const s4v vf = tr_read(kb + K_V * ARR + trow + tile_tr(jt));     // V[4g+e][tile_ch(jt) + 8(x>>2) + (x&3)]
                f4v o = {0.f, 0.f, 0.f, 0.f};
                o = mfma32(cat2(vf, vf), cat2(khf, klf), o);          // (V | V) (Khat_hi ; Khat_lo)
#pragma unroll
                for (int q = 0; q < 4; ++q) Sout[jt][q] = fmaf(e16, Sin[jt][q], e16m8 * o[q]);
            }
        };
        // (as for the column waves' G below: whatever was loaded for the loop is waited for in front of it)
        // (persistent wkv6_bi launch: a call's opening latencies count -- 12 calls per slot at BASELINE configs[2] --: the checkpoint request goes out
        // in front of the wait for the bonus vector, not behind it; a chained call's was made by the call before it)
        if constexpr (CHAIN) { if (!chained_in && ngrp > 0) request_ckpt(ngrp - 1); }
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(ue[q]), "+v"(Rc[q]));
        if constexpr (!CHAIN) { if (ngrp > 0) request_ckpt(ngrp - 1); }
        // The vector-memory counter counts loads and stores alike, in order.  Inside the loop the checkpoint request of a stage is
        // followed by its three gradient stores, so the wait for the first checkpoint register may leave six requests in flight; hipcc
        // sizes the one s_waitcnt at the loop head for the worse of its two entries, and on first entry only the other three checkpoint
        // loads follow -- vmcnt(3 .. 0) would then wait, every stage, for the stores of the stage before to be acknowledged.  Three
        // stores that the bounds check drops give the first entry the same queue.
        {   // (first half of wkv6_bi: four fp32 side stores -- gk, gw of both blocks -- follow a stage's checkpoint request)
            // (persistent wkv6_bi launch: made per call -- as a constant it is hoisted out of the row loop into four scalar registers the loop
            // does not have, and ends up in scratch)
            unsigned none_bytes = 0u;
            if constexpr (CHAIN) { asm volatile("" : "+s"(none_bytes)); none_bytes = __builtin_amdgcn_readfirstlane(none_bytes); }
            const rsrc_t rs_none = make_rsrc(static_cast<bf16_t*>(nullptr), none_bytes);
#pragma unroll
            // (the data are whatever four live registers hold -- the bonus vector: a zero quad is hoisted out of the persistent launch's row loop and spilled)
            for (int i = 0; i < (GEN == 1 ? 4 : 3); ++i)
                buf_store16(rs_none, 16u * i, make_uint4(__float_as_uint(ue[0]), __float_as_uint(ue[1]), __float_as_uint(ue[2]), __float_as_uint(ue[3])));   // (distinct: identical ones are merged)
        }

        __syncthreads();                                          // first stage image is ready
        for (int grp = ngrp - 1; grp >= 0; --grp) {
            int tagv = grp + 1;                                    // this stage's tag value, in a vector register (see publish)
            asm volatile("" : "+v"(tagv));
            WKV6_T(ts0);
            // ---- G-dependent part of gk, decoupled from the epilogues: sum_j (E16m8 G)[i][j] v_b[j] per block from the operand
            //      the column waves publish; block 1's version is taken (and released: GB) here, block 0's behind the pre-phase
            f4v gvb[SBLK];
            auto take_gop = [&](int blk) {
                int base_ = GOP_OFF + (8 * g + (x >> 2)) * GRS + (16 * wv + 4 * (x & 3)) * 2;
                asm volatile("" : "+v"(base_));                    // keeps the reads behind the tag polls that precede the call
                typedef short s8v __attribute__((ext_vector_type(8)));
                const char* const kb = kpart(grp, blk);
                f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {   // lane (x, g): key row 16wv + x, value columns 32s + 8g .. +7 (stored [j][i]: transposing reads)
                    const char* const ph = smem + base_ + 32 * s * GRS;
                    const s8v h8 = __builtin_shufflevector(tr_read(ph), tr_read(ph + 4 * GRS), 0, 1, 2, 3, 4, 5, 6, 7);
                    const s8v l8 = __builtin_shufflevector(tr_read(ph + HEAD * GRS), tr_read(ph + HEAD * GRS + 4 * GRS), 0, 1, 2, 3, 4, 5, 6, 7);
                    const b8v vr = ld_b8(kb + K_V * ARR + x * RSB + (32 * s + 8 * g) * 2);
                    acc = mfma32(__builtin_bit_cast(b8v, h8), vr, acc);
                    acc = mfma32(__builtin_bit_cast(b8v, l8), vr, acc);
                }
                return acc;
            };
            if constexpr (!SPLIT) {   // first of all: block 1's operand is there (published before the stage barrier; the first stage polls) --
                                      // taking it now releases it ~1500 cycles earlier for the column waves to lay block 0's over it
                if (grp == ngrp - 1) await4(TAG_GA, grp + 1);
                WKV6_EV(0);
                gvb[1] = take_gop(1);
                // released at once: the reads are in this wave's LDS queue, which is served in order, and the tag store cannot be moved
                // above loads it may alias (their address is opaque to the compiler) -- no need to wait for their data here
                publish(TAG_GB, tagv);
                WKV6_EV(1);
            }
            // (second half of wkv6_bi: the first half's gr of both blocks, requested a pre-phase ahead of the sums they meet)
            float old_gr[SBLK][4] = {}, old_gk[SBLK][4] = {}, old_gw[SBLK][4] = {};
#pragma unroll
            for (int blk = SBLK - 1; blk >= 0; --blk) fetch_old(0, rs_gr, grp * STG + blk * BLK, REV_R, old_gr[blk]);
            // then, so that it is there long before the others ask for it:
            if (wv >= 2) {   // this wave's dA tiles of the stage: block wv - 2, both orientations (tile 2 blk: dA[a][b], lane col b; 2 blk + 1:
                             // dA^T[b][a], lane col a) from one pair of operand reads.  Row waves 0 and 1 make the score tiles instead.
                const int tb = wv - 2;
                const char* const rb = rpart(grp, tb);
                const char* const kb = kpart(grp, tb);
                f4v dA_ab = {0.f, 0.f, 0.f, 0.f}, dA_ba = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int off = x * RSB + (32 * s + 8 * g) * 2;
                    const b8v gy_ = ld_b8(rb + R_GY * ARR + off), vr = ld_b8(kb + K_V * ARR + off);
                    dA_ab = mfma32(gy_, vr, dA_ab);              // [row a][col b]: lane col b = x, rows a = 4g+q
                    dA_ba = mfma32(vr, gy_, dA_ba);              // [row b][col a]: lane col a = x, rows b = 4g+q
                }
                float dab[4], dba[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = 4 * g + q;
                    dab[q] = x < o ? dA_ab[q] : 0.f;              // dA[a = o][b = x], strictly lower
                    dba[q] = o < x ? dA_ba[q] : 0.f;              // dA^T[b = o][a = x], strictly lower
                }
                uint2 th, tl;
                split4(dab, th, tl, spc);
                tile_store(XT_OFF + (2 * tb) * 1024, th, tl);
                publish_at(tag_tb, TAG_DA, 0, tagv);            // slot TAG_DA + 2 tb
                split4(dba, th, tl, spc);
                tile_store(XT_OFF + (2 * tb + 1) * 1024, th, tl);
                publish_at(tag_tb, TAG_DA, 4, tagv);            // slot TAG_DA + 2 tb + 1
            } else if constexpr (!SPLIT) {   // row waves 0 and 1: the masked score tile of block wv, for the column waves
                const char* const rb = rpart(grp, wv);
                const char* const kb = kpart(grp, wv);
                f4v sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int off = x * RSB + (32 * s + 8 * g) * 2;
                    const b8v rh = ld_b8(rb + R_RH * ARR + off), rl = ld_b8(rb + R_RL * ARR + off);
                    const b8v kh = ld_b8(kb + K_KH * ARR + off), kl = ld_b8(kb + K_KL * ARR + off);
                    sc = mfma32(rh, kh, sc);                  // A[row a][col b]: lane col b = x, rows a = 4g+q
                    sc = mfma32(rh, kl, sc);
                    sc = mfma32(rl, kh, sc);
                }
                const float4 cfa = *reinterpret_cast<const float4*>(rb + ROFF_COEF + 16 * g);
                const float4 cfb = *reinterpret_cast<const float4*>(rb + ROFF_COEF + 64 + 16 * g);
                const float cf[4] = {cfa.x + cfb.x, cfa.y + cfb.y, cfa.z + cfb.z, cfa.w + cfb.w};
                float scm[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = 4 * g + q;                  // query token a; key token b = x
                    scm[q] = x < o ? sc[q] : (x == o ? cf[q] : 0.f);
                }
                uint2 th, tl;
                split4(scm, th, tl, spc);
                tile_store(XS_OFF + wv * 1024, th, tl);
                publish(TAG_SC, tagv);
            }
            // stage-entry forward state: the checkpoint of this 64-token pair (requested a stage's chain ago); the odd stage of a pair
            // first walks it through the two blocks of the even stage in front (their K part is already in the ring)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) ST[0][jt] = CK[jt];
            WKV6_T(ts1);
            if (grp & 1) {
                advance(kpart(grp - 1, 0), ST[0], ST[0]);
                advance(kpart(grp - 1, 1), ST[0], ST[0]);
            }
            // ---- rebuild the entry state of block 1
            [[maybe_unused]] unsigned pk_gc = 0;
#pragma unroll
            for (int blk = 0; blk < SBLK - 1; ++blk) advance(kpart(grp, blk), ST[blk], ST[blk + 1]);
            const unsigned pk_da = peek4(TAG_DA);                    // (examined where the pre-phase starts)
            WKV6_T(ts2);
            // ---- pre-phase: everything that does not depend on G (blocks past the end of the sequence are neutral):
            //      dA, the whole gr path (it needs only the forward states) and the Rhat.dA part of gk.
            f4v ackp[SBLK];
            float at[SBLK][4], vgs[SBLK];
            uint2 held_gr[SBLK];                                  // gr of both blocks, stored behind take_gop(0) (after the checkpoint request: keeps the
                                                                  // vector-memory queue's order loads-then-stores, see the dummy stores in front of the loop)
#pragma unroll
            for (int blk = SBLK - 1; blk >= 0; --blk) {
                const char* const rb = rpart(grp, blk);
                const char* const kb = kpart(grp, blk);
                b8v gyr[2];                                      // gy [token x][32s + 8g .. +7]: the B operand of accs
#pragma unroll
                for (int s = 0; s < 2; ++s) gyr[s] = ld_b8(rb + R_GY * ARR + x * RSB + (32 * s + 8 * g) * 2);
                // vg_x = dA[x][x] = gy_x . v_x: the two half sums
                const float vg = *reinterpret_cast<const float*>(rb + ROFF_VG + x * 4) + *reinterpret_cast<const float*>(rb + ROFF_VG + 64 + x * 4);
                vgs[blk] = vg;
                if (blk == SBLK - 1) settle4(TAG_DA, grp + 1, pk_da);   // all four dA tiles of the stage with one poll
                if constexpr (!SPLIT) { if (blk == 0) pk_gc = peek4(TAG_GC); }   // (examined behind the pre-phase)
                const uint4 fab = tile_load(XT_OFF + (2 * blk) * 1024), fba = tile_load(XT_OFF + (2 * blk + 1) * 1024);
                const s4v dab_hi = __builtin_bit_cast(s4v, make_uint2(fab.x, fab.y)), dab_lo = __builtin_bit_cast(s4v, make_uint2(fab.z, fab.w));
                const s4v dba_hi = __builtin_bit_cast(s4v, make_uint2(fba.x, fba.y)), dba_lo = __builtin_bit_cast(s4v, make_uint2(fba.z, fba.w));
                const s4v rhf_w = tr_read(rb + R_RH * ARR + troff + 32 * wv);      // Rhat[4g+e][16wv + x]
                const s4v rlf_w = tr_read(rb + R_RL * ARR + troff + 32 * wv);
                const s4v khf = tr_read(kb + K_KH * ARR + troff + 32 * wv);
                const s4v klf = tr_read(kb + K_KL * ARR + troff + 32 * wv);
                // gr accumulator [i_local = 4g+q][token x].  E8 scales key rows = output rows here, so it is applied to the 4
                // results instead of the 16 operand values (the state tiles are dead after this: split in place, no copy)
                const float4 e8o = *reinterpret_cast<const float4*>(rb + ROFF_E8 + (16 * wv + 4 * g) * 4);
                f4v accs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float t0[4], t1[4];
                    b8v hi, lo;                                  // k-slot (s, g, e) <-> value channel 32s + 8g + e
#pragma unroll
                    for (int q = 0; q < 4; ++q) { t0[q] = ST[blk][2 * s][q]; t1[q] = ST[blk][2 * s + 1][q]; }
                    split8(t0, t1, hi, lo, spc);
                    accs = mfma32(hi, gyr[s], accs);
                    accs = mfma32(lo, gyr[s], accs);
                }
                f4v accr16 = {0.f, 0.f, 0.f, 0.f};               // separate accumulator per MFMA shape (see wkv6_chunk.hip)
                accr16 = mfma16(khf, dba_hi, accr16);            // sum_b Khat[b][i] dA[a][b]
                accr16 = mfma16(khf, dba_lo, accr16);
                accr16 = mfma16(klf, dba_hi, accr16);
                const f4v accr = {fmaf(e8o.x, accs[0], accr16[0]), fmaf(e8o.y, accs[1], accr16[1]),
                                  fmaf(e8o.z, accs[2], accr16[2]), fmaf(e8o.w, accs[3], accr16[3])};
                f4v ak = {0.f, 0.f, 0.f, 0.f};
                ak = mfma16(rhf_w, dab_hi, ak);                  // sum_a Rhat[a][i] dA[a][b]
                ak = mfma16(rhf_w, dab_lo, ak);
                ak = mfma16(rlf_w, dab_hi, ak);
                ackp[blk] = ak;
                {   // gr, a_t, gu: lane = token x, channels ch .. ch+3
                    const int ch = 16 * wv + 4 * g;
                    const float4 fr4 = *reinterpret_cast<const float4*>(rb + ROFF_FR + x * FRS + ch * 4);
                    const uint2 rr = *reinterpret_cast<const uint2*>(rb + R_R * ARR + x * RSB + ((ch * 2) ^ ((x & 8) << 1)));
                    const uint2 kk = *reinterpret_cast<const uint2*>(rb + R_K * ARR + x * RSB + ((ch * 2) ^ ((x & 8) << 1)));
                    const float frv[4] = {fr4.x, fr4.y, fr4.z, fr4.w};
                    const float rv[4] = {bf_lo(rr.x), bf_hi(rr.x), bf_lo(rr.y), bf_hi(rr.y)};
                    const float kv[4] = {bf_lo(kk.x), bf_hi(kk.x), bf_lo(kk.y), bf_hi(kk.y)};
                    float o_gr[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float dq = frv[q] * accr[q];
                        o_gr[q] = fmaf(vg * ue[q], kv[q], dq);
                        gu_acc[q] = fmaf(vg * rv[q], kv[q], gu_acc[q]);
                        at[blk][q] = rv[q] * dq;
                    }
                    held_gr[blk] = emit(0, grp, blk, REV_R, o_gr, old_gr[blk]);
                }
            }
            WKV6_T(ts3);
            // the next stage's checkpoint: the state registers of this stage are dead from here on
            // (... and its gk, gw, a chain ahead)
#pragma unroll
            for (int blk = SBLK - 1; blk >= 0; --blk) {
                fetch_old(1, rs_gk, grp * STG + blk * BLK, REV_K, old_gk[blk]);
                fetch_old(3, rs_gw, grp * STG + blk * BLK, REV_W, old_gw[blk]);
            }
            WKV6_T5(2, tp0);
            // The even stage of a pair enters with the pair's checkpoint itself -- which its odd stage left untouched in CK (the rebuild
            // above writes ST[0], not CK): only an even stage requests a new one.  (Rounds 4: both stages read it, the second from the
            // L2 -- 16 KB per pair and workgroup through the CU's vector-memory pipe, which these kernels keep ~80 % busy:
            // profiles/r05_memory_pipe.md; same-box -1 %.)
            if (grp > 0 && !(grp & 1)) request_ckpt(grp - 1);
            if constexpr (CHAIN) {
                if (nx_chain && grp == 0) {   // the call that follows: the checkpoint of its last stage pair, a chain's length ahead of its first use
                    const rsrc_t rs_nx = make_rsrc(ch.nx_ckpt + ((long)ch.nx_b * a.H + (long)(ch.nx_bh % (unsigned)a.H)) * nslots * (HEAD * HEAD), nslots * 16384u);
                    const unsigned off = (unsigned)((nxn - 1) >> 1) * 16384u + (unsigned)wv * 4096u + (unsigned)lane * 16u;
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) {
                        const float4 t = buf_load16f(rs_nx, off + jt * 1024u);
                        CK[jt] = f4v{t.x, t.y, t.z, t.w};
                    }
                }
            }
            WKV6_T5(2, tp1); WKV6_ACC5(2, tp1, tp0);
            if constexpr (!SPLIT) {
                WKV6_EV(2);
                settle4(TAG_GC, grp + 1, pk_gc);
                WKV6_EV(3);
                gvb[0] = take_gop(0);
                // Released for the column waves (they lay the NEXT stage's first operand over it): this wave's reads are in the LDS
                // queue, which serves a wave's requests in order, and a store is never moved above a load it may alias.
                publish(TAG_GD, tagv);
            }
#pragma unroll
            for (int blk = SBLK - 1; blk >= 0; --blk) put(0, rs_gr, grp, blk, REV_R, held_gr[blk]);
            // ---- chain: only the work that needs G
#pragma unroll
            for (int blk = SBLK - 1; blk >= 0; --blk) {
                const char* const rb = rpart(grp, blk);
                const char* const kb = kpart(grp, blk);
                f4v acck = {0.f, 0.f, 0.f, 0.f};
                [[maybe_unused]] f4v Oi[4];
                [[maybe_unused]] float e8x = 0.f, e16x = 0.f;
                if constexpr (SPLIT) {
                    // the G-dependent part of gk from this workgroup's own copy of G (by rows), and the (gy^T Rhat) tiles of its update
                    e8x = *reinterpret_cast<const float*>(rb + ROFF_E8 + (16 * wv + x) * 4);
                    e16x = *reinterpret_cast<const float*>(kb + KOFF_E16 + (16 * wv + x) * 4);
                    const float e16m8x = *reinterpret_cast<const float*>(kb + KOFF_E16M8 + (16 * wv + x) * 4);
                    const s4v rhf_w = tr_read(rb + R_RH * ARR + troff + 32 * wv);  // Rhat[4g+e][16wv + x]
                    const s4v rlf_w = tr_read(rb + R_RL * ARR + troff + 32 * wv);
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) {             // [row j_local][col i_local = x]
                        const s4v gyf = tr_read(rb + R_GY * ARR + trow + tile_tr(jt));
                        f4v o = {0.f, 0.f, 0.f, 0.f};
                        o = mfma32(cat2(gyf, gyf), cat2(rhf_w, rlf_w), o);
                        Oi[jt] = o;
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const b8v vr = ld_b8(kb + K_V * ARR + x * RSB + (32 * s + 8 * g) * 2);
                        float t0[4], t1[4];
                        b8v hi, lo;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { t0[q] = GI[2 * s][q] * e16m8x; t1[q] = GI[2 * s + 1][q] * e16m8x; }
                        split8(t0, t1, hi, lo, spc);
                        acck = mfma32(hi, vr, acck);
                        acck = mfma32(lo, vr, acck);
                    }
                } else {
                    acck = gvb[blk];
                }
                acck += ackp[blk];
                {
                    const int ch = 16 * wv + 4 * g;
                    const float4 fn4 = *reinterpret_cast<const float4*>(x < 15 ? rb + ROFF_FR + (x + 1) * FRS + ch * 4 : kb + KOFF_E16M8 + ch * 4);
                    const float4 fk4 = make_float4(__builtin_amdgcn_rcpf(fn4.x), __builtin_amdgcn_rcpf(fn4.y), __builtin_amdgcn_rcpf(fn4.z),
                                                   __builtin_amdgcn_rcpf(fn4.w));      // fK_x = 1 / fR_{x+1}
                    const float4 lw4 = *reinterpret_cast<const float4*>(rb + ROFF_LW + x * FRS + ch * 4);
                    const uint2 rr = *reinterpret_cast<const uint2*>(rb + R_R * ARR + x * RSB + ((ch * 2) ^ ((x & 8) << 1)));
                    const uint2 kk = *reinterpret_cast<const uint2*>(rb + R_K * ARR + x * RSB + ((ch * 2) ^ ((x & 8) << 1)));
                    const float fkv[4] = {fk4.x, fk4.y, fk4.z, fk4.w}, lwv[4] = {lw4.x, lw4.y, lw4.z, lw4.w};
                    const float rv[4] = {bf_lo(rr.x), bf_hi(rr.x), bf_lo(rr.y), bf_hi(rr.y)};
                    const float kv[4] = {bf_lo(kk.x), bf_hi(kk.x), bf_lo(kk.y), bf_hi(kk.y)};
                    const float vg = vgs[blk];
                    float o_gk[4], o_gw[4], bt[4], dl[4], sfx[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float dk = fkv[q] * acck[q];
                        o_gk[q] = fmaf(vg * ue[q], rv[q], dk);
                        bt[q] = kv[q] * dk;
                        dl[q] = at[blk][q] - bt[q];
                        sfx[q] = dl[q];
                    }
                    WKV6_DPP_ACC4(sfx, "row_shl:1");      // inclusive suffix sums over the later tokens of the row
                    WKV6_DPP_ACC4(sfx, "row_shl:2");
                    WKV6_DPP_ACC4(sfx, "row_shl:4");
                    WKV6_DPP_ACC4(sfx, "row_shl:8");
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float total = dpp_mov<DPP_BCAST0>(sfx[q]);   // token 0 of the row holds the block total
                        o_gw[q] = (Rc[q] + (sfx[q] - dl[q]) - bt[q]) * lwv[q];
                        Rc[q] += total;
                    }
                    put(1, rs_gk, grp, blk, REV_K, emit(1, grp, blk, REV_K, o_gk, old_gk[blk]));
                    put(3, rs_gw, grp, blk, REV_W, emit(3, grp, blk, REV_W, o_gw, old_gw[blk]));
                }
                // ---- (SPLIT) G[i = 16wv + x][:] <- E16 G + E8 (Rhat^T gy)
                if constexpr (SPLIT) {
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) GI[jt][q] = fmaf(e16x, GI[jt][q], e8x * Oi[jt][q]);
                }
            }
            WKV6_T(ts4);
            __syncthreads();
            WKV6_T(ts5);
            WKV6_ACC(0, ts1, ts0); WKV6_ACC(1, ts2, ts1); WKV6_ACC(2, ts3, ts2); WKV6_ACC(3, ts4, ts3); WKV6_ACC(4, ts5, ts4);
        }
        if (a.gu) {
            float s4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) s4[q] = row_sum16(gu_acc[q]);
            const long o = (long)b * a.C + h * HEAD + 16 * wv + 4 * g;
            if (x == 0) {
                if (a.part_f32) io4<float>::store(reinterpret_cast<float*>(a.gu) + o, s4);
                else io4<bf16_t>::store(reinterpret_cast<bf16_t*>(a.gu) + o, s4);
            }
        }
    } else {
        // =============== value columns [16wv, 16wv+16): G, gv, gs ======================================
        // GJ[it][q] = G[i = tile_ch(it) + 8g + q][j = 16wv + x]
        f4v GJ[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float t4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.g_in)                                            // adjoint state entering this segment from the future, layout [j][i]
                io4<float>::load(a.g_in + ((long)b * a.H + h) * HEAD * HEAD + (long)(16 * wv + x) * HEAD + tile_ch(t) + 8 * g, t4);
            GJ[t] = f4v{t4[0], t4[1], t4[2], t4[3]};
        }
        // The wait for these loads belongs HERE.  A loop-carried register whose first value is a load still in flight makes hipcc put
        // its s_waitcnt at the first use INSIDE the stage loop, sized for the loop's first entry -- vmcnt(0) -- where it then waits,
        // every stage, for the requests that stage has just issued (measured: the operand of block 1 published 1100 cycles late).
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(GJ[t]));
        // this wave's share of the PREPARATION: v (K part, stage s-2) and gy (R part, stage s-1) of tokens 4wv .. 4wv+3 of both blocks go from
        // global memory into the images, with the two half sums of vg_a = gy_a . v_a (the v of a stage is kept in registers for the one
        // iteration until its gy arrives).  lane = (token tl, channels 4cl .. 4cl+3); requested at the head of a stage, written at its end.
        const int tl = lane >> 4, cl = lane & 15;
        uint2 cv[SBLK], cg[SBLK];
        uint2 (&cvp)[SBLK] = cy.cvp;                           // (BwdCarry: outlives the call in the persistent wkv6_bi launch)
        const int lp_vg = tok.lane(4 * wv + tl, 4 * cl, C_);
        auto load_v = [&](const RowIn& in, int lp, int stg) {
#pragma unroll
            for (int blk = 0; blk < SBLK; ++blk) {
                cv[blk] = buf_load8(in.v, in.tok.off(stg * STG + blk * BLK, 4 * wv + tl, 4 * cl, C_, REV_V, lp) * 2u);
            }
        };
        auto load_gy = [&](const RowIn& in, int lp, int stg) {
#pragma unroll
            for (int blk = 0; blk < SBLK; ++blk) {
                cg[blk] = buf_load8(in.g, in.tok.off(stg * STG + blk * BLK, 4 * wv + tl, 4 * cl, C_, REV_Y, lp) * 2u);
            }
        };
        auto copy_v = [&](char* const k0) {       // cv -> the K part whose block 0 is k0; kept in cvp for the stage's vg
#pragma unroll
            for (int blk = 0; blk < SBLK; ++blk) {
                *reinterpret_cast<uint2*>(k0 + blk * KBLK_BYTES + K_V * ARR + (4 * wv + tl) * RSB + 8 * cl) = cv[blk];
                cvp[blk] = cv[blk];
            }
        };
        auto copy_gy = [&](char* const r0) {      // cg -> the R part whose block 0 is r0, with vg from cvp (= v of the same stage)
#pragma unroll
            for (int blk = 0; blk < SBLK; ++blk) {
                char* const rb = r0 + blk * RBLK_BYTES;
                const int tok = 4 * wv + tl;
                *reinterpret_cast<uint2*>(rb + R_GY * ARR + tok * RSB + 8 * cl) = cg[blk];
                typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                float vg = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, cg[blk].x), __builtin_bit_cast(bf2, cvp[blk].x), 0.f, false);
                vg = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, cg[blk].y), __builtin_bit_cast(bf2, cvp[blk].y), vg, false);
                vg += dpp_mov<DPP_XOR1>(vg);                       // the two 32-channel half sums, formed exactly as the producers of the
                vg += dpp_mov<DPP_XOR2>(vg);                       // split kernels form them (same results bit for bit)
                vg += dpp_mov<DPP_SHL4>(vg);
                if ((cl & 7) == 0) *reinterpret_cast<float*>(rb + ROFF_VG + ((cl >> 3) * 16 + tok) * 4) = vg;
            }
        };
        // (persistent wkv6_bi launch: the call that follows -- this wave moves its first v / gy rows during this call's last two stages)
        [[maybe_unused]] RowIn nx = cur;
        [[maybe_unused]] int lp_vg_nx = lp_vg;
        if constexpr (CHAIN) {
            if (nx_chain) {
                nx = row_in_of(ch.nx_b, (int)(ch.nx_bh % (unsigned)a.H), ch.nx_ntok, ch.nx_rev);
                lp_vg_nx = nx.tok.lane(4 * wv + tl, 4 * cl, C_);
            }
        }
        // (E16m8 of the block whose K part is kb) (.) G, split into the bf16 hi | lo fragments of the two k-steps
        auto scale_split = [&](const char* kb, b8v (&h)[2], b8v (&l)[2]) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float t0[4], t1[4];
                const float4 m0 = *reinterpret_cast<const float4*>(kb + KOFF_E16M8 + (32 * s + 8 * g) * 4);
                const float4 m1 = *reinterpret_cast<const float4*>(kb + KOFF_E16M8 + (32 * s + 8 * g + 4) * 4);
                t0[0] = GJ[2 * s][0] * m0.x; t0[1] = GJ[2 * s][1] * m0.y; t0[2] = GJ[2 * s][2] * m0.z; t0[3] = GJ[2 * s][3] * m0.w;
                t1[0] = GJ[2 * s + 1][0] * m1.x; t1[1] = GJ[2 * s + 1][1] * m1.y;
                t1[2] = GJ[2 * s + 1][2] * m1.z; t1[3] = GJ[2 * s + 1][3] * m1.w;
                split8(t0, t1, h[s], l[s], spc);
            }
        };
        auto gop_write = [&](const b8v (&h)[2], const b8v (&l)[2]) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                char* const dst = smem + GOP_OFF + (16 * wv + x) * GRS + (32 * s + 8 * g) * 2;
                *(lds_vv4u*)dst = __builtin_bit_cast(v4u_t, h[s]);
                *(lds_vv4u*)(dst + HEAD * GRS) = __builtin_bit_cast(v4u_t, l[s]);
            }
        };
        [[maybe_unused]] b8v nh[2] = {}, nl[2] = {};                // block 1's operand of the NEXT stage, across the barrier
        if (!SPLIT && !chained_in && ngrp > 0) {                   // (SPLIT: the producers of this workgroup move v and gy)
            load_v(cur, lp_vg, ngrp - 1);
            load_gy(cur, lp_vg, ngrp - 1);
            copy_v(kpart(ngrp - 1, 0));
            copy_gy(rpart(ngrp - 1, 0));
            if (ngrp > 1) {
                load_v(cur, lp_vg, ngrp - 2);
                copy_v(kpart(ngrp - 2, 0));
            }
        }
        __syncthreads();                                          // first stage image is ready
        if constexpr (!SPLIT) { if (ngrp > 0) scale_split(kpart(ngrp - 1, SBLK - 1), nh, nl); }
        for (int grp = ngrp - 1; grp >= 0; --grp) {
            int tagv = grp + 1;                                    // this stage's tag value, in a vector register (see publish)
            // (column wave w enters a stage w x 128 cycles late: the four waves' operand reads and G updates no longer burst at the same moment --
            // same box -1 %, 64 / 192 cycles -0.4 %, 256 and more slower, the row waves staggered the same way +0.5 ... 1 %: profiles/r05_stagger.txt)
            for (int i_ = 0; i_ < wv; ++i_) __builtin_amdgcn_s_sleep(2);
            asm volatile("" : "+v"(tagv));
            WKV6_T(ts0);

            // ---- the stage's G recurrence: per block (1, then 0) the operand (E16m8 (.) G) hi | lo for the row waves (stored [j][i]: this
            // lane's 16 bytes are key rows 32s + 8g .. +7 of value column 16wv + x), kept as fragments for this wave's own chain, then G
            // moves to the entry of the block.  Block 1's operand of a stage is published in the TAIL of the stage before it (below):
            // a tag takes ~1000 cycles from one wave's LDS queue into another's poll while the LDS is busy, and the row waves' wait for
            // block 0's operand -- which waits for their release of block 1's -- was the longest stall of the stage
            // (profiles/r04_handover_timeline.txt).  Only the first stage publishes it here.
            s4v gyT_w[SBLK], sc_hi[SBLK], sc_lo[SBLK];
            f4v accp[SBLK];
            b8v gh[SBLK][2], gl[SBLK][2];
            [[maybe_unused]] f4v Og[SBLK][4];
            if constexpr (!SPLIT)
#pragma unroll
            for (int blk = SBLK - 1; blk >= 0; --blk) {
                const char* const rb = rpart(grp, blk);
                const char* const kb = kpart(grp, blk);
                if (blk == SBLK - 1) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) { gh[blk][s] = nh[s]; gl[blk][s] = nl[s]; }   // made (and published) in the previous stage's tail
                    if (grp == ngrp - 1) {
                        gop_write(gh[blk], gl[blk]);
                        publish(TAG_GA, tagv);
                    }
                    WKV6_EV(0);
                    WKV6_T5(2, tp0);
                    // this wave's requests for the stages behind (CHAIN: past this call's first stage they go to the call that follows --
                    // stage 1 requests its last stage's v, stage 0 its last stage's gy and the v of the stage in front)
                    if constexpr (CHAIN) {
                        // (unconditional, from a SELECTED row: as branches the two requests make cv / cg merges of "loaded here" and "loaded there",
                        // which hipcc resolves with an s_waitcnt vmcnt(0) inside the stage loop; past the row's first stage an unchained call's
                        // requests lie outside its buffer resources and cost nothing)
                        const bool g_nx = nx_chain && grp == 0, v_nx = nx_chain && grp <= 1;
                        load_gy(g_nx ? nx : cur, g_nx ? lp_vg_nx : lp_vg, g_nx ? nxn - 1 : grp - 1);
                        load_v(v_nx ? nx : cur, v_nx ? lp_vg_nx : lp_vg, v_nx ? (grp == 1 ? nxn - 1 : nxn - 2) : grp - 2);
                    } else {
                        if (grp > 0) load_gy(cur, lp_vg, grp - 1);
                        if (grp > 1) load_v(cur, lp_vg, grp - 2);
                    }
                    WKV6_T5(2, tp1); WKV6_ACC5(2, tp1, tp0);
                } else {
                    scale_split(kb, gh[blk], gl[blk]);
                    WKV6_EV(1);
                    await4(TAG_GB, grp + 1);     // the row waves have taken block 1's version
                    WKV6_EV(2);
                    gop_write(gh[blk], gl[blk]);
                    publish(TAG_GC, tagv);
                    WKV6_EV(3);
                }
                const s4v gyT = tr_read(rb + R_GY * ARR + troff + 32 * wv);           // gy[4g+e][16wv + x]
#pragma unroll
                for (int it = 0; it < 4; ++it) {              // (Rhat^T gy) tile [row i_local][col j_local = x];  G <- E16 G + E8 (Rhat^T gy)
                    const s4v rhf = tr_read(rb + R_RH * ARR + trow + tile_tr(it));
                    const s4v rlf = tr_read(rb + R_RL * ARR + trow + tile_tr(it));
                    f4v o = {0.f, 0.f, 0.f, 0.f};
                    o = mfma32(cat2(rhf, rlf), cat2(gyT, gyT), o);        // (Rhat_hi | Rhat_lo) (gy ; gy)
                    const float4 d16 = *reinterpret_cast<const float4*>(kb + KOFF_E16 + (tile_ch(it) + 8 * g) * 4);
                    const float4 d8 = *reinterpret_cast<const float4*>(rb + ROFF_E8 + (tile_ch(it) + 8 * g) * 4);
                    GJ[it][0] = fmaf(d16.x, GJ[it][0], d8.x * o[0]);
                    GJ[it][1] = fmaf(d16.y, GJ[it][1], d8.y * o[1]);
                    GJ[it][2] = fmaf(d16.z, GJ[it][2], d8.z * o[2]);
                    GJ[it][3] = fmaf(d16.w, GJ[it][3], d8.w * o[3]);
                }
            }
            float old_gv[SBLK][4] = {};                            // (second half of wkv6_bi: the first half's gv, a pre-phase ahead)
#pragma unroll
            for (int blk = SBLK - 1; blk >= 0; --blk) fetch_old(2, rs_gv, grp * STG + blk * BLK, REV_V, old_gv[blk]);
            // ---- pre-phase: everything that does not depend on G
            if (SPLIT && wv < SBLK) {   // (no row waves in this workgroup) this wave's score tile of the stage: block wv
                const char* const rb = rpart(grp, wv);
                const char* const kb = kpart(grp, wv);
                f4v sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int off = x * RSB + (32 * s + 8 * g) * 2;
                    const b8v rh = ld_b8(rb + R_RH * ARR + off), rl = ld_b8(rb + R_RL * ARR + off);
                    const b8v kh = ld_b8(kb + K_KH * ARR + off), kl = ld_b8(kb + K_KL * ARR + off);
                    sc = mfma32(rh, kh, sc);                  // A[row a][col b]: lane col b = x, rows a = 4g+q
                    sc = mfma32(rh, kl, sc);
                    sc = mfma32(rl, kh, sc);
                }
                const float4 cfa = *reinterpret_cast<const float4*>(rb + ROFF_COEF + 16 * g);
                const float4 cfb = *reinterpret_cast<const float4*>(rb + ROFF_COEF + 64 + 16 * g);
                const float cf[4] = {cfa.x + cfb.x, cfa.y + cfb.y, cfa.z + cfb.z, cfa.w + cfb.w};
                float scm[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = 4 * g + q;                  // query token a; key token b = x
                    scm[q] = x < o ? sc[q] : (x == o ? cf[q] : 0.f);
                }
                uint2 th, tl;
                split4(scm, th, tl, spc);
                tile_store(XS_OFF + wv * 1024, th, tl);
                publish(TAG_SC, tagv);
            }
#pragma unroll
            for (int blk = 0; blk < SBLK; ++blk) gyT_w[blk] = tr_read(rpart(grp, blk) + R_GY * ARR + troff + 32 * wv);   // gy[4g+e][16wv + x]
#pragma unroll
            for (int blk = 0; blk < SBLK; ++blk) {
                if (blk == 0) await2(TAG_SC, grp + 1);        // both score tiles with one poll
                const uint4 f = tile_load(XS_OFF + blk * 1024);
                sc_hi[blk] = __builtin_bit_cast(s4v, make_uint2(f.x, f.y));
                sc_lo[blk] = __builtin_bit_cast(s4v, make_uint2(f.z, f.w));
            }
#pragma unroll
            for (int blk = 0; blk < SBLK; ++blk) {
                f4v acc = {0.f, 0.f, 0.f, 0.f};               // gv^T[j][b], first part: sum_a gy[a][j] A[a][b]
                acc = mfma32(cat2(gyT_w[blk], gyT_w[blk]), cat2(sc_hi[blk], sc_lo[blk]), acc);     // (gy | gy) (A_hi ; A_lo)
                accp[blk] = acc;
                if constexpr (SPLIT) {
                    const char* const rb = rpart(grp, blk);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {              // (Rhat^T gy) tile: [row i_local][col j_local = x]
                        const s4v rhf = tr_read(rb + R_RH * ARR + trow + tile_tr(it));
                        const s4v rlf = tr_read(rb + R_RL * ARR + trow + tile_tr(it));
                        f4v o = {0.f, 0.f, 0.f, 0.f};
                        o = mfma32(cat2(rhf, rlf), cat2(gyT_w[blk], gyT_w[blk]), o);
                        Og[blk][it] = o;
                    }
                }
            }
            WKV6_T(ts1);
            // ---- chain: only the work that needs G
            uint2 held_gv[SBLK];
#pragma unroll
            for (int blk = SBLK - 1; blk >= 0; --blk) {
                const char* const kb = kpart(grp, blk);
                f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int off = x * RSB + (32 * s + 8 * g) * 2;
                    const b8v kh = ld_b8(kb + K_KH * ARR + off), kl = ld_b8(kb + K_KL * ARR + off);
                    if constexpr (SPLIT) {                     // this workgroup's own G recurrence runs here, in the chain
                        float t0[4], t1[4];
                        const float4 m0 = *reinterpret_cast<const float4*>(kb + KOFF_E16M8 + (32 * s + 8 * g) * 4);
                        const float4 m1 = *reinterpret_cast<const float4*>(kb + KOFF_E16M8 + (32 * s + 8 * g + 4) * 4);
                        t0[0] = GJ[2 * s][0] * m0.x; t0[1] = GJ[2 * s][1] * m0.y; t0[2] = GJ[2 * s][2] * m0.z; t0[3] = GJ[2 * s][3] * m0.w;
                        t1[0] = GJ[2 * s + 1][0] * m1.x; t1[1] = GJ[2 * s + 1][1] * m1.y;
                        t1[2] = GJ[2 * s + 1][2] * m1.z; t1[3] = GJ[2 * s + 1][3] * m1.w;
                        split8(t0, t1, gh[blk][s], gl[blk][s], spc);
                    }
                    acc = mfma32(gh[blk][s], kh, acc);       // k-slot (s, g, e) <-> key channel 32s + 8g + e
                    acc = mfma32(gh[blk][s], kl, acc);
                    acc = mfma32(gl[blk][s], kh, acc);
                }
                acc += accp[blk];
                {
                    float o[4] = {acc[0], acc[1], acc[2], acc[3]};
                    held_gv[blk] = emit(2, grp, blk, REV_V, o, old_gv[blk]);
                }
                // ---- (SPLIT) G[:][j = 16wv + x] <- E16 G + E8 (Rhat^T gy)
                if constexpr (SPLIT) {
                    const char* const rb = rpart(grp, blk);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const float4 d16 = *reinterpret_cast<const float4*>(kb + KOFF_E16 + (tile_ch(it) + 8 * g) * 4);
                        const float4 d8 = *reinterpret_cast<const float4*>(rb + ROFF_E8 + (tile_ch(it) + 8 * g) * 4);
                        GJ[it][0] = fmaf(d16.x, GJ[it][0], d8.x * Og[blk][it][0]);
                        GJ[it][1] = fmaf(d16.y, GJ[it][1], d8.y * Og[blk][it][1]);
                        GJ[it][2] = fmaf(d16.z, GJ[it][2], d8.z * Og[blk][it][2]);
                        GJ[it][3] = fmaf(d16.w, GJ[it][3], d8.w * Og[blk][it][3]);
                    }
                }
            }
#pragma unroll
            for (int blk = SBLK - 1; blk >= 0; --blk) put(2, rs_gv, grp, blk, REV_V, held_gv[blk]);
            if constexpr (!SPLIT) {
                if (grp > 0) {   // block 1's operand of the next stage (G is at its entry since block 0's update; its K part has been in the ring for a stage)
                    scale_split(kpart(grp - 1, SBLK - 1), nh, nl);
                    await4(TAG_GD, grp + 1);                       // every row wave has taken block 0's operand of this stage
                    gop_write(nh, nl);
                    publish(TAG_GA, tagv - 1);                     // (ordered before the row waves' reads by the stage barrier as well)
                }
            }
            if constexpr (!SPLIT) {
                if (grp > 0) copy_gy(rpart(grp - 1, 0));           // (uses cvp = v of stage s-1, before copy_v replaces it)
                else if (CHAIN && nx_chain) copy_gy(rpart_nx(0, 0));
                if (grp > 1) copy_v(kpart(grp - 2, 0));
                else if (CHAIN && nx_chain) copy_v(grp == 1 ? kpart_nx(0, 0) : kpart_nx(1, 0));
            }
            WKV6_T(ts2);
            __syncthreads();
            WKV6_T(ts3);
            WKV6_ACC(0, ts1, ts0); WKV6_ACC(1, ts2, ts1); WKV6_ACC(2, ts3, ts2);
        }
        if (a.gs) {   // dL/dS0, layout [j][i]
            const long so_ = ((long)b * a.H + h) * HEAD * HEAD + (long)(16 * wv + x) * HEAD + 8 * g;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float t4[4] = {GJ[it][0], GJ[it][1], GJ[it][2], GJ[it][3]};
                if (a.part_f32) io4<float>::store(reinterpret_cast<float*>(a.gs) + so_ + tile_ch(it), t4);
                else io4<bf16_t>::store(reinterpret_cast<bf16_t*>(a.gs) + so_ + tile_ch(it), t4);
            }
        }
    }
#ifdef WKV6_DEBUGBUF
    WKV6_CLK(clk1, rtc1);
    if (a.aux && lane == 0) {
        unsigned long long* const d = reinterpret_cast<unsigned long long*>(a.aux) + ((long)bh * 16 + wid) * 8;
#ifdef WKV6_STAMP
        for (int i = 0; i < 6; ++i) d[i] = stamp_acc[i];
#if WKV6_STAMP5 == 3
        if (wid < 8) for (int i = 0; i < 4; ++i) d[(12 + (wid >> 1) - wid) * 8 + 4 * (wid & 1) + i] = poll_acc[i];  // records 12 .. 15: four events of waves 0 .. 7
#else
        if (wid == 0 || wid == 4) for (int i = 0; i < 8; ++i) d[(12 + (wid >> 2) - wid) * 8 + i] = poll_acc[i];   // records 12 / 13
#endif
#endif
        d[6] = clk1 - clk0;
        d[7] = rtc1 - rtc0;
    }
#endif
    if constexpr (CLK) { if (hwid == 0) clock_stamp(a, slot, 1); }
    if (GEN == 1 && a.zero_tail && part == 0) {
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

template <bool W_RAW, int GEN, bool SPLIT, bool AFF>
__global__ __launch_bounds__(SPLIT ? 512 : 768) void chunk_bwd12k_kernel(const ScanArgs a)
{
    BwdCarry cy;       // (registers that outlive a call only in the persistent wkv6_bi launch: plain locals here)
    chunk_bwd12k_body<W_RAW, GEN, SPLIT, AFF, GEN == 0>(a, blockIdx.x, cy);
}

// the backward of chunk_fwd_pair_kernel (wkv6_chunk.hip): two problems of one shape, slots [0, B H) serve a0, the rest a1
template <bool W_RAW>
__global__ __launch_bounds__(768) void chunk_bwd12k_pair_kernel(const ScanArgs a0, const ScanArgs a1)
{
    const unsigned n = (unsigned)(a0.B * a0.H);
    const bool second = blockIdx.x >= n;
    BwdCarry cy;
    chunk_bwd12k_body<W_RAW, 0, false, false>(second ? a1 : a0, second ? blockIdx.x - n : blockIdx.x, cy);   // (per-tensor reversal maps: general addressing)
}

// Both halves of wkv6_bi in one persistent launch (cuda/wkv6_bi_cuda.cu:363-377 runs one forward and three backward launches): workgroup
// slot s walks the rows s, s + slots, ... of the (length-ordered) batch x head list; per row the forward-direction adjoint goes to the slot's
// fp32 scratch (GEN = 1), the reversed-direction adjoint adds it and rounds once (GEN = 2).  The partials of a row are 4 x T x 256 B
// (512 KB at T = 512) per slot, 128 MB for 256 slots: written and read back within ~50 us by the same CU, they live in the L2 and the
// Infinity Cache instead of making a round trip through HBM (two launches: 16 + 16 B per token-channel).
// (ONE argument block: the reversed-direction problem is the forward-direction one with `reverse`, no bonus, accumulation, no tail
// zeroing, no gu and its own checkpoints -- two blocks kept both in scalar registers across the row loop and spilled 200 of them.)
// (Argument plumbing, round 6.  The two bodies read ~45 fields of the argument block; hipcc hoists those kernarg loads out of the row loop
// and keeps them in scalar registers across both calls -- beside each call's ~14 buffer resources: 134 scalar spills into vector-register
// lanes and, with no vector registers left either, 26 of those to scratch (406 v_readlane / v_writelane, 63 scratch instructions around
// every call; fp32-ew worse: reloads inside the stage loops).  Here every call re-reads the block it needs from the kernarg segment
// through a pointer the compiler cannot see through (laundered per call), so nothing of it is live from call to call: each call opens
// with a handful of s_load_dwordx8/16 -- a few hundred cycles against the ~8 us of a call.)
template <bool W_RAW>
__global__ __launch_bounds__(768) void chunk_bwd12k_bi_kernel(const ScanArgs a1_, float* const ckpt2)
{
    const unsigned n = (unsigned)(a1_.B * a1_.H);
    const int H_ = a1_.H, T_ = a1_.T;
    const int* const order_ = a1_.order;
    const int* const lens_ = a1_.lens;
    // the rows are ordered by decreasing length (a.order): slot j takes row j of the first round of gridDim.x rows, row gridDim.x - 1 - j
    // of the second, ... (boustrophedon), so that every slot gets long and short rows alike -- in plain round-robin order slot 0 would
    // take the longest row of every round and the last slot the shortest (+-12 % of the mean at BASELINE configs[2])
    const auto row_of = [&](unsigned it) { return it * gridDim.x + ((it & 1) ? gridDim.x - 1 - blockIdx.x : blockIdx.x); };
    const auto lookup = [&](unsigned row, int& b, int& ntok) {      // (one row ahead: see BwdChain)
        b = 0; ntok = 0;
        if (row < n) {
            b = order_ ? order_[row / H_] : (int)(row / H_);
            ntok = lens_ ? min(max(lens_[b], 0), T_) : T_;
        }
    };
    const int use_u_ = a1_.use_u;
    const float* const ckpt1_ = a1_.ckpt;
    // one copy of the row loop per role (see ROLE at chunk_bwd12k_body): the three run the same calls in the same order and meet at the same barriers
    const auto rows = [&](auto role_c) {
        constexpr int ROLE = decltype(role_c)::value;
        int b_cur, ntok_cur;
        lookup(row_of(0), b_cur, ntok_cur);
        BwdCarry cy;                                                    // the registers that outlive a call (chained calls: BwdChain)
        int ko = 0, ro = 0;                                             // stage -> buffer offsets of the call that starts: they rotate on across the calls
        bool chained = false;                                           // ... which the call before it has prepared for
        const auto mod3 = [](int x) { return ((x % KRING) + KRING) % KRING; };
        for (unsigned it = 0; it * gridDim.x < n; ++it) {
            const unsigned row = row_of(it);
            if (row >= n) continue;                                     // (the last round may be short; workgroup-uniform)
            const unsigned row_nx = row_of(it + 1);
            int b_nx, ntok_nx;
            lookup(row_nx, b_nx, ntok_nx);
            const int n_cur = (ntok_cur + STG - 1) / STG, n_nx = (ntok_nx + STG - 1) / STG;
            // a call prepares the one behind it when both have at least two stages (the producers' idle time is a call's last two stages)
            const bool chain12 = n_cur >= 2, chain2n = row_nx < n && n_cur >= 2 && n_nx >= 2;
            {
                ScanArgs a1;
                load_kernargs(a1);
                chunk_bwd12k_body<W_RAW, 1, false, true, false, true, ROLE>(a1, row, cy, blockIdx.x,
                                                                            BwdChain{b_cur, ntok_cur, ko, ro, chained, chain12, row, b_cur, ntok_cur, true, false, ckpt2});
            }
            ko = mod3(ko - n_cur); ro = (ro - n_cur) & 1;               // (the call that starts next is this row's reversed half: as many stages)
            // (the producer waves write no global memory, and a chained call has left their next requests in flight: they take no part in the fences)
            if constexpr (ROLE != 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's partial stores are out ...
            __syncthreads();                                            // ... and every wave's (the LDS tags are re-armed behind this barrier)
            if constexpr (ROLE != 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            {
                ScanArgs a2;
                load_kernargs(a2);
                a2.reverse = 1; a2.use_u = 0; a2.accumulate = 1; a2.zero_tail = 0; a2.gu = nullptr; a2.ckpt = ckpt2;
                chunk_bwd12k_body<W_RAW, 2, false, true, false, true, ROLE>(a2, row, cy, blockIdx.x,
                                                                            BwdChain{b_cur, ntok_cur, ko, ro, chain12, chain2n, row_nx, b_nx, ntok_nx, false, use_u_ != 0, ckpt1_});
            }
            ko = mod3(ko - n_nx); ro = (ro - n_nx) & 1;                 // (the call that starts next is the next row's first half)
            chained = chain2n;
            __syncthreads();
            b_cur = b_nx; ntok_cur = ntok_nx;
        }
    };
    const int hw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (hw < 4) rows(std::integral_constant<int, 1>{});
    else if (hw < 8) rows(std::integral_constant<int, 2>{});
    else rows(std::integral_constant<int, 3>{});
}

template <bool W_RAW, int GEN, bool AFF> hipError_t launch_bwd12k_inst2(const ScanArgs& a, hipStream_t st)
{
    constexpr size_t lds = BWD12K_LDS;
    static LdsAttrOnce attr, attr_split;       // per instantiation and device
    if (a.split) {
        if (hipError_t e = attr_split.ensure(reinterpret_cast<const void*>(chunk_bwd12k_kernel<W_RAW, GEN, true, AFF>), lds)) return e;
        hipLaunchKernelGGL((chunk_bwd12k_kernel<W_RAW, GEN, true, AFF>), dim3(2 * a.B * a.H), dim3(512), lds, st, a);
    } else {
        if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(chunk_bwd12k_kernel<W_RAW, GEN, false, AFF>), lds)) return e;
        hipLaunchKernelGGL((chunk_bwd12k_kernel<W_RAW, GEN, false, AFF>), dim3(a.B * a.H), dim3(768), lds, st, a);
    }
    return hipGetLastError();
}
// per-tensor reversal maps (a.rev_n: the compositions' *_rev_ex calls) take the general token addressing; the halves of wkv6_bi never
// carry one
template <bool W_RAW, int GEN> hipError_t launch_bwd12k_inst(const ScanArgs& a, hipStream_t st)
{
    if constexpr (GEN == 0) { if (a.rev_n) return launch_bwd12k_inst2<W_RAW, 0, false>(a, st); }
    else if (a.rev_n) return hipErrorNotSupported;
    return launch_bwd12k_inst2<W_RAW, GEN, true>(a, st);
}
template <bool W_RAW> hipError_t launch_bwd12k_variant(const ScanArgs& a, hipStream_t st)
{
    const bool side = a.g_f32[0] && a.g_f32[1] && a.g_f32[2] && a.g_f32[3];
    if (a.accumulate) return side ? launch_bwd12k_inst<W_RAW, 2>(a, st) : hipErrorInvalidValue;   // the halves of wkv6_bi meet in fp32 side buffers
    const bool first_half = a.zero_tail || a.g_f32[0] || a.g_f32[1] || a.g_f32[2] || a.g_f32[3];
    if (first_half && !side) return hipErrorInvalidValue;
    return first_half ? launch_bwd12k_inst<W_RAW, 1>(a, st) : launch_bwd12k_inst<W_RAW, 0>(a, st);
}

}  // namespace

hipError_t launch_chunk_state_pass(const ScanArgs& a, hipStream_t st);   // wkv6_chunk.hip

// two workgroups per (batch, head) when one each would leave at least half of the CUs without work
int cu_count()              // compute units of the current device (0: unknown)
{
    static int cus[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus[dev] = prop.multiProcessorCount;
    }
    return cus[dev];
}
int want_split(int BH)
{
    if (const char* e = getenv("WKV6_SPLIT")) return atoi(e) != 0;     // A/B switch
    return 2 * BH <= cu_count();
}

// reverse pass over 64-token row-order checkpoints (a.ckpt filled by the forward or by launch_chunk_state_pass); a.split as given
hipError_t launch_chunk_bwd12k(const ScanArgs& a_, hipStream_t st)
{
    if (!a_.ckpt) return hipErrorInvalidValue;
    if (a_.split && (a_.g_in || a_.rc_in)) return hipErrorInvalidValue;     // (segment rows of a two-level scan run one workgroup each)
    ScanArgs a = a_;
    a.clk = clock_claim(1, &a.clk_slots);
#ifdef WKV6_DEBUGBUF
    a.aux = reinterpret_cast<float*>(g_stamp_buffer);
#endif
    return a.wkind ? launch_bwd12k_variant<true>(a, st) : launch_bwd12k_variant<false>(a, st);
}

hipError_t launch_chunk_bwd(const ScanArgs& a_, hipStream_t st)
{
    ScanArgs a = a_;
    a.split = want_split(a.B * a.H);
    if (a.wkind != 1 && ((long)a.T + 64) * a.C >= (1L << 30)) return hipErrorInvalidValue;   // 32-bit byte offsets of the fp32 decay input
    if (!a.ckpt_valid) {                   // self-contained backward: state pass first (same inputs, no outputs)
        ScanArgs sp = a;
        sp.y = nullptr; sp.y_f32 = nullptr; sp.s_out = nullptr; sp.accumulate = 0; sp.zero_tail = 0;
        if (hipError_t e = launch_chunk_state_pass(sp, st)) return e;
    }
    return launch_chunk_bwd12k(a, st);
}

int bi_slots(int BH)
{
    if (want_split(BH)) return 0;
    if (const char* e = getenv("WKV6_BI_FUSED")) { if (atoi(e) == 0) return 0; }     // A/B switch: 0 = the halves as two launches
    const int cus = cu_count();
    return cus > 0 ? (BH < cus ? BH : cus) : 0;
}

hipError_t launch_chunk_bwd_bi(const ScanArgs& a1_, const ScanArgs& a2_, int* slots, hipStream_t st)
{
    const int n = bi_slots(a1_.B * a1_.H);
    if (slots) *slots = n;
    if (!n || !a1_.ckpt || !a2_.ckpt || a1_.g_in || a1_.rc_in) return hipErrorNotSupported;
    if (a1_.wkind != 1 && ((long)a1_.T + 64) * a1_.C >= (1L << 30)) return hipErrorInvalidValue;
    ScanArgs a1 = a1_, a2 = a2_;
    for (ScanArgs* a : {&a1, &a2}) {
        a->split = 0;
        if (!a->ckpt_valid) {                   // self-contained: this half's state pass first (same inputs, no outputs)
            ScanArgs sp = *a;
            sp.y = nullptr; sp.y_f32 = nullptr; sp.s_out = nullptr; sp.accumulate = 0; sp.zero_tail = 0;
            if (hipError_t e = launch_chunk_state_pass(sp, st)) return e;
        }
        a->side_compact = 1;
#ifdef WKV6_DEBUGBUF
        a->aux = reinterpret_cast<float*>(g_stamp_buffer);
#endif
    }
    // (both decay kinds since round 6: the reference-signature symbols pass fp32 ew = -exp(w), cuda/wkv6_bi_op.cpp:5-13)
    constexpr size_t lds = BWD12K_LDS;
    static LdsAttrOnce attr_raw, attr_ew;
    if (a1.wkind == 1) {
        if (hipError_t e = attr_raw.ensure(reinterpret_cast<const void*>(chunk_bwd12k_bi_kernel<true>), lds)) return e;
        hipLaunchKernelGGL((chunk_bwd12k_bi_kernel<true>), dim3(n), dim3(768), lds, st, a1, a2.ckpt);
    } else {
        if (hipError_t e = attr_ew.ensure(reinterpret_cast<const void*>(chunk_bwd12k_bi_kernel<false>), lds)) return e;
        hipLaunchKernelGGL((chunk_bwd12k_bi_kernel<false>), dim3(n), dim3(768), lds, st, a1, a2.ckpt);
    }
    return hipGetLastError();
}

// Backward of both problems of a bidirectional composition in one launch; both checkpoint sets must come from the forward
// (ckpt_valid).  Two launches where a (batch, head) pair is split over two workgroups or the two-level kernel is selected.
hipError_t launch_chunk_bwd_pair(const ScanArgs& a0_, const ScanArgs& a1_, hipStream_t st)
{
    if (a0_.B != a1_.B || a0_.T != a1_.T || a0_.C != a1_.C || a0_.H != a1_.H || a0_.wkind != a1_.wkind) return hipErrorInvalidValue;
    const auto plain = [](const ScanArgs& a) {
        return !a.accumulate && !a.zero_tail && !a.g_f32[0] && !a.g_f32[1] && !a.g_f32[2] && !a.g_f32[3] && a.ckpt && a.ckpt_valid;
    };
    if (!plain(a0_) || !plain(a1_)) return hipErrorNotSupported;
    if (want_split(a0_.B * a0_.H)) {
        if (hipError_t e = launch_chunk_bwd(a0_, st)) return e;
        return launch_chunk_bwd(a1_, st);
    }
    if (a0_.wkind != 1 && ((long)a0_.T + 64) * a0_.C >= (1L << 30)) return hipErrorInvalidValue;
    ScanArgs a0 = a0_, a1 = a1_;
    a0.split = a1.split = 0;
#ifdef WKV6_DEBUGBUF
    a0.aux = a1.aux = reinterpret_cast<float*>(g_stamp_buffer);
#endif
    constexpr size_t lds = BWD12K_LDS;
    static LdsAttrOnce attr_raw, attr_ew;
    if (a0.wkind == 1) {
        if (hipError_t e = attr_raw.ensure(reinterpret_cast<const void*>(chunk_bwd12k_pair_kernel<true>), lds)) return e;
        hipLaunchKernelGGL((chunk_bwd12k_pair_kernel<true>), dim3(2 * a0.B * a0.H), dim3(768), lds, st, a0, a1);
    } else {
        if (hipError_t e = attr_ew.ensure(reinterpret_cast<const void*>(chunk_bwd12k_pair_kernel<false>), lds)) return e;
        hipLaunchKernelGGL((chunk_bwd12k_pair_kernel<false>), dim3(2 * a0.B * a0.H), dim3(768), lds, st, a0, a1);
    }
    return hipGetLastError();
}

}  // namespace wkv6
