// Chunked MFMA backward of WKV6 for gfx950 (bf16 I/O), 16-wave version.  Same block algebra as the forward
// (wkv6_chunk.hip) and as documented below; what differs from a textbook chunked backward is the work split.
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
// Machine facts that shape the kernel (measured on MI355X, scratch microbenchmarks in DESIGN.md section 4): fp32
// VALU issues at ~2.3 cycles per wave64 instruction, integer / DPP / packed-f32 / cvt / transcendental ops at ~4,
// either MFMA shape at 16; the block algebra costs ~3.7 wave-instructions per token-channel, so the kernel is
// bound by instruction issue and dependency latency, not by HBM, LDS or the matrix pipe.  Hence: many waves per
// SIMD, balanced roles, and operand preparation overlapped with the chains instead of serialised before them.
//
// One 1024-thread workgroup (16 wave64, four per SIMD, <= 128 VGPRs) per (batch, head) walks 32-token stages
// (two 16-token blocks) backwards through a double-buffered LDS image; one workgroup barrier per stage.
// Four roles of four waves (one of each role per SIMD):
//   R  waves own key rows [16w, 16w+16) of the FORWARD state (lane = key row).  They need no G: they start each
//      stage from the checkpoint the forward kernel left (fp32, every 32 tokens) and walk its two blocks forwards:
//      gr, gu, and a_t = r (.) dq, which they leave in LDS (in place of fR) for the K wave of the same rows;
//   K  waves own the same key rows of G (lane = key row): dk, gk, then -- once the R partner has flagged its
//      a_t -- gw with its running suffix sum;
//   J0/J1 waves own value columns [16w, 16w+16) of G (lane = value column), J0 the key rows [0,32), J1 [32,64):
//      each produces half of the contraction over i for gv; J1 leaves its partial tile in LDS, J0 adds it after
//      the stage barrier and stores gv.  The score tile of a block is built by one of the two (by parity).
//      The eight J waves also turn the NEXT stage (wave = block x channel quarter, lane = token x 4 channels;
//      cumulative decays are DPP row prefix sums) into MFMA operands in the other LDS buffer and issue the global
//      loads of the stage after that, so preparation overlaps the chains of the other roles.
//   G is held in both orientations (gk contracts it over j, gv over i); the R -> K handoff is a release/acquire
//   flag in LDS (the producer never waits), the J1 -> J0 handoff rides on the stage barrier.
#include "wkv6_chunk.h"
#ifdef DBG_TIME
#include <cstdio>
#endif

namespace wkv6 {
namespace {

using namespace chunk;

enum { B_RH = 0, B_RL, B_KH, B_KL, B_V, B_GY, B_R, B_K, NB_ARR };      // bf16 [16][RSB/2] each
constexpr int FRS = 72 * 4;                                            // bytes per fp32 token row (conflict-free float4 row reads)
constexpr int BOFF_FR = NB_ARR * ARR;                                  // float [16][72]  fR_a = e^{c_a - c_8}; then a_t
constexpr int BOFF_FK = BOFF_FR + BLK * FRS;                           // float [16][72]  fK_a = e^{c_8 - c_{a+1}}
constexpr int BOFF_LW = BOFF_FK + BLK * FRS;                           // float [16][72]  lw_a e^{lw_a - max(lw_a, LW_MIN)}
constexpr int BOFF_E8 = BOFF_LW + BLK * FRS;                           // float [64]; E16 at +256, E16m8 at +512
constexpr int BOFF_E16 = BOFF_E8 + 256;
constexpr int BOFF_E16M8 = BOFF_E16 + 256;
constexpr int BOFF_COEF = BOFF_E16M8 + 256;                            // float [4][16]  per-quarter sum_i r u k
constexpr int BBLK_BYTES = BOFF_COEF + 256;
constexpr int STG = 32;                                                // tokens per stage (= forward checkpoint spacing in this mode)
constexpr int SBLK = STG / BLK;                                        // blocks per stage
constexpr int BUF_BYTES = SBLK * BBLK_BYTES;                           // one stage image; two of them
constexpr int OFF_PART = 2 * BUF_BYTES;                                // float4 [2][SBLK][4][64]  J1's partial gv tiles, by stage parity
constexpr int OFF_FLAG = OFF_PART + 2 * SBLK * 4 * 64 * 16;            // int [4]  last stage whose a_t the R wave has published
constexpr int LDS_BYTES = OFF_FLAG + 64;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

// x += x shifted by n lanes within the 16-lane row, as ONE v_add_f32_dpp (the compiler emits v_mov_dpp + v_add for the
// builtin).  Lanes whose source falls outside the row are disabled by the DPP rules (bound_ctrl off) and keep x.
#define WKV6_DPP_ACC(x, ctrl) asm("v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(x))
__device__ __forceinline__ float row_prefix16(float x)
{   // inclusive prefix sum over the 16 lanes of a DPP row
    WKV6_DPP_ACC(x, "row_shr:1"); WKV6_DPP_ACC(x, "row_shr:2"); WKV6_DPP_ACC(x, "row_shr:4"); WKV6_DPP_ACC(x, "row_shr:8");
    return x;
}
__device__ __forceinline__ float row_suffix16(float x)
{   // inclusive suffix sum over the 16 lanes of a DPP row
    WKV6_DPP_ACC(x, "row_shl:1"); WKV6_DPP_ACC(x, "row_shl:2"); WKV6_DPP_ACC(x, "row_shl:4"); WKV6_DPP_ACC(x, "row_shl:8");
    return x;
}

__device__ __forceinline__ float pick4(const f4v& v, int s)
{
    float o = v[0];
    o = s == 1 ? v[1] : o;
    o = s == 2 ? v[2] : o;
    o = s == 3 ? v[3] : o;
    return o;
}
// split a C-layout tile pair (8 floats) into the hi / lo bf16x8 fragments of one k-step
__device__ __forceinline__ void split8(const float (&t0)[4], const float (&t1)[4], b8v& hi, b8v& lo)
{
    uint2 h0, l0, h1, l1;
    split4(t0, h0, l0);
    split4(t1, h1, l1);
    hi = __builtin_bit_cast(b8v, make_uint4(h0.x, h0.y, h1.x, h1.y));
    lo = __builtin_bit_cast(b8v, make_uint4(l0.x, l0.y, l1.x, l1.y));
}

template <bool W_RAW>
__global__ __launch_bounds__(1024) void chunk_bwd16_kernel(const ScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wid >> 2;                                           // 0 R, 1 K, 2 J0, 3 J1; channel quarter in phase P
    const int wv = wid & 3;                                              // tile owned in phase C, block prepared in phase P
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
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
    if (a.lens) ntok = min(max(a.lens[b], 0), a.T);
    const int nstmax = (a.T + STG - 1) / STG;
    const int nst = (ntok + STG - 1) / STG;
    int* const flags = reinterpret_cast<int*>(smem + OFF_FLAG);
    if (tid < 4) flags[tid] = nst;                                       // no stage has this index

    // ---- preparation role of the J waves: token ptok of block pb, channels ch0..ch0+3
    const int jw = wid & 7;
    const int pb = jw & 1, cq = jw >> 1;
    const int ptok = lane & 15, pc4 = lane >> 4;
    const int ch0 = 16 * cq + 4 * pc4;
    float uu[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.use_u) io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.u) + h * HEAD + ch0, uu);

    uint2 pr, pk, pv, pg, pw;
    float4 pe;
    auto load_stage = [&](int st) {
        const int p = st * STG + pb * BLK + ptok;
        pr = pk = pv = pg = pw = make_uint2(0u, 0u);
        pe = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < ntok) {
            const int t = a.reverse ? ntok - 1 - p : p;
            const unsigned idx = (unsigned)(t * a.C + ch0);
            pr = *reinterpret_cast<const uint2*>(gr_ + idx);
            pk = *reinterpret_cast<const uint2*>(gk_ + idx);
            pv = *reinterpret_cast<const uint2*>(gv_ + idx);
            pg = *reinterpret_cast<const uint2*>(ggy + idx);
            if constexpr (W_RAW) pw = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a.w) + base + idx);
            else pe = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.w) + base + idx);
        }
    };

    auto prep_stage = [&](int st) {
        char* const bb = smem + (st & 1) * BUF_BYTES + pb * BBLK_BYTES;
        const bool valid = st * STG + pb * BLK + ptok < ntok;
        const float r[4] = {bf_lo(pr.x), bf_hi(pr.x), bf_lo(pr.y), bf_hi(pr.y)};
        const float k[4] = {bf_lo(pk.x), bf_hi(pk.x), bf_lo(pk.y), bf_hi(pk.y)};
        float lw[4];
        if constexpr (W_RAW) {
            lw[0] = -__expf(bf_lo(pw.x)); lw[1] = -__expf(bf_hi(pw.x));
            lw[2] = -__expf(bf_lo(pw.y)); lw[3] = -__expf(bf_hi(pw.y));
        } else {
            lw[0] = pe.x; lw[1] = pe.y; lw[2] = pe.z; lw[3] = pe.w;
        }
        float lws[4], lwe[4], inc[4];
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            lws[c] = valid ? fmaxf(lw[c], LW_MIN) : 0.f;                  // the decay the block algebra uses
            lwe[c] = valid ? lw[c] : 0.f;
            part = fmaf(r[c] * uu[c], k[c], part);
            inc[c] = row_prefix16(lws[c]);                                // inclusive prefix over the 16 tokens of the row
        }
        // gw multiplier: the true lw, times d_true / d_clamped where the clamp is active (the algebra yields the exact
        // gradient of the clamped model, d_clamped * X; the true one is d_true * X).  Rare: skipped wave-uniformly.
        if (__builtin_amdgcn_ballot_w64(lwe[0] < LW_MIN || lwe[1] < LW_MIN || lwe[2] < LW_MIN || lwe[3] < LW_MIN)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) lwe[c] *= __expf(fminf(lwe[c] - LW_MIN, 0.f));
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);                                     // the 4 lanes that share this token
        char* const row = bb + ptok * RSB + ch0 * 2;
        if (pc4 == 0) *reinterpret_cast<float*>(bb + BOFF_COEF + (cq * 16 + ptok) * 4) = part;
        *reinterpret_cast<uint2*>(row + B_V * ARR) = pv;
        *reinterpret_cast<uint2*>(row + B_GY * ARR) = pg;
        *reinterpret_cast<uint2*>(row + B_R * ARR) = pr;
        *reinterpret_cast<uint2*>(row + B_K * ARR) = pk;
        *reinterpret_cast<float4*>(bb + BOFF_LW + ptok * FRS + ch0 * 4) = make_float4(lwe[0], lwe[1], lwe[2], lwe[3]);
        float c8[4], c16[4], ev[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            c8[c] = __shfl(inc[c], (lane & 48) + 7);                      // before token 8
            c16[c] = __shfl(inc[c], (lane & 48) + 15);                    // whole block
            ev[c] = __expf(ptok == 0 ? c8[c] : ptok == 1 ? c16[c] : c16[c] - c8[c]);
        }
        if (ptok < 3)                                                     // token lanes 0,1,2 write E8, E16, E16m8
            *reinterpret_cast<float4*>(bb + BOFF_E8 + ptok * 256 + ch0 * 4) = make_float4(ev[0], ev[1], ev[2], ev[3]);
        float rh[4], kh[4], fr[4], fk[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            fr[c] = __expf(inc[c] - lws[c] - c8[c]);
            fk[c] = __expf(c8[c] - inc[c]);
            rh[c] = r[c] * fr[c];
            kh[c] = k[c] * fk[c];
        }
        uint2 hi, lo;
        split4(rh, hi, lo);
        *reinterpret_cast<uint2*>(row + B_RH * ARR) = hi; *reinterpret_cast<uint2*>(row + B_RL * ARR) = lo;
        split4(kh, hi, lo);
        *reinterpret_cast<uint2*>(row + B_KH * ARR) = hi; *reinterpret_cast<uint2*>(row + B_KL * ARR) = lo;
        *reinterpret_cast<float4*>(bb + BOFF_FR + ptok * FRS + ch0 * 4) = make_float4(fr[0], fr[1], fr[2], fr[3]);
        *reinterpret_cast<float4*>(bb + BOFF_FK + ptok * FRS + ch0 * 4) = make_float4(fk[0], fk[1], fk[2], fk[3]);
    };

    // ---- lane roles in the chains
    const int x = lane & 15, g = lane >> 4;
    int troff = (4 * g + (x >> 2)) * RSB + 8 * (x & 3);          // transposed read, natural columns (own tile)
    int trow = (4 * g + (x >> 2)) * RSB + 16 * (x & 3);          // transposed read, tile-labelled columns: + tile_tr(t)
#ifdef DBG_TIME
    long long tacc[4] = {0, 0, 0, 0}, tt0, tt1;
#define TSTART tt0 = clock64();
#define TLAP(i) tt1 = clock64(); tacc[i] += tt1 - tt0; tt0 = tt1;
#else
#define TSTART
#define TLAP(i)
#endif
    if (role >= 2 && nst > 0) {   // first stage image
        load_stage(nst - 1);
        prep_stage(nst - 1);
        if (nst > 1) load_stage(nst - 2);
    }
    __syncthreads();

    if (role == 0) {
        // =============== R: key rows [16wv, 16wv+16) of the forward state: gr, gu, a_t ====================
        // ST[jt][q] = S_entry(block)[i = 16wv + x][j = tile_ch(jt) + 8g + q]   (transposed tiles: lane = key row)
        float ue[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.use_u) io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.u) + h * HEAD + 16 * wv + 4 * g, ue);
        float gu_acc[4] = {0.f, 0.f, 0.f, 0.f};
        f4v SN[4];                                                // next stage's entry state, loaded one stage ahead
        auto load_ckpt = [&](int st) {
            // forward state at the entry of stage st (dumped in the forward kernel's register order):
            // element S[i = 16wv + x][j]: forward wave j>>4, tile 2(i>>5) + ((i>>2)&1), lane 16((i>>3)&3) + (j&15), reg i&3
            const float* const ck = a.ckpt + ((long)blockIdx.x * nstmax + st) * (HEAD * HEAD);
            const int i_ = 16 * wv + x;
            const int fit = 2 * (i_ >> 5) + ((i_ >> 2) & 1), fg = (i_ >> 3) & 3, fq = i_ & 3;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                float t4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j_ = tile_ch(jt) + 8 * g + q;
                    t4[q] = ck[(((j_ >> 4) * 4 + fit) * 64 + 16 * fg + (j_ & 15)) * 4 + fq];
                }
                SN[jt] = f4v{t4[0], t4[1], t4[2], t4[3]};
            }
        };
        if (nst > 0) load_ckpt(nst - 1);
        for (int st = nst - 1; st >= 0; --st) {
            TSTART
            asm volatile("" : "+v"(troff), "+v"(trow));   // pins every transposed LDS read of this stage below the barrier
            f4v ST[4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) ST[jt] = SN[jt];
            if (st > 0) load_ckpt(st - 1);
            char* const buf = smem + (st & 1) * BUF_BYTES;
#pragma unroll
            for (int blk = 0; blk < SBLK; ++blk) {
                char* const bb = buf + blk * BBLK_BYTES;
                f4v dA_ba = {0.f, 0.f, 0.f, 0.f};
                b8v gyr[2];                                      // gy [token x][32s + 8g .. +7]: also the B operand of accr
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int off = x * RSB + (32 * s + 8 * g) * 2;
                    gyr[s] = ld_b8(bb + B_GY * ARR + off);
                    const b8v vr = ld_b8(bb + B_V * ARR + off);
                    dA_ba = mfma32(vr, gyr[s], dA_ba);           // [row b][col a]: lane col a = x, rows b = 4g+q
                }
                // vg_x = dA[x][x]: held by lane (x, g = x>>2) in register x&3
                const float vg = __shfl(pick4(dA_ba, x & 3), 16 * (x >> 2) + x);
                float dba[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) dba[q] = 4 * g + q < x ? dA_ba[q] : 0.f;   // dA^T[b][a = x], strictly lower
                uint2 th, tl;
                split4(dba, th, tl);
                const s4v dba_hi = __builtin_bit_cast(s4v, th), dba_lo = __builtin_bit_cast(s4v, tl);
                const s4v khf = tr_read(bb + B_KH * ARR + troff + 32 * wv);        // Khat[4g+e][16wv + x]
                const s4v klf = tr_read(bb + B_KL * ARR + troff + 32 * wv);
                // gr accumulators [i_local = 4g+q][token x], one per MFMA shape (see wkv6_chunk.hip); the E8 row scale of
                // the state term is applied to the 4 results instead of the 16 operands
                f4v accs_ = {0.f, 0.f, 0.f, 0.f}, acci = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float t0[4] = {ST[2 * s][0], ST[2 * s][1], ST[2 * s][2], ST[2 * s][3]};
                    const float t1[4] = {ST[2 * s + 1][0], ST[2 * s + 1][1], ST[2 * s + 1][2], ST[2 * s + 1][3]};
                    b8v hi, lo;                                  // k-slot (s, g, e) <-> value channel 32s + 8g + e
                    split8(t0, t1, hi, lo);
                    accs_ = mfma32(hi, gyr[s], accs_);
                    accs_ = mfma32(lo, gyr[s], accs_);
                }
                acci = mfma16(khf, dba_hi, acci);                // sum_b Khat[b][i] dA[a][b]
                acci = mfma16(khf, dba_lo, acci);
                acci = mfma16(klf, dba_hi, acci);
                {   // gr, a_t, gu: lane = token x, channels ch .. ch+3
                    const int ch = 16 * wv + 4 * g;
                    float4* const frp = reinterpret_cast<float4*>(bb + BOFF_FR + x * FRS + ch * 4);
                    const float4 fr4 = *frp;
                    const float4 e84 = *reinterpret_cast<const float4*>(bb + BOFF_E8 + ch * 4);
                    const uint2 rr = *reinterpret_cast<const uint2*>(bb + B_R * ARR + x * RSB + ch * 2);
                    const uint2 kk = *reinterpret_cast<const uint2*>(bb + B_K * ARR + x * RSB + ch * 2);
                    const float frv[4] = {fr4.x, fr4.y, fr4.z, fr4.w}, e8v[4] = {e84.x, e84.y, e84.z, e84.w};
                    const float rv[4] = {bf_lo(rr.x), bf_hi(rr.x), bf_lo(rr.y), bf_hi(rr.y)};
                    const float kv[4] = {bf_lo(kk.x), bf_hi(kk.x), bf_lo(kk.y), bf_hi(kk.y)};
                    float o_gr[4], at[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float dq = frv[q] * fmaf(e8v[q], accs_[q], acci[q]);
                        o_gr[q] = fmaf(vg * ue[q], kv[q], dq);
                        gu_acc[q] = fmaf(vg * rv[q], kv[q], gu_acc[q]);
                        at[q] = rv[q] * dq;
                    }
                    *frp = make_float4(at[0], at[1], at[2], at[3]);      // same lane, same address: fR is consumed
                    const int p = st * STG + blk * BLK + x;
                    if (p < ntok) {
                        const int t = a.reverse ? ntok - 1 - p : p;
                        const unsigned idx = (unsigned)(t * a.C + ch);
                        if (a.accumulate) {
                            float o1[4];
                            io4<bf16_t>::load(ogr + idx, o1);
#pragma unroll
                            for (int q = 0; q < 4; ++q) o_gr[q] += o1[q];
                        }
                        io4<bf16_t>::store(ogr + idx, o_gr);
                    }
                }
                if (blk < SBLK - 1) {   // entry state of the next block:  S <- E16 (.) S + E16m8 (.) (Khat^T V)
                    const float e16 = *reinterpret_cast<const float*>(bb + BOFF_E16 + (16 * wv + x) * 4);
                    const float e16m8 = *reinterpret_cast<const float*>(bb + BOFF_E16M8 + (16 * wv + x) * 4);
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) {
                        const s4v vf = tr_read(bb + B_V * ARR + trow + tile_tr(jt));     // V[4g+e][tile_ch(jt) + 8(x>>2) + (x&3)]
                        f4v o = {0.f, 0.f, 0.f, 0.f};
                        o = mfma16(vf, khf, o);
                        o = mfma16(vf, klf, o);
#pragma unroll
                        for (int q = 0; q < 4; ++q) ST[jt][q] = fmaf(e16, ST[jt][q], e16m8 * o[q]);
                    }
                }
            }
            // publish a_t of this stage to the K wave of the same rows
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(flags + wv, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            TLAP(2)
            __syncthreads();
            TLAP(3)
        }
        if (a.gu) {
            float s4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) s4[q] = row_sum16(gu_acc[q]);
            if (x == 0) io4<bf16_t>::store(reinterpret_cast<bf16_t*>(a.gu) + (long)b * a.C + h * HEAD + 16 * wv + 4 * g, s4);
        }
    } else if (role == 1) {
        // =============== K: key rows [16wv, 16wv+16) of G: gk, gw ==========================================
        // GI[jt][q] = G[i = 16wv + x][j = tile_ch(jt) + 8g + q]
        float ue[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.use_u) io4<bf16_t>::load(reinterpret_cast<const bf16_t*>(a.u) + h * HEAD + 16 * wv + 4 * g, ue);
        f4v GI[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) GI[t] = f4v{0.f, 0.f, 0.f, 0.f};
        float Rc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int st = nst - 1; st >= 0; --st) {
            TSTART
            asm volatile("" : "+v"(troff), "+v"(trow));
            char* const buf = smem + (st & 1) * BUF_BYTES;
#pragma unroll
            for (int blk = SBLK - 1; blk >= 0; --blk) {
                char* const bb = buf + blk * BBLK_BYTES;
                f4v dA_ab = {0.f, 0.f, 0.f, 0.f};
                b8v vr[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int off = x * RSB + (32 * s + 8 * g) * 2;
                    const b8v gyr = ld_b8(bb + B_GY * ARR + off);
                    vr[s] = ld_b8(bb + B_V * ARR + off);
                    dA_ab = mfma32(gyr, vr[s], dA_ab);           // [row a][col b]: lane col b = x, rows a = 4g+q
                }
                const float vg = __shfl(pick4(dA_ab, x & 3), 16 * (x >> 2) + x);
                float dab[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) dab[q] = x < 4 * g + q ? dA_ab[q] : 0.f;   // dA[a][b = x], strictly lower
                uint2 th, tl;
                split4(dab, th, tl);
                const s4v dab_hi = __builtin_bit_cast(s4v, th), dab_lo = __builtin_bit_cast(s4v, tl);
                const s4v rhf_w = tr_read(bb + B_RH * ARR + troff + 32 * wv);      // Rhat[4g+e][16wv + x]
                const s4v rlf_w = tr_read(bb + B_RL * ARR + troff + 32 * wv);
                const float e8x = *reinterpret_cast<const float*>(bb + BOFF_E8 + (16 * wv + x) * 4);
                const float e16x = *reinterpret_cast<const float*>(bb + BOFF_E16 + (16 * wv + x) * 4);
                f4v accg = {0.f, 0.f, 0.f, 0.f}, ak = {0.f, 0.f, 0.f, 0.f};   // one accumulator per MFMA shape
                ak = mfma16(rhf_w, dab_hi, ak);                  // sum_a Rhat[a][i] dA[a][b]
                ak = mfma16(rhf_w, dab_lo, ak);
                ak = mfma16(rlf_w, dab_hi, ak);
#pragma unroll
                for (int s = 0; s < 2; ++s) {                    // (G v_b)[i]; its E16m8 row scale is applied to the result
                    const float t0[4] = {GI[2 * s][0], GI[2 * s][1], GI[2 * s][2], GI[2 * s][3]};
                    const float t1[4] = {GI[2 * s + 1][0], GI[2 * s + 1][1], GI[2 * s + 1][2], GI[2 * s + 1][3]};
                    b8v hi, lo;
                    split8(t0, t1, hi, lo);
                    accg = mfma32(hi, vr[s], accg);
                    accg = mfma32(lo, vr[s], accg);
                }
                // ---- G[i = 16wv + x][:] <- E16 G + E8 (Rhat^T gy)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {                 // [row j_local][col i_local = x]
                    const s4v gyf = tr_read(bb + B_GY * ARR + trow + tile_tr(jt));
                    f4v o = {0.f, 0.f, 0.f, 0.f};
                    o = mfma16(gyf, rhf_w, o);
                    o = mfma16(gyf, rlf_w, o);
#pragma unroll
                    for (int q = 0; q < 4; ++q) GI[jt][q] = fmaf(e16x, GI[jt][q], e8x * o[q]);
                }
                {   // gk now; b_t = k (.) dk goes to LDS (in place of fK) for the gw pass
                    const int ch = 16 * wv + 4 * g;
                    float4* const fkp = reinterpret_cast<float4*>(bb + BOFF_FK + x * FRS + ch * 4);
                    const float4 fk4 = *fkp;
                    const float4 m84 = *reinterpret_cast<const float4*>(bb + BOFF_E16M8 + ch * 4);
                    const uint2 rr = *reinterpret_cast<const uint2*>(bb + B_R * ARR + x * RSB + ch * 2);
                    const uint2 kk = *reinterpret_cast<const uint2*>(bb + B_K * ARR + x * RSB + ch * 2);
                    const float fkv[4] = {fk4.x, fk4.y, fk4.z, fk4.w}, m8v[4] = {m84.x, m84.y, m84.z, m84.w};
                    const float rv[4] = {bf_lo(rr.x), bf_hi(rr.x), bf_lo(rr.y), bf_hi(rr.y)};
                    const float kv[4] = {bf_lo(kk.x), bf_hi(kk.x), bf_lo(kk.y), bf_hi(kk.y)};
                    float o_gk[4], bt[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float dk = fkv[q] * fmaf(m8v[q], accg[q], ak[q]);
                        o_gk[q] = fmaf(vg * ue[q], rv[q], dk);
                        bt[q] = kv[q] * dk;
                    }
                    *fkp = make_float4(bt[0], bt[1], bt[2], bt[3]);      // same lane, same address: fK is consumed
                    const int p = st * STG + blk * BLK + x;
                    if (p < ntok) {
                        const int t = a.reverse ? ntok - 1 - p : p;
                        const unsigned idx = (unsigned)(t * a.C + ch);
                        if (a.accumulate) {
                            float o2[4];
                            io4<bf16_t>::load(ogk + idx, o2);
#pragma unroll
                            for (int q = 0; q < 4; ++q) o_gk[q] += o2[q];
                        }
                        io4<bf16_t>::store(ogk + idx, o_gk);
                    }
                }
            }
#ifndef DBG_NOGW
            // ---- gw: needs a_t of the R wave that owns the same key rows
            while (__hip_atomic_load(flags + wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != st)
                __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int blk = SBLK - 1; blk >= 0; --blk) {
                const char* const bb = buf + blk * BBLK_BYTES;
                const int ch = 16 * wv + 4 * g;
                const float4 at4 = *reinterpret_cast<const float4*>(bb + BOFF_FR + x * FRS + ch * 4);
                const float4 bt4 = *reinterpret_cast<const float4*>(bb + BOFF_FK + x * FRS + ch * 4);
                const float4 lw4 = *reinterpret_cast<const float4*>(bb + BOFF_LW + x * FRS + ch * 4);
                const float atv[4] = {at4.x, at4.y, at4.z, at4.w}, btv[4] = {bt4.x, bt4.y, bt4.z, bt4.w};
                const float lwv[4] = {lw4.x, lw4.y, lw4.z, lw4.w};
                float o_gw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float dl = atv[q] - btv[q];
                    const float sfx = row_suffix16(dl);           // inclusive suffix sum over the later tokens of the row
                    const float total = __shfl(sfx, lane & 48);
                    o_gw[q] = (Rc[q] + sfx - atv[q]) * lwv[q];    // Rc + (sfx - dl) - b_t
                    Rc[q] += total;
                }
                const int p = st * STG + blk * BLK + x;
                if (p < ntok) {
                    const int t = a.reverse ? ntok - 1 - p : p;
                    const unsigned idx = (unsigned)(t * a.C + ch);
                    if (a.accumulate) {
                        float o3[4];
                        io4<bf16_t>::load(ogw + idx, o3);
#pragma unroll
                        for (int q = 0; q < 4; ++q) o_gw[q] += o3[q];
                    }
                    io4<bf16_t>::store(ogw + idx, o_gw);
                }
            }
#endif
            TLAP(2)
            __syncthreads();
            TLAP(3)
        }
    } else {
        // =============== J0 / J1: value columns [16wv, 16wv+16) x key rows [32ih, 32ih+32) of G: gv, gs =====
        // GJ[t][q] = G[i = 32ih + 4t + 8g + q][j = 16wv + x]        (state tiles 2ih + t)
        const int ih = role - 2;
        f4v GJ[2];
        GJ[0] = GJ[1] = f4v{0.f, 0.f, 0.f, 0.f};
        float4* const part = reinterpret_cast<float4*>(smem + OFF_PART) + wv * 64 + lane;      // + (parity * SBLK + blk) * 256
        for (int st = nst - 1; st >= 0; --st) {
            TSTART
            if (st > 0) {   // operands of the next stage into the other buffer; global loads of the one after
                prep_stage(st - 1);
                if (st > 1) load_stage(st - 2);
            }
            TLAP(0)
            asm volatile("" : "+v"(troff), "+v"(trow));
            const char* const buf = smem + (st & 1) * BUF_BYTES;
            f4v accs[SBLK];
#pragma unroll
            for (int blk = SBLK - 1; blk >= 0; --blk) {
                const char* const bb = buf + blk * BBLK_BYTES;
                const s4v gyT_w = tr_read(bb + B_GY * ARR + troff + 32 * wv);      // gy[4g+e][16wv + x]
                f4v acc = {0.f, 0.f, 0.f, 0.f}, acci = {0.f, 0.f, 0.f, 0.f};   // gv^T[j][b]; one accumulator per MFMA shape
                if ((blk & 1) == ih) {   // the intra-block part, sum_a gy[a][j] A[a][b]: one of the two J waves per block
                    f4v sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int off = x * RSB + (32 * s + 8 * g) * 2;
                        const b8v rh = ld_b8(bb + B_RH * ARR + off), rl = ld_b8(bb + B_RL * ARR + off);
                        const b8v kh = ld_b8(bb + B_KH * ARR + off), kl = ld_b8(bb + B_KL * ARR + off);
                        sc = mfma32(rh, kh, sc);                  // A[row a][col b]: lane col b = x, rows a = 4g+q
                        sc = mfma32(rh, kl, sc);
                        sc = mfma32(rl, kh, sc);
                    }
                    float cf[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float4 c4 = *reinterpret_cast<const float4*>(bb + BOFF_COEF + c * 64 + 16 * g);
                        cf[0] += c4.x; cf[1] += c4.y; cf[2] += c4.z; cf[3] += c4.w;
                    }
                    float scm[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int o = 4 * g + q;                  // query token a; key token b = x
                        scm[q] = x < o ? sc[q] : (x == o ? cf[q] : 0.f);
                    }
                    uint2 th, tl;
                    split4(scm, th, tl);
                    acci = mfma16(gyT_w, __builtin_bit_cast(s4v, th), acci);
                    acci = mfma16(gyT_w, __builtin_bit_cast(s4v, tl), acci);
                }
                {   // + sum_{i in half} Khat[b][i] (E16m8 (.) G)[i][j]; k-slot (g, e) <-> key channel 32ih + 8g + e
                    const int off = x * RSB + (32 * ih + 8 * g) * 2;
                    const b8v kh = ld_b8(bb + B_KH * ARR + off), kl = ld_b8(bb + B_KL * ARR + off);
                    const float4 m0 = *reinterpret_cast<const float4*>(bb + BOFF_E16M8 + (32 * ih + 8 * g) * 4);
                    const float4 m1 = *reinterpret_cast<const float4*>(bb + BOFF_E16M8 + (32 * ih + 8 * g + 4) * 4);
                    const float t0[4] = {GJ[0][0] * m0.x, GJ[0][1] * m0.y, GJ[0][2] * m0.z, GJ[0][3] * m0.w};
                    const float t1[4] = {GJ[1][0] * m1.x, GJ[1][1] * m1.y, GJ[1][2] * m1.z, GJ[1][3] * m1.w};
                    b8v gh, gl;
                    split8(t0, t1, gh, gl);
                    acc = mfma32(gh, kh, acc);
                    acc = mfma32(gh, kl, acc);
                    acc = mfma32(gl, kh, acc);
                }
                acc += acci;
                if (ih) part[((st & 1) * SBLK + blk) * 256] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                accs[blk] = acc;
                // ---- G[half][j = 16wv + x] <- E16 G + E8 (Rhat^T gy)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const s4v rhf = tr_read(bb + B_RH * ARR + trow + tile_tr(2 * ih + t));
                    const s4v rlf = tr_read(bb + B_RL * ARR + trow + tile_tr(2 * ih + t));
                    f4v o = {0.f, 0.f, 0.f, 0.f};                 // [row i_local][col j_local = x]
                    o = mfma16(rhf, gyT_w, o);
                    o = mfma16(rlf, gyT_w, o);
                    const float4 d16 = *reinterpret_cast<const float4*>(bb + BOFF_E16 + (32 * ih + 4 * t + 8 * g) * 4);
                    const float4 d8 = *reinterpret_cast<const float4*>(bb + BOFF_E8 + (32 * ih + 4 * t + 8 * g) * 4);
                    GJ[t][0] = fmaf(d16.x, GJ[t][0], d8.x * o[0]);
                    GJ[t][1] = fmaf(d16.y, GJ[t][1], d8.y * o[1]);
                    GJ[t][2] = fmaf(d16.z, GJ[t][2], d8.z * o[2]);
                    GJ[t][3] = fmaf(d16.w, GJ[t][3], d8.w * o[3]);
                }
            }
            TLAP(2)
            __syncthreads();
            TLAP(3)
            if (ih == 0) {   // J1's halves are in LDS now (it rewrites this parity only two stages later)
#pragma unroll
                for (int blk = 0; blk < SBLK; ++blk) {
                    const int p = st * STG + blk * BLK + x;
                    const float4 o1 = part[((st & 1) * SBLK + blk) * 256];
                    if (p < ntok) {
                        const int t = a.reverse ? ntok - 1 - p : p;
                        const unsigned idx = (unsigned)(t * a.C + 16 * wv + 4 * g);
                        float o[4] = {accs[blk][0] + o1.x, accs[blk][1] + o1.y, accs[blk][2] + o1.z, accs[blk][3] + o1.w};
                        if (a.accumulate) {
                            float old[4];
                            io4<bf16_t>::load(ogv + idx, old);
#pragma unroll
                            for (int q = 0; q < 4; ++q) o[q] += old[q];
                        }
                        io4<bf16_t>::store(ogv + idx, o);
                    }
                }
            }
        }
        if (a.gs) {   // dL/dS0, layout [j][i]
            bf16_t* const og = reinterpret_cast<bf16_t*>(a.gs) + ((long)b * a.H + h) * HEAD * HEAD;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float t4[4] = {GJ[t][0], GJ[t][1], GJ[t][2], GJ[t][3]};
                io4<bf16_t>::store(og + (long)(16 * wv + x) * HEAD + 32 * ih + 4 * t + 8 * g, t4);
            }
        }
    }
#ifdef DBG_TIME
    if (a.aux && blockIdx.x == 7 && lane == 0)
        for (int i = 0; i < 4; ++i) a.aux[wid * 4 + i] = (float)tacc[i];
#endif
    if (a.zero_tail && !a.accumulate) {
        const float z[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = ntok + (tid >> 4); t < a.T; t += 64) {
            const unsigned idx = (unsigned)(t * a.C + 4 * (tid & 15));
            io4<bf16_t>::store(ogr + idx, z);
            io4<bf16_t>::store(ogk + idx, z);
            io4<bf16_t>::store(ogv + idx, z);
            io4<bf16_t>::store(ogw + idx, z);
        }
    }
}

template <bool W_RAW> hipError_t launch_bwd16_variant(const ScanArgs& a, hipStream_t st)
{
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chunk_bwd16_kernel<W_RAW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        configured = true;
    }
#ifdef DBG_TIME
    {
        static float* dbg = nullptr;
        if (!dbg) hipMalloc(&dbg, 64 * 4);
        ScanArgs a2 = a; a2.aux = dbg;
        hipLaunchKernelGGL((chunk_bwd16_kernel<W_RAW>), dim3(a.B * a.H), dim3(1024), (size_t)LDS_BYTES, st, a2);
        hipDeviceSynchronize();
        float host[64];
        hipMemcpy(host, dbg, 64 * 4, hipMemcpyDeviceToHost);
        static int calls = 0;
        if (++calls == 3) {
            const int ng = (a.T + STG - 1) / STG;
            const char* rn[4] = {"R ", "K ", "J0", "J1"};
            for (int w = 0; w < 16; ++w)
                printf("%s wave %d: per stage  prep %7.0f  wait %7.0f  work %7.0f  wait %7.0f\n", rn[w >> 2], w & 3,
                       host[w * 4] / ng, host[w * 4 + 1] / ng, host[w * 4 + 2] / ng, host[w * 4 + 3] / ng);
        }
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((chunk_bwd16_kernel<W_RAW>), dim3(a.B * a.H), dim3(1024), (size_t)LDS_BYTES, st, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_chunk_bwd16(const ScanArgs& a, hipStream_t st)
{
    if (a.ckpt_tok != STG) return hipErrorInvalidValue;
    return a.wkind ? launch_bwd16_variant<true>(a, st) : launch_bwd16_variant<false>(a, st);
}

}  // namespace wkv6
