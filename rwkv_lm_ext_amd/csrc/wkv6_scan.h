// Exact-fp32 token-serial WKV6 kernels for gfx950 ("scan" path).
//
// These kernels evaluate the recurrence of cuda/wkv6_cuda.cu:6-227 literally (fp32 state, one
// token after another), re-laid-out for CDNA4:
//   * one 512-thread workgroup (8 wave64) per (batch, head); the 64x64 state is tiled over the
//     waves, every lane owns a small i x j sub-block in registers;
//   * r/k/v/w(/gy) rows are staged 16 tokens at a time through a double-buffered LDS image
//     (coalesced 128-B row segments from HBM), decays exp(-exp(w)) are formed once per element
//     while staging;
//   * the per-token reductions use DPP row operations and v_permlane{16,32}_swap, not LDS;
//   * backward = two sweeps instead of the reference's five (kernel_backward_111 has three scans,
//     kernel_backward_222 two): sweep S (forward in scan order) recomputes the state and emits gr
//     and a_t = r_t (.) sum_j gy_t[j] S_t[.][j]; sweep G (reverse) carries G, emits gk, gv and
//     b_t = k_t (.) sum_j G_t[.][j] v_t[j] and forms gw from the suffix-sum identity
//     gew_t = sum_{s>t} a_s - sum_{s>=t} b_s   (fla/ops/rwkv6/recurrent_fuse.py:394-396),
//     which removes the reference's per-thread float[T] scratch array and its T <= _T_ limit.
//
// They are the reference-exact path (any T >= 1, any decay magnitude, optional initial / final
// state, either time direction, per-row lengths).
#pragma once
#include <cstdlib>
#include "wkv6_common.h"

namespace wkv6 {

// The chunked forward leaves fp32 state checkpoints for the chunked backward (wkv6_chunk_bwd12k.hip, one or two workgroups per
// (batch, head)): one 64x64 state per CKPT_TOK = 64 tokens and head (4 B per token-channel), laid out in the register order of
// that kernel's row waves -- [row wave i>>4][column tile jt][lane 16 g + (i & 15)][4 columns tile_ch(jt) + 8 g + q] -- so that a row
// wave takes its 16x64 slice with four coalesced 16-byte loads.  (Rounds 3-4 had a second, opt-in layout for the two-level backward
// experiment, now parked under tools/experiments/.)
constexpr int CKPT_TOK = 64;

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device): the attribute is per device.
struct LdsAttrOnce {
    bool done[64] = {};
    hipError_t ensure(const void* fn, size_t bytes)
    {
        int dev = 0;
        if (hipError_t e = hipGetDevice(&dev)) return e;
        if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
        if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)) return e;
        if (dev >= 0 && dev < 64) done[dev] = true;
        return hipSuccess;
    }
};

struct ScanArgs {
    int B, T, C, H;
    const void *r, *k, *v, *w, *u;   // w: float ew = -exp(w) when wkind == 0, raw w in the I/O type when 1,
    int wkind;                        //    float decay d = exp(-exp(w)) when 2 (inference kernel, cuda/rwkv6.cu:38)
    int state_f32;                    // s0 / s_out are fp32 regardless of the I/O type (cuda/rwkv6.cu:8)
    const void* s0;                   // initial state [.., H, N(j), N(i)] in the I/O type, or null (zero)
    long s0_bstride;                  // elements between batch entries of s0 (0: shared over the batch)
    void* s_out;                      // final state [B,H,N,N] in the I/O type, or null   (forward only)
    void* y;                          // forward output
    float* y_f32;                     // optional fp32 side buffer [B,T,C]: written INSTEAD of y when accumulate == 0,
                                      // read as the addend (instead of y) when accumulate == 1  (wkv6_bi halves)
    const void* gy;                   // backward input
    void *gr, *gk, *gv, *gw;          // backward outputs, I/O type
    float* g_f32[4];                  // optional fp32 side buffers [B,T,C] for gr,gk,gv,gw (chunked backward): written INSTEAD
                                      // of the outputs when accumulate == 0, read as the addend when accumulate == 1, so the
                                      // two halves of wkv6_bi are summed in fp32 and rounded once
    void* gu;                         // [B,C] per-batch partials, I/O type (null: skip)
    void* gs;                         // [B,H,N,N] per-batch dL/dS0, I/O type (null: skip)
    int part_f32;                     // gu and gs are fp32 whatever the I/O type (WKV6_PARTIALS_F32): the caller sums them over the batch
    float* aux;                       // [B,T,C] fp32 scratch carrying a_t from sweep S to sweep G
    float* ckpt;                      // chunked path: [B*H][ceil(T/CKPT_TOK)][4096] fp32 states at 64-token boundaries (forward / state pass -> backward)
    int ckpt_valid;                   // backward: ckpt was filled by the forward, skip the state pass
    const int* lens;                  // per-batch number of tokens to scan (null: T)
    const int* order;                 // chunked kernels: batch row served by workgroup slot blockIdx / H (null: identity) -- rows
                                      // sorted by decreasing length, so that the longest sequences start first (wkv6_bi)
    int reverse;                      // 1: scan tokens lens-1 .. 0
    const int* rev_n;                 // chunked kernels: per-batch number of leading tokens that the tensors named in rev_mask
    unsigned rev_mask;                //   hold in reverse order (scan position p < rev_n[b] <-> token rev_n[b]-1-p; positions
                                      //   beyond keep their place); every token is still scanned.  Bits: REV_*
    int use_u;                        // 0: bonus u treated as 0 (reverse half of wkv6_bi)
    int accumulate;                   // 1: add into y / gr,gk,gv,gw instead of overwriting
    int zero_tail;                    // 1: write zeros for tokens >= lens[b]
    int ckpt_segs;                    // chunked forward as a two-level scan: batch row b is segment b % ckpt_segs of sequence
                                      // b / ckpt_segs, and its checkpoints go to that sequence's slots (0 / 1: plain)
    float* dsum;                      // chunked forward / state pass: [B*H][4][64] per-block-slot sums of the (clamped) log2-decays
                                      // over the whole sequence (the segment summaries of the T-split forward), or null
    // chunked forward with the GroupNorm(H) * gate epilogue of the time-mix block fused into its store (src/model.py:462-468):
    const void *gn_gate, *gn_gamma, *gn_beta;   // gate [B,T,C], ln_x.weight / bias [C] (I/O type); gn_out != null enables it
    float gn_eps;
    void* gn_out;                     // [B,T,C]  GroupNorm_H(y) * gate  (y itself is stored too unless y == null)
    float* gn_stats;                  // [B*T, H, 2] mean, rstd of every (token, head) for the backward, or null
    const float* g_in;                // backward as a two-level scan over T (wkv6_api.hip: chunk_backward): fp32 [B,H,N(j),N(i)] adjoint state
                                      // entering this row (= segment) from the future, or null (zero)
    const float* rc_in;               // ... and fp32 [B,C]: the gw suffix sum at the segment's end, sum_{s >= end} (a_s - b_s) =
                                      // Phi[i] = sum_j G[i][j] S[i][j] at the boundary, or null (zero)
    int side_compact;                 // chunked wkv6_bi halves in one launch: y_f32 / g_f32 are per-workgroup scratch, fp32 [slot][T][64]
                                      // (token stride 64), instead of [B,T,C] arrays addressed by (batch, head)
    unsigned long long* clk;          // chunked kernels: clock stamps of wave 0 of workgroup slots < clk_slots ({memtime, memrealtime} at start
    int clk_slots;                    //   and end: wkv6_set_clock_buffer, include/wkv6_amd.h), or null (the default: no stamp executes)
    int split;                        // chunked kernels, set by the launcher when B*H leaves half the chip idle: two workgroups per
                                      // (batch, head), each with its own producers and half of the consuming waves
};

enum { REV_R = 1, REV_K = 2, REV_V = 4, REV_W = 8, REV_Y = 16, REV_ALL = 31 };   // REV_Y: y in the forward, gy in the backward;
                                                                                // every gradient follows its tensor's bit
// token a tensor holds at scan position p
struct RevMap {
    int n;
    unsigned mask;
    __device__ __forceinline__ int operator()(int p, unsigned bit) const { return ((mask & bit) && p < n) ? n - 1 - p : p; }
};
__device__ __forceinline__ RevMap make_revmap(const ScanArgs& a, int b, int ntok)
{
    if (a.reverse) return RevMap{ntok, REV_ALL};
    if (a.rev_n) return RevMap{min(max(a.rev_n[b], 0), ntok), a.rev_mask};
    return RevMap{0, 0u};
}

// Token addressing of the chunked kernels' global accesses.  AFF (chosen by the launchers: no per-tensor reversal map, a.rev_n == nullptr):
// scan position p of the row is token t0 + sgn * p of EVERY tensor (0, +1; or ntok - 1, -1 under a.reverse), so the element offset of
// (position pu + plane, channel ch) splits into a loop-invariant lane part, computed once in front of the stage loops, and a wave-uniform
// part that the scalar unit forms: one vector add per access where the general map costs a compare, a select, a subtraction and an
// integer multiply (half rate: 4.2 cycles, profiles/r05_issue2_microbench.txt) per access (round 5: ~45 of the backward's 1066 vector instructions per SIMD and stage).  Positions past the
// end map past the end (or below zero = far above it as unsigned): outside the row's buffer resources, as with the general map.
template <bool AFF>
struct TokAddr {
    RevMap m;
    int t0, sgn;
    __device__ __forceinline__ TokAddr(const ScanArgs& a, int b, int ntok) : m(make_revmap(a, b, ntok))
    {
        t0 = m.mask ? ntok - 1 : 0;
        sgn = m.mask ? -1 : 1;
    }
    // (explicit form: the whole row reversed or not -- the persistent wkv6_bi launches address the call that FOLLOWS the one whose argument
    // block they hold)
    __device__ __forceinline__ TokAddr(int ntok, bool rev) : m(RevMap{rev ? ntok : 0, rev ? (unsigned)REV_ALL : 0u})
    {
        t0 = rev ? ntok - 1 : 0;
        sgn = rev ? -1 : 1;
    }
    // lane part of the element offset: position `plane` within the uniform base, channel ch, `stride` elements between tokens
    __device__ __forceinline__ int lane(int plane, int ch, int stride) const { return AFF ? (t0 + sgn * plane) * stride + ch : 0; }
    __device__ __forceinline__ int token(int p, unsigned bit) const { return AFF ? t0 + sgn * p : m(p, bit); }
    // element offset of (position pu + plane, channel ch): pu wave-uniform; lane_part = lane(plane, ch, stride)
    __device__ __forceinline__ unsigned off(int pu, int plane, int ch, int stride, unsigned bit, int lane_part) const
    {
        if constexpr (AFF) return (unsigned)(lane_part + pu * (sgn * stride));
        else return (unsigned)(m(pu + plane, bit) * stride + ch);
    }
};

enum { IO_BF16 = 0, IO_F32 = 1, IO_F16 = 2 };   // I/O element type of the scan forward (fp16: inference entry point only)
hipError_t launch_scan_fwd(const ScanArgs& a, int io, hipStream_t st);
hipError_t launch_scan_bwd(const ScanArgs& a, bool io_f32, hipStream_t st);
hipError_t launch_selftest(int* result, hipStream_t st);
// chunked MFMA forward (bf16 I/O only), wkv6_chunk.hip
hipError_t launch_chunk_fwd(const ScanArgs& a, hipStream_t st);
// two problems of one shape in one launch; the backward needs both problems' forward checkpoints
hipError_t launch_chunk_fwd_pair(const ScanArgs& a0, const ScanArgs& a1, hipStream_t st);
hipError_t launch_chunk_bwd_pair(const ScanArgs& a0, const ScanArgs& a1, hipStream_t st);
// chunked MFMA backward (bf16 I/O only): state pass + reverse pass; a.ckpt must hold chunk_ckpt_floats() floats
hipError_t launch_chunk_bwd(const ScanArgs& a, hipStream_t st);
// both halves of wkv6_bi in ONE persistent launch (one workgroup slot per CU walks (batch, head) rows: first half into the slot's fp32
// scratch, second half adds it and rounds): a1 / a2 = the forward-direction / reversed-direction problem; hipErrorNotSupported where
// the launch does not apply (two workgroups per pair) -- the caller then runs the halves as two launches
hipError_t launch_chunk_fwd_bi(const ScanArgs& a1, const ScanArgs& a2, int* slots, hipStream_t st);
hipError_t launch_chunk_bwd_bi(const ScanArgs& a1, const ScanArgs& a2, int* slots, hipStream_t st);
int bi_slots(int BH);                        // workgroup slots such a launch uses (0: does not apply)
hipError_t launch_chunk_bwd12k(const ScanArgs& a, hipStream_t st);    // reverse pass over 64-token row-order checkpoints, a.split as given (wkv6_chunk_bwd12k.hip)
size_t chunk_ckpt_floats(int B, int T, int H);
hipError_t launch_chunk_state_pass(const ScanArgs& a, hipStream_t st);   // state recurrence only (s_out, ckpt, dsum)
// In-run clock probe (wkv6_set_clock_ring, wkv6_api.hip): where launch number n of kind (0: chunked forward, 1: chunked backward) stamps,
// or null; takes the launch's place in the ring (host side, one atomic increment per launch)
unsigned long long* clock_claim(int kind, int* slots);
int cu_count();
int want_split(int BH);                      // two workgroups per (batch, head)?  (wkv6_chunk_bwd12k.hip)

}  // namespace wkv6
