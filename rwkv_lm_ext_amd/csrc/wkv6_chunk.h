// Shared definitions of the chunked MFMA kernels (wkv6_chunk.hip, wkv6_chunk_bwd.hip).
#pragma once
#include "wkv6_scan.h"

// Diagnostic build (-DWKV6_STAMP, tools/ablate.sh): waves accumulate s_memtime cycles per phase into the buffer set through
// wkv6_set_debug_buffer(); no stamp executes in the normal build.
// -DWKV6_CLOCK: only one (s_memtime, s_memrealtime) pair around each wave's whole life, slots 6 / 7 of its record: the in-kernel
// shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz with no per-phase stamp in the loop (tools/clock_probe.py).
// -DWKV6_DEBUG: the LDS tag polls of the backward count their tries and trap with a record in the debug buffer instead of spinning forever.
#if defined(WKV6_STAMP) || defined(WKV6_CLOCK) || defined(WKV6_DEBUG)
#define WKV6_DEBUGBUF 1
#define WKV6_CLK(c, r) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r) :: "memory")
namespace wkv6 { extern unsigned long long* g_stamp_buffer; }
#endif
#ifdef WKV6_STAMP
#define WKV6_T(var) do { __builtin_amdgcn_sched_barrier(0); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WKV6_ACC(k, t1, t0) stamp_acc[k] += (t1) - (t0)
#else
#define WKV6_T(var) do { } while (0)
#define WKV6_ACC(k, t1, t0) do { } while (0)
#endif

namespace wkv6 {
namespace chunk {

typedef short s4v __attribute__((ext_vector_type(4)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int BLK = 16;                       // tokens per block
constexpr int NBLK = 4;                       // blocks per group (= waves)
constexpr int GRP = BLK * NBLK;               // 64 tokens per group
constexpr int RSB = 160;                      // bytes per staged token row (64 bf16 + 16 pad): ds_read_b128 row reads and
                                              // ds_read_b64_tr_b16 are both conflict-free at this stride (144 B is 2-way on both)
constexpr int ARR = BLK * RSB;                // one operand array
enum { A_RH = 0, A_RL, A_KH, A_KL, A_V, N_ARR };       // forward operand arrays, bf16 [16][72] each
constexpr int OFF_E8 = N_ARR * ARR;                    // float[64]  e^{c_8}
constexpr int OFF_E16 = OFF_E8 + 256;                  // float[64]  e^{c_16}
constexpr int OFF_E16M8 = OFF_E16 + 256;               // float[64]  e^{c_16 - c_8}
constexpr int OFF_COEF = OFF_E16M8 + 256;              // float[16]  sum_i r u k
constexpr int OFF_SC = OFF_COEF + 16 * 4;              // uint4 [64]  masked scores^T, bf16x4 hi | bf16x4 lo: the B fragment of each lane
constexpr int BLK_BYTES = OFF_SC + 2 * 64 * 8;
constexpr float LW_MIN = -9.0f;
// The producers keep the log-decays in log2 units (lw * log2 e): every later exponential is then a bare v_exp_f32 (= 2^x)
// instead of v_mul + v_exp, at one extra multiply per token-channel where lw is formed.
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float LW_MIN2 = LW_MIN * LOG2E;
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// State tiles are 16x16 C-layout MFMA tiles; tile t = 2s + hb, row rho = 4g + q of a tile stands for channel
//     tile_ch(t) + 8g + q        with   tile_ch(t) = 32 (t >> 1) + 4 (t & 1)
// (not 16t + rho): with this labelling the 8 values a lane feeds into k-step s of a 16x16x32 MFMA (tiles 2s and
// 2s+1, rows 4g..4g+3) are the 8 CONTIGUOUS channels 32s + 8g .. +7, so the other operand is a plain 16-byte row
// read -- conflict-free at the 160-B row stride, unlike the ds_read2_b64 pairs a permuted k order needs.
__device__ __forceinline__ constexpr int tile_ch(int t) { return 32 * (t >> 1) + 4 * (t & 1); }
// byte offset (within a bf16 row) of the column chunk lane p supplies to a transposed read of tile t
__device__ __forceinline__ constexpr int tile_tr(int t) { return 64 * (t >> 1) + 8 * (t & 1); }

__device__ __forceinline__ s4v tr_read(const char* p)
{   // ds_read_b64_tr_b16: lane x of each 16-lane group receives column x of a 4-row x 16-column block whose
    // row q / columns 4p..4p+3 are addressed by lane 4q+p of the group
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s4v __attribute__((address_space(3)))*)(const_cast<char*>(p)));
}
__device__ __forceinline__ f4v mfma16(s4v a, s4v b, f4v c)
{
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f4v mfma32(b8v a, b8v b, f4v c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// Exchanges among the token lanes of a producer wave (lane bits 3..5) on the vector ALU instead of the LDS crossbar
// (ds_bpermute: ~100+ cycles of latency each on the producers' dependent chain, which is the critical path of a stage):
// bit 3 = the other half of the DPP row (row_ror:8), bits 4 / 5 = v_permlane16_swap / v_permlane32_swap of two copies of the value,
// which hand every lane both its own row's (half's) value and its partner's.
constexpr int DPP_SHL4 = 0x104;      // row_shl:4 (lane i <- lane i + 4)
__device__ __forceinline__ void rows16(float x, float& even, float& odd)    // even = x of the even row of each row pair, in both rows
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    even = __uint_as_float(r[0]);
    odd = __uint_as_float(r[1]);
}
__device__ __forceinline__ void halves32(float x, float& lower, float& upper)   // lower = x of lane & 31, in both halves
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lower = __uint_as_float(r[0]);
    upper = __uint_as_float(r[1]);
}

// The persistent wkv6_bi kernels (chunk_fwd_bi_kernel, chunk_bwd12k_bi_kernel) run two kernel bodies per row in a loop.  Read directly,
// hipcc hoists the bodies' ~45 kernarg loads out of that loop and keeps the whole argument block in scalar registers across both calls,
// beside each call's ~14 buffer resources: 110-134 scalar spills into vector-register lanes (and, in the backward, vector registers to
// scratch).  Instead every call re-reads the block from the kernarg segment -- the kernel's first parameter, at offset 0 -- through a pointer
// the compiler cannot see through, so nothing of it is live from call to call: a call opens with a handful of s_load_dwordx4/8, a few
// hundred cycles against the microseconds of a call.
__device__ __forceinline__ void load_kernargs(ScanArgs& dst)
{
    typedef const __attribute__((address_space(4))) char kchar;
    const unsigned long p0 = (unsigned long)(kchar*)__builtin_amdgcn_kernarg_segment_ptr();     // the kernel's first parameter sits at offset 0
    unsigned lo = (unsigned)p0, hi = (unsigned)(p0 >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    // (an inline-asm result counts as divergent: re-assert uniformity, or the block is fetched with vector loads)
    lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);
    typedef const __attribute__((address_space(4))) unsigned kword;
    kword* const kw = (kword*)(((unsigned long)hi << 32) | lo);
    // (dword by dword: a memcpy is expanded after the uniformity annotation and comes out as vector loads)
    static_assert(sizeof(ScanArgs) % 4 == 0, "whole dwords");
    unsigned* const d = reinterpret_cast<unsigned*>(&dst);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(ScanArgs) / 4); ++i) d[i] = kw[i];
}
// Workgroup slot -> (batch, head) row of the plain kernels.  Workgroups are dealt round-robin over the 8 XCDs (slots s and s + 8 share one:
// MI355X_MICROARCH.md), so in slot order an XCD's L2 sees heads h, h + 8, h + 16, h + 24 of a token row: 128-byte pieces at a 1 KB stride.
// WKV6_XCD_REMAP = 1 gives every XCD a contiguous range of (batch, head) rows instead -- at B = 8, H = 32 all 32 heads of one batch row,
// i.e. whole 4 KB token rows per L2 (profiles/r06_xcd_remap.txt has the A/B).  A permutation of the rows: results cannot change.
#ifndef WKV6_XCD_REMAP
#define WKV6_XCD_REMAP 0
#endif
__device__ __forceinline__ unsigned xcd_row_of_slot(unsigned slot, unsigned n)
{
#if WKV6_XCD_REMAP == 1
    return (n & 7u) ? slot : (slot & 7u) * (n >> 3) + (slot >> 3);
#else
    return slot;
#endif
}

// In-run clock probe (wkv6_set_clock_buffer): hardware wave 0 of a workgroup stamps {s_memtime, s_memrealtime} at its start (which = 0)
// and its end (which = 1) straight into the buffer -- nothing stays in registers in between.  a.clk == null: one scalar branch.
__device__ __forceinline__ void clock_stamp(const ScanArgs& a, unsigned slot, int which)
{
    if (a.clk && (int)slot < a.clk_slots) {
        unsigned long long c, r;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r) :: "memory");
        if ((threadIdx.x & 63) == 0) {
            a.clk[slot * 4 + 2 * which] = c;
            a.clk[slot * 4 + 2 * which + 1] = r;
        }
    }
}

__device__ __forceinline__ b8v ld_b8(const char* p) { return *reinterpret_cast<const b8v*>(p); }
__device__ __forceinline__ b8v ld_b8_2x4(const char* p0, const char* p1)
{
    const uint2 a = *reinterpret_cast<const uint2*>(p0);
    const uint2 b = *reinterpret_cast<const uint2*>(p1);
    const uint4 v = make_uint4(a.x, a.y, b.x, b.y);
    return __builtin_bit_cast(b8v, v);
}
// split 4 floats into packed bf16 hi and lo parts.  lo = x - float(hi) is formed by v_dot2(c)_f32_bf16 straight from
// the packed hi pair (hi.(-1,0) + x, exact: every partial result is representable), ~2 cycles per element where the
// unpack (shift 4, and 2.3) + subtract (2) path costs ~6 (issue rates: DESIGN.md section 4).
// The two constants of the trick -- (-1, 0) and (0, -1) as packed bf16 -- are made ONCE per wave (split_const) and handed to every
// split: kept opaque (hipcc 7.2 folds such a pair into the inline constant -1.0, which the hardware does not expand to (bf16 -1, 0) for
// this instruction: results were off by whole terms), they used to be re-materialised by two v_mov_b32 in front of every call --
// ~45 of the backward's 1058 vector instructions per SIMD and stage, ~80 of the forward's 913 per group.
struct SplitConst { unsigned c10, c01; };
__device__ __forceinline__ SplitConst split_const()
{
    unsigned c10 = 0x0000bf80u, c01 = 0xbf800000u;
    asm volatile("" : "+v"(c10), "+v"(c01));
    return SplitConst{c10, c01};
}
__device__ __forceinline__ void split4(const float (&x)[4], uint2& hi, uint2& lo, const SplitConst& sc)
{
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 m10 = __builtin_bit_cast(bf2, sc.c10), m01 = __builtin_bit_cast(bf2, sc.c01);
    hi.x = pack_bf2(x[0], x[1]);
    hi.y = pack_bf2(x[2], x[3]);
    // (lo = x - float(hi) by unpack + subtract on the plain vector ALU instead: 1.5 instructions per element, forward +3-5 %, backward
    // +2-4 %: profiles/r05_pk_dot2c_ab.txt)
    const bf2 h0 = __builtin_bit_cast(bf2, hi.x), h1 = __builtin_bit_cast(bf2, hi.y);
    lo.x = pack_bf2(__builtin_amdgcn_fdot2_f32_bf16(h0, m10, x[0], false), __builtin_amdgcn_fdot2_f32_bf16(h0, m01, x[1], false));
    lo.y = pack_bf2(__builtin_amdgcn_fdot2_f32_bf16(h1, m10, x[2], false), __builtin_amdgcn_fdot2_f32_bf16(h1, m01, x[3], false));
}

}  // namespace chunk
}  // namespace wkv6
