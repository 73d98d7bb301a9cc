// Shared device helpers for the gfx950 WKV6 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wkv6 {

constexpr int HEAD = 64;             // head size N (= _N_ of the reference build, src/model.py:189)

typedef unsigned short bf16_t;       // raw bfloat16 bits
typedef _Float16 f16_t;              // IEEE half (inference kernel with fp16 I/O: cuda/rwkv6_op.cpp:9, 16-19)

__device__ __forceinline__ float bf_lo(uint32_t x) { return __uint_as_float(x << 16); }
__device__ __forceinline__ float bf_hi(uint32_t x) { return __uint_as_float(x & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi)
{   // round-to-nearest-even (v_cvt_pk_bf16_f32), same rounding as at::BFloat16(float)
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v = {lo, hi};
    bf2 h = __builtin_convertvector(v, bf2);
    return __builtin_bit_cast(uint32_t, h);
}

// ---- buffer-resource access: a 128-bit descriptor (base, byte count) in SGPRs + a 32-bit byte offset per lane.  Against
// flat global_load/store this saves the 64-bit address arithmetic per access, and the hardware drops accesses at or past the
// byte count (loads return 0), which replaces the per-lane "token < length" predicates with their exec-mask branches.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);   // raw buffer, gfx9 flags
}
__device__ __forceinline__ uint2 buf_load8(rsrc_t r, unsigned off)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ float4 buf_load16f(rsrc_t r, unsigned off)
{
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void buf_store16f(rsrc_t r, unsigned off, const float (&v)[4])
{
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(v4u{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                                           r, (int)off, 0, 0);
}
__device__ __forceinline__ void buf_store16(rsrc_t r, unsigned off, uint4 v)
{
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(v4u{v.x, v.y, v.z, v.w}, r, (int)off, 0, 0);
}
__device__ __forceinline__ void buf_store8(rsrc_t r, unsigned off, uint2 v)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(v2u{v.x, v.y}, r, (int)off, 0, 0);
}

// ---- 4-wide channel I/O in the operator's I/O type (bf16 or float) ------------------------------
template <typename T> struct io4;
template <> struct io4<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&o)[4])
    {
        const uint2 raw = *reinterpret_cast<const uint2*>(p);
        o[0] = bf_lo(raw.x); o[1] = bf_hi(raw.x); o[2] = bf_lo(raw.y); o[3] = bf_hi(raw.y);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4])
    {
        uint2 raw; raw.x = pack_bf2(v[0], v[1]); raw.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = raw;
    }
};
template <> struct io4<f16_t> {      // widened to fp32 on load (exact), rounded to nearest even on store, as cuda/rwkv6.cu:8-71
    typedef f16_t h4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void load(const f16_t* p, float (&o)[4])
    {
        const h4 raw = *reinterpret_cast<const h4*>(p);
        o[0] = (float)raw[0]; o[1] = (float)raw[1]; o[2] = (float)raw[2]; o[3] = (float)raw[3];
    }
    static __device__ __forceinline__ void store(f16_t* p, const float (&v)[4])
    {
        const h4 raw = {(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3]};
        *reinterpret_cast<h4*>(p) = raw;
    }
};
template <> struct io4<float> {
    static __device__ __forceinline__ void load(const float* p, float (&o)[4])
    {
        const float4 raw = *reinterpret_cast<const float4*>(p);
        o[0] = raw.x; o[1] = raw.y; o[2] = raw.z; o[3] = raw.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4])
    {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};

// ---- cross-lane helpers ---------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int DPP_ROR4 = 0x124;      // row_ror:4
constexpr int DPP_ROR8 = 0x128;      // row_ror:8
constexpr int DPP_BCAST0 = 0x150;    // row_newbcast:0 (lane 0 of each row of 16 to the whole row)

// Sum over the 16 lanes of a DPP row (lanes 16q..16q+15).  Every lane gets the total.
__device__ __forceinline__ float row_sum16(float x)
{
    x += dpp_mov<DPP_XOR1>(x);
    x += dpp_mov<DPP_XOR2>(x);
    x += dpp_mov<DPP_ROR4>(x);
    x += dpp_mov<DPP_ROR8>(x);
    return x;
}
// Transposing row reductions: NV values per lane are summed over the 16 lanes of the row; on
// return lane `c` (= lane & 15) holds the total of value index row_sel<NV>(c).
template <int NV> __device__ __forceinline__ int row_sel(int c);
template <> __device__ __forceinline__ int row_sel<4>(int c) { return ((c & 1) << 1) | ((c >> 1) & 1); }
template <> __device__ __forceinline__ int row_sel<2>(int c) { return c & 1; }

__device__ __forceinline__ float row_reduce(const float (&y)[4], int c)
{
    const bool b0 = c & 1, b1 = c & 2;
    const float send0 = b0 ? y[0] : y[2], send1 = b0 ? y[1] : y[3];
    const float keep0 = b0 ? y[2] : y[0], keep1 = b0 ? y[3] : y[1];
    const float z0 = keep0 + dpp_mov<DPP_XOR1>(send0);
    const float z1 = keep1 + dpp_mov<DPP_XOR1>(send1);
    const float send = b1 ? z0 : z1, keep = b1 ? z1 : z0;
    float q = keep + dpp_mov<DPP_XOR2>(send);
    q += dpp_mov<DPP_ROR4>(q);
    q += dpp_mov<DPP_ROR8>(q);
    return q;
}
__device__ __forceinline__ float row_reduce(const float (&y)[2], int c)
{
    const bool b0 = c & 1;
    const float send = b0 ? y[0] : y[1], keep = b0 ? y[1] : y[0];
    float q = keep + dpp_mov<DPP_XOR1>(send);
    q += dpp_mov<DPP_XOR2>(q);
    q += dpp_mov<DPP_ROR4>(q);
    q += dpp_mov<DPP_ROR8>(q);
    return q;
}

// Sum 4 values over the 4 rows of the wave (lanes l, l^16, l^32, l^48); on return a lane in row
// rho (= lane >> 4) holds the total of value index col_sel(rho) = bit-reverse2(rho).
__device__ __forceinline__ int col_sel(int rho) { return ((rho & 1) << 1) | (rho >> 1); }
__device__ __forceinline__ float col_reduce(const float (&x)[4])
{
    // v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[0]), __float_as_uint(x[1]), false, false);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[2]), __float_as_uint(x[3]), false, false);
    const float s01 = __uint_as_float(a[0]) + __uint_as_float(a[1]);   // lanes<32: x0, lanes>=32: x1
    const float s23 = __uint_as_float(b[0]) + __uint_as_float(b[1]);   // lanes<32: x2, lanes>=32: x3
    // v_permlane16_swap: odd rows of the first operand <-> even rows of the second
    auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
    return __uint_as_float(c[0]) + __uint_as_float(c[1]);
}
__device__ __forceinline__ float col_reduce_ref(const float (&x)[4], int rho)
{   // same contract through ds_bpermute shuffles (used by the device self-test)
    float t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float s = x[q];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        t[q] = s;
    }
    const int sel = col_sel(rho);
    return sel == 0 ? t[0] : sel == 1 ? t[1] : sel == 2 ? t[2] : t[3];
}

}  // namespace wkv6
