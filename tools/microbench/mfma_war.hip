#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4v __attribute__((ext_vector_type(4)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
// (a) overwrite SrcB / SrcA N wait states after issue.  a,b ones -> 16 + c.
template <int N> __device__ void war_ab(f4v c, float* o)
{
    asm volatile(
        "v_mov_b32 v20, 0x3f803f80\n v_mov_b32 v21, 0x3f803f80\n v_mov_b32 v22, 0x3f803f80\n v_mov_b32 v23, 0x3f803f80\n"
        "s_nop 7\n"
        "v_mfma_f32_16x16x16_bf16 v[10:13], v[20:21], v[22:23], %4\n"
        "s_nop %5\n"
        "v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n"
        "s_nop 7\n s_nop 7\n"
        "v_mov_b32 %0, v10\n v_mov_b32 %1, v11\n v_mov_b32 %2, v12\n v_mov_b32 %3, v13"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(c), "n"(N)
        : "v10", "v11", "v12", "v13", "v20", "v21", "v22", "v23");
}
// (b) dst partially overlapping SrcC: C in v[12:15], D in v[10:13]  and the other direction C in v[10:13], D in v[12:15]
__device__ void overlap_lo(f4v c, float* o)
{
    asm volatile(
        "v_mov_b32 v20, 0x3f803f80\n v_mov_b32 v21, 0x3f803f80\n"
        "v_mov_b32 v12, %4\n v_mov_b32 v13, %5\n v_mov_b32 v14, %6\n v_mov_b32 v15, %7\n s_nop 7\n"
        "v_mfma_f32_16x16x16_bf16 v[10:13], v[20:21], v[20:21], v[12:15]\n"
        "s_nop 7\n s_nop 7\n"
        "v_mov_b32 %0, v10\n v_mov_b32 %1, v11\n v_mov_b32 %2, v12\n v_mov_b32 %3, v13"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3])
        : "v10", "v11", "v12", "v13", "v14", "v15", "v20", "v21");
}
__device__ void overlap_hi(f4v c, float* o)
{
    asm volatile(
        "v_mov_b32 v20, 0x3f803f80\n v_mov_b32 v21, 0x3f803f80\n"
        "v_mov_b32 v10, %4\n v_mov_b32 v11, %5\n v_mov_b32 v12, %6\n v_mov_b32 v13, %7\n s_nop 7\n"
        "v_mfma_f32_16x16x16_bf16 v[12:15], v[20:21], v[20:21], v[10:13]\n"
        "s_nop 7\n s_nop 7\n"
        "v_mov_b32 %0, v12\n v_mov_b32 %1, v13\n v_mov_b32 %2, v14\n v_mov_b32 %3, v15"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3])
        : "v10", "v11", "v12", "v13", "v14", "v15", "v20", "v21");
}
__global__ void k(float* out)
{
    f4v c = {100.f, 200.f, 300.f, 400.f};
    float r[6][4];
    war_ab<0>(c, r[0]); war_ab<1>(c, r[1]); war_ab<2>(c, r[2]); war_ab<4>(c, r[3]);
    overlap_lo(c, r[4]); overlap_hi(c, r[5]);
    for (int i = 0; i < 6; ++i) for (int q = 0; q < 4; ++q) out[(i * 4 + q) * 64 + threadIdx.x] = r[i][q];
}
int main()
{
    float* d; hipMalloc(&d, 24 * 64 * 4);
    k<<<1, 64>>>(d);
    float h[24 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"war_ab nop0", "war_ab nop1", "war_ab nop2", "war_ab nop4", "overlap D below C", "overlap D above C"};
    for (int i = 0; i < 6; ++i) {
        printf("%-20s", names[i]);
        for (int q = 0; q < 4; ++q) {
            int bad = 0; float want = 116.f + 100.f * q;
            for (int l = 0; l < 64; ++l) bad += h[(i * 4 + q) * 64 + l] != want;
            printf("  reg%d bad %2d (lane0 %g, lane63 %g)", q, bad, h[(i * 4 + q) * 64], h[(i * 4 + q) * 64 + 63]);
        }
        printf("\n");
    }
    return 0;
}
