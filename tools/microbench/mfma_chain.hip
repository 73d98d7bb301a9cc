#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
// ones in v[20:23] (bf16 pairs).  first MFMA: D = ones*ones + c ; second: D2 = ones*ones + D.
#define SETUP "v_mov_b32 v20, 0x3f803f80\n v_mov_b32 v21, 0x3f803f80\n v_mov_b32 v22, 0x3f803f80\n v_mov_b32 v23, 0x3f803f80\n" \
              "v_mov_b32 v10, %4\n v_mov_b32 v11, %5\n v_mov_b32 v12, %6\n v_mov_b32 v13, %7\n s_nop 7\n s_nop 7\n"
#define READ "s_nop 7\n s_nop 7\n s_nop 7\n v_mov_b32 %0, v10\n v_mov_b32 %1, v11\n v_mov_b32 %2, v12\n v_mov_b32 %3, v13"
#define CLOB "v10", "v11", "v12", "v13", "v20", "v21", "v22", "v23"
#define M32 "v_mfma_f32_16x16x32_bf16 v[10:13], v[20:23], v[20:23], v[10:13]\n"
#define M16 "v_mfma_f32_16x16x16_bf16 v[10:13], v[20:21], v[20:21], v[10:13]\n"
#define DEF(NAME, FIRST, GAP, SECOND) __device__ void NAME(f4v c, float* o) { \
    asm volatile(SETUP FIRST GAP SECOND READ : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) \
                 : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]) : CLOB); }
DEF(a0, M32, "", M16) DEF(a1, M32, "s_nop 0\n", M16) DEF(a2, M32, "s_nop 1\n", M16) DEF(a3, M32, "s_nop 2\n", M16)
DEF(a4, M32, "s_nop 3\n", M16) DEF(a5, M32, "s_nop 4\n", M16) DEF(a6, M32, "s_nop 5\n", M16) DEF(a7, M32, "s_nop 6\n", M16)
DEF(b0, M16, "", M32) DEF(b1, M16, "s_nop 0\n", M32) DEF(b2, M16, "s_nop 1\n", M32) DEF(b3, M16, "s_nop 2\n", M32)
DEF(b4, M16, "s_nop 3\n", M32) DEF(b5, M16, "s_nop 4\n", M32)
DEF(c0, M32, "", M32) DEF(c1, M16, "", M16)
__global__ void k(float* out)
{
    f4v c = {100.f, 200.f, 300.f, 400.f};
    float r[16][4];
    a0(c, r[0]); a1(c, r[1]); a2(c, r[2]); a3(c, r[3]); a4(c, r[4]); a5(c, r[5]); a6(c, r[6]); a7(c, r[7]);
    b0(c, r[8]); b1(c, r[9]); b2(c, r[10]); b3(c, r[11]); b4(c, r[12]); b5(c, r[13]); c0(c, r[14]); c1(c, r[15]);
    for (int i = 0; i < 16; ++i) for (int q = 0; q < 4; ++q) out[(i * 4 + q) * 64 + threadIdx.x] = r[i][q];
}
int main()
{
    float* d; hipMalloc(&d, 64 * 64 * 4);
    k<<<1, 64>>>(d);
    static float h[64 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[16] = {"32->16 none", "32->16 nop0", "32->16 nop1", "32->16 nop2", "32->16 nop3", "32->16 nop4", "32->16 nop5", "32->16 nop6",
                             "16->32 none", "16->32 nop0", "16->32 nop1", "16->32 nop2", "16->32 nop3", "16->32 nop4", "32->32 none", "16->16 none"};
    const float add[16] = {48, 48, 48, 48, 48, 48, 48, 48, 48, 48, 48, 48, 48, 48, 64, 32};
    for (int i = 0; i < 16; ++i) {
        printf("%-14s", names[i]);
        for (int q = 0; q < 4; ++q) {
            int bad = 0; float want = add[i] + 100.f * (q + 1);
            for (int l = 0; l < 64; ++l) bad += h[(i * 4 + q) * 64 + l] != want;
            printf("  reg%d bad %2d (lane0 %g)", q, bad, h[(i * 4 + q) * 64]);
        }
        printf("\n");
    }
    return 0;
}
