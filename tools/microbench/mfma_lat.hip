#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4v __attribute__((ext_vector_type(4)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
template <int N> __device__ void probe16(s4v a, s4v b, f4v c, float& o0, float& o3)
{
    asm volatile("v_mfma_f32_16x16x16_bf16 v[10:13], %2, %3, %4\n s_nop %5\n v_mov_b32 %0, v10\n v_mov_b32 %1, v13"
                 : "=&v"(o0), "=&v"(o3) : "v"(a), "v"(b), "v"(c), "n"(N) : "v10", "v11", "v12", "v13");
}
template <int N> __device__ void probe32(b8v a, b8v b, f4v c, float& o0, float& o3)
{
    asm volatile("v_mfma_f32_16x16x32_bf16 v[10:13], %2, %3, %4\n s_nop %5\n v_mov_b32 %0, v10\n v_mov_b32 %1, v13"
                 : "=&v"(o0), "=&v"(o3) : "v"(a), "v"(b), "v"(c), "n"(N) : "v10", "v11", "v12", "v13");
}
__global__ void k(float* out)
{
    const short one = 0x3f80;
    s4v a = {one, one, one, one}, b = a;
    b8v a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)1.0f; b8[i] = (__bf16)1.0f; }
    f4v c = {100.f, 100.f, 100.f, 100.f};
    float r16[16], r32[16], q16[16], q32[16];
#define P(N) probe16<N>(a, b, c, r16[N], q16[N]); probe32<N>(a8, b8, c, r32[N], q32[N]);
    P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7) P(8) P(9) P(10) P(11) P(12) P(13) P(14) P(15)
    // report lane-wise: any lane wrong?
    for (int n = 0; n < 16; ++n) {
        out[(n * 4 + 0) * 64 + threadIdx.x] = r16[n];
        out[(n * 4 + 1) * 64 + threadIdx.x] = r32[n];
        out[(n * 4 + 2) * 64 + threadIdx.x] = q16[n];
        out[(n * 4 + 3) * 64 + threadIdx.x] = q32[n];
    }
}
int main()
{
    float* d; hipMalloc(&d, 64 * 64 * 4);
    k<<<1, 64>>>(d);
    float h[64 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int n = 0; n < 16; ++n) {
        int bad16 = 0, bad32 = 0, badq16 = 0, badq32 = 0;
        for (int l = 0; l < 64; ++l) {
            bad16 += h[(n * 4) * 64 + l] != 116.f; bad32 += h[(n * 4 + 1) * 64 + l] != 132.f;
            badq16 += h[(n * 4 + 2) * 64 + l] != 116.f; badq32 += h[(n * 4 + 3) * 64 + l] != 132.f;
        }
        printf("s_nop %2d: 16x16x16 bad lanes reg0 %2d reg3 %2d (lane0 %g)   16x16x32 bad lanes reg0 %2d reg3 %2d (lane0 %g)\n", n, bad16, badq16, h[(n * 4) * 64], bad32, badq32, h[(n * 4 + 1) * 64]);
    }
    return 0;
}
