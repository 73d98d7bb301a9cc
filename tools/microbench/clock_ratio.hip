// s_memtime (shader clock) against s_memrealtime (100 MHz reference) over a busy loop: the effective shader clock.
// hipcc --offload-arch=gfx950 -O3 tools/microbench/clock_ratio.hip -o /tmp/clock_ratio && /tmp/clock_ratio
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, float* sink, int iters, int heavy)
{
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0));
    float x = threadIdx.x, y = 1.0001f, z = 0.5f, w = 0.25f;
    for (int i = 0; i < iters; ++i) {
        x = fmaf(x, y, z); w = fmaf(w, y, x);
        if (heavy) { z = fmaf(z, y, w); y = fmaf(y, 0.99999f, 1e-6f); x = fmaf(x, z, w); w = fmaf(w, x, y); }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1));
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
    if (x + w == 12345.f) sink[0] = x;
}
int main()
{
    unsigned long long* d; float* s;
    hipMalloc(&d, 16); hipMalloc(&s, 4);
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int blocks = (cfg & 1) ? 2048 : 1, heavy = cfg >> 1;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, d, s, 2000000, heavy);
            hipDeviceSynchronize();
        }
        unsigned long long h[2];
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("blocks %4d heavy %d: shader clocks %llu, ref ticks %llu -> %.0f MHz\n", blocks, heavy, h[0], h[1], 100.0 * h[0] / h[1]);
    }
    return 0;
}
