// What HBM bandwidth does the operator's OWN access pattern get?  The tensors are [B, T, C] with a head's slice of a token =
// 64 bf16 = 128 contiguous bytes, the next token of the same head 2 C bytes (4 KB at C = 2048) further: a (batch, head)
// workgroup streams 128-byte pieces at a 4 KB stride from four tensors and writes one (forward: 10 B per token-channel), or reads
// five and writes four (backward: 18 B).  This program does exactly that and nothing else -- one workgroup per (batch, head), 8
// waves, 8 bytes per lane (a wave-instruction = 4 token rows x 128 B, as the kernels' producers issue them), loads for the next 32 /
// 96 / 224 tokens in flight while earlier ones are stored -- and prints GB/s, against the same bytes moved as one flat stream.
//     hipcc --offload-arch=gfx950 -O3 tools/microbench/head_slices.hip -o /tmp/head_slices && /tmp/head_slices
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NIN, int NOUT, int DEPTH>
__global__ __launch_bounds__(512) void slices(const uint2* const* in, uint2* const* out, int T, int C, int H)
{
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tq = lane >> 4, c4 = lane & 15;                   // 4 token rows per wave-instruction, 16 lanes x 8 B per row
    const long base = ((long)b * T * C + (long)h * 64) / 4;     // in uint2 (4 bf16) units
    const long rowstride = C / 4;
    // each wave owns tokens  t = 32 i + 4 wave + tq  (8 waves x 4 rows = 32 tokens per workgroup iteration)
    // DEPTH iterations (of 32 tokens per workgroup) of loads in flight ahead of the one being stored: a ring of register sets
    uint2 ring[DEPTH + 1][NIN];
    auto load = [&](int i, uint2 (&r)[NIN]) {
        const long off = base + (long)(32 * i + 4 * wave + tq) * rowstride + c4;
#pragma unroll
        for (int k = 0; k < NIN; ++k) r[k] = in[k][off];
    };
    const int n = T / 32;                                       // (T is a multiple of 32 (DEPTH + 1))
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(d, ring[d]);
    for (int i0 = 0; i0 < n; i0 += DEPTH + 1) {
#pragma unroll
        for (int j = 0; j <= DEPTH; ++j) {                      // unrolled: ring slots are static register sets
            const int i = i0 + j;
            if (i + DEPTH < n) load(i + DEPTH, ring[(j + DEPTH) % (DEPTH + 1)]);
            uint2 acc = ring[j][0];
#pragma unroll
            for (int k = 1; k < NIN; ++k) { acc.x ^= ring[j][k].x; acc.y += ring[j][k].y; }
            const long off = base + (long)(32 * i + 4 * wave + tq) * rowstride + c4;
#pragma unroll
            for (int k = 0; k < NOUT; ++k) out[k][off] = make_uint2(acc.x + k, acc.y);
        }
    }
}

// variants: 16 bytes per lane (8 lanes per 128-byte row, 8 token rows per wave-instruction), and the sequence cut into `parts`
// pieces that run as separate workgroups (2 per CU, LDS-free here)
template <int NIN, int NOUT, int DEPTH>
__global__ __launch_bounds__(512) void slices16(const uint4* const* in, uint4* const* out, int T, int C, int H, int parts)
{
    const int bh = blockIdx.x / parts, part = blockIdx.x % parts;
    const int b = bh / H, h = bh % H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tq = lane >> 3, c8 = lane & 7;                    // 8 token rows per wave-instruction
    const int Tp = T / parts;
    const long base = ((long)b * T * C + (long)part * Tp * C + (long)h * 64) / 8;     // in uint4 (8 bf16) units
    const long rowstride = C / 8;
    uint4 ring[DEPTH + 1][NIN];
    auto load = [&](int i, uint4 (&r)[NIN]) {                   // 8 waves x 8 rows = 64 tokens per workgroup iteration
        const long off = base + (long)(64 * i + 8 * wave + tq) * rowstride + c8;
#pragma unroll
        for (int k = 0; k < NIN; ++k) r[k] = in[k][off];
    };
    const int n = Tp / 64;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(d, ring[d]);
    for (int i0 = 0; i0 < n; i0 += DEPTH + 1) {
#pragma unroll
        for (int j = 0; j <= DEPTH; ++j) {
            const int i = i0 + j;
            if (i + DEPTH < n) load(i + DEPTH, ring[(j + DEPTH) % (DEPTH + 1)]);
            uint4 acc = ring[j][0];
#pragma unroll
            for (int k = 1; k < NIN; ++k) { acc.x ^= ring[j][k].x; acc.y += ring[j][k].y; acc.z ^= ring[j][k].z; acc.w += ring[j][k].w; }
            const long off = base + (long)(64 * i + 8 * wave + tq) * rowstride + c8;
#pragma unroll
            for (int k = 0; k < NOUT; ++k) out[k][off] = make_uint4(acc.x + k, acc.y, acc.z, acc.w);
        }
    }
}

// The 12-wave backward's OWN instruction shapes, nothing else: per 32-token stage of a (batch, head)
//   4 "producer" waves (block, channel half): r, k, w as 8 bytes per lane, 8 lanes = 64-byte half rows, 8 rows per instruction x 2;
//   4 "column" waves: v, gy of 4 tokens per block as 8 bytes per lane, 16 lanes = full 128-byte rows; gv stores as 8 bytes per lane,
//     4 lanes = 32-byte pieces (a wave owns 16 of the 64 channels), 16 token rows per instruction, per block;
//   4 "row" waves: gr, gk, gw stores, 32-byte pieces, per block.
// WIDE = 1: the same bytes with 16 bytes per lane and full rows wherever a wave could own them: producers (block) x (token half):
//   8 lanes x 16 B = full rows, 8 rows per instruction; stores of full 128-byte rows, 16 bytes per lane, 8 rows per instruction
//   (what an LDS transposition of the 16-channel tiles would allow).
template <int WIDE, int SHAPE = 0>
__global__ __launch_bounds__(768) void bwd_shapes(const char* const* in, char* const* out, int T, int C, int H, unsigned* sink)
{
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long base = ((long)b * T * C + (long)h * 64) * 2;      // bytes
    const long row = (long)C * 2;
    unsigned acc = 0;
    for (int s = T / 32 - 1; s >= 0; --s) {
        const long st = base + (long)(32 * s) * row;
        if (wave >= 8) {                                         // producers: r, k, w (tensors 0, 1, 3)
            const int pw = wave - 8;
            if (!(WIDE & 1)) {
                const int pb = pw & 1, half = pw >> 1, tq = lane >> 3, c8 = lane & 7;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const long off = st + (long)(16 * pb + 2 * tq + tt) * row + 64 * half + 8 * c8;
                    acc ^= ((const uint2*)(in[0] + off))->x ^ ((const uint2*)(in[1] + off))->y ^ ((const uint2*)(in[3] + off))->x;
                }
            } else {
                const int tok = 8 * pw + (lane >> 3);            // 8 tokens x full rows per wave
                const long off = st + (long)tok * row + 16 * (lane & 7);
                acc ^= ((const uint4*)(in[0] + off))->x ^ ((const uint4*)(in[1] + off))->y ^ ((const uint4*)(in[3] + off))->w;
            }
        } else if (wave >= 4) {                                  // column waves: v, gy loads (tensors 2, 4), gv stores (out 2)
            const int wv = wave - 4;
            if (!(WIDE & 1)) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const long off = st + (long)(16 * blk + 4 * wv + (lane >> 4)) * row + 8 * (lane & 15);
                    acc ^= ((const uint2*)(in[2] + off))->x ^ ((const uint2*)(in[4] + off))->y;
                }
            } else {
                const long off = st + (long)(8 * wv + (lane >> 3)) * row + 16 * (lane & 7);
                acc ^= ((const uint4*)(in[2] + off))->x ^ ((const uint4*)(in[4] + off))->y;
            }
            if (SHAPE == 4) {                                     // loads only
            } else if (!(WIDE & 2) && SHAPE == 1) {                      // 32-byte pieces, 16 B per lane, the stage's 32 tokens in one instruction
                const long so = st + (long)(lane >> 1) * row + 32 * wv + 16 * (lane & 1);
                *(uint4*)(out[2] + so) = make_uint4(acc, s, lane, wv);
            } else if (!(WIDE & 2) && SHAPE == 2) {               // 64-byte pieces (a wave pair's 32 channels), 16 B per lane, 16 tokens per instruction
                const long so = st + (long)(16 * (wv & 1) + (lane >> 2)) * row + 64 * (wv >> 1) + 16 * (lane & 3);
                *(uint4*)(out[2] + so) = make_uint4(acc, s, lane, wv);
            } else if (!(WIDE & 2)) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const long so = st + (long)(16 * blk + (lane & 15)) * row + 32 * wv + 8 * (lane >> 4);
                    if (SHAPE == 3) __builtin_nontemporal_store(((unsigned long long)s << 32) | acc, (unsigned long long*)(out[2] + so));
                    else *(uint2*)(out[2] + so) = make_uint2(acc, s);
                }
            } else {
                const long off = st + (long)(8 * wv + (lane >> 3)) * row + 16 * (lane & 7);
                *(uint4*)(out[2] + off) = make_uint4(acc, s, lane, wv);
            }
        } else {                                                 // row waves: gr, gk, gw stores (out 0, 1, 3)
            if (SHAPE == 4) {
            } else if (!(WIDE & 2) && SHAPE == 1) {
                const long so = st + (long)(lane >> 1) * row + 32 * wave + 16 * (lane & 1);
                *(uint4*)(out[0] + so) = make_uint4(s, lane, wave, 0);
                *(uint4*)(out[1] + so) = make_uint4(s, lane, wave, 1);
                *(uint4*)(out[3] + so) = make_uint4(s, lane, wave, 3);
            } else if (!(WIDE & 2) && SHAPE == 2) {
                const long so = st + (long)(16 * (wave & 1) + (lane >> 2)) * row + 64 * (wave >> 1) + 16 * (lane & 3);
                *(uint4*)(out[0] + so) = make_uint4(s, lane, wave, 0);
                *(uint4*)(out[1] + so) = make_uint4(s, lane, wave, 1);
                *(uint4*)(out[3] + so) = make_uint4(s, lane, wave, 3);
            } else if (!(WIDE & 2)) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const long so = st + (long)(16 * blk + (lane & 15)) * row + 32 * wave + 8 * (lane >> 4);
                    if (SHAPE == 3) {
                        __builtin_nontemporal_store(((unsigned long long)s << 32) | lane, (unsigned long long*)(out[0] + so));
                        __builtin_nontemporal_store(((unsigned long long)s << 32) | wave, (unsigned long long*)(out[1] + so));
                        __builtin_nontemporal_store(((unsigned long long)lane << 32) | wave, (unsigned long long*)(out[3] + so));
                    } else {
                    *(uint2*)(out[0] + so) = make_uint2(s, lane);
                    *(uint2*)(out[1] + so) = make_uint2(s, wave);
                    *(uint2*)(out[3] + so) = make_uint2(lane, wave);
                    }
                }
            } else {
                const long off = st + (long)(8 * wave + (lane >> 3)) * row + 16 * (lane & 7);
                *(uint4*)(out[0] + off) = make_uint4(s, lane, wave, 0);
                *(uint4*)(out[1] + off) = make_uint4(s, lane, wave, 1);
                *(uint4*)(out[3] + off) = make_uint4(s, lane, wave, 3);
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}


// The 8-wave forward's OWN instruction shapes, nothing else (round 5): per 64-token group of a (batch, head)
//   4 "producer" waves (wave = 16-token block): r, k, v, w as 8 bytes per lane, 16 lanes = one 128-byte row, 4 rows per instruction,
//     4 instructions per tensor (lane = 4 channels x tokens 4 tq .. 4 tq + 3), the next group's loads in flight;
//   4 "consumer" waves (wave = 16 of the 64 value channels): y stores per block, 8 bytes per lane, 4 lanes = 32-byte pieces of 16
//     token rows; and per group one fp32 state checkpoint, four 16-byte-per-lane stores of 1 KB contiguous each.
// WIDE & 1: loads as 16 bytes per lane (8 lanes = one row, 8 rows per instruction, 2 instructions per tensor);
// WIDE & 2: y as full 128-byte rows, 16 bytes per lane (a consumer wave stores 16 token rows of the group: 2 instructions per group);
// WIDE & 4: no checkpoint stores.
template <int WIDE>
__global__ __launch_bounds__(512) void fwd_shapes(const char* const* in, char* const* out, float* ckpt, int T, int C, int H, unsigned* sink)
{
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long base = ((long)b * T * C + (long)h * 64) * 2;      // bytes
    const long row = (long)C * 2;
    unsigned acc = 0;
    const int ng = T / 64;
    if (wave >= 4) {
        const int wv = wave - 4;
        if (!(WIDE & 1)) {
            uint2 cur[4][4], nxt[4][4];
            auto load = [&](int g, uint2 (&r)[4][4]) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const long off = base + (long)(64 * g + 16 * wv + 4 * (lane >> 4) + tt) * row + 8 * (lane & 15);
#pragma unroll
                    for (int k = 0; k < 4; ++k) r[k][tt] = *(const uint2*)(in[k] + off);
                }
            };
            load(0, cur);
            for (int g = 0; g < ng; ++g) {
                if (g + 1 < ng) load(g + 1, nxt);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) { acc ^= cur[k][tt].x; acc += cur[k][tt].y; cur[k][tt] = nxt[k][tt]; }
            }
        } else {
            uint4 cur[4][2], nxt[4][2];
            auto load = [&](int g, uint4 (&r)[4][2]) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const long off = base + (long)(64 * g + 16 * wv + 2 * (lane >> 3) + tt) * row + 16 * (lane & 7);
#pragma unroll
                    for (int k = 0; k < 4; ++k) r[k][tt] = *(const uint4*)(in[k] + off);
                }
            };
            load(0, cur);
            for (int g = 0; g < ng; ++g) {
                if (g + 1 < ng) load(g + 1, nxt);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) { acc ^= cur[k][tt].x ^ cur[k][tt].z; acc += cur[k][tt].y + cur[k][tt].w; cur[k][tt] = nxt[k][tt]; }
            }
        }
    } else {
        const int wv = wave, x = lane & 15, g4 = lane >> 4;
        float* const ck = ckpt + (long)blockIdx.x * ng * 4096;
        for (int g = 0; g < ng; ++g) {
            if (!(WIDE & 4)) {
#pragma unroll
                for (int wb = 0; wb < 4; ++wb)
                    *(uint4*)(ck + (long)g * 4096 + ((wb * 4 + wv) * 64 + lane) * 4) = make_uint4(g, lane, wv, wb);
            }
            if (!(WIDE & 2)) {
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    const long so = base + (long)(64 * g + 16 * blk + x) * row + 32 * wv + 8 * g4;
                    *(uint2*)(out[0] + so) = make_uint2(g, lane);
                }
            } else {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const long so = base + (long)(64 * g + 16 * wv + 8 * half + (lane >> 3)) * row + 16 * (lane & 7);
                    *(uint4*)(out[0] + so) = make_uint4(g, lane, wv, half);
                }
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int NIN, int NOUT>
__global__ __launch_bounds__(512) void flat(const uint4* const* in, uint4* const* out, long n16)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
        uint4 acc = in[0][i];
#pragma unroll
        for (int k = 1; k < NIN; ++k) { const uint4 t = in[k][i]; acc.x ^= t.x; acc.y += t.y; acc.z ^= t.z; acc.w += t.w; }
#pragma unroll
        for (int k = 0; k < NOUT; ++k) out[k][i] = make_uint4(acc.x + k, acc.y, acc.z, acc.w);
    }
}

template <int NIN, int NOUT, int DEPTH> void run(const char* name, int B, int T, int C, int H, void** bufs)
{
    const void** din; void** dout;
    hipMalloc(&din, NIN * sizeof(void*)); hipMalloc(&dout, NOUT * sizeof(void*));
    hipMemcpy(din, bufs, NIN * sizeof(void*), hipMemcpyHostToDevice);
    hipMemcpy(dout, bufs + NIN, NOUT * sizeof(void*), hipMemcpyHostToDevice);
    const double bytes = (double)B * T * C * 2 * (NIN + NOUT);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < (DEPTH == 1 ? 4 : 1); ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 20; ++it) {
                if (mode == 0) hipLaunchKernelGGL((slices<NIN, NOUT, DEPTH>), dim3(B * H), dim3(512), 0, 0, (const uint2* const*)din, (uint2* const*)dout, T, C, H);
                else if (mode == 2) hipLaunchKernelGGL((slices16<NIN, NOUT, 1>), dim3(B * H), dim3(512), 0, 0, (const uint4* const*)din, (uint4* const*)dout, T, C, H, 1);
                else if (mode == 3) hipLaunchKernelGGL((slices16<NIN, NOUT, 1>), dim3(2 * B * H), dim3(512), 0, 0, (const uint4* const*)din, (uint4* const*)dout, T, C, H, 2);
                else hipLaunchKernelGGL((flat<NIN, NOUT>), dim3(4096), dim3(512), 0, 0, (const uint4* const*)din, (uint4* const*)dout, (long)B * T * C * 2 / 16);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2 && ms / 20 < best) best = ms / 20;
        }
        if (mode == 0) printf("%-30s 128-byte head slices, 4 KB stride, %d x 32 tokens of loads in flight (%3d KB per CU): %.4f ms  %.0f GB/s\n", name,
                              DEPTH, DEPTH * 32 * 128 * NIN / 1024, best, bytes / best / 1e6);
        else if (mode == 1) printf("%-30s the same bytes as flat 16-byte streams: %.4f ms  %.0f GB/s\n", name, best, bytes / best / 1e6);
        else if (mode == 2) printf("%-30s head slices with 16 bytes per lane (8 rows per wave-instruction): %.4f ms  %.0f GB/s\n", name, best, bytes / best / 1e6);
        else printf("%-30s ... and two workgroups per (batch, head), half the sequence each: %.4f ms  %.0f GB/s\n", name, best, bytes / best / 1e6);
    }
    hipFree(din); hipFree(dout);
}

int main()
{
    const int B = 8, T = 4096, C = 2048, H = 32;
    void* bufs[9];
    for (int i = 0; i < 9; ++i) { hipMalloc(&bufs[i], (size_t)B * T * C * 2); hipMemset(bufs[i], i + 1, (size_t)B * T * C * 2); }
    printf("B=%d T=%d C=%d H=%d, one 512-thread workgroup per (batch, head), bf16 tensors of %.0f MB\n", B, T, C, H, B * T * C * 2 / 1e6);
    run<4, 1, 1>("forward pattern (4 in, 1 out)", B, T, C, H, bufs);
    run<4, 1, 3>("forward pattern (4 in, 1 out)", B, T, C, H, bufs);
    run<4, 1, 7>("forward pattern (4 in, 1 out)", B, T, C, H, bufs);
    run<5, 4, 1>("backward pattern (5 in, 4 out)", B, T, C, H, bufs);
    run<5, 4, 3>("backward pattern (5 in, 4 out)", B, T, C, H, bufs);
    run<5, 4, 7>("backward pattern (5 in, 4 out)", B, T, C, H, bufs);
    {   // the 12-wave backward's own instruction shapes
        const void** din; void** dout; unsigned* sink;
        hipMalloc(&din, 5 * sizeof(void*)); hipMalloc(&dout, 4 * sizeof(void*)); hipMalloc(&sink, 4);
        hipMemcpy(din, bufs, 5 * sizeof(void*), hipMemcpyHostToDevice);
        hipMemcpy(dout, bufs + 5, 4 * sizeof(void*), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const double bytes = (double)B * T * C * 2 * 9;
        for (int wide = 0; wide < 8; ++wide) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                for (int it = 0; it < 20; ++it) {
                    if (wide == 7) hipLaunchKernelGGL((bwd_shapes<0, 4>), dim3(B * H), dim3(768), 0, 0, (const char* const*)din, (char* const*)dout, T, C, H, sink);
                    else if (wide == 6) hipLaunchKernelGGL((bwd_shapes<0, 3>), dim3(B * H), dim3(768), 0, 0, (const char* const*)din, (char* const*)dout, T, C, H, sink);
                    else if (wide == 5) hipLaunchKernelGGL((bwd_shapes<0, 2>), dim3(B * H), dim3(768), 0, 0, (const char* const*)din, (char* const*)dout, T, C, H, sink);
                    else if (wide == 4) hipLaunchKernelGGL((bwd_shapes<0, 1>), dim3(B * H), dim3(768), 0, 0, (const char* const*)din, (char* const*)dout, T, C, H, sink);
                    else if (wide == 3) hipLaunchKernelGGL(bwd_shapes<3>, dim3(B * H), dim3(768), 0, 0, (const char* const*)din, (char* const*)dout, T, C, H, sink);
                    else if (wide == 2) hipLaunchKernelGGL(bwd_shapes<2>, dim3(B * H), dim3(768), 0, 0, (const char* const*)din, (char* const*)dout, T, C, H, sink);
                    else if (wide == 1) hipLaunchKernelGGL(bwd_shapes<1>, dim3(B * H), dim3(768), 0, 0, (const char* const*)din, (char* const*)dout, T, C, H, sink);
                    else hipLaunchKernelGGL(bwd_shapes<0>, dim3(B * H), dim3(768), 0, 0, (const char* const*)din, (char* const*)dout, T, C, H, sink);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2 && ms / 20 < best) best = ms / 20;
            }
            const char* names[8] = {"backward, 12 waves, the kernel's own instruction shapes:", "... loads as 16 B per lane / full rows, stores as the kernel's:",
                                    "... loads as the kernel's, stores as 16 B per lane / full rows:", "... 16 B per lane and full rows everywhere:",
                                    "... stores: 32-byte pieces but 16 B per lane (32 tokens per instruction):", "... stores: 64-byte pieces, 16 B per lane (16 tokens per instruction):", "... the kernel's shapes with non-temporal stores:", "... the kernel's loads, no stores at all (the bytes counted are the loads' 5/9):"};
            printf("%-66s %.4f ms  %.0f GB/s\n", names[wide], best, (wide == 7 ? bytes * 5 / 9 : bytes) / best / 1e6);
        }
    }
    {   // the 8-wave forward's own instruction shapes
        const void** din; void** dout; unsigned* sink; float* ck;
        hipMalloc(&din, 4 * sizeof(void*)); hipMalloc(&dout, sizeof(void*)); hipMalloc(&sink, 4); hipMalloc(&ck, (size_t)B * H * (T / 64) * 4096 * 4);
        hipMemcpy(din, bufs, 4 * sizeof(void*), hipMemcpyHostToDevice);
        hipMemcpy(dout, bufs + 4, sizeof(void*), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const char* names[6] = {"forward, 8 waves, the kernel's own instruction shapes (+ checkpoints):", "... loads as 16 B per lane (8 rows per instruction):",
                                "... y as full 128-byte rows, 16 B per lane:", "... both:", "... the kernel's shapes without checkpoint stores:", "... both, without checkpoint stores:"};
        const int modes[6] = {0, 1, 2, 3, 4, 7};
        for (int m = 0; m < 6; ++m) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                for (int it = 0; it < 20; ++it) {
#define FW(W) hipLaunchKernelGGL(fwd_shapes<W>, dim3(B * H), dim3(512), 0, 0, (const char* const*)din, (char* const*)dout, ck, T, C, H, sink)
                    switch (modes[m]) { case 0: FW(0); break; case 1: FW(1); break; case 2: FW(2); break; case 3: FW(3); break; case 4: FW(4); break; default: FW(7); }
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2 && ms / 20 < best) best = ms / 20;
            }
            const double bytes = (double)B * T * C * 2 * 5 + ((modes[m] & 4) ? 0.0 : (double)B * T * C * 4);
            printf("%-76s %.4f ms  %.0f GB/s\n", names[m], best, bytes / best / 1e6);
        }
    }
    {   // the 8-wave forward's own instruction shapes
        const void** din; void** dout; unsigned* sink; float* ck;
        hipMalloc(&din, 4 * sizeof(void*)); hipMalloc(&dout, sizeof(void*)); hipMalloc(&sink, 4); hipMalloc(&ck, (size_t)B * H * (T / 64) * 4096 * 4);
        hipMemcpy(din, bufs, 4 * sizeof(void*), hipMemcpyHostToDevice);
        hipMemcpy(dout, bufs + 4, sizeof(void*), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const char* names[6] = {"forward, 8 waves, the kernel's own instruction shapes (+ checkpoints):", "... loads as 16 B per lane (8 rows per instruction):",
                                "... y as full 128-byte rows, 16 B per lane:", "... both:", "... the kernel's shapes without checkpoint stores:", "... both, without checkpoint stores:"};
        const int modes[6] = {0, 1, 2, 3, 4, 7};
        for (int m = 0; m < 6; ++m) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                for (int it = 0; it < 20; ++it) {
#define FW(W) hipLaunchKernelGGL(fwd_shapes<W>, dim3(B * H), dim3(512), 0, 0, (const char* const*)din, (char* const*)dout, ck, T, C, H, sink)
                    switch (modes[m]) { case 0: FW(0); break; case 1: FW(1); break; case 2: FW(2); break; case 3: FW(3); break; case 4: FW(4); break; default: FW(7); }
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2 && ms / 20 < best) best = ms / 20;
            }
            const double bytes = (double)B * T * C * 2 * 5 + ((modes[m] & 4) ? 0.0 : (double)B * T * C * 4);
            printf("%-76s %.4f ms  %.0f GB/s\n", names[m], best, bytes / best / 1e6);
        }
    }
    return 0;
}
