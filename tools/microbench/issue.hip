#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND> __global__ void k(long long* out, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    long long t0 = clock64();
    for (int it = 0; it < 16; ++it) {
        if constexpr (KIND == 0) asm volatile(REP64("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %6, %6, %6, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 1) asm volatile(REP64("v_pk_fma_f32 v[10:11], v[10:11], v[10:11], v[12:13]\n v_pk_fma_f32 v[14:15], v[14:15], v[14:15], v[16:17]\n v_pk_fma_f32 v[18:19], v[18:19], v[18:19], v[20:21]\n v_pk_fma_f32 v[22:23], v[22:23], v[22:23], v[24:25]\n") ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25");
        if constexpr (KIND == 2) asm volatile(REP64("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %4, %5\n v_cvt_pk_bf16_f32 %6, %7, %1\n v_cvt_pk_bf16_f32 %0, %2, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 3) asm volatile(REP64("v_exp_f32 %0, %1\n v_exp_f32 %2, %3\n v_exp_f32 %4, %5\n v_exp_f32 %6, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 4) asm volatile(REP64("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 5) asm volatile(REP64("v_and_b32 %0, 0xffff0000, %1\n v_lshlrev_b32 %2, 16, %3\n v_and_b32 %4, 0xffff0000, %5\n v_lshlrev_b32 %6, 16, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 6) asm volatile(REP64("v_pk_mul_f32 v[10:11], v[10:11], v[12:13]\n v_pk_add_f32 v[14:15], v[14:15], v[16:17]\n v_pk_mul_f32 v[18:19], v[18:19], v[20:21]\n v_pk_add_f32 v[22:23], v[22:23], v[24:25]\n") ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25");
        if constexpr (KIND == 7) asm volatile(REP64("v_mfma_f32_16x16x32_bf16 v[10:13], v[30:33], v[34:37], v[10:13]\n v_mfma_f32_16x16x32_bf16 v[14:17], v[30:33], v[34:37], v[14:17]\n v_mfma_f32_16x16x32_bf16 v[18:21], v[30:33], v[34:37], v[18:21]\n v_mfma_f32_16x16x32_bf16 v[22:25], v[30:33], v[34:37], v[22:25]\n") ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v30","v31","v32","v33","v34","v35","v36","v37");
        if constexpr (KIND == 8) asm volatile(REP64("v_mfma_f32_16x16x16_bf16 v[10:13], v[30:31], v[34:35], v[10:13]\n v_mfma_f32_16x16x16_bf16 v[14:17], v[30:31], v[34:35], v[14:17]\n v_mfma_f32_16x16x16_bf16 v[18:21], v[30:31], v[34:35], v[18:21]\n v_mfma_f32_16x16x16_bf16 v[22:25], v[30:31], v[34:35], v[22:25]\n") ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v30","v31","v34","v35");
        if constexpr (KIND == 9) asm volatile(REP64("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 10) asm volatile(REP64("v_cvt_f32_bf16 %0, %1\n v_cvt_f32_bf16 %2, %3\n v_cvt_f32_bf16 %4, %5\n v_cvt_f32_bf16 %6, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 11) asm volatile(REP64("v_cvt_f32_bf16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_bf16_sdwa %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_bf16_sdwa %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_bf16_sdwa %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 12) asm volatile(REP64("v_dot2c_f32_bf16 %0, %1, %2\n v_dot2c_f32_bf16 %3, %4, %5\n v_dot2c_f32_bf16 %6, %7, %1\n v_dot2c_f32_bf16 %2, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 13) asm volatile(REP64("v_dot2_f32_bf16 %0, %1, %2, %3\n v_dot2_f32_bf16 %4, %5, %6, %7\n v_dot2_f32_bf16 %1, %2, %3, %4\n v_dot2_f32_bf16 %5, %6, %7, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 14) asm volatile(REP64("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_cndmask_b32 %6, %7, %1, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
        if constexpr (KIND == 15) asm volatile(REP64("v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %4, %5, %6, %7\n v_perm_b32 %1, %2, %3, %4\n v_perm_b32 %5, %6, %7, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 16) asm volatile(REP64("v_mul_f32 %0, %1, %2\n v_sub_f32 %3, %4, %5\n v_mul_f32 %6, %7, %1\n v_add_f32 %2, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 17) asm volatile(REP64("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 18) asm volatile(REP64("v_add_u32 %0, %1, %2\n v_add_u32 %3, %4, %5\n v_add_u32 %6, %7, %1\n v_add_u32 %2, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 20) asm volatile(REP64("v_lshlrev_b32 %0, 16, %1\n v_lshlrev_b32 %2, 16, %3\n v_lshlrev_b32 %4, 16, %5\n v_lshlrev_b32 %6, 16, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 21) asm volatile(REP64("v_and_b32 %0, 0xffff0000, %1\n v_and_b32 %2, 0xffff0000, %3\n v_and_b32 %4, 0xffff0000, %5\n v_and_b32 %6, 0xffff0000, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if constexpr (KIND == 22) asm volatile("s_mov_b32 s40, 0xffff0000\n" REP64("v_and_b32 %0, s40, %1\n v_and_b32 %2, s40, %3\n v_and_b32 %4, s40, %5\n v_and_b32 %6, s40, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s40");
        if constexpr (KIND == 23) asm volatile("s_mov_b64 s[40:41], 0x5555\n" REP64("v_cndmask_b32_e64 %0, %1, %2, s[40:41]\n v_cndmask_b32_e64 %3, %4, %5, s[40:41]\n v_cndmask_b32_e64 %6, %7, %1, s[40:41]\n v_cndmask_b32_e64 %2, %4, %5, s[40:41]\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s40", "s41");
        if constexpr (KIND == 24) asm volatile(REP64("v_cndmask_b32_e64 %0, 0, %2, s[40:41]\n v_cndmask_b32_e64 %3, 0, %5, s[40:41]\n v_cndmask_b32_e64 %6, 0, %1, s[40:41]\n v_cndmask_b32_e64 %2, 0, %5, s[40:41]\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s40", "s41");
        if constexpr (KIND == 25) asm volatile(REP64("v_mul_f32 %0, s40, %2\n v_mul_f32 %3, s40, %5\n v_mul_f32 %6, s40, %1\n v_mul_f32 %2, s40, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s40", "s41");
        if constexpr (KIND == 26) asm volatile(REP64("v_fma_f32 %0, %1, %2, 1.0\n v_fma_f32 %3, %4, %5, 1.0\n v_max_f32 %6, %7, %1\n v_min_f32 %2, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[0] = 0;
}
template <int KIND> void run(const char* name, long long* d)
{
    for (int nthr : {256, 512, 1024}) {
        k<KIND><<<1, nthr>>>(d, 1.0f);
        long long h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        long long mx = 0; for (int i = 0; i < nthr / 64; ++i) mx = h[i] > mx ? h[i] : mx;
        // 16 iterations x 64 x 4 instructions per wave
        printf("%-22s waves/SIMD %d: %.2f clk64-ticks per instr per wave, SIMD throughput %.2f ticks/instr\n", name, nthr / 256 ? nthr / 256 : 1,
               (double)mx / (16 * 256), (double)mx / (16 * 256) / (nthr >= 256 ? nthr / 256 : 1));
    }
}
int main()
{
    long long* d; hipMalloc(&d, 16 * 16 * 8);
    run<20>("v_lshlrev_b32 16", d); run<21>("v_and_b32 literal", d); run<22>("v_and_b32 sgpr", d); run<23>("v_cndmask_e64 sgpr", d); run<24>("v_cndmask_e64 0,v,sgpr", d); run<25>("v_mul_f32 sgpr", d); run<26>("fma/max/min", d);
    return 0;
}
