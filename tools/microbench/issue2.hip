// Issue cost per instruction class on one gfx950 SIMD at 1 / 2 / 3 / 4 waves per SIMD, alone and beside a wave that issues
// v_mfma_f32_16x16x32_bf16 back to back.  One workgroup on one CU; wave w and w + 4 share a SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue2 tools/microbench/issue2.hip && /tmp/issue2 [substring-of-class-name]
//
// Every class is a body of 8 instructions with 8 different destination registers (v80..v95) whose sources (v64..v79, s40..s43)
// are never written: an INDEPENDENT stream (the "dep" variants chain one register instead).  A wave runs `it` iterations of 16
// bodies (128 instructions + 3 scalar loop instructions) between two s_memtime stamps; all waves start at a barrier.
//   "cyc/instr/wave" = a wave's own cycles per instruction; "SIMD cyc/instr" = the same divided by the waves per SIMD that run the
//   class = the inverse issue throughput of the SIMD for that class.
// Beside MFMAs: W waves per SIMD run the class for a time in which a further wave per SIMD issues MFMAs throughout (it runs
// 6 x longer), giving the class's cost beside MFMAs; and the reverse (the class runs 6 x longer), giving the MFMA's cost beside it.
// Output line: class | config | dispatch index (the order of launches = Dispatch_Id order in a rocprofv3 counter CSV).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>

#define R2(x) x x
#define R4(x) R2(R2(x))
#define R16(x) R4(R4(x))

// ---- the classes ---------------------------------------------------------------------------------------------------
#define B_FMA    "v_fma_f32 v80, v64, v65, v66\n v_fma_f32 v81, v65, v66, v67\n v_fma_f32 v82, v66, v67, v68\n v_fma_f32 v83, v67, v68, v69\n" \
                 "v_fma_f32 v84, v68, v69, v70\n v_fma_f32 v85, v69, v70, v71\n v_fma_f32 v86, v70, v71, v72\n v_fma_f32 v87, v71, v72, v73\n"
#define B_FMA_DEP "v_fma_f32 v80, v80, v65, v66\n v_fma_f32 v80, v80, v66, v67\n v_fma_f32 v80, v80, v67, v68\n v_fma_f32 v80, v80, v68, v69\n" \
                 "v_fma_f32 v80, v80, v69, v70\n v_fma_f32 v80, v80, v70, v71\n v_fma_f32 v80, v80, v71, v72\n v_fma_f32 v80, v80, v72, v73\n"
#define OP2(op)  op " v80, v64, v65\n " op " v81, v65, v66\n " op " v82, v66, v67\n " op " v83, v67, v68\n " \
                 op " v84, v68, v69\n " op " v85, v69, v70\n " op " v86, v70, v71\n " op " v87, v71, v72\n"
#define OP2DEP(op) op " v80, v80, v65\n " op " v80, v80, v66\n " op " v80, v80, v67\n " op " v80, v80, v68\n " \
                 op " v80, v80, v69\n " op " v80, v80, v70\n " op " v80, v80, v71\n " op " v80, v80, v72\n"
#define OP1(op)  op " v80, v64\n " op " v81, v65\n " op " v82, v66\n " op " v83, v67\n " op " v84, v68\n " op " v85, v69\n " op " v86, v70\n " op " v87, v71\n"
#define OP1DEP(op) op " v80, v80\n " op " v80, v80\n " op " v80, v80\n " op " v80, v80\n " op " v80, v80\n " op " v80, v80\n " op " v80, v80\n " op " v80, v80\n"
#define B_MUL_S  "v_mul_f32 v80, s40, v65\n v_mul_f32 v81, s41, v66\n v_mul_f32 v82, s40, v67\n v_mul_f32 v83, s41, v68\n" \
                 "v_mul_f32 v84, s40, v69\n v_mul_f32 v85, s41, v70\n v_mul_f32 v86, s40, v71\n v_mul_f32 v87, s41, v72\n"
#define B_MUL_C  "v_mul_f32 v80, 0x3fb8aa3b, v65\n v_mul_f32 v81, 0x3fb8aa3b, v66\n v_mul_f32 v82, 0x3fb8aa3b, v67\n v_mul_f32 v83, 0x3fb8aa3b, v68\n" \
                 "v_mul_f32 v84, 0x3fb8aa3b, v69\n v_mul_f32 v85, 0x3fb8aa3b, v70\n v_mul_f32 v86, 0x3fb8aa3b, v71\n v_mul_f32 v87, 0x3fb8aa3b, v72\n"
#define B_CND64  "v_cndmask_b32_e64 v80, v64, v65, s[40:41]\n v_cndmask_b32_e64 v81, v65, v66, s[42:43]\n v_cndmask_b32_e64 v82, v66, v67, s[40:41]\n v_cndmask_b32_e64 v83, v67, v68, s[42:43]\n" \
                 "v_cndmask_b32_e64 v84, v68, v69, s[40:41]\n v_cndmask_b32_e64 v85, v69, v70, s[42:43]\n v_cndmask_b32_e64 v86, v70, v71, s[40:41]\n v_cndmask_b32_e64 v87, v71, v72, s[42:43]\n"
#define B_CNDVCC "v_cndmask_b32 v80, v64, v65, vcc\n v_cndmask_b32 v81, v65, v66, vcc\n v_cndmask_b32 v82, v66, v67, vcc\n v_cndmask_b32 v83, v67, v68, vcc\n" \
                 "v_cndmask_b32 v84, v68, v69, vcc\n v_cndmask_b32 v85, v69, v70, vcc\n v_cndmask_b32 v86, v70, v71, vcc\n v_cndmask_b32 v87, v71, v72, vcc\n"
#define B_CMPCND "v_cmp_lt_f32 vcc, v64, v65\n v_cndmask_b32 v80, v64, v65, vcc\n v_cmp_lt_f32 vcc, v66, v67\n v_cndmask_b32 v81, v65, v66, vcc\n" \
                 "v_cmp_lt_f32 vcc, v68, v69\n v_cndmask_b32 v82, v66, v67, vcc\n v_cmp_lt_f32 vcc, v70, v71\n v_cndmask_b32 v83, v67, v68, vcc\n"
#define B_AND    "v_and_b32 v80, 0xffff0000, v64\n v_and_b32 v81, 0xffff0000, v65\n v_and_b32 v82, 0xffff0000, v66\n v_and_b32 v83, 0xffff0000, v67\n" \
                 "v_and_b32 v84, 0xffff0000, v68\n v_and_b32 v85, 0xffff0000, v69\n v_and_b32 v86, 0xffff0000, v70\n v_and_b32 v87, 0xffff0000, v71\n"
#define B_LSHL   "v_lshlrev_b32 v80, 16, v64\n v_lshlrev_b32 v81, 16, v65\n v_lshlrev_b32 v82, 16, v66\n v_lshlrev_b32 v83, 16, v67\n" \
                 "v_lshlrev_b32 v84, 16, v68\n v_lshlrev_b32 v85, 16, v69\n v_lshlrev_b32 v86, 16, v70\n v_lshlrev_b32 v87, 16, v71\n"
#define PK2(op)  op " v[80:81], v[64:65], v[66:67]\n " op " v[82:83], v[66:67], v[68:69]\n " op " v[84:85], v[68:69], v[70:71]\n " op " v[86:87], v[70:71], v[72:73]\n " \
                 op " v[88:89], v[72:73], v[74:75]\n " op " v[90:91], v[74:75], v[76:77]\n " op " v[92:93], v[76:77], v[78:79]\n " op " v[94:95], v[64:65], v[78:79]\n"
#define B_PKFMA  "v_pk_fma_f32 v[80:81], v[64:65], v[66:67], v[68:69]\n v_pk_fma_f32 v[82:83], v[66:67], v[68:69], v[70:71]\n v_pk_fma_f32 v[84:85], v[68:69], v[70:71], v[72:73]\n v_pk_fma_f32 v[86:87], v[70:71], v[72:73], v[74:75]\n" \
                 "v_pk_fma_f32 v[88:89], v[72:73], v[74:75], v[76:77]\n v_pk_fma_f32 v[90:91], v[74:75], v[76:77], v[78:79]\n v_pk_fma_f32 v[92:93], v[76:77], v[78:79], v[64:65]\n v_pk_fma_f32 v[94:95], v[64:65], v[78:79], v[66:67]\n"
#define DPPC " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define B_ADDDPP "v_add_f32_dpp v80, v64, v65" DPPC "v_add_f32_dpp v81, v65, v66" DPPC "v_add_f32_dpp v82, v66, v67" DPPC "v_add_f32_dpp v83, v67, v68" DPPC \
                 "v_add_f32_dpp v84, v68, v69" DPPC "v_add_f32_dpp v85, v69, v70" DPPC "v_add_f32_dpp v86, v70, v71" DPPC "v_add_f32_dpp v87, v71, v72" DPPC
#define B_ADDDPP_DEP "v_add_f32_dpp v80, v80, v80" DPPC "s_nop 1\n v_add_f32_dpp v80, v80, v80" DPPC "s_nop 1\n v_add_f32_dpp v80, v80, v80" DPPC "s_nop 1\n v_add_f32_dpp v80, v80, v80" DPPC "s_nop 1\n"
#define B_MOVDPP "v_mov_b32_dpp v80, v64" DPPC "v_mov_b32_dpp v81, v65" DPPC "v_mov_b32_dpp v82, v66" DPPC "v_mov_b32_dpp v83, v67" DPPC \
                 "v_mov_b32_dpp v84, v68" DPPC "v_mov_b32_dpp v85, v69" DPPC "v_mov_b32_dpp v86, v70" DPPC "v_mov_b32_dpp v87, v71" DPPC
#define B_PL16   "v_permlane16_swap_b32 v80, v81\n v_permlane16_swap_b32 v82, v83\n v_permlane16_swap_b32 v84, v85\n v_permlane16_swap_b32 v86, v87\n" \
                 "v_permlane16_swap_b32 v88, v89\n v_permlane16_swap_b32 v90, v91\n v_permlane16_swap_b32 v92, v93\n v_permlane16_swap_b32 v94, v95\n"
#define B_PL32   "v_permlane32_swap_b32 v80, v81\n v_permlane32_swap_b32 v82, v83\n v_permlane32_swap_b32 v84, v85\n v_permlane32_swap_b32 v86, v87\n" \
                 "v_permlane32_swap_b32 v88, v89\n v_permlane32_swap_b32 v90, v91\n v_permlane32_swap_b32 v92, v93\n v_permlane32_swap_b32 v94, v95\n"
#define B_PERM   "v_perm_b32 v80, v64, v65, v66\n v_perm_b32 v81, v65, v66, v67\n v_perm_b32 v82, v66, v67, v68\n v_perm_b32 v83, v67, v68, v69\n" \
                 "v_perm_b32 v84, v68, v69, v70\n v_perm_b32 v85, v69, v70, v71\n v_perm_b32 v86, v70, v71, v72\n v_perm_b32 v87, v71, v72, v73\n"
#define B_SNOP   "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
#define B_SNOP1  "s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n"
#define B_SALU   "s_add_u32 s44, s44, 1\n s_add_u32 s45, s45, 1\n s_add_u32 s44, s44, 1\n s_add_u32 s45, s45, 1\n s_add_u32 s44, s44, 1\n s_add_u32 s45, s45, 1\n s_add_u32 s44, s44, 1\n s_add_u32 s45, s45, 1\n"
// the split of four floats as the kernels do it (wkv6_chunk.h: split4): 2 cvt_pk, 4 dot2c in place, 2 cvt_pk; x in v80..83 are re-made by v_mov
#define B_SPLIT4 "v_cvt_pk_bf16_f32 v88, v64, v65\n v_cvt_pk_bf16_f32 v89, v66, v67\n v_mov_b32 v80, v64\n v_mov_b32 v81, v65\n v_mov_b32 v82, v66\n v_mov_b32 v83, v67\n" \
                 "v_dot2c_f32_bf16 v80, v88, v78\n v_dot2c_f32_bf16 v81, v88, v79\n v_dot2c_f32_bf16 v82, v89, v78\n v_dot2c_f32_bf16 v83, v89, v79\n" \
                 "v_cvt_pk_bf16_f32 v90, v80, v81\n v_cvt_pk_bf16_f32 v91, v82, v83\n"       /* 12 instructions */
#define B_MFMA32 "v_mfma_f32_16x16x32_bf16 v[104:107], v[96:99], v[100:103], v[104:107]\n v_mfma_f32_16x16x32_bf16 v[108:111], v[96:99], v[100:103], v[108:111]\n" \
                 "v_mfma_f32_16x16x32_bf16 v[112:115], v[96:99], v[100:103], v[112:115]\n v_mfma_f32_16x16x32_bf16 v[116:119], v[96:99], v[100:103], v[116:119]\n" \
                 "v_mfma_f32_16x16x32_bf16 v[104:107], v[96:99], v[100:103], v[104:107]\n v_mfma_f32_16x16x32_bf16 v[108:111], v[96:99], v[100:103], v[108:111]\n" \
                 "v_mfma_f32_16x16x32_bf16 v[112:115], v[96:99], v[100:103], v[112:115]\n v_mfma_f32_16x16x32_bf16 v[116:119], v[96:99], v[100:103], v[116:119]\n"
#define B_MFMA16 "v_mfma_f32_16x16x16_bf16 v[104:107], v[96:97], v[100:101], v[104:107]\n v_mfma_f32_16x16x16_bf16 v[108:111], v[96:97], v[100:101], v[108:111]\n" \
                 "v_mfma_f32_16x16x16_bf16 v[112:115], v[96:97], v[100:101], v[112:115]\n v_mfma_f32_16x16x16_bf16 v[116:119], v[96:97], v[100:101], v[116:119]\n" \
                 "v_mfma_f32_16x16x16_bf16 v[104:107], v[96:97], v[100:101], v[104:107]\n v_mfma_f32_16x16x16_bf16 v[108:111], v[96:97], v[100:101], v[108:111]\n" \
                 "v_mfma_f32_16x16x16_bf16 v[112:115], v[96:97], v[100:101], v[112:115]\n v_mfma_f32_16x16x16_bf16 v[116:119], v[96:97], v[100:101], v[116:119]\n"
#define B_MFMA32_DEP "v_mfma_f32_16x16x32_bf16 v[104:107], v[96:99], v[100:103], v[104:107]\n v_mfma_f32_16x16x32_bf16 v[104:107], v[96:99], v[100:103], v[104:107]\n" \
                 "v_mfma_f32_16x16x32_bf16 v[104:107], v[96:99], v[100:103], v[104:107]\n v_mfma_f32_16x16x32_bf16 v[104:107], v[96:99], v[100:103], v[104:107]\n" \
                 "v_mfma_f32_16x16x32_bf16 v[104:107], v[96:99], v[100:103], v[104:107]\n v_mfma_f32_16x16x32_bf16 v[104:107], v[96:99], v[100:103], v[104:107]\n" \
                 "v_mfma_f32_16x16x32_bf16 v[104:107], v[96:99], v[100:103], v[104:107]\n v_mfma_f32_16x16x32_bf16 v[104:107], v[96:99], v[100:103], v[104:107]\n"
// LDS: conflict-free per-lane addresses in v120 (16 B per lane), one counted wait per body
#define B_DSR128 "ds_read_b128 v[80:83], v120\n ds_read_b128 v[84:87], v120 offset:1024\n ds_read_b128 v[88:91], v120 offset:2048\n ds_read_b128 v[92:95], v120 offset:3072\n" \
                 "ds_read_b128 v[80:83], v120 offset:4096\n ds_read_b128 v[84:87], v120 offset:5120\n ds_read_b128 v[88:91], v120 offset:6144\n ds_read_b128 v[92:95], v120 offset:7168\n s_waitcnt lgkmcnt(4)\n"
#define B_DSRTR  "ds_read_b64_tr_b16 v[80:81], v121\n ds_read_b64_tr_b16 v[82:83], v121 offset:512\n ds_read_b64_tr_b16 v[84:85], v121 offset:1024\n ds_read_b64_tr_b16 v[86:87], v121 offset:1536\n" \
                 "ds_read_b64_tr_b16 v[88:89], v121 offset:2048\n ds_read_b64_tr_b16 v[90:91], v121 offset:2560\n ds_read_b64_tr_b16 v[92:93], v121 offset:3072\n ds_read_b64_tr_b16 v[94:95], v121 offset:3584\n s_waitcnt lgkmcnt(4)\n"
#define B_DSW64  "ds_write_b64 v121, v[64:65]\n ds_write_b64 v121, v[66:67] offset:512\n ds_write_b64 v121, v[68:69] offset:1024\n ds_write_b64 v121, v[70:71] offset:1536\n" \
                 "ds_write_b64 v121, v[72:73] offset:2048\n ds_write_b64 v121, v[74:75] offset:2560\n ds_write_b64 v121, v[76:77] offset:3072\n ds_write_b64 v121, v[78:79] offset:3584\n s_waitcnt lgkmcnt(4)\n"

// integer multiplies (the general token map's offset arithmetic, gone from the benched kernels in round 5)
#define B_MULLO  "v_mul_lo_u32 v80, v64, v65\n v_mul_lo_u32 v81, v65, v66\n v_mul_lo_u32 v82, v66, v67\n v_mul_lo_u32 v83, v67, v68\n" \
                 "v_mul_lo_u32 v84, v68, v69\n v_mul_lo_u32 v85, v69, v70\n v_mul_lo_u32 v86, v70, v71\n v_mul_lo_u32 v87, v71, v72\n"
#define B_MAD64  "v_mad_u64_u32 v[80:81], vcc, v64, v65, v[66:67]\n v_mad_u64_u32 v[82:83], vcc, v65, v66, v[68:69]\n v_mad_u64_u32 v[84:85], vcc, v66, v67, v[70:71]\n v_mad_u64_u32 v[86:87], vcc, v67, v68, v[72:73]\n" \
                 "v_mad_u64_u32 v[88:89], vcc, v68, v69, v[74:75]\n v_mad_u64_u32 v[90:91], vcc, v69, v70, v[76:77]\n v_mad_u64_u32 v[92:93], vcc, v70, v71, v[64:65]\n v_mad_u64_u32 v[94:95], vcc, v71, v72, v[66:67]\n"
// own-wave interleaving: one MFMA followed by six vector instructions (all independent)
#define M1 "v_mfma_f32_16x16x32_bf16 v[104:107], v[96:99], v[100:103], v[104:107]\n"
#define B_M_PKFMA M1 "v_pk_fma_f32 v[80:81], v[64:65], v[66:67], v[68:69]\n v_pk_fma_f32 v[82:83], v[66:67], v[68:69], v[70:71]\n v_pk_fma_f32 v[84:85], v[68:69], v[70:71], v[72:73]\n" \
                  "v_pk_fma_f32 v[86:87], v[70:71], v[72:73], v[74:75]\n v_pk_fma_f32 v[88:89], v[72:73], v[74:75], v[76:77]\n v_pk_fma_f32 v[90:91], v[74:75], v[76:77], v[78:79]\n"
#define B_M_FMA6  M1 "v_fma_f32 v80, v64, v65, v66\n v_fma_f32 v81, v65, v66, v67\n v_fma_f32 v82, v66, v67, v68\n v_fma_f32 v83, v67, v68, v69\n v_fma_f32 v84, v68, v69, v70\n v_fma_f32 v85, v69, v70, v71\n"
#define B_M_FMA12 B_M_FMA6 "v_fma_f32 v86, v64, v65, v66\n v_fma_f32 v87, v65, v66, v67\n v_fma_f32 v88, v66, v67, v68\n v_fma_f32 v89, v67, v68, v69\n v_fma_f32 v90, v68, v69, v70\n v_fma_f32 v91, v69, v70, v71\n"
#define B_M_DOT6  M1 "v_dot2c_f32_bf16 v80, v64, v65\n v_dot2c_f32_bf16 v81, v65, v66\n v_dot2c_f32_bf16 v82, v66, v67\n v_dot2c_f32_bf16 v83, v67, v68\n v_dot2c_f32_bf16 v84, v68, v69\n v_dot2c_f32_bf16 v85, v69, v70\n"
#define B_M_CVT6  M1 "v_cvt_pk_bf16_f32 v80, v64, v65\n v_cvt_pk_bf16_f32 v81, v65, v66\n v_cvt_pk_bf16_f32 v82, v66, v67\n v_cvt_pk_bf16_f32 v83, v67, v68\n v_cvt_pk_bf16_f32 v84, v68, v69\n v_cvt_pk_bf16_f32 v85, v69, v70\n"
#define CLOB "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
             "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
             "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111", \
             "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","s40","s41","s42","s43","s44","s45","vcc","memory"

#define KINDS(X) \
    X(0, "v_fma_f32", B_FMA, 8) X(1, "v_fma_f32 dep", B_FMA_DEP, 8) X(2, "v_mul_f32", OP2("v_mul_f32"), 8) X(3, "v_mul_f32 sgpr", B_MUL_S, 8) \
    X(4, "v_mul_f32 literal", B_MUL_C, 8) X(5, "v_add_f32", OP2("v_add_f32"), 8) X(6, "v_sub_f32 dep", OP2DEP("v_sub_f32"), 8) X(7, "v_max_f32", OP2("v_max_f32"), 8) \
    X(8, "v_mov_b32", OP1("v_mov_b32"), 8) X(9, "v_cndmask_b32_e64 sgpr", B_CND64, 8) X(10, "v_cndmask_b32 vcc", B_CNDVCC, 8) X(11, "v_cmp+v_cndmask vcc", B_CMPCND, 8) \
    X(12, "v_cvt_pk_bf16_f32", OP2("v_cvt_pk_bf16_f32"), 8) X(13, "v_and_b32 literal", B_AND, 8) X(14, "v_lshlrev_b32", B_LSHL, 8) X(15, "v_add_u32", OP2("v_add_u32"), 8) \
    X(16, "v_pk_mul_f32", PK2("v_pk_mul_f32"), 8) X(17, "v_pk_add_f32", PK2("v_pk_add_f32"), 8) X(18, "v_pk_fma_f32", B_PKFMA, 8) X(19, "v_exp_f32", OP1("v_exp_f32"), 8) \
    X(20, "v_exp_f32 dep", OP1DEP("v_exp_f32"), 8) X(21, "v_rcp_f32", OP1("v_rcp_f32"), 8) X(22, "v_add_f32_dpp", B_ADDDPP, 8) X(23, "v_add_f32_dpp dep+s_nop1", B_ADDDPP_DEP, 4) \
    X(24, "v_mov_b32_dpp", B_MOVDPP, 8) X(25, "v_permlane16_swap", B_PL16, 8) X(26, "v_permlane32_swap", B_PL32, 8) X(27, "v_dot2c_f32_bf16", OP2("v_dot2c_f32_bf16"), 8) \
    X(28, "v_perm_b32", B_PERM, 8) X(29, "v_ldexp_f32", OP2("v_ldexp_f32"), 8) X(30, "s_nop 0", B_SNOP, 8) X(31, "s_nop 1", B_SNOP1, 8) X(32, "s_add_u32", B_SALU, 8) \
    X(33, "split4 sequence (12 instr)", B_SPLIT4, 12) X(34, "mfma 16x16x32 bf16", B_MFMA32, 8) X(35, "mfma 16x16x16 bf16", B_MFMA16, 8) X(36, "mfma 16x16x32 dep", B_MFMA32_DEP, 8) \
    X(37, "ds_read_b128", B_DSR128, 8) X(38, "ds_read_b64_tr_b16", B_DSRTR, 8) X(39, "ds_write_b64", B_DSW64, 8) X(40, "v_fmac_f32", OP2("v_fmac_f32"), 8) \
    X(41, "own wave: mfma + 6 v_pk_fma_f32", B_M_PKFMA, 7) X(42, "own wave: mfma + 6 v_fma_f32", B_M_FMA6, 7) X(43, "own wave: mfma + 12 v_fma_f32", B_M_FMA12, 13) \
    X(44, "own wave: mfma + 6 v_dot2c", B_M_DOT6, 7) X(45, "own wave: mfma + 6 v_cvt_pk", B_M_CVT6, 7) \
    X(46, "v_mul_lo_u32", B_MULLO, 8) X(47, "v_mad_u64_u32", B_MAD64, 8)
constexpr int NKIND = 46;

// role 0 = the class, role 1 = back-to-back MFMAs (16x16x32, four accumulators)
template <int KIND> __global__ __launch_bounds__(1024) void k(long long* out, int wclass, int it_class, int it_mfma, int mfmode)
{
    __shared__ float lds[4 * 2048 + 64 * 4 * 16];
    lds[threadIdx.x] = 1.0f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mf = (wave >> 2) >= wclass;
    const int it = mf ? it_mfma : it_class;
    const unsigned a128 = (unsigned)(size_t)(void*)lds & 0xffff, dummy = 0;
    (void)dummy;
    asm volatile(
        "v_mov_b32 v64, 1.0\n v_mov_b32 v65, 0.5\n v_mov_b32 v66, 2.0\n v_mov_b32 v67, 0.5\n v_mov_b32 v68, 1.0\n v_mov_b32 v69, 0.5\n v_mov_b32 v70, 2.0\n v_mov_b32 v71, 0.5\n"
        "v_mov_b32 v72, 1.0\n v_mov_b32 v73, 0.5\n v_mov_b32 v74, 2.0\n v_mov_b32 v75, 0.5\n v_mov_b32 v76, 1.0\n v_mov_b32 v77, 0.5\n v_mov_b32 v78, 0x0000bf80\n v_mov_b32 v79, 0xbf800000\n"
        "v_mov_b32 v80, 1.0\n v_mov_b32 v81, 1.0\n v_mov_b32 v82, 1.0\n v_mov_b32 v83, 1.0\n v_mov_b32 v84, 1.0\n v_mov_b32 v85, 1.0\n v_mov_b32 v86, 1.0\n v_mov_b32 v87, 1.0\n"
        "v_mov_b32 v88, 1.0\n v_mov_b32 v89, 1.0\n v_mov_b32 v90, 1.0\n v_mov_b32 v91, 1.0\n v_mov_b32 v92, 1.0\n v_mov_b32 v93, 1.0\n v_mov_b32 v94, 1.0\n v_mov_b32 v95, 1.0\n"
        "v_mov_b32 v96, 0x3f803f80\n v_mov_b32 v97, 0x3f803f80\n v_mov_b32 v98, 0x3f803f80\n v_mov_b32 v99, 0x3f803f80\n v_mov_b32 v100, 0x3c003c00\n v_mov_b32 v101, 0x3c003c00\n v_mov_b32 v102, 0x3c003c00\n v_mov_b32 v103, 0x3c003c00\n"
        "v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n"
        "v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n"
        "v_lshlrev_b32 v120, 4, %0\n v_add_u32 v120, v120, %1\n v_lshlrev_b32 v121, 3, %0\n v_add_u32 v121, v121, %1\n"
        "s_mov_b32 s40, 0x3f000000\n s_mov_b32 s41, 0x3f800000\n s_mov_b32 s42, 0x55555555\n s_mov_b32 s43, 0x33333333\n s_mov_b32 s44, 0\n s_mov_b32 s45, 0\n s_mov_b64 vcc, 0x5555\n"
        :: "v"(lane), "v"(a128 + 64 * 4 * 16 + (wave & 3) * 8192) : CLOB);
    __syncthreads();
    long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (!mf) {
        for (int i = 0; i < it; ++i) {
#define X(id, name, body, n) if constexpr (KIND == id) asm volatile(R16(body) ::: CLOB);
            KINDS(X)
#undef X
        }
    } else {
        // mfmode 0: back to back (matrix pipe saturated); 1: one MFMA per ~64 cycles (25 % duty: 16 + s_nop 7 + s_nop 3 = 16 + 32 + 16);
        // 2: one per ~32 cycles (50 %)
        if (mfmode == 0) for (int i = 0; i < it; ++i) asm volatile(R16(B_MFMA32) ::: CLOB);
        else if (mfmode == 1) for (int i = 0; i < it; ++i) asm volatile(R16(R4(M1 "s_nop 7\n s_nop 3\n")) ::: CLOB);
        else for (int i = 0; i < it; ++i) asm volatile(R16(R4(M1 "s_nop 3\n")) ::: CLOB);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) out[wave] = t1 - t0;
    __syncthreads();
}

struct Kind { int id; const char* name; int n; void (*fn)(long long*, int, int, int, int); };
static const Kind kinds[] = {
#define X(id, name, body, n) {id, name, n, k<id>},
    KINDS(X)
#undef X
};

int main(int argc, char** argv)
{
    const char* filt = (argc > 1 && strcmp(argv[1], "all")) ? argv[1] : "";
    const bool quick = argc > 2 && !strcmp(argv[2], "quick");     // profiler passes: fewer configurations
    long long* d; hipMalloc(&d, 16 * 8);
    long long h[16];
    int disp = 0;
    const int IT = 24;
    printf("%-28s %-34s %8s %14s %14s %10s\n", "class", "config", "dispatch", "cyc/instr/wave", "SIMD cyc/instr", "mfma cyc");
    for (const Kind& kd : kinds) {
        if (!strstr(kd.name, filt)) continue;
        const bool is_mfma = (kd.id >= 34 && kd.id <= 36) || kd.id >= 41;
        // alone, W waves per SIMD
        for (int W = 1; W <= 4; ++W) {
            if (quick && W == 3) continue;
            hipLaunchKernelGGL(kd.fn, dim3(1), dim3(256 * W), 0, 0, d, W, IT, 0, 0);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            long long mx = 0; for (int i = 0; i < 4 * W; ++i) mx = h[i] > mx ? h[i] : mx;
            const double per = (double)mx / (IT * 16.0 * kd.n);
            char cfg[64]; snprintf(cfg, sizeof cfg, "alone, %d wave(s)/SIMD", W);
            printf("%-28s %-34s %8d %14.2f %14.2f %10s\n", kd.name, cfg, disp++, per, per / W, "");
        }
        if (is_mfma || quick) continue;
        // beside one MFMA wave per SIMD: (a) class short, MFMA long -> class cost; (b) class long, MFMA short -> MFMA cost
        for (int W = 1; W <= 3; ++W) {
            hipLaunchKernelGGL(kd.fn, dim3(1), dim3(256 * (W + 1)), 0, 0, d, W, IT, IT * 6 * W, 0);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            long long mx = 0; for (int i = 0; i < 4 * W; ++i) mx = h[i] > mx ? h[i] : mx;
            long long mm = 0; for (int i = 4 * W; i < 4 * W + 4; ++i) mm = h[i] > mm ? h[i] : mm;
            const double per = (double)mx / (IT * 16.0 * kd.n);
            const bool covered = mm > mx;                              // the MFMA wave outlasted the class waves
            hipLaunchKernelGGL(kd.fn, dim3(1), dim3(256 * (W + 1)), 0, 0, d, W, IT * 12, IT, 0);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            long long m2 = 0; for (int i = 4 * W; i < 4 * W + 4; ++i) m2 = h[i] > m2 ? h[i] : m2;
            long long c2 = 0; for (int i = 0; i < 4 * W; ++i) c2 = h[i] > c2 ? h[i] : c2;
            char cfg[64]; snprintf(cfg, sizeof cfg, "%d wave(s) + 1 MFMA wave /SIMD%s%s", W, covered ? "" : " (!short)", c2 > m2 ? "" : " (!short2)");
            char mc[32]; snprintf(mc, sizeof mc, "%.2f", (double)m2 / (IT * 16.0 * 8));
            printf("%-28s %-34s %8d %14.2f %14.2f %10s\n", kd.name, cfg, disp, per, per / W, mc);
            disp += 2;
        }
        // beside a wave that issues one MFMA per ~64 / ~32 cycles (25 % / 50 % matrix-pipe duty, what the product kernels run at)
        for (int mode = 1; mode <= 2; ++mode)
            for (int W = 1; W <= 2; ++W) {
                // the MFMA wave: 64 MFMAs per iteration at ~64 (32) cycles each; keep it running ~3x as long as the class waves
                const int itm = (int)(IT * 128.0 * 6.0 * W * 3.0 / (64.0 * (mode == 1 ? 64 : 32))) + 2;
                hipLaunchKernelGGL(kd.fn, dim3(1), dim3(256 * (W + 1)), 0, 0, d, W, IT, itm, mode);
                hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
                long long mx = 0; for (int i = 0; i < 4 * W; ++i) mx = h[i] > mx ? h[i] : mx;
                long long mm = 0; for (int i = 4 * W; i < 4 * W + 4; ++i) mm = h[i] > mm ? h[i] : mm;
                const double per = (double)mx / (IT * 16.0 * kd.n);
                char cfg[64]; snprintf(cfg, sizeof cfg, "%d wave(s) + MFMA wave at %d %% duty%s", W, mode == 1 ? 25 : 50, mm > mx ? "" : " (!short)");
                char mc[32]; snprintf(mc, sizeof mc, "%.2f", (double)mm / (itm * 64.0));
                printf("%-28s %-34s %8d %14.2f %14.2f %10s\n", kd.name, cfg, disp++, per, per / W, mc);
            }
    }
    return 0;
}
