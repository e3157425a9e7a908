// Issue-rate probe for gfx950: cycles per wave64 instruction and SIMD for the instruction classes the byte kernels use or could
// move to, with CONTROL rows (v_fma_f32, v_pk_fma_f32, v_add_u32, v_and_b32) and a sweep over 1 / 2 / 4 / 8 resident waves per SIMD.
// Two clocks: the host's (HIP events, priced at 2.4 GHz) and the wave's own (s_memtime ticks = shader cycles, median over waves), so a
// chip that holds a lower clock under load does not read as a slower instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > profiles/r02_valu_issue_rates.txt
// Four independent dependency chains per lane; residency is pinned by dynamic LDS (one 256-thread workgroup = one wave per SIMD, n
// workgroups per CU fit when each asks for 160 KiB / n minus slack) and the grid is exactly 256 CUs x n workgroups.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#define CHAIN4(INS) \
    asm volatile(INS " %0, %0, %4\n\t" INS " %1, %1, %4\n\t" INS " %2, %2, %4\n\t" INS " %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
#define CHAIN4_3(INS) \
    asm volatile(INS " %0, %0, %4, %5\n\t" INS " %1, %1, %4, %5\n\t" INS " %2, %2, %4, %5\n\t" INS " %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
#define CHAIN4_1(INS) \
    asm volatile(INS " %0, %0\n\t" INS " %1, %1\n\t" INS " %2, %2\n\t" INS " %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
// 64-bit register pairs (packed f32)
#define CHAIN4_P3(INS) \
    asm volatile(INS " %0, %0, %4, %5\n\t" INS " %1, %1, %4, %5\n\t" INS " %2, %2, %4, %5\n\t" INS " %3, %3, %4, %5" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(E), "v"(F));
#define CHAIN4_P2(INS) \
    asm volatile(INS " %0, %0, %4\n\t" INS " %1, %1, %4\n\t" INS " %2, %2, %4\n\t" INS " %3, %3, %4" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(E));
// LDS: four reads in flight per step, drained every 16 (the address registers are the chain; data registers are sinks)
#define LDS4(INS) \
    asm volatile(INS " %0, %4\n\t" INS " %1, %4 offset:256\n\t" INS " %2, %4 offset:512\n\t" INS " %3, %4 offset:768" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(e) : "memory");
#define LDS4_64(INS) \
    asm volatile(INS " %0, %4\n\t" INS " %1, %4 offset:512\n\t" INS " %2, %4 offset:1024\n\t" INS " %3, %4 offset:1536" : "=v"(A), "=v"(B), "=v"(C), "=v"(D) : "v"(e) : "memory");

enum Op {
    FMA_F32, PK_FMA_F32, ADD_F32, PK_ADD_F32, MIN3_F32, MAX3_F32, MINIMUM3_F32, SUB_F32, CVT_F32_UBYTE0, CVT_F32_UBYTE3, CVT_U32_F32,
    ADD_U32, AND_B32, XOR_B32, LSHL_OR_B32, ADD3_U32, MIN_I32, MIN3_I32, MAX3_I32, MAD_I32_I24, MAD_U32_U24, MUL_LO_U32,
    PK_MIN_I16, PK_MAX_I16, PK_SUB_I16, PK_ADD_U16, MIN3_I16, PK_MINIMUM3_F16, PK_MAXIMUM3_F16, PK_MIN_F16, PK_ADD_F16,
    DOT4_U32_U8, BCNT, PERM, ALIGNBYTE, SAD_U8, CNDMASK, CMP_GT_I32, MBCNT,
    OR_B32, SUB_U32, LSHLREV, LSHRREV, ASHRREV, MOV_B32, NOT_B32, MUL_F32, MAX_F32, MIN_F32, MAX_U32, MIN_U32, MAX_I32, MIN_U16, SUB_U16, MUL_U32_U24, FMAC_F32,
    BFE_U32, AND_OR_B32, OR3_B32, LSHL_ADD_U32, XAD_U32, BFI_B32, MED3_I32, MAX3_U32, MUL_HI_U32, CVT_F32_I32, CNDMASK_S, CMP_GT_U32_S, CMP_GT_F32,
    SDWA_SUB_U32_BYTES, SDWA_ADD_U32_W1, SDWA_AND_BYTE, SDWA_MIN_U16_BYTES, SDWA_MAX_I32_WORDS, SDWA_SUB_F32, SDWA_CVT_UBYTE, SDWA_MOV_W1, SDWA_CMP_BYTES, DPP_MOV_SHR1, DPP_ADD_SHR1, DPP_MOV_BCAST, READLANE,
    FMA_F64, MUL_F64, ADD_F64, FMA_F64_DEP, RSQ_F64, RCP_F64,
    DS_READ_U8, DS_READ_U8_D16_HI, DS_READ_B32, DS_READ_B64, DS_READ_B32_MIS1, DS_READ_B64_MIS1, DS_READ_B64_MIS4, DS_READ_B32_STRIDE5, DS_READ_B64_STRIDE7, NOPS
};

template <int OP> __global__ __launch_bounds__(256) void k(unsigned *out, unsigned long long *ticks, unsigned seed, int iters) {
    extern __shared__ unsigned lds[];
    unsigned a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 7, d = seed * 11, e = seed * 13 + threadIdx.x, f = seed * 17;
    unsigned long long A = a | (unsigned long long)b << 32, B = c | (unsigned long long)d << 32, C = e, D = f, E = 0x3f8000003f800000ull, F = 0x3a8000003a800000ull;
    if (OP >= DS_READ_U8 && OP < NOPS) {
        for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = i * seed;
        e = (threadIdx.x & 63) * ((OP == DS_READ_B64 || OP == DS_READ_B64_MIS1 || OP == DS_READ_B64_MIS4) ? 8 : 4);   // conflict-free: consecutive (d)words
        if (OP == DS_READ_B32_MIS1 || OP == DS_READ_B64_MIS1) e += 1;      // every lane's access straddles a dword boundary
        if (OP == DS_READ_B64_MIS4) e += 4;                                 // 8-byte reads on odd dword boundaries
        if (OP == DS_READ_B32_STRIDE5) e = (threadIdx.x & 63) * 5;          // byte-granular gather: 4-byte reads at 5-byte steps
        if (OP == DS_READ_B64_STRIDE7) e = (threadIdx.x & 63) * 7;          // 8-byte reads at 7-byte steps
        __syncthreads();
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == FMA_F32) CHAIN4_3("v_fma_f32")
            if (OP == PK_FMA_F32) CHAIN4_P3("v_pk_fma_f32")
            if (OP == ADD_F32) CHAIN4("v_add_f32")
            if (OP == PK_ADD_F32) CHAIN4_P2("v_pk_add_f32")
            if (OP == MIN3_F32) CHAIN4_3("v_min3_f32")
            if (OP == MAX3_F32) CHAIN4_3("v_max3_f32")
            if (OP == MINIMUM3_F32) CHAIN4_3("v_minimum3_f32")
            if (OP == SUB_F32) CHAIN4("v_sub_f32")
            if (OP == CVT_F32_UBYTE0) CHAIN4_1("v_cvt_f32_ubyte0")
            if (OP == CVT_F32_UBYTE3) CHAIN4_1("v_cvt_f32_ubyte3")
            if (OP == CVT_U32_F32) CHAIN4_1("v_cvt_u32_f32")
            if (OP == ADD_U32) CHAIN4("v_add_u32")
            if (OP == AND_B32) CHAIN4("v_and_b32")
            if (OP == XOR_B32) CHAIN4("v_xor_b32")
            if (OP == LSHL_OR_B32) CHAIN4_3("v_lshl_or_b32")
            if (OP == ADD3_U32) CHAIN4_3("v_add3_u32")
            if (OP == MIN_I32) CHAIN4("v_min_i32")
            if (OP == MIN3_I32) CHAIN4_3("v_min3_i32")
            if (OP == MAX3_I32) CHAIN4_3("v_max3_i32")
            if (OP == MAD_I32_I24) CHAIN4_3("v_mad_i32_i24")
            if (OP == MAD_U32_U24) CHAIN4_3("v_mad_u32_u24")
            if (OP == MUL_LO_U32) CHAIN4("v_mul_lo_u32")
            if (OP == PK_MIN_I16) CHAIN4("v_pk_min_i16")
            if (OP == PK_MAX_I16) CHAIN4("v_pk_max_i16")
            if (OP == PK_SUB_I16) CHAIN4("v_pk_sub_i16")
            if (OP == PK_ADD_U16) CHAIN4("v_pk_add_u16")
            if (OP == MIN3_I16) CHAIN4_3("v_min3_i16")
            if (OP == PK_MINIMUM3_F16) CHAIN4_3("v_pk_minimum3_f16")
            if (OP == PK_MAXIMUM3_F16) CHAIN4_3("v_pk_maximum3_f16")
            if (OP == PK_MIN_F16) CHAIN4("v_pk_min_f16")
            if (OP == PK_ADD_F16) CHAIN4("v_pk_add_f16")
            if (OP == DOT4_U32_U8) CHAIN4_3("v_dot4_u32_u8")
            if (OP == BCNT) CHAIN4("v_bcnt_u32_b32")
            if (OP == PERM) CHAIN4_3("v_perm_b32")
            if (OP == ALIGNBYTE) CHAIN4_3("v_alignbyte_b32")
            if (OP == SAD_U8) CHAIN4_3("v_sad_u8")
            if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");
            if (OP == CMP_GT_I32) asm volatile("v_cmp_gt_i32 vcc, %0, %4\n\tv_cmp_gt_i32 vcc, %1, %4\n\tv_cmp_gt_i32 vcc, %2, %4\n\tv_cmp_gt_i32 vcc, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");
            if (OP == MBCNT) CHAIN4("v_mbcnt_lo_u32_b32")

            if (OP == OR_B32) CHAIN4("v_or_b32")
            if (OP == SUB_U32) CHAIN4("v_sub_u32")
            if (OP == LSHLREV) asm volatile("v_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_lshlrev_b32 %2, 1, %2\n\tv_lshlrev_b32 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == LSHRREV) asm volatile("v_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == ASHRREV) asm volatile("v_ashrrev_i32 %0, 1, %0\n\tv_ashrrev_i32 %1, 1, %1\n\tv_ashrrev_i32 %2, 1, %2\n\tv_ashrrev_i32 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == NOT_B32) CHAIN4_1("v_not_b32")
            if (OP == MUL_F32) CHAIN4("v_mul_f32")
            if (OP == MAX_F32) CHAIN4("v_max_f32")
            if (OP == MIN_F32) CHAIN4("v_min_f32")
            if (OP == MAX_U32) CHAIN4("v_max_u32")
            if (OP == MIN_U32) CHAIN4("v_min_u32")
            if (OP == MAX_I32) CHAIN4("v_max_i32")
            if (OP == MIN_U16) CHAIN4("v_min_u16")
            if (OP == SUB_U16) CHAIN4("v_sub_u16")
            if (OP == MUL_U32_U24) CHAIN4("v_mul_u32_u24")
            if (OP == FMAC_F32) CHAIN4("v_fmac_f32")
            if (OP == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 1, 31\n\tv_bfe_u32 %1, %1, 1, 31\n\tv_bfe_u32 %2, %2, 1, 31\n\tv_bfe_u32 %3, %3, 1, 31" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == AND_OR_B32) CHAIN4_3("v_and_or_b32")
            if (OP == OR3_B32) CHAIN4_3("v_or3_b32")
            if (OP == LSHL_ADD_U32) asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n\tv_lshl_add_u32 %1, %1, 1, %4\n\tv_lshl_add_u32 %2, %2, 1, %4\n\tv_lshl_add_u32 %3, %3, 1, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
            if (OP == XAD_U32) CHAIN4_3("v_xad_u32")
            if (OP == BFI_B32) CHAIN4_3("v_bfi_b32")
            if (OP == MED3_I32) CHAIN4_3("v_med3_i32")
            if (OP == MAX3_U32) CHAIN4_3("v_max3_u32")
            if (OP == MUL_HI_U32) CHAIN4("v_mul_hi_u32")
            if (OP == CVT_F32_I32) CHAIN4_1("v_cvt_f32_i32")
            if (OP == CNDMASK_S) asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n\tv_cndmask_b32_e64 %1, %1, %4, %5\n\tv_cndmask_b32_e64 %2, %2, %4, %5\n\tv_cndmask_b32_e64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "s"(0x5555aaaa5555aaaaull));
            if (OP == CMP_GT_U32_S) asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %4\n\tv_cmp_gt_u32_e64 s[22:23], %1, %4\n\tv_cmp_gt_u32_e64 s[24:25], %2, %4\n\tv_cmp_gt_u32_e64 s[26:27], %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            if (OP == CMP_GT_F32) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %4\n\tv_cmp_gt_f32_e64 s[22:23], %1, %4\n\tv_cmp_gt_f32_e64 s[24:25], %2, %4\n\tv_cmp_gt_f32_e64 s[26:27], %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
#define SDWA4(INS, MODS) asm volatile(INS " %0, %0, %4 " MODS "\n\t" INS " %1, %1, %4 " MODS "\n\t" INS " %2, %2, %4 " MODS "\n\t" INS " %3, %3, %4 " MODS : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
            if (OP == SDWA_SUB_U32_BYTES) SDWA4("v_sub_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2")
            if (OP == SDWA_ADD_U32_W1) SDWA4("v_add_u32_sdwa", "dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_2")
            if (OP == SDWA_AND_BYTE) SDWA4("v_and_b32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
            if (OP == SDWA_MIN_U16_BYTES) SDWA4("v_min_u16_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")
            if (OP == SDWA_MAX_I32_WORDS) SDWA4("v_max_i32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")
            if (OP == SDWA_SUB_F32) SDWA4("v_sub_f32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")
            if (OP == SDWA_CVT_UBYTE) asm volatile("v_cvt_f32_ubyte0_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n\tv_cvt_f32_ubyte0_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n\tv_cvt_f32_ubyte0_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n\tv_cvt_f32_ubyte0_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == SDWA_MOV_W1) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n\tv_mov_b32_sdwa %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n\tv_mov_b32_sdwa %2, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n\tv_mov_b32_sdwa %3, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == SDWA_CMP_BYTES) asm volatile("v_cmp_gt_u32_sdwa s[20:21], %0, %4 src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_cmp_gt_u32_sdwa s[22:23], %1, %4 src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_cmp_gt_u32_sdwa s[24:25], %2, %4 src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_cmp_gt_u32_sdwa s[26:27], %3, %4 src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            if (OP == DPP_MOV_SHR1) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == DPP_ADD_SHR1) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %3, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == DPP_MOV_BCAST) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == READLANE) asm volatile("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 5\n\tv_readlane_b32 s22, %2, 7\n\tv_readlane_b32 s23, %3, 9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "s20", "s21", "s22", "s23");
            if (OP == FMA_F64) CHAIN4_P3("v_fma_f64")
            if (OP == MUL_F64) CHAIN4_P2("v_mul_f64")
            if (OP == ADD_F64) CHAIN4_P2("v_add_f64")
            if (OP == FMA_F64_DEP) asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(A) : "v"(E), "v"(F));
            if (OP == RSQ_F64) asm volatile("v_rsq_f64 %0, %0\n\tv_rsq_f64 %1, %1\n\tv_rsq_f64 %2, %2\n\tv_rsq_f64 %3, %3" : "+v"(A), "+v"(B), "+v"(C), "+v"(D));
            if (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3" : "+v"(A), "+v"(B), "+v"(C), "+v"(D));
            if (OP == DS_READ_U8) LDS4("ds_read_u8")
            if (OP == DS_READ_U8_D16_HI) LDS4("ds_read_u8_d16_hi")
            if (OP == DS_READ_B32) LDS4("ds_read_b32")
            if (OP == DS_READ_B64 || OP == DS_READ_B64_MIS1 || OP == DS_READ_B64_MIS4 || OP == DS_READ_B64_STRIDE7) LDS4_64("ds_read_b64")
            if (OP == DS_READ_B32_MIS1 || OP == DS_READ_B32_STRIDE5) LDS4("ds_read_b32")
        }
        if (OP >= DS_READ_U8 && OP < NOPS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (unsigned)(A + B + C + D) + (unsigned)((A + B + C + D) >> 32);
}

template <int OP> void run(const char *name, int lanesNote = 0) {
    std::printf("%-20s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        unsigned *d; unsigned long long *tk;
        (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
        (void)hipMalloc(&tk, (size_t)blocks * 4 * 8);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int iters = 2048;
        const size_t lds = std::max<size_t>(4096, (size_t)(160 * 1024 / wps) - 2048);        // exactly wps workgroups fit per CU
        (void)hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, tk, 1u, 16);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, tk, 1u, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)blocks * 4);
        (void)hipMemcpy(h.data(), tk, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double perWave = (double)iters * 16 * 4;                       // instructions one wave issued
        const double waveInstr = (double)blocks * 4 * perWave;
        // wall: all SIMDs busy the whole time; ticks: one wave's own loop, of which it gets 1 / wps of its SIMD
        const double cycWall = 1024 * 2.4e9 / (waveInstr / (ms * 1e-3));
        const double cycTick = (double)h[h.size() / 2] / perWave / wps;
        std::printf("  %dw: %5.2f wall %5.2f tick", wps, cycWall, cycTick);
        (void)hipFree(d); (void)hipFree(tk);
    }
    std::printf("\n");
}

int main(int argc, char **argv) {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    if (argc > 1 && std::string(argv[1]) == "f64") {                        // the double-precision rows alone (the optimiser kernels)
        std::printf("# tools/valu_rate.hip f64 on %s: cycles per wave64 instruction and SIMD at 1 / 2 / 4 / 8 resident waves per SIMD (four independent chains per lane; 'dep' = one chain)\n", p.gcnArchName);
        run<FMA_F32>("v_fma_f32 (control)"); run<FMA_F64>("v_fma_f64"); run<MUL_F64>("v_mul_f64"); run<ADD_F64>("v_add_f64"); run<FMA_F64_DEP>("v_fma_f64 dep");
        run<RSQ_F64>("v_rsq_f64"); run<RCP_F64>("v_rcp_f64");
        return 0;
    }
    std::printf("# tools/valu_rate.hip on %s (%s), %d CUs, clockRate %d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
    std::printf("# cycles per wave64 instruction and SIMD at 1 / 2 / 4 / 8 resident waves per SIMD; 'wall' = HIP-event time priced at 2.4 GHz over 1024 SIMDs,\n");
    std::printf("# 'tick' = the median wave's own s_memtime delta / instructions / waves per SIMD (clock-independent).  LDS rows: per SIMD, so a CU's LDS serves 4x that rate.\n");
    std::printf("# -- controls (float pipe, plain integer) --\n");
    run<FMA_F32>("v_fma_f32"); run<PK_FMA_F32>("v_pk_fma_f32"); run<ADD_F32>("v_add_f32"); run<PK_ADD_F32>("v_pk_add_f32"); run<SUB_F32>("v_sub_f32");
    run<ADD_U32>("v_add_u32"); run<AND_B32>("v_and_b32"); run<XOR_B32>("v_xor_b32");
    std::printf("# -- float min/max and conversions (contrasts <= 255 are exact in f32 and f16) --\n");
    run<MIN3_F32>("v_min3_f32"); run<MAX3_F32>("v_max3_f32"); run<MINIMUM3_F32>("v_minimum3_f32"); run<CVT_F32_UBYTE0>("v_cvt_f32_ubyte0"); run<CVT_F32_UBYTE3>("v_cvt_f32_ubyte3");
    run<CVT_U32_F32>("v_cvt_u32_f32"); run<PK_MINIMUM3_F16>("v_pk_minimum3_f16"); run<PK_MAXIMUM3_F16>("v_pk_maximum3_f16"); run<PK_MIN_F16>("v_pk_min_f16"); run<PK_ADD_F16>("v_pk_add_f16");
    std::printf("# -- integer classes used by the kernels --\n");
    run<MIN_I32>("v_min_i32"); run<MIN3_I32>("v_min3_i32"); run<MAX3_I32>("v_max3_i32"); run<MIN3_I16>("v_min3_i16"); run<PK_MIN_I16>("v_pk_min_i16"); run<PK_MAX_I16>("v_pk_max_i16");
    run<PK_SUB_I16>("v_pk_sub_i16"); run<PK_ADD_U16>("v_pk_add_u16"); run<LSHL_OR_B32>("v_lshl_or_b32"); run<ADD3_U32>("v_add3_u32"); run<MAD_I32_I24>("v_mad_i32_i24"); run<MAD_U32_U24>("v_mad_u32_u24");
    run<MUL_LO_U32>("v_mul_lo_u32"); run<DOT4_U32_U8>("v_dot4_u32_u8"); run<BCNT>("v_bcnt_u32_b32"); run<PERM>("v_perm_b32"); run<ALIGNBYTE>("v_alignbyte_b32"); run<SAD_U8>("v_sad_u8");
    run<CNDMASK>("v_cndmask_b32"); run<CMP_GT_I32>("v_cmp_gt_i32"); run<MBCNT>("v_mbcnt_lo_u32_b32");
    std::printf("# -- more two-operand and three-operand classes --\n");
    run<OR_B32>("v_or_b32"); run<SUB_U32>("v_sub_u32"); run<LSHLREV>("v_lshlrev_b32"); run<LSHRREV>("v_lshrrev_b32"); run<ASHRREV>("v_ashrrev_i32"); run<MOV_B32>("v_mov_b32"); run<NOT_B32>("v_not_b32");
    run<MUL_F32>("v_mul_f32"); run<FMAC_F32>("v_fmac_f32"); run<MAX_F32>("v_max_f32"); run<MIN_F32>("v_min_f32"); run<MAX_U32>("v_max_u32"); run<MIN_U32>("v_min_u32"); run<MAX_I32>("v_max_i32");
    run<MIN_U16>("v_min_u16"); run<SUB_U16>("v_sub_u16"); run<MUL_U32_U24>("v_mul_u32_u24"); run<BFE_U32>("v_bfe_u32"); run<AND_OR_B32>("v_and_or_b32"); run<OR3_B32>("v_or3_b32");
    run<LSHL_ADD_U32>("v_lshl_add_u32"); run<XAD_U32>("v_xad_u32"); run<BFI_B32>("v_bfi_b32"); run<MED3_I32>("v_med3_i32"); run<MAX3_U32>("v_max3_u32"); run<MUL_HI_U32>("v_mul_hi_u32"); run<CVT_F32_I32>("v_cvt_f32_i32");
    run<CNDMASK_S>("v_cndmask_b32 sgpr"); run<CMP_GT_U32_S>("v_cmp_gt_u32 sgpr"); run<CMP_GT_F32>("v_cmp_gt_f32 sgpr");
    std::printf("# -- SDWA (byte / word operand selects) and DPP --\n");
    run<SDWA_SUB_U32_BYTES>("sub_u32 sdwa b0,b2"); run<SDWA_ADD_U32_W1>("add_u32 sdwa ->w1"); run<SDWA_AND_BYTE>("and_b32 sdwa b1"); run<SDWA_MIN_U16_BYTES>("min_u16 sdwa b0,b1");
    run<SDWA_MAX_I32_WORDS>("max_i32 sdwa w0,w1"); run<SDWA_SUB_F32>("sub_f32 sdwa dword"); run<SDWA_CVT_UBYTE>("cvt_f32_ubyte0 sdwa"); run<SDWA_MOV_W1>("mov_b32 sdwa ->w1"); run<SDWA_CMP_BYTES>("cmp_gt_u32 sdwa");
    run<DPP_MOV_SHR1>("mov_b32 dpp row_shr"); run<DPP_ADD_SHR1>("add_u32 dpp row_shr"); run<DPP_MOV_BCAST>("mov_b32 dpp wave_shr"); run<READLANE>("v_readlane_b32");
    std::printf("# -- LDS issue (conflict-free, 4 in flight, drained every 64) --\n");
    run<DS_READ_U8>("ds_read_u8"); run<DS_READ_U8_D16_HI>("ds_read_u8_d16_hi"); run<DS_READ_B32>("ds_read_b32"); run<DS_READ_B64>("ds_read_b64");
    std::printf("# -- LDS reads at addresses that are not multiples of their size (unaligned-ds-access) --\n");
    run<DS_READ_B32_MIS1>("b32 at 4 l + 1"); run<DS_READ_B64_MIS1>("b64 at 8 l + 1"); run<DS_READ_B64_MIS4>("b64 at 8 l + 4"); run<DS_READ_B32_STRIDE5>("b32 at 5 l"); run<DS_READ_B64_STRIDE7>("b64 at 7 l");
    return 0;
}
