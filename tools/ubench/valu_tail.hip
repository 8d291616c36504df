// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the integer VALU
// instructions the all-pairs engine is built from, independent streams (8 registers) and
// dependent chains, at 1 / 2 / 4 resident waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 valu_ops.hip -o valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define IND8(OP) \
  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

#define KERNEL(NAME, BODY)                                                                     \
  __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned r, unsigned s, int iters) { \
    unsigned v0 = threadIdx.x, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 11, v5 = v0 * 13, v6 = v0 * 17,   \
             v7 = v0 * 19;                                                                    \
    for (int i = 0; i < iters; ++i) {                                                         \
      asm volatile(REP8(BODY)                                                                 \
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)      \
                   : "v"(r), "v"(s)                                                           \
                   : "vcc", "s20", "s21", "s22", "s23");                                                    \
    }                                                                                         \
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;              \
  }

// 8 instructions per BODY, each on its own register (independent)
KERNEL(k_xor, "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7\n")
KERNEL(k_bcnt0, "v_bcnt_u32_b32 %0, %0, 0\n v_bcnt_u32_b32 %1, %1, 0\n v_bcnt_u32_b32 %2, %2, 0\n v_bcnt_u32_b32 %3, %3, 0\n v_bcnt_u32_b32 %4, %4, 0\n v_bcnt_u32_b32 %5, %5, 0\n v_bcnt_u32_b32 %6, %6, 0\n v_bcnt_u32_b32 %7, %7, 0\n")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %8, %9, %0 bitop3:0xbe\n v_bitop3_b32 %1, %8, %9, %1 bitop3:0xbe\n v_bitop3_b32 %2, %8, %9, %2 bitop3:0xbe\n v_bitop3_b32 %3, %8, %9, %3 bitop3:0xbe\n v_bitop3_b32 %4, %8, %9, %4 bitop3:0xbe\n v_bitop3_b32 %5, %8, %9, %5 bitop3:0xbe\n v_bitop3_b32 %6, %8, %9, %6 bitop3:0xbe\n v_bitop3_b32 %7, %8, %9, %7 bitop3:0xbe\n")
KERNEL(k_min, "v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8\n")
KERNEL(k_min3, "v_min3_u32 %0, %0, %8, %9\n v_min3_u32 %1, %1, %8, %9\n v_min3_u32 %2, %2, %8, %9\n v_min3_u32 %3, %3, %8, %9\n v_min3_u32 %4, %4, %8, %9\n v_min3_u32 %5, %5, %8, %9\n v_min3_u32 %6, %6, %8, %9\n v_min3_u32 %7, %7, %8, %9\n")
KERNEL(k_add, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
KERNEL(k_add3, "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9\n")
KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8\n")
KERNEL(k_cmps, "v_cmp_lt_u32 s[20:21], %0, %8\n v_cmp_lt_u32 s[22:23], %1, %8\n v_cmp_lt_u32 s[20:21], %2, %8\n v_cmp_lt_u32 s[22:23], %3, %8\n v_cmp_lt_u32 s[20:21], %4, %8\n v_cmp_lt_u32 s[22:23], %5, %8\n v_cmp_lt_u32 s[20:21], %6, %8\n v_cmp_lt_u32 s[22:23], %7, %8\n")
KERNEL(k_sad, "v_sad_u8 %0, %8, %9, %0\n v_sad_u8 %1, %8, %9, %1\n v_sad_u8 %2, %8, %9, %2\n v_sad_u8 %3, %8, %9, %3\n v_sad_u8 %4, %8, %9, %4\n v_sad_u8 %5, %8, %9, %5\n v_sad_u8 %6, %8, %9, %6\n v_sad_u8 %7, %8, %9, %7\n")
KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 3, %8\n v_lshl_or_b32 %1, %1, 3, %8\n v_lshl_or_b32 %2, %2, 3, %8\n v_lshl_or_b32 %3, %3, 3, %8\n v_lshl_or_b32 %4, %4, 3, %8\n v_lshl_or_b32 %5, %5, 3, %8\n v_lshl_or_b32 %6, %6, 3, %8\n v_lshl_or_b32 %7, %7, 3, %8\n")
KERNEL(k_perm, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n")
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
KERNEL(k_pkadd, "v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n")
KERNEL(k_dot4, "v_dot4_u32_u8 %0, %8, %9, %0\n v_dot4_u32_u8 %1, %8, %9, %1\n v_dot4_u32_u8 %2, %8, %9, %2\n v_dot4_u32_u8 %3, %8, %9, %3\n v_dot4_u32_u8 %4, %8, %9, %4\n v_dot4_u32_u8 %5, %8, %9, %5\n v_dot4_u32_u8 %6, %8, %9, %6\n v_dot4_u32_u8 %7, %8, %9, %7\n")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n v_readlane_b32 s22, %6, 3\n v_readlane_b32 s23, %7, 3\n")
// dependent chains: every instruction reads the previous result
KERNEL(k_xor_dep, "v_xor_b32 %0, %0, %8\n v_xor_b32 %0, %0, %9\n v_xor_b32 %0, %0, %8\n v_xor_b32 %0, %0, %9\n v_xor_b32 %0, %0, %8\n v_xor_b32 %0, %0, %9\n v_xor_b32 %0, %0, %8\n v_xor_b32 %0, %0, %9\n")
KERNEL(k_bcnt_dep, "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %0, %9, %0\n v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %0, %9, %0\n v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %0, %9, %0\n v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %0, %9, %0\n")
KERNEL(k_bitop3_dep, "v_bitop3_b32 %0, %8, %9, %0 bitop3:0xbe\n v_bitop3_b32 %0, %9, %8, %0 bitop3:0xbe\n v_bitop3_b32 %0, %8, %9, %0 bitop3:0xbe\n v_bitop3_b32 %0, %9, %8, %0 bitop3:0xbe\n v_bitop3_b32 %0, %8, %9, %0 bitop3:0xbe\n v_bitop3_b32 %0, %9, %8, %0 bitop3:0xbe\n v_bitop3_b32 %0, %8, %9, %0 bitop3:0xbe\n v_bitop3_b32 %0, %9, %8, %0 bitop3:0xbe\n")
// the stage-1 pattern of the engine: xor, bcnt, xor, bcnt, min, cmp (per row, two columns)
KERNEL(k_stage1, "v_xor_b32 %0, %8, %4\n v_bcnt_u32_b32 %1, %0, 0\n v_xor_b32 %0, %8, %5\n v_bcnt_u32_b32 %2, %0, 0\n v_min_u32 %0, %1, %2\n v_cmp_lt_u32 vcc, %0, %9\n v_xor_b32 %3, %9, %4\n v_bcnt_u32_b32 %6, %3, 0\n")

KERNEL(k_or, "v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8\n")
KERNEL(k_and, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
KERNEL(k_sub, "v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8\n")
KERNEL(k_or3, "v_or3_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_or3_b32 %2, %2, %8, %9\n v_or3_b32 %3, %3, %8, %9\n v_or3_b32 %4, %4, %8, %9\n v_or3_b32 %5, %5, %8, %9\n v_or3_b32 %6, %6, %8, %9\n v_or3_b32 %7, %7, %8, %9\n")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7\n")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
KERNEL(k_max, "v_max_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n v_max_u32 %4, %4, %8\n v_max_u32 %5, %5, %8\n v_max_u32 %6, %6, %8\n v_max_u32 %7, %7, %8\n")
KERNEL(k_xad, "v_xad_u32 %0, %0, %8, %9\n v_xad_u32 %1, %1, %8, %9\n v_xad_u32 %2, %2, %8, %9\n v_xad_u32 %3, %3, %8, %9\n v_xad_u32 %4, %4, %8, %9\n v_xad_u32 %5, %5, %8, %9\n v_xad_u32 %6, %6, %8, %9\n v_xad_u32 %7, %7, %8, %9\n")
KERNEL(k_andor, "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9\n")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8\n")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 5\n v_bfe_u32 %1, %1, 3, 5\n v_bfe_u32 %2, %2, 3, 5\n v_bfe_u32 %3, %3, 3, 5\n v_bfe_u32 %4, %4, 3, 5\n v_bfe_u32 %5, %5, 3, 5\n v_bfe_u32 %6, %6, 3, 5\n v_bfe_u32 %7, %7, 3, 5\n")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %8, %0\n v_mbcnt_lo_u32_b32 %1, %8, %1\n v_mbcnt_lo_u32_b32 %2, %8, %2\n v_mbcnt_lo_u32_b32 %3, %8, %3\n v_mbcnt_lo_u32_b32 %4, %8, %4\n v_mbcnt_lo_u32_b32 %5, %8, %5\n v_mbcnt_lo_u32_b32 %6, %8, %6\n v_mbcnt_lo_u32_b32 %7, %8, %7\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %8, 7\n v_alignbit_b32 %1, %1, %8, 7\n v_alignbit_b32 %2, %2, %8, 7\n v_alignbit_b32 %3, %3, %8, 7\n v_alignbit_b32 %4, %4, %8, 7\n v_alignbit_b32 %5, %5, %8, 7\n v_alignbit_b32 %6, %6, %8, 7\n v_alignbit_b32 %7, %7, %8, 7\n")
KERNEL(k_addco, "v_add_co_u32 %0, vcc, %0, %8\n v_add_co_u32 %1, vcc, %1, %8\n v_add_co_u32 %2, vcc, %2, %8\n v_add_co_u32 %3, vcc, %3, %8\n v_add_co_u32 %4, vcc, %4, %8\n v_add_co_u32 %5, vcc, %5, %8\n v_add_co_u32 %6, vcc, %6, %8\n v_add_co_u32 %7, vcc, %7, %8\n")
KERNEL(k_subrev, "v_subrev_u32 %0, %8, %0\n v_subrev_u32 %1, %8, %1\n v_subrev_u32 %2, %8, %2\n v_subrev_u32 %3, %8, %3\n v_subrev_u32 %4, %8, %4\n v_subrev_u32 %5, %8, %5\n v_subrev_u32 %6, %8, %6\n v_subrev_u32 %7, %8, %7\n")
KERNEL(k_not, "v_not_b32 %0, %0\n v_not_b32 %1, %1\n v_not_b32 %2, %2\n v_not_b32 %3, %3\n v_not_b32 %4, %4\n v_not_b32 %5, %5\n v_not_b32 %6, %6\n v_not_b32 %7, %7\n")
KERNEL(k_xnor, "v_xnor_b32 %0, %0, %8\n v_xnor_b32 %1, %1, %8\n v_xnor_b32 %2, %2, %8\n v_xnor_b32 %3, %3, %8\n v_xnor_b32 %4, %4, %8\n v_xnor_b32 %5, %5, %8\n v_xnor_b32 %6, %6, %8\n v_xnor_b32 %7, %7, %8\n")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 3, %0\n v_ashrrev_i32 %1, 3, %1\n v_ashrrev_i32 %2, 3, %2\n v_ashrrev_i32 %3, 3, %3\n v_ashrrev_i32 %4, 3, %4\n v_ashrrev_i32 %5, 3, %5\n v_ashrrev_i32 %6, 3, %6\n v_ashrrev_i32 %7, 3, %7\n")
KERNEL(k_fma, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
KERNEL(k_addf, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
KERNEL(k_mulf, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
KERNEL(k_bcnt_s, "v_bcnt_u32_b32 %0, %0, s20\n v_bcnt_u32_b32 %1, %1, s20\n v_bcnt_u32_b32 %2, %2, s20\n v_bcnt_u32_b32 %3, %3, s20\n v_bcnt_u32_b32 %4, %4, s20\n v_bcnt_u32_b32 %5, %5, s20\n v_bcnt_u32_b32 %6, %6, s20\n v_bcnt_u32_b32 %7, %7, s20\n")
KERNEL(k_cmp_i, "v_cmp_gt_i32 vcc, 0, %0\n v_cmp_gt_i32 vcc, 0, %1\n v_cmp_gt_i32 vcc, 0, %2\n v_cmp_gt_i32 vcc, 0, %3\n v_cmp_gt_i32 vcc, 0, %4\n v_cmp_gt_i32 vcc, 0, %5\n v_cmp_gt_i32 vcc, 0, %6\n v_cmp_gt_i32 vcc, 0, %7\n")
__global__ __launch_bounds__(256) void k_tA(unsigned *out, unsigned r, unsigned s, int iters) {
 unsigned v0 = threadIdx.x, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 11, v5 = v0 * 13, v6 = v0 * 17, v7 = v0 * 19;
 for (int i = 0; i < iters; ++i) { asm volatile("v_xor_b32 %0, %8, %9\n v_xor_b32 %1, %8, %9\n v_xor_b32 %2, %8, %9\n v_xor_b32 %3, %8, %9\n v_xor_b32 %4, %8, %9\n v_xor_b32 %5, %8, %9\n v_xor_b32 %6, %8, %9\n v_xor_b32 %7, %8, %9\n v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n v_or_b32 %0, %0, %1\n v_or_b32 %2, %2, %3\n v_or_b32 %4, %4, %5\n v_or_b32 %6, %6, %7\n v_or_b32 %0, %0, %2\n v_or3_b32 %0, %0, %4, %6\n v_cmp_gt_i32 vcc, 0, %0\n v_xor_b32 %0, %8, %9\n v_xor_b32 %1, %8, %9\n v_xor_b32 %2, %8, %9\n v_xor_b32 %3, %8, %9\n v_xor_b32 %4, %8, %9\n v_xor_b32 %5, %8, %9\n v_xor_b32 %6, %8, %9\n v_xor_b32 %7, %8, %9\n v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n v_or_b32 %0, %0, %1\n v_or_b32 %2, %2, %3\n v_or_b32 %4, %4, %5\n v_or_b32 %6, %6, %7\n v_or_b32 %0, %0, %2\n v_or3_b32 %0, %0, %4, %6\n v_cmp_gt_i32 vcc, 0, %0\n v_xor_b32 %0, %8, %9\n v_xor_b32 %1, %8, %9\n v_xor_b32 %2, %8, %9\n v_xor_b32 %3, %8, %9\n v_xor_b32 %4, %8, %9\n v_xor_b32 %5, %8, %9\n v_xor_b32 %6, %8, %9\n v_xor_b32 %7, %8, %9\n v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n v_or_b32 %0, %0, %1\n v_or_b32 %2, %2, %3\n v_or_b32 %4, %4, %5\n v_or_b32 %6, %6, %7\n v_or_b32 %0, %0, %2\n v_or3_b32 %0, %0, %4, %6\n v_cmp_gt_i32 vcc, 0, %0\n " : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(r), "v"(s) : "vcc"); }
 out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7; }
__global__ __launch_bounds__(256) void k_tB(unsigned *out, unsigned r, unsigned s, int iters) {
 unsigned v0 = threadIdx.x, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 11, v5 = v0 * 13, v6 = v0 * 17, v7 = v0 * 19;
 for (int i = 0; i < iters; ++i) { asm volatile("v_xor_b32 %0, %8, %9\n v_xor_b32 %1, %8, %9\n v_xor_b32 %2, %8, %9\n v_xor_b32 %3, %8, %9\n v_xor_b32 %4, %8, %9\n v_xor_b32 %5, %8, %9\n v_xor_b32 %6, %8, %9\n v_xor_b32 %7, %8, %9\n v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xfe\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0xfe\n v_bitop3_b32 %6, %6, %7, %0 bitop3:0xfe\n v_or_b32 %3, %3, %6\n v_cmp_gt_i32 vcc, 0, %3\n v_xor_b32 %0, %8, %9\n v_xor_b32 %1, %8, %9\n v_xor_b32 %2, %8, %9\n v_xor_b32 %3, %8, %9\n v_xor_b32 %4, %8, %9\n v_xor_b32 %5, %8, %9\n v_xor_b32 %6, %8, %9\n v_xor_b32 %7, %8, %9\n v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xfe\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0xfe\n v_bitop3_b32 %6, %6, %7, %0 bitop3:0xfe\n v_or_b32 %3, %3, %6\n v_cmp_gt_i32 vcc, 0, %3\n v_xor_b32 %0, %8, %9\n v_xor_b32 %1, %8, %9\n v_xor_b32 %2, %8, %9\n v_xor_b32 %3, %8, %9\n v_xor_b32 %4, %8, %9\n v_xor_b32 %5, %8, %9\n v_xor_b32 %6, %8, %9\n v_xor_b32 %7, %8, %9\n v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xfe\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0xfe\n v_bitop3_b32 %6, %6, %7, %0 bitop3:0xfe\n v_or_b32 %3, %3, %6\n v_cmp_gt_i32 vcc, 0, %3\n " : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(r), "v"(s) : "vcc"); }
 out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7; }
__global__ __launch_bounds__(256) void k_tC(unsigned *out, unsigned r, unsigned s, int iters) {
 unsigned v0 = threadIdx.x, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 11, v5 = v0 * 13, v6 = v0 * 17, v7 = v0 * 19;
 for (int i = 0; i < iters; ++i) { asm volatile("v_xor_b32 %0, %8, %9\n v_xor_b32 %1, %8, %9\n v_xor_b32 %2, %8, %9\n v_xor_b32 %3, %8, %9\n v_xor_b32 %4, %8, %9\n v_xor_b32 %5, %8, %9\n v_xor_b32 %6, %8, %9\n v_xor_b32 %7, %8, %9\n v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %3, %3, %4, %5\n v_or3_b32 %6, %6, %7, %0\n v_or_b32 %3, %3, %6\n v_cmp_gt_i32 vcc, 0, %3\n v_xor_b32 %0, %8, %9\n v_xor_b32 %1, %8, %9\n v_xor_b32 %2, %8, %9\n v_xor_b32 %3, %8, %9\n v_xor_b32 %4, %8, %9\n v_xor_b32 %5, %8, %9\n v_xor_b32 %6, %8, %9\n v_xor_b32 %7, %8, %9\n v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %3, %3, %4, %5\n v_or3_b32 %6, %6, %7, %0\n v_or_b32 %3, %3, %6\n v_cmp_gt_i32 vcc, 0, %3\n v_xor_b32 %0, %8, %9\n v_xor_b32 %1, %8, %9\n v_xor_b32 %2, %8, %9\n v_xor_b32 %3, %8, %9\n v_xor_b32 %4, %8, %9\n v_xor_b32 %5, %8, %9\n v_xor_b32 %6, %8, %9\n v_xor_b32 %7, %8, %9\n v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %3, %3, %4, %5\n v_or3_b32 %6, %6, %7, %0\n v_or_b32 %3, %3, %6\n v_cmp_gt_i32 vcc, 0, %3\n " : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(r), "v"(s) : "vcc"); }
 out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7; }

typedef void (*kern_t)(unsigned *, unsigned, unsigned, int);

static void run(const char *name, kern_t kern, unsigned *out, int NI) {
  const int iters = 4000, cus = 256;
  printf("%-12s", name);
  for (int w : {1, 2, 3, 4, 8}) {
    const int grid = cus * w;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<grid, 256>>>(out, 0x12345u, 0x777u, 50);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<grid, 256>>>(out, 0x12345u, 0x777u, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)w * iters * NI;                      // wave-instructions per SIMD
    printf("  w%d: %5.2f", w, ms * 1e-3 * 2.4e9 / instr);
  }
  printf("   cycles/instr/SIMD @2.4GHz\n");
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned *out;
  (void)hipMalloc(&out, 256 * 16 * 256 * sizeof(unsigned));
  run("k_tA", k_tA, out, 69);
  run("k_tB", k_tB, out, 63);
  run("k_tC", k_tC, out, 63);
  return 0;
}
