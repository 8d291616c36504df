// Device-side building blocks shared by every kernel of the hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int u32;
typedef unsigned long long u64;

#define PG_WAVE 64
#define PG_WG_WAVES 4
#define PG_WG_THREADS (PG_WAVE * PG_WG_WAVES)
#define PG_RB 16    // rows per wave pass of the all-pairs engine
#define PG_RBD 64   // rows per workgroup of the dense kernel

enum { PG_MODE_EPS = 0, PG_MODE_KNN = 1 };

// ---------------------------------------------------------------------------------------
// Mismatch counting.  One dword holds 4 byte tokens; x = a ^ b has a non-zero byte exactly
// where the tokens differ.
//   7-bit alphabets (every token <= 127, so every byte of x <= 0x7f):
//       (x + 0x7f7f7f7f) sets bit 7 of a byte iff that byte of x is non-zero and never
//       carries into the next byte.  v_xad_u32 does the xor and the add in one VALU op,
//       v_and_b32 isolates the flags, v_bcnt_u32_b32 popcounts AND accumulates:
//       3 VALU ops per 4 tokens.
//   8-bit alphabets: ((x & 0x7f..) + 0x7f..) | x has bit 7 set iff the byte is non-zero:
//       5 VALU ops per 4 tokens.
// ---------------------------------------------------------------------------------------
// The byte masks are passed to the kernels as ARGUMENTS (struct MisK), not literals: gfx9
// VOP3 cannot encode a 32-bit literal, and with literals hipcc (ROCm 7.2) splits
// (a ^ b) + k into v_xor + v_add and turns the popcount chain into v_bcnt(x, 0) + v_add3
// trees.  With the constants opaque in SGPRs and the row operand NOT provably wave-uniform
// (see `opaque_zero`) it selects v_xad_u32 and v_bfi_b32 by itself.  Only the accumulating
// popcount is pinned with a one-instruction asm: ISel prefers v_bcnt(x,0)+v_add3 otherwise.
// (hipcc pads an asm result with s_nop only in front of a compiler-generated consumer; the
// chain's consumers are the next asm, so one pad per sequence pair remains.)
__device__ __forceinline__ u32 bcnt_acc(u32 x, u32 acc) {
  u32 t;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(acc));
  return t;
}
// A VGPR that holds 0 in every lane but that the compiler must treat as divergent.  LLVM's
// uniformity analysis marks a broadcast LDS read (same address in all lanes) as uniform and
// then refuses the three-operand VALU patterns (two "scalar" operands would exceed the gfx9
// constant-bus limit) although the value lives in a VGPR; adding this to the LDS index
// keeps the loaded row operand divergent at zero cost (it folds into the address VGPR).
__device__ __forceinline__ int opaque_zero() {
  int z;
  asm("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}
struct MisK {
  u32 k7f, k3f, k1f;      // 0x7f7f7f7f 0x3f3f3f3f 0x1f1f1f1f
  u32 m80, mc0, me0;      // 0x80808080 0xc0c0c0c0 0xe0e0e0e0
};

template <int ALPHA>
__device__ __forceinline__ u32 mis_acc(const MisK &K, u32 a, u32 b, u32 acc) {
  if constexpr (ALPHA == 8) {
    u32 x = a ^ b;
    u32 t = ((x & K.k7f) + K.k7f) | x;
    return bcnt_acc(t & K.m80, acc);
  } else {
    return bcnt_acc(((a ^ b) + K.k7f) & K.m80, acc);
  }
}

//   5-bit alphabets (every token <= 31, which covers the 20 amino acids + pad): three
//       dwords share one mask and one popcount.  Adding 0x7f / 0x3f / 0x1f puts the
//       "differs" flag of a byte in bit 7 / 6 / 5 with zeros above it, v_bfi_b32 merges the
//       three words: 3 xad + 2 bfi + and + bcnt = 7 VALU ops per 12 tokens.
//       hipcc lowers the merge to v_and + v_bitop3 pairs (10 ops), so the group is one asm
//       block; VALU->VALU dependencies are interlocked in hardware, no wait states needed.
__device__ __forceinline__ u32 mis_acc3_5bit(const MisK &K, u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2, u32 acc) {
  u32 t0, t1, t2, o;
  asm("v_xad_u32 %0, %4, %5, %10\n\t"
      "v_xad_u32 %1, %6, %7, %11\n\t"
      "v_xad_u32 %2, %8, %9, %12\n\t"
      "v_bfi_b32 %0, %13, %0, %1\n\t"
      "v_bfi_b32 %0, %14, %0, %2\n\t"
      "v_and_b32 %0, %15, %0\n\t"
      "v_bcnt_u32_b32 %3, %0, %16"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=v"(o)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "s"(K.k7f), "s"(K.k3f), "s"(K.k1f),
        "s"(K.m80), "s"(K.mc0), "s"(K.me0), "v"(acc));
  return o;
}
__device__ __forceinline__ u32 mis_acc2_5bit(const MisK &K, u32 a0, u32 b0, u32 a1, u32 b1, u32 acc) {
  u32 t0, t1, o;
  asm("v_xad_u32 %0, %3, %4, %7\n\t"
      "v_xad_u32 %1, %5, %6, %8\n\t"
      "v_bfi_b32 %0, %9, %0, %1\n\t"
      "v_and_b32 %0, %10, %0\n\t"
      "v_bcnt_u32_b32 %2, %0, %11"
      : "=&v"(t0), "=&v"(t1), "=v"(o)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "s"(K.k7f), "s"(K.k3f), "s"(K.m80), "s"(K.mc0), "v"(acc));
  return o;
}

// per-byte "differs" flags (bit 7 of each byte) of one dword pair
template <int ALPHA>
__device__ __forceinline__ u32 mis_flags(u32 a, u32 b) {
  if constexpr (ALPHA == 8) {
    u32 x = a ^ b;
    return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
  } else {
    return ((a ^ b) + 0x7f7f7f7fu) & 0x80808080u;
  }
}

// mismatches of one sequence pair, added to `init` (callers fold a bias into it)
template <int Q, int ALPHA>
__device__ __forceinline__ u32 mismatch(const MisK &K, const uint4 (&r)[Q], const uint4 (&c)[Q], u32 init = 0) {
  u32 acc = init;
  if constexpr (ALPHA == 5) {
    constexpr int W = 4 * Q;
    u32 rw[W], cw[W];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      rw[4 * q] = r[q].x; rw[4 * q + 1] = r[q].y; rw[4 * q + 2] = r[q].z; rw[4 * q + 3] = r[q].w;
      cw[4 * q] = c[q].x; cw[4 * q + 1] = c[q].y; cw[4 * q + 2] = c[q].z; cw[4 * q + 3] = c[q].w;
    }
    constexpr int G = W / 3;
#pragma unroll
    for (int g = 0; g < G; ++g)
      acc = mis_acc3_5bit(K, rw[3 * g], cw[3 * g], rw[3 * g + 1], cw[3 * g + 1], rw[3 * g + 2], cw[3 * g + 2], acc);
    if constexpr (W - 3 * G == 2) acc = mis_acc2_5bit(K, rw[W - 2], cw[W - 2], rw[W - 1], cw[W - 1], acc);
    if constexpr (W - 3 * G == 1) acc = mis_acc<7>(K, rw[W - 1], cw[W - 1], acc);
  } else {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      acc = mis_acc<ALPHA>(K, r[q].x, c[q].x, acc);
      acc = mis_acc<ALPHA>(K, r[q].y, c[q].y, acc);
      acc = mis_acc<ALPHA>(K, r[q].z, c[q].z, acc);
      acc = mis_acc<ALPHA>(K, r[q].w, c[q].w, acc);
    }
  }
  return acc;
}

// lane's rank among the set bits of a 64-bit wave mask (exclusive prefix popcount)
__device__ __forceinline__ u32 mask_rank(u64 m) {
  return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// lane i <- lane i-1 (lane 0 <- `fill`) with one DPP move: wave_shr:1
__device__ __forceinline__ u32 wave_shr1(u32 v, u32 fill) {
  return (u32)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);
}

// Parameters of the all-pairs engine (one struct so the per-Q translation units share it)
struct NsqParams {
  MisK K;
  const uint4 *rowPlanes;
  long long rowNpad, row0, nrows;
  const uint4 *colPlanes;
  long long colNpad, ncols;
  int rowsPerWave, rowsPerPass;
  // eps
  u32 lo, span, cap;
  int *slotIdx;
  unsigned char *slotW;
  u32 *counts;
  // knn
  int k;
  int *knnIdx;
  unsigned char *knnDist;
};

struct DenseParams {
  MisK K;
  const uint4 *xPlanes;   // columns of the output (N)
  long long xNpad, n;
  const uint4 *yPlanes;   // rows of the output (M)
  long long yNpad, m;
  void *out;
  long long ldo;
  int outBytes;
};

struct CompactParams {
  NsqParams e;
  const long long *indptr;
  int *indices;
  unsigned char *weights;
};

// per-Q launchers (pg_nsq_inst.hip compiled once per Q)
#define PG_DECL_Q(Q)                                                                      \
  int pg_launch_nsq_q##Q(int mode, int alpha, const NsqParams &p, int grid, hipStream_t s); \
  int pg_launch_dense_q##Q(int alpha, const DenseParams &p, hipStream_t s);               \
  int pg_launch_compact_q##Q(int alpha, const CompactParams &p, hipStream_t s);
PG_DECL_Q(1) PG_DECL_Q(2) PG_DECL_Q(3) PG_DECL_Q(4)
PG_DECL_Q(5) PG_DECL_Q(6) PG_DECL_Q(7) PG_DECL_Q(8)
