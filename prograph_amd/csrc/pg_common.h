// Device-side building blocks shared by every kernel of the hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int u32;
typedef unsigned long long u64;

#define PG_WAVE 64
#define PG_WG_WAVES 4
#define PG_WG_THREADS (PG_WAVE * PG_WG_WAVES)
#define PG_AUX_BYTES 64   // per sequence after the chunk arrays: MFMA signature fragments (32 B) + plane folds (32 B)
#define PG_RB 32    // rows per wave pass of the all-pairs engine (<= 64: per-row state is lane indexed)
#define PG_RB_KNN 28 // kNN passes: 28 rows, so that lists + candidate queue keep 4 workgroups per CU in LDS
#define PG_SORT_MAX 512 // symmetric eps: a row's back part (entries from lower rows) up to this size is rank-sorted in LDS
#ifndef PG_QCAP
#define PG_QCAP 128  // kNN: entries of the per-wave candidate queue (flushed in batches of 64)
#endif
#define PG_QCAP_EPS 576 // eps: every passing lane is queued (order!), a group can add 4 rows x 2 x 64
#ifndef PG_PUSH_MAX
#define PG_PUSH_MAX 8 // kNN: a triggered sub-tile with more passing lanes than this is evaluated in place
#endif
#define PG_RBD 64   // rows per workgroup of the dense kernel

enum { PG_MODE_EPS = 0, PG_MODE_KNN = 1, PG_MODE_EPS_SYM = 2 };   // EPS_SYM: square self-graph, upper triangle only
#define PG_MODE_KNN_SHORT 3   // launcher code only: pg_mm_kernel<.., PG_MODE_KNN, PG_MM_KL> (k + 1 <= PG_MM_KL, ranks from 1)
#define PG_MODE_KNN_SHORT2 4  // the same with two row blocks (64 rows) per pass: pg_mm_kernel<.., PG_MODE_KNN, PG_MM_KL, 2>
#ifndef PG_MM_KL
#define PG_MM_KL 20           // list entries of the short-list kNN instance of pg_mm.h (a multiple of 4)
#endif

// ---------------------------------------------------------------------------------------
// Mismatch counting on the BIT-SLICED layout.
//
// The all-pairs kernels are bound by VALU instruction count, not by bytes.  A SIMD-32 issues a wave64
// instruction in 2 cycles at best (MI355X_MICROARCH.md: the peak bench.py prices against); measured here
// (profiles/r01_valu_issue_microbench.txt) v_xor / v_or / v_bitop3 on VGPR operands take 2.4-2.9 cycles, v_or3 /
// v_bcnt / v_alignbit and anything with an SGPR source 4.2-4.4.  The first version of this kernel (byte tokens,
// xor/add/and/popcount, 7 ops per 12 tokens) already sat at 78 % of the 4-cycle rate.  The way down is fewer
// instructions per token:
//
//   a sequence is stored as G = ceil(L/32) groups of B bit planes; plane p of group g is one
//   dword whose bit j is bit p of token 32g+j.  Record order is PLANE MAJOR (dword p*G + g), so
//   chunk 0 of a record is plane 0 of every group: the lower-bound stage of the engine (below)
//   touches only that chunk.  Two sequences differ at position 32g+j iff
//   some plane differs there, so   t = OR_p (a[g][p] ^ b[g][p])   has one bit per mismatching
//   position and popcount(t) is the group's Hamming distance.  gfx950's v_bitop3_b32 evaluates
//   (a ^ b) | t in ONE instruction (truth table 0xBE for inputs a=0xF0, b=0xCC, c=0xAA), so a
//   group costs  1 v_xor + (B-1) v_bitop3 + 1 v_bcnt(+acc)  =  B+1 VALU ops per 32 tokens:
//   6 ops for 5-bit alphabets (the 20 amino acids + pad), 9 for full bytes.
// ---------------------------------------------------------------------------------------
#define PG_MAX_G 8          // groups of 32 positions: L <= 255 for 5-bit tokens (G <= 4, L <= 128 for bytes)
#define PG_BITOP_XOR_OR 0xBE

// A VGPR that holds 0 in every lane but that the compiler must treat as divergent.  LLVM's
// uniformity analysis marks a broadcast LDS read (same address in all lanes) as uniform and may
// then pull the row operand through scalar registers (v_readfirstlane) or refuse three-operand
// VALU forms on the gfx9 constant-bus limit, although the value already lives in a VGPR; adding
// this to the LDS index keeps the loaded row operand divergent at zero cost.
__device__ __forceinline__ int opaque_zero() {
  int z;
  asm("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}

// copy of a (possibly uniform) value into a VGPR the compiler cannot see through
__device__ __forceinline__ u32 opaque_vgpr(u32 x) {
  u32 v;
  asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
  return v;
}

template <int G, int B>
struct Rec {
  static constexpr int W = G * B;            // dwords per sequence record
  static constexpr int Q = (W + 3) / 4;      // 16-byte chunks per record (tail dwords are zero)
};

template <int Q>
__device__ __forceinline__ void unpack(const uint4 (&v)[Q], u32 (&w)[4 * Q]) {
#pragma unroll
  for (int q = 0; q < Q; ++q) { w[4 * q] = v[q].x; w[4 * q + 1] = v[q].y; w[4 * q + 2] = v[q].z; w[4 * q + 3] = v[q].w; }
}

// per-position "differs" bitmask of group g (record dwords are plane major: p*G + g)
template <int G, int B>
__device__ __forceinline__ u32 diff_bits(const u32 *a, const u32 *b, int g) {
  u32 t = a[g] ^ b[g];
#pragma unroll
  for (int p = 1; p < B; ++p) t = __builtin_amdgcn_bitop3_b32(a[p * G + g], b[p * G + g], t, PG_BITOP_XOR_OR);
  return t;
}

// Hamming distance of one sequence pair, added to `init` (callers fold a bias into it)
template <int G, int B>
__device__ __forceinline__ u32 mismatch(const uint4 (&r)[Rec<G, B>::Q], const uint4 (&c)[Rec<G, B>::Q], u32 init = 0) {
  constexpr int Q = Rec<G, B>::Q;
  u32 rw[4 * Q], cw[4 * Q];
  unpack<Q>(r, rw);
  unpack<Q>(c, cw);
  u32 acc = init;
#pragma unroll
  for (int g = 0; g < G; ++g) acc += __builtin_popcount(diff_bits<G, B>(rw, cw, g));
  return acc;
}

// LOWER BOUND of the Hamming distance from plane 0 of the first PG_LB_GROUPS groups (chunk 0 of
// both records): positions whose tokens differ in bit 0 certainly differ.  1 v_xor + 1 v_bcnt
// per 32 tokens looked at.  Unrelated sequences differ in bit 0 at about half of their positions
// (~16 +- 3 per group), far above the thresholds graph construction uses, so ONE group already
// rejects them; looking at more groups only costs instructions.
#ifndef PG_LB_GROUPS
#define PG_LB_GROUPS 1
#endif
template <int G>
__device__ __forceinline__ u32 mismatch_lb(const uint4 &r0, const uint4 &c0, u32 seed) {
  constexpr int GL = G < PG_LB_GROUPS ? G : PG_LB_GROUPS;
  u32 acc = __builtin_popcount(r0.x ^ c0.x) + seed;       // v_bcnt's addend carries the seed for free
  if constexpr (GL > 1) acc += __builtin_popcount(r0.y ^ c0.y);
  if constexpr (GL > 2) acc += __builtin_popcount(r0.z ^ c0.z);
  if constexpr (GL > 3) acc += __builtin_popcount(r0.w ^ c0.w);
  return acc;
}

// Metric policies of the all-pairs engine: record size in 16-byte chunks + the pair function.
template <int G, int B>
struct HammingMetric {
  static constexpr int Q = Rec<G, B>::Q;
  static constexpr int kGroups = G, kBits = B;
  static constexpr bool kHasLB = true;
  static __device__ __forceinline__ u32 dist(const uint4 (&r)[Q], const uint4 (&c)[Q], u32 init) {
    return mismatch<G, B>(r, c, init);
  }
  // seed + (lower bound of the distance); the engine seeds with -bound and tests the sign
  static __device__ __forceinline__ u32 lower_bound(const uint4 &r0, const uint4 &c0, u32 seed) {
    return mismatch_lb<G>(r0, c0, seed);
  }
  // the same bound in two steps, so that the engine can issue the 2-cycle logic ops of a whole row
  // group as one run and the 4-cycle popcounts as another (mixed streams run everything at the
  // 4-cycle rate: tools/ubench/valu_s1.hip)
  // Filter signature of a sequence: the XOR of the plane-0 words of all its groups (record dwords
  // 0..G-1).  A differing signature bit means that an odd number of the tokens folded onto it
  // differ in bit 0, so popcount(sig_a ^ sig_b) is a lower bound of the Hamming distance that looks
  // at EVERY position (the plane-0 word of group 0 alone only sees the first 32): near pairs are
  // rejected more often, unrelated pairs as before (each bit still differs with probability 1/2).
  // The engine folds a column once per tile and a row once per pass.
  static constexpr bool kSigFold = G > 1;                 // one group: the plane-0 word is the signature
  static __device__ __forceinline__ u32 fold(const uint4 (&rec)[Q]) {
    u32 w[4 * Q];
    unpack<Q>(rec, w);
    u32 s = w[0];
#pragma unroll
    for (int g = 1; g < G; ++g) s ^= w[g];
    return s;
  }
  // plane-0 words folded to two: XOR over the even groups, XOR over the odd groups (the 64 bits the MFMA
  // engine's signature is cut from: pg_sig54)
  static __device__ __forceinline__ void fold2(const uint4 (&rec)[Q], u32 &even, u32 &odd) {
    u32 w[4 * Q];
    unpack<Q>(rec, w);
    even = w[0];
    odd = G > 1 ? w[1] : 0u;
#pragma unroll
    for (int g = 2; g < G; ++g) {
      if (g & 1) odd ^= w[g]; else even ^= w[g];
    }
  }
  // XOR fold of plane `pl`'s words over all groups (the signature of that plane; pl = 0: fold())
  static __device__ __forceinline__ u32 fold_plane(const uint4 (&rec)[Q], int pl) {
    u32 w[4 * Q];
    unpack<Q>(rec, w);
    if (pl >= B) return 0u;
    u32 s = w[pl * G];
#pragma unroll
    for (int g = 1; g < G; ++g) s ^= w[pl * G + g];
    return s;
  }
  static __device__ __forceinline__ u32 lb_prep(const uint4 &r0, const uint4 &c0) { return r0.x ^ c0.x; }
  static __device__ __forceinline__ u32 lb_finish(u32 x, const uint4 &r0, const uint4 &c0, u32 seed) {
    constexpr int GL = G < PG_LB_GROUPS ? G : PG_LB_GROUPS;
    u32 acc = __builtin_popcount(x) + seed;
    if constexpr (GL > 1) acc += __builtin_popcount(r0.y ^ c0.y);
    if constexpr (GL > 2) acc += __builtin_popcount(r0.z ^ c0.z);
    if constexpr (GL > 3) acc += __builtin_popcount(r0.w ^ c0.w);
    return acc;
  }
};

// Edit-distance LOWER BOUND from the bag-of-symbols profile of a sequence (Levenshtein filter):
// record = 32 symbol counts (bytes, dwords 0..7) + the length stored in two bytes of dword 8.
//   sad  = sum_s |cnt_a(s) - cnt_b(s)|   (every edit changes it by at most 2)
//   len2 = 2 * |len_a - len_b|           (every edit changes the length by at most 1)
// so   max(sad, len2) <= 2 * d_edit;   9 v_sad_u8 + 1 v_max per pair.
struct BagMetric {
  static constexpr int Q = 3;
  // stage-1 bound of the engine: the SAD over the first 16 symbols (chunk 0) already exceeds
  // 2*band for almost every unrelated pair: 4 v_sad_u8 instead of 9 + v_max
  static constexpr bool kHasLB = true;
  static constexpr bool kSigFold = false;
  static __device__ __forceinline__ u32 fold(const uint4 (&)[Q]) { return 0u; }
  static __device__ __forceinline__ u32 lb_prep(const uint4 &, const uint4 &) { return 0u; }
  static __device__ __forceinline__ u32 lb_finish(u32, const uint4 &r0, const uint4 &c0, u32 seed) {
    return lower_bound(r0, c0, seed);
  }
  static __device__ __forceinline__ u32 lower_bound(const uint4 &r0, const uint4 &c0, u32 seed) {
    u32 s = __builtin_amdgcn_sad_u8(r0.x, c0.x, seed);
    s = __builtin_amdgcn_sad_u8(r0.y, c0.y, s);
    s = __builtin_amdgcn_sad_u8(r0.z, c0.z, s);
    return __builtin_amdgcn_sad_u8(r0.w, c0.w, s);
  }
  static __device__ __forceinline__ u32 dist(const uint4 (&r)[Q], const uint4 (&c)[Q], u32 init) {
    u32 rw[12], cw[12];
    unpack<Q>(r, rw);
    unpack<Q>(c, cw);
    u32 s = init;
#pragma unroll
    for (int i = 0; i < 8; ++i) s = __builtin_amdgcn_sad_u8(rw[i], cw[i], s);
    const u32 l2 = __builtin_amdgcn_sad_u8(rw[8], cw[8], init);
    return s > l2 ? s : l2;
  }
};

// lane's rank among the set bits of a 64-bit wave mask (exclusive prefix popcount)
__device__ __forceinline__ u32 mask_rank(u64 m) {
  return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// lane i <- lane i-1 (lane 0 <- `fill`) with one DPP move: wave_shr:1
// OR of v over the wave; the result is valid in lane 63 (DPP row shifts, then row_bcast:15 / :31)
__device__ __forceinline__ u32 wave_or_to63(u32 v) {
  v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8   -> lane 15 of a row: the row's OR
  v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
  v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2, 3
  return v;
}
__device__ __forceinline__ u32 wave_shr1(u32 v, u32 fill) {
  return (u32)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);
}

// ---- filter signature of the MFMA engine (pg_mm.h; written by pg_pack_planes) -------------------------------
// v_mfma_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands: K = 64 four-bit elements in the 16 bytes a lane holds,
// at the cycles of the int8 32x32x32 form.  54 of them carry the signature - the plane-0 bits of the sequence,
// 64 positions folded to 54 by XOR (a differing signature bit = an odd number of the positions folded onto it
// differ in bit 0, so popcount(sig_a ^ sig_b) <= Hamming distance) - and 10 carry the row's bias.
#define PG_SIG_BITS 54
#define PG_SIG_BIAS 10       // bias elements: sums of +-{1, 2, 3, 4, 6} reach every integer in [-60, 58] but -59
__host__ __device__ __forceinline__ unsigned long long pg_sig54(u32 even, u32 odd) {
  const unsigned long long s = (unsigned long long)even | ((unsigned long long)odd << 32);
  return (s & ((1ull << PG_SIG_BITS) - 1ull)) ^ (s >> PG_SIG_BITS);
}
// 8 bits -> 8 nibbles, bit i in bit 0 of nibble i
__host__ __device__ __forceinline__ u32 pg_nib8(u32 x) {
  u32 t = x & 0xFFu;
  t = (t | (t << 12)) & 0x000F000Fu;
  t = (t | (t << 6)) & 0x03030303u;
  t = (t | (t << 3)) & 0x11111111u;
  return t;
}
// ten FP4 nibbles (40 bits, element k = 54 + i in nibble i) whose sum is the largest representable value
// <= b, b clamped to [-60, 58]: q sixes, then the remainder (5 = 4 + 1); exact except b = -59 (-> -60) and 59
__host__ __device__ __forceinline__ unsigned long long pg_bias_nibbles(int b) {
  const bool neg = b < 0;
  u32 v = (u32)(neg ? -b : b);
  if (neg) v = v > 60u ? 60u : (v == 59u ? 60u : v);
  else v = v > 58u ? 58u : v;
  const u32 q = (v * 43u) >> 8, r = v - 6u * q;            // v <= 60
  unsigned long long s = q ? (0x7777777777ull >> (4u * (10u - q))) : 0ull;
  s |= ((0x260605040200ull >> (8u * r)) & 0xFFull) << (4u * q);
  const u32 cnt = q + (r == 0u ? 0u : (r == 5u ? 2u : 1u));
  if (neg && cnt) s |= 0x8888888888ull >> (4u * (10u - cnt));
  return s;
}

// Parameters of the all-pairs engine (one struct so the per-Q translation units share it)
struct NsqParams {
  const uint4 *rowPlanes;
  long long rowNpad, row0, nrows;
  // pg_mm.h, eps fill pass (pg_eps_fill_rows): the i-th row of the launch is row rowList[i] (relative to
  // row0) and its matches go straight to the CSR at fillIndptr[rowList[i]] (slotIdx/slotW = indices/weights)
  const long long *rowList, *fillIndptr;
  const uint4 *colPlanes;
  long long colNpad, ncols;
  unsigned long long *stats;   // debug builds (-DPG_MM_STATS): event counters of pg_mm_kernel, else unused
  const uint4 *colFold, *rowFold;   // fold sections (two uint4 arrays of npad: planes 0..3, 4..7 of the plane folds)
  const uint4 *colSig;  // signature section of the column operand: MFMA B fragments, 1 KiB per 32 columns (pg_mm.h)
  int rowsPerWave, rowsPerPass;
  long long mmTailFrom;   // pg_mm.h: waves from this index on take mmTailRows rows each (the last, partly filled round of the grid)
  int mmTailRows;
  long long mmPasses, mmGridWaves;   // pg_mm.h: passes in all; waves in the grid (wave w starts with pass w, then takes
  unsigned *mmPassCounter;           // mmGridWaves + atomicAdd(counter) until none is left); the counter starts at 0
  // pg_mm.h, short-list kNN instances: column pieces.  mmPieces > 1: pass mmPieceFrom + b * mmPieces + piece sweeps only its
  // piece of the super-tiles for row block mmPieceFrom + b and leaves its list (k + 1 keys) in mmPartial[(row' * mmPieces + piece) * (k + 1)],
  // row' counted from the first row in pieces;
  // pg_knn_merge_kernel makes the rows' results of them.  (The rows one full round of waves cannot hold, pg_api.hip.)
  // A piece never gives up the optimistic cap (no checkpoints, no second phase): its list is exact for every column
  // below the cap, so a merged list whose (k+1)-th distance ends below the cap is exact; the merge hands the rows
  // where it does not to pg_knn_rows_kernel (mmEvict below).
  // pg_mm.h kNN: rows that lose their optimistic cap at a checkpoint are EVICTED when they are few in their pass (an open bound
  // in a pass of 63 settled rows turns every remaining tile of the pass into dense-form work: 0.1 % of unrelated sequences among
  // cfg3's took the launch from 1.7 to 28 ms): their index goes to mmEvictRows[atomicAdd(mmEvict, n) ...], their bound to 0,
  // their results come from pg_knn_rows_kernel (pg_nsq.h) behind the launch.  Null: the second phase inside the pass, as before.
  u32 *mmEvict;                     // the counter (zeroed with the pass counters)
  u32 *mmEvictRows;
  int mmPieces;
  long long mmPieceFrom;            // passes from this index on are column pieces (the ones before: plain passes of rowsPerWave rows)
  u32 *mmPartial;                   // ... of the rows from mmPieceFrom * rowsPerWave on
  int mmDenseL1, mmDenseL2, mmDirectRun;   // pg_mm.h: density rules of the filter hierarchy
  // data-driven choice between engines / paths WITHOUT a host round trip (pg_api.hip: probe): when `gate` is not
  // NULL the kernel runs only if bit *gate of gateMask is set (a device word the probe's decision kernel wrote on the same
  // stream); the alternatives are all launched, all but one leave at once
  const u32 *gate;
  u32 gateMask;
  int filter;   // 1 = plane-0 lower-bound filter allowed (adaptive per tile), 0 = always direct
  u32 knnGuess; // kNN: optimistic cap on the stage-1 bound until a row's list is full (0 = off), see pg_nsq.h
  // eps
  u32 lo, span, cap;
  u32 hi1;      // lo + span + 1 (saturating): a pair whose lower bound reaches it cannot match
  int *slotIdx;
  unsigned char *slotW;
  u32 *counts;
  u32 *countsLo;  // EPS_SYM: per row, matches found from the other side (column < row), filled with atomics
  int epsOrdered; // pg_mm.h eps: 1 = a row's matches reach its slot in ascending column order (PG_EPS_ORDERED=1, A/B runs); 0: by tile
  int *slotAux;   // EPS_SYM, optional: for a front entry the back position of its mirror entry in the other row's slot
  // knn: lanes [knnFirst, knnFirst + k) of the sorted 64-key list are written; keys <= floorKeys[row]
  // are ignored (continuation rounds for k > 63); lastKeys[row] receives the last written key
  int k, knnFirst;
  const u32 *floorKeys;
  u32 *lastKeys;
  int *knnIdx;
  unsigned char *knnDist;
};

// Probe of the data in front of an all-pairs launch (pg_api.hip): exact distances of `nsample` evenly spaced rows of the
// launch against all columns, counted per sample row: columns nearer than `near` and columns inside the eps interval.
#define PG_PROBE_STRIDE 8   // the probe looks at every 8th tile of 64 columns
struct ProbeParams {
  const uint4 *rowPlanes, *colPlanes;
  long long rowNpad, colNpad, row0, nrows, ncols;
  int nsample, wavesPerRow;
  u32 near, lo, span;
  u32 *counts;   // [nsample * wavesPerRow][2]: per wave its near count and its eps count
};

struct DenseParams {
  const uint4 *xPlanes;   // columns of the output (N)
  long long xNpad, n;
  const uint4 *yPlanes;   // rows of the output (M)
  long long yNpad, m;
  void *out;
  long long ldo;
  int outBytes;
  int accumulate;   // 1: out += distance (sequences longer than one record are summed segment by segment)
};

// pg_knn_rows_kernel: exact kNN of single rows (the rows the MFMA engine evicted), one wave per row
struct KnnRowsParams {
  const uint4 *rowPlanes, *colPlanes;
  long long rowNpad, colNpad, ncols;
  const u32 *count, *rows;   // *count rows, rows[i] = row index (as in rowPlanes)
  long long baseRow;         // output row of row index r: r - baseRow
  int k;
  int *knnIdx;
  unsigned char *knnDist;
  const u32 *gate;
  u32 gateMask;
};

struct CompactParams {
  NsqParams e;
  int skipOverflow;   // 1: rows beyond their slot are left to pg_eps_fill_rows instead of being recomputed here
  const long long *indptr;
  int *indices;
  unsigned char *weights;
};

void pg_set_error(const char *msg);   // thread-local message behind pg_last_error() (pg_api.hip)

// per-(G,B) launchers (pg_nsq_inst.hip is compiled once per group count G = 1..4)
#define PG_DECL_G(G)                                                                          \
  int pg_launch_nsq_g##G(int mode, int bits, const NsqParams &p, int grid, hipStream_t s);   \
  int pg_occ_nsq_g##G(int mode, int bits); /* resident workgroups per CU of that instance */ \
  int pg_launch_mm_g##G(int mode, int bits, const NsqParams &p, int grid, hipStream_t s); /* MFMA stage 1 (pg_mm.h) */ \
  int pg_launch_dense_g##G(int bits, const DenseParams &p, hipStream_t s);                    \
  int pg_launch_compact_g##G(int bits, const CompactParams &p, hipStream_t s);                \
  int pg_launch_probe_g##G(int bits, const ProbeParams &p, hipStream_t s);                    \
  int pg_launch_knn_rows_g##G(int bits, const KnnRowsParams &p, int grid, hipStream_t s);
PG_DECL_G(1) PG_DECL_G(2) PG_DECL_G(3) PG_DECL_G(4) PG_DECL_G(5) PG_DECL_G(6) PG_DECL_G(7) PG_DECL_G(8)
int pg_launch_nsq_bag(const NsqParams &p, int grid, hipStream_t s);   // pg_lev.hip
int pg_launch_nsq_bag_sym(const NsqParams &p, int grid, hipStream_t s);
int pg_occ_nsq_bag();
int pg_launch_lev_profile(const unsigned char *tok, long long n, int l, long long ld, u32 *prof, long long npad,
                          int *lens, u32 *flags, hipStream_t s);
int pg_launch_lev_select(const unsigned char *tok, long long n, int l, long long ld, const uint4 *planes,
                         long long npad, const int *lens, long long row0,
                         long long nrows, int band, int k, u32 cap, const int *slotIdx, unsigned char *slotW, const int *slotAux, const u32 *counts, const u32 *countsLo,
                         int *knnIdx, unsigned char *knnDist, hipStream_t s);
