// The all-pairs engine, second generation: stage 1 (signature lower bound) on the MATRIX cores.
//
// Replaces the hot loop of Prograph.build_graph (prograph/prograph.py:731-739 and :756-762 of the
// reference): distance(X, batch) -> mask/where or sort -> gather.  Same contract, same per-row
// ownership and output order as pg_nsq.h (one wave owns a pass of rows and sweeps ALL columns in
// ascending order, so eps matches come out in `torch.where` order and kNN lists need no merge);
// what changed is how the ~97 % of pairs that cannot match are rejected.
//
// Stage 1 as an MFMA.  The filter signature of a sequence (its plane-0 bits, 64 positions XOR-folded to
// 54: pg_sig54 in pg_common.h) gives   lb(i,j) = popcount(s_i ^ s_j) <= d(i,j).   That popcount is bilinear:
//     lb = pa_i - sum_k a'_ik * b_jk        a' = +1 / -1 per signature bit of the row,
//                                           b  =  1 /  0 per signature bit of the column,
// so with  A_ik = -a'_ik  (k < 54),  sum_{k >= 54} A_ik = pa_i - bound_i,  B_kj = b_jk,  B_kj = 1 (k >= 54)
//     D = A x B = lb(i,j) - bound_i,        "may be below the row's bound"  <=>  D < 0.
// The instruction is v_mfma_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands: K = 64 elements in the SAME 16
// bytes per lane and the SAME cycles as the int8 32x32x32 form the first version used (31 bits + bias byte;
// tools/ubench/mfma_fp4.hip checks operand layout and exactness: +-1, 0 and the bias values +-{1,2,3,4,6}
// are FP4 numbers, sums of 64 of them are exact in f32, D = 0 comes out as +0).  54 bits instead of 31 cost
// nothing and make unrelated pairs invisible up to bounds of ~12 (one in 3e6 at 10; the 31-bit form let one
// in 2e3 pass at 7): cfg3 3.73 -> 3.55 ms, a 125k x 1M slice 9.3 -> 6.1 ms, and the optimistic kNN cap could
// go from 7 to 10.
// One MFMA evaluates 32 rows x 32 columns; the 16 result registers are OR-ed (8 v_or3) and ONE sign test +
// scalar branch per four of them decides whether the super-tile holds any candidate: ~0.56 VALU instructions
// per 64 pairs instead of ~2.9 (pg_nsq.h), with the multiply-adds on a pipe the VALU does not compete for.
// Measured on MI355X for the int8 form (tools/ubench/mfma_s1.hip, N = 200k full sweep): 1.45 ms against
// 3.6 ms for xor + bcnt; insensitive to occupancy (2..8 waves per SIMD) and to how many row blocks share a
// column fragment, i.e. not bound by the 1 KiB-per-MFMA fragment stream from L2.
//   * the row operand A (4 VGPRs) is built once per pass and stays in registers; a row's bias is ten nibbles
//     of it (lane 32+row: top byte of A[2] and A[3]; pg_bias_nibbles).  When a row's distance threshold moves
//     (kNN) the operand is only marked stale and re-encoded for all rows before the next super-tile: a stale
//     bias is looser, never wrong;
//   * the column operand comes from the signature section of the plane buffer (written by pg_pack_planes):
//     per 32 columns one 1 KiB block in MFMA fragment order, so a B operand is a single coalesced
//     global_load_dwordx4; a ring of four tiles is in flight;
//   * candidates (D < 0) are queued as (row, column) in LDS and evaluated exactly 64 at a time, one per lane,
//     with gathered records - as in pg_nsq.h, but now for every candidate (the column records are no longer
//     in registers);
//   * DENSE data (mutant libraries, one big cluster: most pairs pass the plane-0 bound) leaves the MFMA form
//     per run of super-tiles for the folded form (plane folds of all bit planes, B + 1 instructions per 64
//     pairs, hits into the same candidate queue) or, where even that is not selective, the exact form (every
//     distance in place, pg_nsq.h's direct form); see "dense forms" below.
// kNN keeps the optimistic cap / checkpoint / second-phase scheme of pg_nsq.h (exactness argument
// there and in DESIGN.md §4.1); only the representation of the bound changed.
#pragma once
#include "pg_common.h"

typedef int pg_v4i __attribute__((ext_vector_type(4)));
typedef int pg_v16i __attribute__((ext_vector_type(16)));

#define PG_MM_RB 32          // rows per pass = M of the MFMA tile
#define PG_MM_QCAP 128       // candidate queue entries per wave: < 64 before a push, <= 64 per push
#define PG_MM_ST 128         // columns per super-tile: 4 MFMA tiles = one direct-form tile (C = 2)
// defaults of the density rules (NsqParams carries them: PG_MM_L1 / PG_MM_RUN override for experiments)
#define PG_MM_DENSE_L1 56    // of 256 lane slots per super-tile with a candidate: leave the MFMA form (tools/dense_knobs.py: 40..64 flat, 96 costs dense data 25 %)
#define PG_MM_DENSE_L2 48    // (unused since the MFMA level 2 was dropped; NsqParams still carries the field)
#define PG_MM_GROUP_ROWS 4    // folded form: rows per group of straight-line code (their folds: 20 SGPRs in flight)
#define PG_MM_DIRECT_RUN 8   // super-tiles of dense form before the MFMA filter is probed again
#define PG_MM_PRIO_STEPS 16  // R = 2: steps of the progress-driven issue priority along a sweep
#define PG_MM_EVICT_MAX 48   // kNN: up to this many rows of a pass that lose their cap at a checkpoint are evicted (NsqParams::mmEvict); more (nearly the
                             // whole pass: data without neighbours inside the cap, e.g. clusters smaller than k + 1): the second phase in the pass
#ifndef PG_EXP_SAMETILE
#define PG_EXP_SAMETILE 0   // experiment builds: 1 = every fragment load reads the same super-tile (L1 hits; wrong results)
#endif
#define PG_NSTAT 24          // debug builds (-DPG_MM_STATS): event / cycle counters per launch, then (start, duration) per pass

static_assert(PG_MM_QCAP >= 63 + 64, "a register push adds up to 64 candidates to a queue holding up to 63");
static_assert(PG_QCAP >= 63 + 4 * 2 * PG_PUSH_MAX, "pg_nsq.h kNN queue: a group pushes up to 4 rows x 2 columns x PG_PUSH_MAX");
static_assert(PG_QCAP_EPS >= 63 + 4 * 2 * 64, "pg_nsq.h eps queue: a group pushes up to 4 rows x 2 x 64 lanes");

typedef int pg_v8i __attribute__((ext_vector_type(8)));
typedef float pg_v16f __attribute__((ext_vector_type(16)));

// D = A x B over K = 64 FP4 elements (only the first four registers of each operand are read); the result
// comes back as its bit patterns: the filter looks at signs only, and sums of these small integers are exact
__device__ __forceinline__ pg_v16i pg_mfma_fp4(const pg_v4i &a, const pg_v4i &b) {
  const pg_v8i a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
  const pg_v16f zero = {0};
  const pg_v16f d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, zero, 4, 4, 0, 0, 0, 0);
  return __builtin_bit_cast(pg_v16i, d);
}

// a | b | c as v_bitop3_b32 (truth table 0xFE): the 2-cycle issue class of xor / or / bitop3, where v_or3_b32 is in
// the 4-cycle class (profiles/r01_valu_issue_microbench.txt: k_bitop3 2.4-2.9 cycles, k_or3 4.2-4.4)
#ifndef PG_OR3_PLAIN
__device__ __forceinline__ int pg_or3(int a, int b, int c) {
  return (int)__builtin_amdgcn_bitop3_b32((u32)a, (u32)b, (u32)c, 0xFE);
}
#else
__device__ __forceinline__ int pg_or3(int a, int b, int c) { return a | b | c; }
#endif
// median of three unsigned values (sorted insertion: new[i] = med3(old[i-1], key, old[i]))
__device__ __forceinline__ u32 pg_med3(u32 a, u32 b, u32 c) {
  u32 r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ int pg_or16(const pg_v16i &d) {
  int a = pg_or3(d[0], d[1], d[2]);
  int b = pg_or3(d[3], d[4], d[5]);
  int c = pg_or3(d[6], d[7], d[8]);
  int e = pg_or3(d[9], d[10], d[11]);
  int f = pg_or3(d[12], d[13], d[14]);
  a = pg_or3(a, b, c);
  e = pg_or3(e, f, d[15]);
  return a | e;
}

// Records of up to three chunks (L <= 64 with 5 bit planes): held at 4 waves per SIMD (amdgpu_waves_per_eu(4, 8) =
// 128 VGPRs).  What the compiler reports per instance (-Rpass-analysis=kernel-resource-usage) is committed in
// profiles/r03_kernel_resource_usage.txt: the 64-row kNN instance takes all 128 and spills 5 VGPRs (24 B per lane of
// scratch) in the slow paths; tools/check_ring_asm.py checks on the generated ISA that nothing - no spill, no copy - touches
// a fragment register between its load and the wait in front of the MFMA.
// KL (kNN only): entries of a row's list in LDS.  64 = the list lives across the wave's lanes (lane j = j-th smallest
// key; insertion = one DPP shift, one candidate at a time).  KL < 64 (k + 1 <= KL, PG_MM_KL): the list is KL consecutive
// dwords, RIGHT aligned (the (k+1)-th smallest key, i.e. the row's threshold, is always entry KL-1; unused entries in
// front hold 0), and a flush inserts up to 64 candidates of different rows AT ONCE, one per lane: the winner lane of a row
// reads its list (KL/4 ds_read_b128), new[i] = med3(old[i-1], key, old[i]), writes it back.
// R: row blocks of 32 per pass (1 or 2).  The hot loop is bound by the vector-memory return path (64 B per clock and
// CU: every MFMA needs a fresh 1 KiB fragment; measured ~70 % of that rate, with the loads hitting L1 or L2 alike), so
// R = 2 - two row operands, every fragment feeds two MFMAs - halves that traffic per pair.  Per-row state is lane
// indexed, a wave has 64 lanes: R <= 2.  Short lists only (LDS).
template <class M, int MODE, int KL = 64, int R = 1>
// (records of four chunks - byte alphabets at L <= 64 - in the 32-row short-list kNN instance: held at three waves per SIMD,
//  168 VGPRs; left alone it takes 173 since the eviction code: two waves, N = 200k 2.05 -> 2.5 ms)
__global__ __launch_bounds__(PG_WG_THREADS) __attribute__((amdgpu_waves_per_eu(
    M::Q <= 3 ? 4 : ((M::Q == 4 && MODE == PG_MODE_KNN && KL < 64 && R == 1) ? 3 : 1), 8))) void pg_mm_kernel(const NsqParams p) {
  constexpr int Q = M::Q;
  constexpr bool kPar = MODE == PG_MODE_KNN && KL < 64;    // lane-per-candidate insertion
  static_assert(KL == 64 || (KL % 4 == 0 && KL >= 8 && KL <= 32), "list entries: 64, or a multiple of 4 in 8..32");
  static_assert(R == 1 || (R == 2 && kPar), "two row blocks per pass: the short-list kNN instance only");
  constexpr int C = Q <= 4 ? 2 : 1;                        // direct form: columns per lane
  constexpr bool kEps = MODE != PG_MODE_KNN;
  constexpr bool kSym = MODE == PG_MODE_EPS_SYM;
  constexpr int RB = PG_MM_RB * R;
  constexpr int LROWS = MODE == PG_MODE_KNN ? RB : 1;
  __shared__ uint4 rowbuf[PG_WG_WAVES][RB][Q];
  __shared__ __attribute__((aligned(16))) u32 lstbuf[PG_WG_WAVES][LROWS][KL];   // kNN: per row the sorted keys (see KL above)
  __shared__ u32 cqbuf[PG_WG_WAVES][PG_MM_QCAP];           // deferred candidates: row << SH | column
  __shared__ u32 claimbuf[PG_WG_WAVES][kPar ? RB : 1];     // kPar: per row the lane that inserts in this turn
  const int lane = threadIdx.x & 63;
  const u32 ulane = (u32)lane;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long gw = (long long)blockIdx.x * PG_WG_WAVES + wv;
  if (gw >= p.mmPasses) return;   // whole wave leaves; no workgroup barrier is used below
  if (p.gate && !((p.gateMask >> __builtin_nontemporal_load(p.gate)) & 1u)) return;   // the probe chose another engine / path
  const uint4 *__restrict__ colp = p.colPlanes;
  const pg_v4i *__restrict__ colsig = reinterpret_cast<const pg_v4i *>(p.colSig);
  const u32 ncols = (u32)p.ncols;
  const int nst = (int)((p.ncols + PG_MM_ST - 1) / PG_MM_ST);   // super-tiles of 128 columns
  const uint4 *rows = &rowbuf[wv][0][0] + opaque_zero();   // broadcast reads, kept "divergent"
  u32 *cq = &cqbuf[wv][0];
  const u32 bias = kEps ? opaque_vgpr(0u - p.lo) : 0u;     // eps: -lo rides in the popcount accumulator
  // eps entries carry the row in 5 bits above a 27-bit column; wider problems run the direct form
  const bool canFilterK = p.filter != 0 && (!kEps || p.ncols < (1ll << 27));   // (per pass: canFilter below)
  const bool epsOrdered = kEps && (p.fillIndptr != nullptr || p.epsOrdered != 0);   // (see push_signs)
  constexpr int SH = kEps ? 27 : 24;
  // arguments that only cold code needs (staging a pass, storing results, the dense forms) are read from the
  // kernel-argument segment where they are used: held in SGPRs across the sweep they crowd the loop state of
  // the MFMA form out into spill lanes
  typedef const NsqParams __attribute__((address_space(4))) *pg_kargs;
  auto K = [&]() -> const NsqParams __attribute__((address_space(4))) & {
    pg_kargs kp = (pg_kargs)__builtin_amdgcn_kernarg_segment_ptr();   // the kernel's one argument: NsqParams by value
    asm volatile("" : "+s"(kp));                          // (opaque: not the values the prologue already loaded)
    return *kp;
  };

#ifdef PG_MM_STATS
  u32 st[PG_NSTAT] = {0};   // 0 L1 super-tiles, 1 with candidates, 2 exact tiles, 3 dense runs, 4 dense super-tiles, 5 candidates of
                      // folded tiles, 6 candidates queued, 7 flushes, 8 insertions / eps matches, 9 resweep super-tiles, 10 passes,
                      // 11 folded tiles; x64 cycles (s_memtime) of this wave: 12 in flush, 13 in its insertion part, 14 in folded tiles,
                      // 15 in passes, 16 in scan() (the hot loop), 17 in slow_mfma (queueing, its flushes included), 18 scan() calls,
                      // 19 bias refreshes, 20 queueing turns, 21 x64 cycles from a pass's start to its first super-tile
#define PG_ST(i, n) st[i] += (u32)(n)
#define PG_T0(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define PG_T1(i, v) st[i] += (u32)((__builtin_amdgcn_s_memtime() - v) >> 6)
#else
#define PG_ST(i, n)
#define PG_T0(v)
#define PG_T1(i, v)
#endif
  // Passes are handed out dynamically: the grid holds one wave per slot of the chip (pg_api.hip: plan_mm),
  // wave gw starts with pass gw and fetches further ones from a counter.  The hardware dispatcher refills the
  // CUs whose workgroups end first to full occupancy and leaves the others idle for the last round of a longer
  // grid; persistent waves spread that round over all SIMDs, where its waves run faster (fewer per SIMD).
  for (long long pass = gw; pass < p.mmPasses;) {
    // the rows of the pass: rowsPerWave each, mmTailRows from pass mmTailFrom on
    // (column pieces, kPar instances: passes from mmPieceFrom on; uniform passes of rowsPerWave rows)
    const int npieces = kPar && p.mmPieces > 1 && pass >= p.mmPieceFrom ? p.mmPieces : 1;
    const long long rpass = npieces > 1 ? p.mmPieceFrom + (pass - p.mmPieceFrom) / npieces : pass;
    const int piece = npieces > 1 ? (int)((pass - p.mmPieceFrom) % npieces) : 0;
    const bool tailPass = rpass >= p.mmTailFrom;
    const long long prows = tailPass ? p.mmTailRows : p.rowsPerWave;
    const long long pr0 = tailPass ? p.mmTailFrom * p.rowsPerWave + (rpass - p.mmTailFrom) * p.mmTailRows : rpass * p.rowsPerWave;
    // the super-tiles of this pass: all of them, or the piece's share (whole pairs of super-tiles: the folded form's tiles)
    const int npair = (nst + 1) >> 1;
    const int sb = npieces > 1 ? 2 * (int)((long long)npair * piece / npieces) : 0;
    const int se = npieces > 1 ? (2 * (int)((long long)npair * (piece + 1) / npieces) < nst ? 2 * (int)((long long)npair * (piece + 1) / npieces) : nst) : nst;
    const long long left = (pr0 + prows < p.nrows ? pr0 + prows : p.nrows) - pr0;
    PG_ST(10, 1);
    PG_T0(tp0);
#ifdef PG_MM_STATS
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
    const u32 st5_0 = st[5], st2_0 = st[2], st7_0 = st[7];
#endif
    const int nr = __builtin_amdgcn_readfirstlane((int)(left < RB ? left : RB));

    // ---- stage the pass's rows into the wave's LDS region (wave private) ----
    const auto &ka = K();
    for (int e = lane; e < RB * Q; e += 64) {
      const int rr = e % RB, q = e / RB;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (rr < nr) v = ka.rowPlanes[(long long)q * ka.rowNpad + ka.row0 + (ka.rowList ? ka.rowList[pr0 + rr] : pr0 + rr)];
      rowbuf[wv][rr][q] = v;
    }
    // eps: where row `lane` of the pass stores its matches: its slot, or (fill pass) its place in the CSR
    u32 baselo = 0, basehi = 0;
    bool passDense = false;
    if constexpr (kEps) {
      long long b = 0;
      if (lane < nr) b = ka.fillIndptr ? ka.fillIndptr[ka.rowList ? ka.rowList[pr0 + lane] : pr0 + lane] : (pr0 + lane) * (long long)ka.cap;
      baselo = (u32)b;
      basehi = (u32)((unsigned long long)b >> 32);
      // fill pass: the rows' degrees are known (their CSR extents).  A pass whose rows match one column in 32 or
      // more on average is swept in the exact form from the start: no filter can beat "every distance" there
      if (ka.fillIndptr) {
        long long deg = 0;
        if (lane < nr) deg = ka.fillIndptr[(ka.rowList ? ka.rowList[pr0 + lane] : pr0 + lane) + 1] - b;
        u32 dsum = (u32)(deg > 0x3FFFFFF ? 0x3FFFFFF : deg);  // (32 rows x 2^26 fits 32 bits)
        for (int o = 32; o > 0; o >>= 1) dsum += (u32)__shfl_xor((int)dsum, o);
        passDense = (unsigned long long)dsum * 32ull >= (unsigned long long)nr * (unsigned long long)p.ncols;
      }
    }
    const bool canFilter = canFilterK && !passDense;
    auto row_base = [&](int row) -> long long {
      // (readlane returns a signed int: without the u32 casts a low half with bit 31 set sign-extends into the high half)
      const u32 lo32 = (u32)__builtin_amdgcn_readlane((int)baselo, row), hi32 = (u32)__builtin_amdgcn_readlane((int)basehi, row);
      return (long long)(((unsigned long long)hi32 << 32) | (unsigned long long)lo32);
    };
    const u32 G0 = (MODE == PG_MODE_KNN && canFilter) ? p.knnGuess : 0u;
    u64 failed = 0;                                         // kNN: rows that lost their optimistic cap (bit = row)
    u64 evicted = 0;                                        // kNN: rows handed to pg_knn_rows_kernel (NsqParams::mmEvict): bound 0, no results from here
    u32 resweep = 0;                                        // kNN: 1 in phase 1 (early super-tiles again for the failed rows)
    int sredo = 0;                                          // kNN: super-tiles [0, sredo) are swept again for them
    // list geometry: lanes / entries [lfirst, lfirst + k) are written out, entry thrLane is the row's threshold
    const int thrLane = kPar ? KL - 1 : p.knnFirst + p.k - 1;   // last list entry that is still needed
    const int lfirst = kPar ? KL - p.k : p.knnFirst;            // (kPar: the host guarantees knnFirst == 1, k + 1 <= KL)
    const int lzero = kPar ? KL - 1 - p.k : 0;                   // kPar: entries below hold 0 (never displaced: keys are >= 0)
    if constexpr (MODE == PG_MODE_KNN) {
      if constexpr (kPar) {
        for (int e = lane; e < nr * KL; e += 64) (&lstbuf[wv][0][0])[e] = (e % KL) >= lzero ? 0xFFFFFFFFu : 0u;
      } else {
        for (int rr = 0; rr < nr; ++rr) lstbuf[wv][rr][lane] = 0xFFFFFFFFu;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- row operand of the MFMA: lane l holds row l & 31, elements k = 32*(l >> 5) .. +31 as FP4 nibbles:
    // signature bit set -> -1.0 (0xA), clear -> +1.0 (0x2); k = 54..63 (top byte of A0[2] and A0[3] of lanes
    // 32..) = the bias pa - bound as a sum of up to ten elements, so D = lb - bound and its sign is the answer ----
    // (named scalars per row block, not arrays: an array indexed inside a loop the compiler unrolls late ends up in
    //  scratch memory; block 1 is dead code for R == 1)
    u32 pa0 = 0, pa1 = 0;
    pg_v4i A0 = {0, 0, 0, 0}, A1 = {0, 0, 0, 0};            // block b: rows 32b .. 32b+31 of the pass
    auto build_operand = [&](int blk, u32 &pa, pg_v4i &A) {
      uint4 rec[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) rec[q] = rowbuf[wv][32 * blk + (lane & 31)][q];
      u32 even, odd;
      M::fold2(rec, even, odd);
      const unsigned long long sig = pg_sig54(even, odd);
      pa = (u32)__builtin_popcountll(sig);
      const u32 half = lane >> 5 ? (u32)(sig >> 32) : (u32)sig;
      A[0] = (int)(0x22222222u | (pg_nib8(half) << 3));
      A[1] = (int)(0x22222222u | (pg_nib8(half >> 8) << 3));
      A[2] = (int)(0x22222222u | (pg_nib8(half >> 16) << 3));
      A[3] = (int)(0x22222222u | (pg_nib8(half >> 24) << 3));
    };
    build_operand(0, pa0, A0);
    if constexpr (R == 2) build_operand(1, pa1, A1);
    // A row's bound: lanes 32.. of its block's boundv hold it (authoritative) and, as bias nibbles, their operand.
    // A bound beyond pa + 60 passes everything either way (lb <= 54).  A bound that moves (kNN: always down)
    // only marks the operand stale; the next stretch of the MFMA form re-encodes all rows at once - one
    // copy of the encoder, one pass per flush instead of one per insertion; a stale bias is merely looser.
    u32 boundv0 = 0, boundv1 = 0;
    bool stale = false;
    auto encode_bias = [&](u32 pa, u32 bound, pg_v4i &A) {
      const bool up = lane >= 32;
      const unsigned long long nb = pg_bias_nibbles(bound > 255u ? -60 : (int)pa - (int)bound);
      A[2] = up ? (int)(((u32)A[2] & 0x00FFFFFFu) | ((u32)nb << 24)) : A[2];
      A[3] = up ? (int)(u32)(nb >> 8) : A[3];
    };
    auto refresh_bias = [&]() {
      encode_bias(pa0, boundv0, A0);
      if constexpr (R == 2) encode_bias(pa1, boundv1, A1);
      stale = false;
    };
    auto set_bound = [&](int row, u32 bound) {              // row, bound wave uniform
      boundv0 = (row < 32 && lane == 32 + row) ? bound : boundv0;
      if constexpr (R == 2) boundv1 = (row >= 32 && lane == row) ? bound : boundv1;
      stale = true;
    };
    auto set_all_bounds = [&](u32 bv) {                     // bv: lane r < RB holds row r's bound
      boundv0 = (u32)__builtin_amdgcn_ds_bpermute((lane & 31) << 2, (int)bv);
      if constexpr (R == 2) boundv1 = (u32)__builtin_amdgcn_ds_bpermute(((lane & 31) + 32) << 2, (int)bv);
      stale = true;                                         // (may loosen bounds: the refresh comes before the next scan)
    };

    // Per-row state, lane indexed (lane = row in pass), touched with v_readlane / lane selects:
    //   eps: cntv = matches so far;   knn: thrv = current (k+1)-th smallest key, capv = cap on the bound
    u32 cntv = 0u, thrv = 0xFFFFFFFFu;
    u32 capv = G0 ? G0 : 255u;
    u32 floorv = 0u;                                        // kNN continuation rounds (k > 63)
    if constexpr (MODE == PG_MODE_KNN) {
      if (ka.floorKeys && lane < nr) floorv = ka.floorKeys[pr0 + lane];
    }
    set_all_bounds(lane < nr ? (kEps ? p.hi1 : capv) : 0u); // rows past nr: bound 0, nothing passes

    auto publish = [&](int row, u32 thr) {                  // kNN: a row's threshold moved
      const u32 cp = __builtin_amdgcn_readlane(capv, row);
      const u32 b = (thr >> 24) + resweep;                  // phase 1: lb <= distance bound may still win a tie
      set_bound(row, b < cp ? b : cp);
    };
    // kNN: the bounds of ALL rows from the row-indexed state (after a flush that moved several thresholds at once):
    // min(threshold distance [+1 in phase 1], cap); 0 for rows past nr and, in phase 1, for the frozen rows
    auto republish_all = [&]() {
      const bool live = lane < nr && (!resweep || ((failed >> lane) & 1ull)) && !((evicted >> lane) & 1ull);
      const u32 b = (thrv >> 24) + resweep;                 // open lists read 255
      set_all_bounds(live ? (b < capv ? b : capv) : 0u);
    };
    // EPS_SYM: a match (row, col), col > row, also belongs to row `col` (owned by another wave): its
    // entry goes to the BACK of that row's slot through an atomic counter (pg_compact_kernel sorts it)
    auto emit_lower = [&](bool on, u32 rowg, u32 col, u32 w) -> u32 {
      u32 pos = 0xFFFFFFFFu;
      if constexpr (kSym) {
        if (on) {
          pos = atomicAdd(&p.countsLo[col], 1u);
          if (pos < p.cap) {
            const long long o = (long long)col * p.cap + (p.cap - 1u - pos);
            p.slotIdx[o] = (int)rowg;
            p.slotW[o] = (unsigned char)w;
          }
        }
      }
      return pos;
    };
    // in-place epilogue of the direct form (exact distances of a whole 64-column sub-tile)
    auto epilogue = [&](u32 d, u32 col, int rr) {
      if constexpr (kEps) {
        bool h2 = (d <= p.span) && (col < ncols);
        if constexpr (kSym) h2 = h2 && col > (u32)(pr0 + rr);
        const u64 m2 = __builtin_amdgcn_ballot_w64(h2);
        if (m2) {
          const u32 cnt = __builtin_amdgcn_readlane(cntv, rr);
          const u32 pos = cnt + mask_rank(m2);
          const u32 posb = emit_lower(h2, (u32)(pr0 + rr), col, d + p.lo);
          if (h2 && pos < p.cap) {
            const long long o = row_base(rr) + pos;
            p.slotIdx[o] = (int)col;
            p.slotW[o] = (unsigned char)(d + p.lo);
            if constexpr (kSym) {
              if (p.slotAux) p.slotAux[o] = (int)posb;
            }
          }
          cntv = (lane == rr) ? cnt + (u32)__popcll(m2) : cntv;
        }
      } else {
        u32 thr = __builtin_amdgcn_readlane(thrv, rr);
        const u32 key = (d << 24) | col;
        bool cand = (key < thr) && (col < ncols);
        if (p.floorKeys) cand = cand && key > __builtin_amdgcn_readlane(floorv, rr);
        u64 m = __builtin_amdgcn_ballot_w64(cand);
        if (m) {
          // (kPar: the list is entries 0..KL-1 = lanes 0..KL-1 here; the zeros in front of it stay where they are)
          u32 lst = (!kPar || lane < KL) ? lstbuf[wv][rr][kPar ? (lane < KL ? lane : 0) : lane] : 0xFFFFFFFFu;
          do {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const u32 x = __builtin_amdgcn_readlane(key, j);
            if (x < thr && !(resweep && __builtin_amdgcn_ballot_w64(lst == x && lane >= lzero))) {
              const u32 prev = wave_shr1(lst, 0u);
              lst = (lst <= x) ? lst : (prev > x ? prev : x);
              thr = __builtin_amdgcn_readlane(lst, thrLane);
            }
          } while (m);
          if (!kPar || lane < KL) lstbuf[wv][rr][kPar ? (lane < KL ? lane : 0) : lane] = lst;
          thrv = (lane == rr) ? thr : thrv;
          publish(rr, thr);
        }
      }
    };

    // ---- deferred candidates: 64 at a time, one per lane, records gathered (row: LDS, column: L2) ----
    int qn = 0;                                             // queue fill, wave uniform
    auto flush = [&]() {
      const int nbat = qn < 64 ? qn : 64;
      PG_ST(7, 1);
      PG_T0(tf0);
      const u32 e = cq[lane];
      if constexpr (kPar) {
        // every candidate of the batch at once: exact distance, then insertion by the candidate's own lane.  Two
        // candidates of one row take turns (the row's claim word says whose turn it is; any order gives the same
        // list: keys are totally ordered).
        const u32 col = e & 0x00FFFFFFu;
        const u32 erow = (e >> 24) & (u32)(RB - 1);
        const bool act = lane < nbat && col < ncols;
        u32 key = 0xFFFFFFFFu;
        if (act) {
          uint4 cr[Q], rw[Q];                               // all gathers in flight at once
#pragma unroll
          for (int q = 0; q < Q; ++q) cr[q] = colp[(long long)q * p.colNpad + col];
#pragma unroll
          for (int q = 0; q < Q; ++q) rw[q] = rowbuf[wv][erow][q];
          key = (M::dist(rw, cr, 0u) << 24) | col;
        }
        bool pend = act && key < lstbuf[wv][erow][KL - 1];  // (a row's threshold is the last entry of its list)
        PG_T0(ti0);
        while (__builtin_amdgcn_ballot_w64(pend)) {
          // (LDS operations of a wave are processed in order: the last writer of a row's word wins.  The fences keep
          // the compiler from forwarding a lane's own store to its load; a volatile pointer would lose the LDS address
          // space: flat accesses and a full vmcnt wait per turn)
          if (pend) claimbuf[wv][erow] = (u32)lane;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const bool win = pend && claimbuf[wv][erow] == (u32)lane;
          [[maybe_unused]] bool ins = false;
          if (win) {
            uint4 *lp = reinterpret_cast<uint4 *>(&lstbuf[wv][erow][0]);
            u32 o[KL];
#pragma unroll
            for (int i = 0; i < KL / 4; ++i) {
              const uint4 v = lp[i];
              o[4 * i] = v.x; o[4 * i + 1] = v.y; o[4 * i + 2] = v.z; o[4 * i + 3] = v.w;
            }
            bool ok = key < o[KL - 1];                      // (the row's threshold may have moved on since the first look)
            if (resweep) {                                  // phase 1 meets columns again: no duplicates
              bool dup = false;
#pragma unroll
              for (int i = 0; i < KL; ++i) dup = dup || (o[i] == key && i >= lzero);
              ok = ok && !dup;
            }
            if (ok) {
#pragma unroll
              for (int i = KL - 1; i > 0; --i) o[i] = pg_med3(o[i - 1], key, o[i]);
              o[0] = o[0] < key ? o[0] : key;
#pragma unroll
              for (int i = 0; i < KL / 4; ++i) lp[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
              ins = true;
            }
          }
          PG_ST(8, __popcll(__builtin_amdgcn_ballot_w64(ins)));
          pend = pend && !win;
        }
        // the thresholds back into the row-indexed vector, straight from the lists; bounds follow where a distance moved
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const u32 nthr = lane < nr ? lstbuf[wv][lane < RB ? lane : 0][KL - 1] : thrv;
        const bool moved = (nthr >> 24) != (thrv >> 24);
        thrv = nthr;
        if (__builtin_amdgcn_ballot_w64(moved)) republish_all();
        PG_T1(13, ti0);
      } else if constexpr (MODE == PG_MODE_KNN) {
        const u32 col = e & 0x00FFFFFFu;
        const u32 erow = (e >> 24) & 31u;
        const bool act = lane < nbat && col < ncols;
        u32 key = 0xFFFFFFFFu, thr = 0u;
        if (act) {
          uint4 cr[Q], rw[Q];                               // all gathers in flight at once
#pragma unroll
          for (int q = 0; q < Q; ++q) cr[q] = colp[(long long)q * p.colNpad + col];
#pragma unroll
          for (int q = 0; q < Q; ++q) rw[q] = rowbuf[wv][erow][q];
          const u32 d = M::dist(rw, cr, 0u);
          key = (d << 24) | col;
          thr = lstbuf[wv][erow][thrLane];
        }
        const bool cand = act && key < thr;
        u64 m = __builtin_amdgcn_ballot_w64(cand);
        PG_T0(ti0);
        while (m) {
          const int j = __builtin_ctzll(m);
          m &= m - 1;
          const int row = (int)__builtin_amdgcn_readlane(erow, j);
          const u32 x = __builtin_amdgcn_readlane(key, j);
          if (p.floorKeys && x <= __builtin_amdgcn_readlane(floorv, row)) continue;   // continuation round
          const u32 othr = (u32)__builtin_amdgcn_readlane((int)thrv, row);
          if (x >= othr) continue;                          // the row's threshold moved on meanwhile (no LDS trip)
          u32 lst = lstbuf[wv][row][lane];
          if (!(resweep && __builtin_amdgcn_ballot_w64(lst == x))) {
            const u32 prev = wave_shr1(lst, 0u);
            lst = (lst <= x) ? lst : (prev > x ? prev : x);
            lstbuf[wv][row][lane] = lst;
            PG_ST(8, 1);
            const u32 nthr = __builtin_amdgcn_readlane(lst, thrLane);
            thrv = (lane == row) ? nthr : thrv;
            if ((nthr >> 24) != (othr >> 24)) publish(row, nthr);   // the filter's bound only knows the distance
          }
        }
        PG_T1(13, ti0);
      } else {
        const u32 col = e & 0x07FFFFFFu;
        const u32 erow = e >> 27;
        const bool act = lane < nbat && col < ncols;
        u32 d = 0xFFFFFFFFu;
        if (act) {
          uint4 cr[Q], rw[Q];
#pragma unroll
          for (int q = 0; q < Q; ++q) cr[q] = colp[(long long)q * p.colNpad + col];
#pragma unroll
          for (int q = 0; q < Q; ++q) rw[q] = rowbuf[wv][erow][q];
          d = M::dist(rw, cr, bias);
        }
        bool match = act && d <= p.span;
        if constexpr (kSym) match = match && col > (u32)pr0 + erow;
        const u32 posb = emit_lower(match, (u32)pr0 + erow, col, d + p.lo);
        u64 m = __builtin_amdgcn_ballot_w64(match);
        while (m) {                                          // one turn per row present in the batch
          const u32 row = __builtin_amdgcn_readlane(erow, __builtin_ctzll(m));
          const bool mine = match && erow == row;
          const u64 same = __builtin_amdgcn_ballot_w64(mine);
          const u32 cnt = __builtin_amdgcn_readlane(cntv, (int)row);
          const u32 pos = cnt + mask_rank(same);
          if (mine && pos < p.cap) {
            const long long o = row_base((int)row) + pos;
            p.slotIdx[o] = (int)col;
            p.slotW[o] = (unsigned char)(d + p.lo);
            if constexpr (kSym) {
              if (p.slotAux) p.slotAux[o] = (int)posb;
            }
          }
          cntv = (lane == (int)row) ? cnt + (u32)__popcll(same) : cntv;
          PG_ST(8, __popcll(same));
          m &= ~same;
        }
      }
      if (qn > 64) {                                         // keep the tail (at most 63 entries), in order
        const u32 tail = cq[64 + lane];
        cq[lane] = tail;
      }
      qn -= nbat;
      PG_T1(12, tf0);
    };

    // ---- filtered form (the MFMA form) ----
    // C/D layout of the 32x32 MFMA: register r, lane l -> row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), column l & 31.
    // The hot loop (scan) only looks at the OR of a tile's 16 result registers.  At a super-tile with candidates the
    // flagged tiles are evaluated AGAIN from the ring and the SIGNS of the 16 registers are shifted into one word per
    // lane (bit r = register r holds a candidate); the queueing code works from these four words - nothing else of
    // the MFMA form is live across it.
    auto signs16 = [&](const pg_v16i &d) -> u32 {
      u32 a = 0;
#pragma unroll
      for (int r = 15; r >= 0; --r) a = __builtin_amdgcn_alignbit(a, (u32)d[r], 31);   // (a << 1) | sign
      return a;
    };
    // eps slots (pg_eps_slots[_sym]) take the lane-parallel form as well: a row's matches then reach its slot in ascending
    // order of their 32-column TILE but in any order within one - pg_compact_kernel puts every entry in its place by a
    // rank among its 31 neighbours on either side.  The fill pass (pg_eps_fill_rows: straight into the CSR, nothing
    // behind it that could sort) keeps the ordered form below: `epsOrdered`.
    // kNN (the order of a row's candidates does not matter to its list): LANE PARALLEL queueing - per turn every
    // lane that has any candidate queues its lowest one.  The cluster mates of 32 consecutive rows sit on a tile's
    // diagonal - 32 different lanes - so a tile takes one or two turns: 16 (sign words) + ~12 per turn vector
    // instructions where round 2's register-by-register form issued ~250, most of that kernel's instructions.
    // eps (slot positions follow queue order = ascending columns per row): register by register, but only the
    // registers that hold a candidate in some lane (wave OR of the sign words), lanes in order.
    // One copy of the loop (and of flush) for the four tiles of a super-tile.
    // scan() -> push_signs(): one sign word per tile of the super-tile - bit 16 * block + r = result register r of that
    // row block holds a candidate (named scalars: an array selected by a run-time index ends up in scratch memory)
    u32 am0 = 0, am1 = 0, am2 = 0, am3 = 0;
    u32 amMask = 0;                                         // bit t = tile t is flagged
    auto push_signs = [&](int S) {
      u32 tm = amMask;
      while (tm) {
        const int t = __builtin_ctz(tm);
        tm &= tm - 1u;
        u32 a = t == 1 ? am1 : (t == 2 ? am2 : (t == 3 ? am3 : am0));
        const u32 ebase = ((u32)(4 * (lane >> 5)) << SH) | (u32)((S * 4 + t) * 32 + (lane & 31));
        if (MODE == PG_MODE_KNN || !epsOrdered) {
          u64 mb = __builtin_amdgcn_ballot_w64(a != 0u);
          while (mb) {
            // bit b: row block b >> 4, register b & 15; the register's row in its block: (r & 3) + 8 * (r >> 2) = r + (r & 12)
            const u32 b = (u32)__builtin_ctz(a | 0x80000000u);
            if (a != 0u) cq[qn + mask_rank(mb)] = ebase + ((b + (b & 12u) + (b & 16u)) << SH);
            const int n = (int)__popcll(mb);
            PG_ST(6, n);
            PG_ST(20, 1);
            qn += n;
            a &= a - 1u;
            if (qn >= 64) flush();                          // (at most 63 + 64 entries before it)
            mb = __builtin_amdgcn_ballot_w64(a != 0u);
          }
        } else {
          u32 regs = (u32)__builtin_amdgcn_readlane((int)wave_or_to63(a), 63);
          while (regs) {
            const u32 b = (u32)__builtin_ctz(regs);
            regs &= regs - 1u;
            const bool hit = (a >> b) & 1u;
            const u64 mb = __builtin_amdgcn_ballot_w64(hit);
            if (hit) cq[qn + mask_rank(mb)] = ebase + ((b + (b & 12u)) << SH);
            const int n = (int)__popcll(mb);
            PG_ST(6, n);
            PG_ST(20, 1);
            qn += n;
            while (qn >= 64) flush();                       // (at most 63 + 64 entries before it)
          }
        }
      }
    };
    // The fragment ring: the column operands of one super-tile (4 x 1 KiB per wave).  It lives only inside scan().
    // Its loads are inline assembly - destination = the ring register itself, so the hot loop has no copies at its
    // back edge; address = the section's base (a kernel argument: always scalar) + a 32-bit vector offset (super-tile
    // * 4096 + 16 * lane; the section is below 4 GiB: ncols < 2^27) + immediate - which the compiler does not count:
    // the wait at the loop's head waits for them (the extra outstanding loads only make the compiler's own waits
    // stricter).  RULE: between an assembly load and that wait there is no code in which the compiler could move or
    // spill a ring register (it would read it before the data has landed): the loads are the last thing before it.
    // HAZARD: "VALU writes an SGPR, a memory instruction reads it" needs five wait states, and nobody inserts them
    // in front of inline assembly - while the compiler does hand over scalar operands it has just produced with
    // v_readfirstlane (where it takes a loop for divergent) or v_readlane (an SGPR spill reload): that was a
    // wrong-address fault.  So the base is copied by an s_mov INSIDE the statement (a scalar write has no such
    // hazard), all four loads are ONE statement, and the moving part of the address is the vector offset.
#ifndef PG_EXP_HALFLOAD
#define PG_EXP_LD23 "global_load_dwordx4 %2, %5, %4 offset:2048\n\tglobal_load_dwordx4 %3, %5, %4 offset:3072"
#define PG_EXP_VMCNT "4"
#else   // experiment: only two of the four fragment loads (half the vector-memory traffic; wrong results, timing only)
#define PG_EXP_LD23 "s_nop 0"
#define PG_EXP_VMCNT "2"
#endif
#define PG_RING_LOAD(c, x0, x1, x2, x3, voff)                                                    \
  asm volatile("s_mov_b64 %4, %6\n\t"                                                            \
               "global_load_dwordx4 %0, %5, %4 offset:0\n\t"                                     \
               "global_load_dwordx4 %1, %5, %4 offset:1024\n\t"                                  \
               PG_EXP_LD23                                                                       \
               : c(x0), c(x1), c(x2), c(x3), "=&s"(ringBase) : "v"(voff), "s"(colsig) : "memory")
    // THE HOT LOOP: super-tiles S, S+1, .. below `stop` in the MFMA form while none holds a candidate - per tile and
    // row block an MFMA on the ring, the OR of its result; one sign test per super-tile; the ring refilled with the
    // super-tile after next.  A tight loop of its own: nothing of the slow paths' state lives in registers across it, no
    // spill code inside.
    // Returns 0: S reached `stop`;  1: super-tile S holds candidates, their sign words are in am[] / amMask;
    // 2: most lane slots of S hold a candidate - the signature is not selective here, the dense form takes over.
    auto scan = [&](int &S, int stop, int last) -> int {   // last = send - 1: nothing is fetched beyond it
      u32 voff = ulane * 16u + (u32)S * 4096u;
      // TWO rings: while one super-tile is evaluated the next one's fragments are already on their way - loads are
      // issued a whole iteration before they are needed; the loop is unrolled by two so the rings swap roles instead
      // of being copied.  vmcnt(4): the older ring has landed, the younger one (four loads) may still be out.
      pg_v4i r0, r1, r2, r3, q0, q1, q2, q3;
      unsigned long long ringBase;                          // (scratch SGPR pair of the load statement)
      int av0 = 0, av1 = 0, av2 = 0, av3 = 0, av4 = 0, av5 = 0, av6 = 0, av7 = 0;   // OR of MFMA m = tile * R + block
      bool found;
      // two result sets in turn (the 128-VGPR budget), two MFMAs in the pipe before the first OR
#define PG_SB __builtin_amdgcn_sched_barrier(0)
#define PG_RING_STEP(x0, x1, x2, x3)                                                               \
      {                                                                                            \
        pg_v16i d0, d1;                                                                            \
        if constexpr (R == 1) {                                                                    \
          d0 = pg_mfma_fp4(A0, x0); d1 = pg_mfma_fp4(A0, x1); PG_SB;                               \
          av0 = pg_or16(d0); PG_SB; d0 = pg_mfma_fp4(A0, x2);                                      \
          av1 = pg_or16(d1); PG_SB; d1 = pg_mfma_fp4(A0, x3);                                      \
          av2 = pg_or16(d0);                                                                       \
          av3 = pg_or16(d1);                                                                       \
          found = __builtin_amdgcn_ballot_w64((av0 | av1 | av2 | av3) < 0) != 0;                   \
        } else {                                                                                   \
          d0 = pg_mfma_fp4(A0, x0); d1 = pg_mfma_fp4(A1, x0); PG_SB;                               \
          av0 = pg_or16(d0); PG_SB; d0 = pg_mfma_fp4(A0, x1);                                      \
          av1 = pg_or16(d1); PG_SB; d1 = pg_mfma_fp4(A1, x1);                                      \
          av2 = pg_or16(d0); PG_SB; d0 = pg_mfma_fp4(A0, x2);                                      \
          av3 = pg_or16(d1); PG_SB; d1 = pg_mfma_fp4(A1, x2);                                      \
          av4 = pg_or16(d0); PG_SB; d0 = pg_mfma_fp4(A0, x3);                                      \
          av5 = pg_or16(d1); PG_SB; d1 = pg_mfma_fp4(A1, x3);                                      \
          av6 = pg_or16(d0);                                                                       \
          av7 = pg_or16(d1);                                                                       \
          found = __builtin_amdgcn_ballot_w64((av0 | av1 | av2 | av3 | av4 | av5 | av6 | av7) < 0) != 0; \
        }                                                                                          \
      }
      // (no branch around a load statement: where two paths with different statements meet, the compiler copies
      //  ring registers - while their loads are still out.  So the prefetch is unconditional; at the last super-tile
      //  of the sweep it fetches that super-tile again.)
      PG_RING_LOAD("=&v", r0, r1, r2, r3, voff);
      bool inQ = false;                                     // the super-tile the loop stopped at sits in q0..3
      for (;;) {
        voff += (S < last && !PG_EXP_SAMETILE) ? 4096u : 0u;
        PG_RING_LOAD("=&v", q0, q1, q2, q3, voff);
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
        PG_RING_STEP(r0, r1, r2, r3);
        if (found || S + 1 >= stop) break;
        ++S;
        voff += (S < last && !PG_EXP_SAMETILE) ? 4096u : 0u;
        PG_RING_LOAD("=&v", r0, r1, r2, r3, voff);
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : : "memory");
        PG_RING_STEP(q0, q1, q2, q3);
        if (found || S + 1 >= stop) { inQ = true; break; }
        ++S;
      }
      // whatever is still on its way lands before anything else touches the registers (RULE); the prefetched
      // super-tile is given up (the next scan() fetches it again)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : : "memory");
      if (inQ) { r0 = q0; r1 = q1; r2 = q2; r3 = q3; }
#undef PG_RING_STEP
#undef PG_SB
      if (!found) {
        ++S;
        return 0;
      }
      PG_ST(1, 1);
      // MFMAs with a candidate; the lane slots (column x 16-row half) that hold one
      int nslots = 0;
      u32 mask = 0;                                         // bit m = MFMA m (tile m / R, row block m % R)
#define PG_FLAG(m, v) { const u64 mm = __builtin_amdgcn_ballot_w64((v) < 0); nslots += (int)__popcll(mm); mask |= mm ? (1u << (m)) : 0u; }
      PG_FLAG(0, av0) PG_FLAG(1, av1) PG_FLAG(2, av2) PG_FLAG(3, av3)
      if constexpr (R == 2) { PG_FLAG(4, av4) PG_FLAG(5, av5) PG_FLAG(6, av6) PG_FLAG(7, av7) }
#undef PG_FLAG
#ifdef PG_MM_KNOBS
      if (nslots >= R * K().mmDenseL1) return 2;            // (tuning builds: PG_MM_L1 at run time)
#else
      if (nslots >= R * PG_MM_DENSE_L1) return 2;
#endif
      // the flagged MFMAs again, from the ring: the sign word of every tile
      auto tile_signs = [&](int t, const pg_v4i &x) -> u32 {
        u32 a = 0;
        if constexpr (R == 1) {
          if ((mask >> t) & 1u) a = signs16(pg_mfma_fp4(A0, x));
        } else {
          if ((mask >> (2 * t + 1)) & 1u) a = signs16(pg_mfma_fp4(A1, x)) << 16;
          if ((mask >> (2 * t)) & 1u) a |= signs16(pg_mfma_fp4(A0, x));
        }
        return a;
      };
      am0 = tile_signs(0, r0); am1 = tile_signs(1, r1); am2 = tile_signs(2, r2); am3 = tile_signs(3, r3);
      if constexpr (R == 1) amMask = mask;
      else amMask = ((mask & 3u) ? 1u : 0u) | ((mask & 12u) ? 2u : 0u) | ((mask & 48u) ? 4u : 0u) | ((mask & 192u) ? 8u : 0u);
      return 1;
    };

    // ---- dense forms (the signature is not selective: mutant libraries, one cluster) ----
    //  exact : the round-1 direct form - the records of 64*C columns per lane in registers (two sets, the
    //          next tile's loads in flight), every distance of every row-step, in-place epilogue
    //  folded: per lane the PLANE FOLDS of four columns (256 columns a tile, B words a column: the G group
    //          words of a plane XOR-ed into one, from the fold section of the plane buffer);
    //          popcount(OR_p(fold_p(row) ^ fold_p(col))) <= d - an exact test of "differs nowhere" per folded
    //          position, B + 1 ops a column whatever L is.  Rows go four to a group of straight-line code, their
    //          folds as SGPR operands (scalar loads), the negated bound seeds the popcount and the sign is shifted
    //          into a per-slice hit mask: no branch in the row loop.  After the rows the hits join the candidate
    //          queue of the MFMA form (exact distance from gathered records, 64 a batch).
    // A run starts in the folded form and falls back to `exact` where the bound is not selective (hits in three
    // lanes of four, or more than 16 candidates a row: bounds still loose, eps graphs of data this dense,
    // unrelated sequences), probing again after 16 .. 256 super-tiles.
    constexpr int B = M::kBits;
    constexpr int CF = 4;                                   // folded form: columns per lane
    // plane folds of sequences come from the fold section of the plane buffer (pg_pack_planes): the columns'
    // as per-lane loads (B words a column), the rows' as SCALAR loads - wave-uniform addresses in the
    // constant address space, so a row's folds arrive in SGPRs and cost no VALU, LDS or VGPR at all.
    // The section pointers are read from the kernel arguments again in every tile (K(), above).
    typedef const u32 __attribute__((address_space(4))) *pg_kptr;
    struct DenseArgs { pg_kptr rowA, rowB; const uint4 *colA, *colB; };
    auto dense_args = [&]() -> DenseArgs {
      const auto &k = K();
      const uint4 *rf = k.rowFold, *cfp = k.colFold;
      return DenseArgs{(pg_kptr)(unsigned long long)rf, (pg_kptr)(unsigned long long)(rf + k.rowNpad), cfp, cfp + k.colNpad};
    };
    u32 seqv;                                               // lane r: sequence index of pass row r (rows past nr: the last row's)
    {
      const long long i = pr0 + ((lane & (RB - 1)) < nr ? (lane & (RB - 1)) : nr - 1);
      seqv = (u32)(ka.row0 + (ka.rowList ? ka.rowList[i] : i));
    }
    auto load_row_fold = [&](const DenseArgs &da, u32 (&f)[B], int rr) {
      const unsigned long long sq = (u32)__builtin_amdgcn_readlane((int)seqv, rr < RB ? rr : RB - 1);
#pragma unroll
      for (int pl = 0; pl < (B < 4 ? B : 4); ++pl) f[pl] = da.rowA[sq * 4 + pl];
#pragma unroll
      for (int pl = 4; pl < B; ++pl) f[pl] = da.rowB[sq * 4 + (pl - 4)];
    };
    auto load_col_fold = [&](const DenseArgs &da, u32 (&f)[B], long long col) {
      const uint4 a = da.colA[col];
      f[0] = a.x;
      if constexpr (B > 1) f[1] = a.y;
      if constexpr (B > 2) f[2] = a.z;
      if constexpr (B > 3) f[3] = a.w;
      if constexpr (B > 4) {
        const u32 *hi = reinterpret_cast<const u32 *>(da.colB + col);
#pragma unroll
        for (int pl = 4; pl < B; ++pl) f[pl] = hi[pl - 4];
      }
    };
    // lane r: minus the bound of row r for the dense forms (pairs at or beyond it cannot matter); 0 = nothing
    // can: rows past nr and, in phase 1, the frozen rows (their lists are final)
    auto neg_bounds = [&]() -> u32 {
      if constexpr (kEps) return lane < nr ? 0u - p.hi1 : 0u;
      const u32 t = thrv >> 24;                             // open lists read 255
      const bool live = lane < nr && (!resweep || ((failed >> lane) & 1ull)) && !((evicted >> lane) & 1ull);
      return live ? 0u - ((t < capv ? t : capv) + resweep) : 0u;
    };
    auto load_rec = [&](uint4 (&dst)[Q], long long col) {
#pragma unroll
      for (int q = 0; q < Q; ++q) dst[q] = colp[(long long)q * p.colNpad + col];
    };
    // exact distances of one row-step + the in-place epilogue (the round-1 direct form)
    auto row_exact = [&](const uint4 (&c)[C][Q], int rr, u32 col0, u32 bound) {
      uint4 r[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) r[q] = rows[rr * Q + q];
      u32 d[C];
#pragma unroll
      for (int b = 0; b < C; ++b) d[b] = M::dist(r, c[b], bias);
      u32 dmin = d[0];
#pragma unroll
      for (int b = 1; b < C; ++b) dmin = dmin < d[b] ? dmin : d[b];
      if (__builtin_amdgcn_ballot_w64(dmin < bound)) {
#pragma unroll
        for (int b = 0; b < C; ++b) epilogue(d[b], col0 + b * 64, rr);
      }
    };
    auto load_cols = [&](uint4 (&dst)[C][Q], int dt) {     // exact tile dt: columns dt * 64 * C ..
#pragma unroll
      for (int b = 0; b < C; ++b) load_rec(dst[b], (long long)dt * (64 * C) + b * 64 + lane);
    };
    auto rows_exact = [&](const uint4 (&c)[C][Q], int dt) {
      PG_ST(2, 1);
      const u32 col0 = (u32)(dt * (64 * C)) + lane;
      const u32 nbv = neg_bounds();                         // a row's bound only matters before its own step
      // (round 1's two-rows-in-turn pipelining of the LDS reads costs more here than it saves: two more record
      // sets push the kernel into spills - dense eps 30.9 -> 34.1 ms, random kNN 22.6 -> 23.7)
      for (int rr = 0; rr < nr; ++rr) {
        const u32 bnd = 0u - (u32)__builtin_amdgcn_readlane((int)nbv, rr);
        if (bnd) row_exact(c, rr, col0, kEps ? p.span + 1u : bnd);
      }
    };
    // one folded tile: 256 columns from colbase, folds in cf; the next tile's folds (from nxt) load into cfn
    // meanwhile.  The rows go GR to a group of straight-line code, the next group's folds in flight; the sign of
    // every lb - bound shifts into a per-slice hit mask (v_alignbit: one instruction a pair-of-64, no branch in
    // the row loop).  After the rows the pairs inside the bound join the candidate queue of the MFMA form
    // (exact distance from gathered records, 64 a batch).  Returns their number.
    constexpr int GR = PG_MM_GROUP_ROWS;
    auto tile_folded = [&](const u32 (&cf)[CF][B], u32 (&cfn)[CF][B], long long colbase, long long nxt) -> int {
      PG_ST(11, 1);
      PG_T0(tt0);
      const DenseArgs da = dense_args();
      const u32 nbv = neg_bounds();
#pragma unroll
      for (int b = 0; b < CF; ++b) load_col_fold(da, cfn[b], nxt + b * 64 + lane);
      // per row block (32 rows share a hit word): after the block's npb rows, bit npb-1-r = its row r is inside its bound
      u32 acc[R * CF];
      int npb[R];
      u32 all = 0;
#pragma unroll
      for (int blk = 0; blk < R; ++blk) {
        const int nrb = nr - 32 * blk < 0 ? 0 : (nr - 32 * blk > 32 ? 32 : nr - 32 * blk);   // rows of this block
        const int np = (nrb + GR - 1) / GR * GR;
        npb[blk] = np;
#pragma unroll
        for (int b = 0; b < CF; ++b) acc[blk * CF + b] = 0;
        u32 rf[GR][B], rn[GR][B];
#pragma unroll
        for (int u = 0; u < GR; ++u) load_row_fold(da, rf[u], 32 * blk + u);
        for (int r0 = 0; r0 < np; r0 += GR) {
#pragma unroll
          for (int u = 0; u < GR; ++u) load_row_fold(da, rn[u], 32 * blk + r0 + GR + u);
          u32 t[GR][CF];
#pragma unroll
          for (int u = 0; u < GR; ++u)
#pragma unroll
            for (int b = 0; b < CF; ++b) {
              t[u][b] = rf[u][0] ^ cf[b][0];
#pragma unroll
              for (int pl = 1; pl < B; ++pl) t[u][b] = __builtin_amdgcn_bitop3_b32(rf[u][pl], cf[b][pl], t[u][b], PG_BITOP_XOR_OR);
            }
#pragma unroll
          for (int u = 0; u < GR; ++u) {
            const u32 nb = (u32)__builtin_amdgcn_readlane((int)nbv, 32 * blk + r0 + u);
#pragma unroll
            for (int b = 0; b < CF; ++b)                    // lb - bound: negative = may be within the bound
              acc[blk * CF + b] = __builtin_amdgcn_alignbit(acc[blk * CF + b], (u32)__builtin_popcount(t[u][b]) + nb, 31);
          }
#pragma unroll
          for (int u = 0; u < GR; ++u)
#pragma unroll
            for (int pl = 0; pl < B; ++pl) rf[u][pl] = rn[u][pl];
        }
#pragma unroll
        for (int b = 0; b < CF; ++b) all |= acc[blk * CF + b];
      }
      int ncand = 0;
      // three lanes of four with a hit: the bound is not selective here (bounds still loose, eps graphs of dense
      // data, unrelated sequences).  Nothing is queued - the caller takes the tile again in the exact form
      if (__popcll(__builtin_amdgcn_ballot_w64(all != 0)) >= 48) return -1;
      if constexpr (MODE == PG_MODE_KNN) {
        // the hits, lane-parallel: per hit word every lane that holds any queues its lowest one, until none is left
        // (one or two turns as a rule; the order of a row's candidates does not matter to its list)
#pragma unroll
        for (int blk = 0; blk < R; ++blk)
#pragma nounroll
        for (int b = 0; b < CF; ++b) {
          u32 a = b == 1 ? acc[blk * CF + 1] : (b == 2 ? acc[blk * CF + 2] : (b == 3 ? acc[blk * CF + 3] : acc[blk * CF]));
          const u32 rowTop = (u32)(32 * blk + npb[blk] - 1);
          u64 mb = __builtin_amdgcn_ballot_w64(a != 0);
          while (mb) {
            const u32 j = (u32)__builtin_ctz(a | 0x80000000u);
            if (a != 0) cq[qn + mask_rank(mb)] = (rowTop - j) << SH | ((u32)colbase + b * 64 + lane);
            const int n = (int)__popcll(mb);
            qn += n;
            ncand += n;
            a &= a - 1;
            if (qn >= 64) flush();                          // (at most 63 + 64 entries before it)
            mb = __builtin_amdgcn_ballot_w64(a != 0);
          }
        }
      } else {
        // the hits, row by row (a row's columns stay in ascending order in the queue: slices, then lanes)
        const int np = npb[0];
        u32 rowsHit = (u32)__builtin_amdgcn_readlane((int)wave_or_to63(all), 63);
        while (rowsHit) {
          const int j = __builtin_ctz(rowsHit);
          rowsHit &= rowsHit - 1;
          const u32 erow = (u32)(np - 1 - j) << SH;
#pragma nounroll
          for (int b = 0; b < CF; ++b) {
            const u32 ab = b == 0 ? acc[0] : (b == 1 ? acc[1] : (b == 2 ? acc[2] : acc[3]));
            const bool hit = (ab >> j) & 1u;
            const u64 mb = __builtin_amdgcn_ballot_w64(hit);
            if (mb) {
              if (hit) cq[qn + mask_rank(mb)] = erow | ((u32)colbase + b * 64 + lane);
              const int n = (int)__popcll(mb);
              qn += n;
              ncand += n;
              if (qn >= 64) flush();                        // (at most 63 + 64 entries before it)
            }
          }
        }
      }
      PG_ST(5, ncand);
      PG_T1(14, tt0);
      return ncand;
    };

    // kNN checkpoints (first super-tile after them): after 1/32 of the sweep a row without any near
    // column yet is taken to be unclustered and loses the cap; after 1/8 every row whose list is
    // not settled below G0 does.  Phase 1 covers the larger range in use.
    // `nextCk` = the super-tile at (or after) which the next checkpoint is due; the sweep loop below services it -
    // queue drained first - at the first position it reaches from there (the one site shared with the end-of-sweep
    // drain: one copy of flush).  No checkpoints without a cap and in phase 1.
    constexpr int kNoCk = 0x7FFFFFFF;
    // (a column piece keeps the cap to its end: what the cap hides from it, the merge detects - NsqParams::mmPieces)
    // With eviction (NsqParams::mmEvict) a row is judged ONCE, and not before the sweep has passed the pass's own rows
    // (a self graph: column index = row index): where similar sequences sit together (a file sorted by family or by name)
    // a row meets its neighbours around its own position, and judged at 1/32 or 1/8 of the sweep a whole pass would look
    // unclustered (cfg3's sequences in lexicographic order: 17.9 ms).  Waiting costs nothing - a capped row has a bound of
    // G0, no candidates among unrelated sequences - and a row that fails is evicted, not swept again.  A pass whose own
    // rows lie at the end of the columns is judged at the end of the sweep.
    const bool judgeOnce = MODE == PG_MODE_KNN && p.mmEvict != nullptr;
    int sw2 = (nst + 7) >> 3;
    if (judgeOnce) {
      const long long own = (K().row0 + pr0) / PG_MM_ST + 8;   // (eight super-tiles past: neighbours on both sides)
      if (own > sw2) sw2 = own < nst ? (int)own : nst;
    }
    int nextCk = (MODE == PG_MODE_KNN && G0 && npieces == 1) ? (judgeOnce ? sw2 : ((nst + 31) >> 5)) : kNoCk;
    auto checkpoint = [&](int snext) {                      // the sweep has reached super-tile snext >= nextCk; queue empty
      if constexpr (MODE == PG_MODE_KNN) {
        const bool at2 = snext >= sw2;
        nextCk = at2 ? kNoCk : sw2;
        const bool mine = lane < nr && !((failed >> lane) & 1ull) && !((evicted >> lane) & 1ull);
        u32 dref = thrv >> 24;                             // open lists read 255
        if (!at2) dref = mine ? lstbuf[wv][lane < RB ? lane : 0][lfirst] >> 24 : 0u;
        const bool late = mine && dref >= G0;
        const u64 now = __builtin_amdgcn_ballot_w64(late);
        const auto &ke = K();
        if (now && ke.mmEvict && __popcll(now) <= PG_MM_EVICT_MAX) {
          // a few rows of the pass: out with them (an open bound among settled rows costs the whole pass its filter)
          u32 base = 0;
          if (lane == 0) base = atomicAdd(ke.mmEvict, (u32)__popcll(now));
          base = (u32)__builtin_amdgcn_readfirstlane((int)base);
          if (late) ke.mmEvictRows[base + mask_rank(now)] = (u32)(ke.row0 + pr0 + lane);
          evicted |= now;
          const u32 b = thrv >> 24;
          set_all_bounds((lane < nr && !((evicted >> lane) & 1ull)) ? (b < capv ? b : capv) : 0u);
        } else if (now) {
          failed |= now;
          sredo = snext;
          if (late) capv = 255u;
          const u32 b = thrv >> 24;
          set_all_bounds(lane < nr ? (b < capv ? b : capv) : 0u);
        }
      }
    };
    const bool canFold = canFilter;                         // PG_LB_FILTER=0 (or too many columns for queue entries): exact form only
    bool prefilter = canFold;
    int exact_left = 0, exact_run = 16;
    // super-tiles [S0, S1); returns where it stopped: >= S1 (a folded tile may end one past it), or earlier at a
    // position >= nextCk (a checkpoint is due: the caller services it and calls again)
    auto run_dense = [&](int S0, int S1) -> int {
      constexpr int F = PG_MM_ST / (64 * C);                // exact tiles per super-tile
      int s = S0;
      while (s < S1) {
        if (prefilter && !(s & 1)) {
          // folded tiles of two super-tiles each while the bound stays selective
          u32 cf[CF][B], cfn[CF][B];
#pragma unroll
          for (int b = 0; b < CF; ++b) load_col_fold(dense_args(), cf[b], (long long)s * PG_MM_ST + b * 64 + lane);
          while (s < S1) {
            const long long colbase = (long long)s * PG_MM_ST;
            const int ncand = tile_folded(cf, cfn, colbase, s + 2 < S1 ? colbase + 64 * CF : colbase);
            if (ncand < 0) {                                // not selective at all: this tile again, in the exact form,
              prefilter = false;                            // and twice as long until the next probe (16 .. 256 super-tiles)
              exact_left = exact_run;
              exact_run = exact_run < 256 ? exact_run * 2 : 256;
              break;
            }
            s += 2;
            exact_run = 16;
            // a candidate costs about three instructions of a gather batch, an exact row-step 25 per 64 columns
            if (ncand > 16 * nr) { prefilter = false; exact_left = 16; break; }
            if (s >= nextCk) return s;
#pragma unroll
            for (int b = 0; b < CF; ++b)
#pragma unroll
              for (int pl = 0; pl < B; ++pl) cf[b][pl] = cfn[b][pl];
          }
        } else {
          // exact form: to the end of the run, or until the bound is due for another probe
          if constexpr (kEps) {
            while (qn > 0) flush();                         // in-place results must come after queued ones (slot order)
          }
          int s1 = S1;
          if (prefilter) s1 = s + 1;                        // an odd super-tile in front of the folded tiles
          else if (canFold && s + exact_left < S1) s1 = s + exact_left;
          if (s < nextCk && nextCk < s1) s1 = nextCk;       // a run ends where a checkpoint is due
          const int t0 = s * F, t1 = s1 * F;
          uint4 ca[C][Q], cb[C][Q];
          load_cols(ca, t0);
          for (int dt = t0; dt < t1; dt += 2) {
            load_cols(cb, dt + 1 < t1 ? dt + 1 : dt);
            rows_exact(ca, dt);
            if (dt + 1 < t1) {
              load_cols(ca, dt + 2 < t1 ? dt + 2 : dt + 1);
              rows_exact(cb, dt + 1);
            }
          }
          if (!prefilter && canFold) {
            exact_left -= s1 - s;
            if (exact_left <= 0) prefilter = true;
          }
          s = s1;
          if (s >= nextCk) return s;
        }
      }
      return s;
    };

    PG_T1(21, tp0);
    int send = se;
    int drun = 0;                                           // super-tiles per dense run (0: the first of a series)
    const int sbeg = kSym ? (int)(pr0 / PG_MM_ST) : sb;    // EPS_SYM: from the super-tile that holds the pass's first row
    for (;;) {
      int S = sbeg;
      int dEnd = canFilter ? 0 : send;                      // a dense run is in progress up to here (no filter: throughout)
      for (;;) {
        if (S >= send || S >= nextCk) {                     // end of the sweep / a checkpoint: the queue is drained HERE
          while (qn > 0) flush();
          if (S >= nextCk) checkpoint(S);
          if (S >= send) break;
        }
        if (S < dEnd) {
          S = run_dense(S, dEnd);                           // (returns early where a checkpoint is due)
          continue;
        }
        if (stale) { PG_ST(19, 1); refresh_bias(); }
        // Issue priority falls with a wave's progress (16 steps along the sweep, the priority cycling 3..0 within four of
        // them): the waves of a SIMD advance together.  Left alone the arbiter favours the oldest wave, the three waves of
        // a SIMD finish at 0.70 / 0.83 / 1.0 of the launch (profiles/r03_pass_timeline.txt) and its last third runs with
        // two, then one wave per SIMD.  cfg3 1.74 -> 1.65 ms, N = 270k 2.62 -> 2.42 (4 steps: 1.69; 64: 1.69); the 32-row
        // instances do not gain (N = 100k L = 128: 0.98 -> 0.99; N = 50k: 0.50 -> 0.52), nor do eps launches (3.02 / 3.00 ms): left alone.
        if constexpr (R == 2) {
          const int step = ((S - sb) * PG_MM_PRIO_STEPS) / (se - sb > 0 ? se - sb : 1);
          const int pr = step & 3;
          if (pr == 0) __builtin_amdgcn_s_setprio(3);
          else if (pr == 1) __builtin_amdgcn_s_setprio(2);
          else if (pr == 2) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
        const int Sin = S;
        PG_ST(18, 1);
        PG_T0(ts0);
        const int rc = scan(S, send < nextCk ? send : nextCk, send - 1);
        PG_T1(16, ts0);
        PG_ST(0, S - Sin + (rc ? 1 : 0));                   // (counted here: nothing but the loop's own state lives in scan())
        PG_ST(9, resweep ? S - Sin + (rc ? 1 : 0) : 0);
        if (S != Sin) drun = 0;                             // a clean stretch ends a series of dense runs
        if (rc == 0) continue;
        if (rc == 1) {                                      // candidates queued, the MFMA form goes on
          PG_T0(tq0);
          push_signs(S);
          PG_T1(17, tq0);
          ++S;
          drun = 0;
          continue;
        }
        // the signature is not selective here: a run of the dense form
        const int drun0 = K().mmDirectRun;
        if (!drun) drun = drun0;
        dEnd = S + drun < send ? S + drun : send;
        drun = drun * 2 < 8 * drun0 ? drun * 2 : 8 * drun0;  // back off while every probe is dense
        PG_ST(3, 1);
        PG_ST(4, dEnd - S);
      }
      if constexpr (MODE == PG_MODE_KNN) {
        if (!failed || resweep) break;
        // phase 1: the rows that lost their cap see super-tiles [0, sredo) again; the others are frozen
        resweep = 1;
        nextCk = kNoCk;
        set_all_bounds((lane < nr && ((failed >> lane) & 1ull)) ? (thrv >> 24) + 1u : 0u);
        send = sredo;
      } else {
        break;
      }
    }

    // ---- per-row results of this pass ----
    if constexpr (kEps) {
      if (lane < nr) p.counts[pr0 + lane] = cntv;
    } else {
      const auto &kr = K();
      if constexpr (kPar) {
        const int kk = kr.k;
        if (npieces > 1) {
          // a column piece: the list as it stands (k + 1 keys, rank 0 included) for the merge
          const int k1 = kk + 1;
          for (int e = lane; e < nr * k1; e += 64) {
            const int rr = e / k1, j = e - rr * k1;
            kr.mmPartial[((pr0 - kr.mmPieceFrom * kr.rowsPerWave + rr) * (long long)npieces + piece) * k1 + j] = lstbuf[wv][rr][KL - k1 + j];
          }
        } else
        // the pass's nr x k results are one contiguous stretch of the output: coalesced stores
        for (int e = lane; e < nr * kk; e += 64) {
          const int rr = e / kk, j = e - rr * kk;
          const u32 key = lstbuf[wv][rr][KL - kk + j];
          const long long o = pr0 * (long long)kk + e;
          if ((evicted >> rr) & 1ull) continue;             // (pg_knn_rows_kernel writes these)
          kr.knnIdx[o] = (key == 0xFFFFFFFFu) ? -1 : (int)(key & 0x00FFFFFFu);
          kr.knnDist[o] = (unsigned char)(key >> 24);
        }
      } else {
        for (int rr = 0; rr < nr; ++rr) {
          if ((evicted >> rr) & 1ull) continue;             // (pg_knn_rows_kernel writes these)
          const u32 key = lstbuf[wv][rr][lane];
          if (lane >= kr.knnFirst && lane < kr.knnFirst + kr.k) {
            const long long o = (pr0 + rr) * (long long)kr.k + (lane - kr.knnFirst);
            kr.knnIdx[o] = (key == 0xFFFFFFFFu) ? -1 : (int)(key & 0x00FFFFFFu);
            kr.knnDist[o] = (unsigned char)(key >> 24);
          }
          if (kr.lastKeys && lane == thrLane) kr.lastKeys[pr0 + rr] = key;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    PG_T1(15, tp0);
#ifdef PG_MM_STATS
    if (p.stats && lane == 0 && pass < 65536) {
      p.stats[PG_NSTAT + 2 * pass] = tr0;              // 100 MHz wall clock at the start of the pass
      p.stats[PG_NSTAT + 2 * pass + 1] = (__builtin_amdgcn_s_memrealtime() - tr0) | ((unsigned long long)(st[5] - st5_0) << 24) |
                                         ((unsigned long long)(st[2] - st2_0) << 44) | ((unsigned long long)(st[7] - st7_0) << 54);
    }
#endif
    // the next pass of this wave
    u32 nx = 0;
    if (lane == 0) nx = atomicAdd(p.mmPassCounter, 1u);
    pass = p.mmGridWaves + (long long)(u32)__builtin_amdgcn_readfirstlane((int)nx);
  }
#ifdef PG_MM_STATS
  {
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < PG_NSTAT; ++i) v = lane == i ? st[i] : v;
    if (p.stats && lane < PG_NSTAT) atomicAdd(&p.stats[lane], (unsigned long long)v);
  }
#endif
}
#undef PG_ST
