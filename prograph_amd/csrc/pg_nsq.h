// The all-pairs engine: pairwise Hamming + fused epsilon-threshold / kNN selection.
//
// Replaces the hot loop of Prograph.build_graph (prograph/prograph.py:731-739 and
// :756-762 of the reference): distance(X, batch) -> mask/where or sort -> gather.
//
// Mapping (CDNA4, wave64):
//   * one LANE owns C (1 or 2) COLUMN sequences: their bit-sliced records (pg_common.h) sit in
//     VGPRs, loaded from the chunk-major layout as fully coalesced global_load_dwordx4 (1 KiB per
//     wave instruction); the next column tile is prefetched into a second register set while the
//     current one is being compared;
//   * one WAVE owns a pass of up to 32 (eps) / 28 (kNN) ROW sequences, staged once per pass into a
//     wave-private LDS region and read back as broadcast LDS reads (all lanes the same address);
//   * the wave sweeps ALL column tiles in ascending order for its rows, therefore every row is owned
//     by exactly one wave: eps matches come out in ascending column order (the reference's
//     `torch.where` order, prograph/prograph.py:736) without a sort, kNN lists are kept sorted by
//     (distance, column) keys in LDS (64 lanes = 64 list slots, insertion = one DPP shift);
//   * a row-step (64*C pairs) normally costs only STAGE 1: a lower bound of the distance from 32-bit
//     filter signatures (v_xor + seeded v_bcnt, sign bit = "may be below the row's bound"), four
//     rows per group, one v_cmp + one branch per group;
//   * lanes that pass stage 1 are QUEUED (row, column) in LDS and evaluated exactly 64 at a time,
//     one candidate per lane, with gathered records; sub-tiles where many lanes pass, and whole
//     tiles while most row-steps trigger (dense data), are evaluated in place instead ("direct
//     form": all exact distances of the sub-tile, the column records are already in registers);
//   * kNN rows start with an optimistic cap on their stage-1 bound and, if they lose it at a
//     checkpoint, see the early tiles a second time at the end of the pass (exactness argument at
//     the definition of G0 below).
// No MFMA: the inner loop is boolean bit-plane logic (xor / bitop3 / popcount); the operand matrix
// is cache resident and the kernel is VALU-instruction bound, see DESIGN.md §4.1 for the roofline.
#pragma once
#include "pg_common.h"
#include <type_traits>

// kNN instances are held at the occupancy they had before the optimistic-start state was added
// (the extra scalar state spills a few SGPRs outside the loops instead of costing a wave): two
// columns per lane 4 (records of <= 3 chunks) / 3 waves per SIMD, one column per lane 3 (<= 5 chunks)
template <class M, int C, int MODE>
__global__ __launch_bounds__(PG_WG_THREADS)
__attribute__((amdgpu_waves_per_eu(MODE == PG_MODE_KNN ? (C == 2 ? (M::Q <= 3 ? 4 : 3) : (M::Q <= 5 ? 3 : 1)) : 1, 8))) void pg_nsq_kernel(const NsqParams p) {
  constexpr int Q = M::Q;
  constexpr bool kEps = MODE != PG_MODE_KNN;              // eps slots, rectangular or symmetric
  constexpr bool kSym = MODE == PG_MODE_EPS_SYM;          // every unordered pair once: columns above the row only
  constexpr int RB = MODE == PG_MODE_KNN ? PG_RB_KNN : PG_RB;   // rows per pass
  constexpr int LROWS = MODE == PG_MODE_KNN ? RB : 1;
  constexpr int QCAP = MODE == PG_MODE_KNN ? PG_QCAP : (M::kHasLB ? PG_QCAP_EPS : 1);
  __shared__ uint4 rowbuf[PG_WG_WAVES][RB + 4][Q];         // +4: the row prefetch runs up to four past
  __shared__ u32 lstbuf[PG_WG_WAVES][LROWS][64];           // kNN: per row, lane j = j-th smallest key
  __shared__ uint4 bndbuf[PG_WG_WAVES][RB / 4 + 2];        // kNN: per row minus the current (k+1)-th distance
  __shared__ u32 capbuf[PG_WG_WAVES][MODE == PG_MODE_KNN ? RB + 4 : 1];   // kNN: per row cap on the published bound
  __shared__ uint4 sigbuf[PG_WG_WAVES][M::kSigFold ? RB / 4 + 2 : 1];   // filter signatures of the pass's rows, 4 per read
  __shared__ u32 cqbuf[PG_WG_WAVES][QCAP];                 // deferred candidates: kNN row << 24 | column, eps {row, column} pairs
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long gw = (long long)blockIdx.x * PG_WG_WAVES + wv;
  const long long wr0 = gw * p.rowsPerWave;
  if (wr0 >= p.nrows) return;   // whole wave leaves; no workgroup barrier is used below
  if (p.gate && !((p.gateMask >> __builtin_nontemporal_load(p.gate)) & 1u)) return;   // the probe chose another engine / path
  const long long wr1 = (wr0 + p.rowsPerWave < p.nrows) ? wr0 + p.rowsPerWave : p.nrows;
  const int ntiles = (int)((p.ncols + 64 * C - 1) / (64 * C));   // ncols < 2^31
  const uint4 *__restrict__ colp = p.colPlanes;
  const u32 ncols = (u32)p.ncols;
  const uint4 *rows = &rowbuf[wv][0][0] + opaque_zero();   // broadcast reads, kept "divergent"
  const uint4 *bnds = &bndbuf[wv][0] + opaque_zero();
  u32 *bndw = reinterpret_cast<u32 *>(&bndbuf[wv][0]);
  // eps: the bias -lo seeds the popcount accumulator (v_bcnt's addend) from an opaque VGPR, so
  // hipcc cannot re-associate it into an extra v_sub per pair
  const u32 bias = kEps ? opaque_vgpr(0u - p.lo) : 0u;

  for (long long pr0 = wr0; pr0 < wr1; pr0 += p.rowsPerPass) {
    const long long left = wr1 - pr0;
    const int nr = __builtin_amdgcn_readfirstlane((int)(left < p.rowsPerPass ? left : p.rowsPerPass));

    // ---- stage this pass's rows into the wave's LDS region (wave private: LDS operations
    // of one wave are processed in order, the fences only pin the compiler) ----
    for (int e = lane; e < (RB + 4) * Q; e += 64) {
      const int rr = e % (RB + 4), q = e / (RB + 4);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (rr < nr) v = p.rowPlanes[(long long)q * p.rowNpad + p.row0 + pr0 + rr];
      rowbuf[wv][rr][q] = v;
    }
    // kNN, optimistic start: until a row has seen k+1 near columns its threshold is useless (any
    // unrelated pair passes) and the sweep would have to run in the direct form (~8 % of the tiles,
    // ~20 % of the instructions at cfg3).  Instead the published stage-1 bound is capped at G0 while
    // the row is "optimistic": only columns whose lower bound is below G0 are looked at.  That is
    // exact for every row whose final (k+1)-th distance is below G0 (a skipped column has
    // d >= lb >= G0).  At a checkpoint after 1/8 of the tiles the rows that are not there yet lose
    // the cap, and the tiles before the checkpoint are swept again for them at the end (phase 1)
    // with the relaxed rule "lb <= distance bound, full key comparison, no duplicates": the list
    // then holds columns from everywhere, so ties can no longer be decided by sweep position.
    const u32 G0 = (MODE == PG_MODE_KNN && M::kHasLB) ? p.knnGuess : 0u;   // the host passes 0 when the filter is off
    u32 failed = 0;                                         // kNN: rows that lost their optimistic cap (bit = row); bit 31 = phase 1
#define resweep (failed >> 31)
    int tredo = 0;                                          // kNN: tiles [0, tredo) are swept again for them
    if constexpr (MODE == PG_MODE_KNN) {
      for (int rr = 0; rr < nr; ++rr) lstbuf[wv][rr][lane] = 0xFFFFFFFFu;
      if (lane < RB + 8) bndw[lane] = 0u - (G0 ? G0 : 255u);   // bounds are stored negated
      if (lane < RB + 4) capbuf[wv][lane] = G0 ? G0 : 255u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if constexpr (M::kSigFold) {
      if (lane < RB + 8) {                                  // rows >= nr are zero records: signature 0
        uint4 rec[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) rec[q] = lane < RB + 4 ? rowbuf[wv][lane][q] : make_uint4(0, 0, 0, 0);
        reinterpret_cast<u32 *>(&sigbuf[wv][0])[lane] = M::fold(rec);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // Per-row state lives in ONE VGPR each, indexed by lane = row-in-pass and touched with
    // v_readlane / a lane-select (row index is wave uniform):
    //   eps: cntv = matches so far;   knn: thrv = current (k+1)-th smallest key (0xFFFFFFFF = open)
    // The kNN lists themselves sit in LDS and are only visited in the slow path.
    u32 cntv = 0u, thrv = 0xFFFFFFFFu;
    // kNN continuation rounds (k > 63): per-row floor key, only larger keys are candidates
    u32 floorv = 0u;
    if constexpr (MODE == PG_MODE_KNN) {
      if (p.floorKeys && lane < nr) floorv = p.floorKeys[pr0 + lane];
    }
    const int thrLane = p.knnFirst + p.k - 1;            // last list lane that is still needed

    // Two register sets hold the current and the next column tile; the tile loop is unrolled by
    // two so the sets swap roles instead of being copied, and the prefetch is unconditional (the
    // last tile re-reads itself) so that the waitcnt pass leaves it in flight across the rows.
    uint4 ca[C][Q], cb[C][Q];
    auto load_tile = [&](uint4 (&dst)[C][Q], int t) {
      const long long tt = t < ntiles ? t : ntiles - 1;
#pragma unroll
      for (int b = 0; b < C; ++b)
#pragma unroll
        for (int q = 0; q < Q; ++q) dst[b][q] = colp[(long long)q * p.colNpad + tt * (64 * C) + b * 64 + lane];
    };

    // exact distance of row rr against ONE 64-column sub-tile b + epilogue (slot append / sorted
    // insertion).  eps: (comp(d, eps) & (d > 0)) is one unsigned range test lo <= d <= lo+span, the
    // bias -lo rides in the popcount accumulator.  knn: keys are (distance << 24 | column); columns
    // only grow along the sweep, so a candidate beats the current (k+1)-th key iff its distance is
    // strictly smaller.
    // stage-1 bound of a row after its threshold moved: min(distance bound, cap), negated
    auto publish = [&](int row, u32 thr) {
      if (lane == 0) {
        const u32 cp = capbuf[wv][row];
        const u32 b = (thr >> 24) + (u32)resweep;           // phase 1: lb <= distance bound may still win a tie
        bndw[row] = 0u - (b < cp ? b : cp);
      }
    };
    // EPS_SYM: a match (row, col), col > row, also belongs to row `col`.  That row is owned by another
    // wave, so its entry goes to the BACK of its slot (position from an atomic counter, any order;
    // pg_compact_sym_kernel sorts that part) while the owner fills the front in column order.  Front
    // and back can only collide when the row overflows its slot, and then it is recomputed anyway.
    auto emit_lower = [&](bool on, u32 rowg, u32 col, u32 w) -> u32 {   // returns the back position
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
    auto epilogue = [&](u32 d, u32 col, int rr) {
      if constexpr (kEps) {
        bool h2 = (d <= p.span) && (col < ncols);
        if constexpr (kSym) h2 = h2 && col > (u32)(pr0 + rr);   // the pair's other half comes from emit_lower
        const u64 m2 = __builtin_amdgcn_ballot_w64(h2);
        if (m2) {
          const u32 cnt = __builtin_amdgcn_readlane(cntv, rr);
          const u32 pos = cnt + mask_rank(m2);
          const u32 posb = emit_lower(h2, (u32)(pr0 + rr), col, d + p.lo);
          if (h2 && pos < p.cap) {
            const long long o = (pr0 + rr) * (long long)p.cap + pos;
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
        // full key comparison: along the forward sweep it equals "d < current distance" (every list
        // entry has a smaller column), in phase 1 it also decides ties against later columns
        bool cand = (key < thr) && (col < ncols);
        if (p.floorKeys) cand = cand && key > __builtin_amdgcn_readlane(floorv, rr);   // continuation round
        u64 m = __builtin_amdgcn_ballot_w64(cand);
        if (m) {
          u32 lst = lstbuf[wv][rr][lane];
          do {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const u32 x = __builtin_amdgcn_readlane(key, j);
            if (x < thr && !(resweep && __builtin_amdgcn_ballot_w64(lst == x))) {
              const u32 prev = wave_shr1(lst, 0u);
              lst = (lst <= x) ? lst : (prev > x ? prev : x);
              thr = __builtin_amdgcn_readlane(lst, thrLane);
            }
          } while (m);
          lstbuf[wv][rr][lane] = lst;
          thrv = (lane == rr) ? thr : thrv;
          publish(rr, thr);
        }
      }
    };

    // kNN, filtered sweep: a triggered sub-tile usually holds ONE lane worth an exact distance (a
    // cluster mate that passes the plane-0 bound).  Evaluating it in place spends a full wave on it
    // (~23 VALU per trigger); instead the passing lanes are queued as (row, column) and 64 queued
    // candidates are evaluated together, one per lane, with both records gathered (row from the
    // wave's LDS, column from L2).  Keys are totally ordered, so insertion order does not matter for
    // the lists; the in-place test "d < current distance" stays exact because every list entry,
    // queued or not, has a smaller column than the sweep position.  Until a queued candidate is
    // inserted the row's bound is merely looser than it could be.
    int qn = 0;                                             // queue fill, wave uniform
    auto flush_batch = [&]() {
      if constexpr (MODE == PG_MODE_KNN) {
        const int nbat = qn < 64 ? qn : 64;
        const u32 e = cqbuf[wv][lane];
        const u32 col = e & 0x00FFFFFFu;
        const u32 erow = (e >> 24) & 31u;
        const bool act = lane < nbat && col < ncols;
        u32 key = 0xFFFFFFFFu, thr = 0u;
        if (act) {
          uint4 cr[Q], rw[Q];                             // all gathers in flight at once
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
        while (m) {
          const int j = __builtin_ctzll(m);
          m &= m - 1;
          const int row = (int)__builtin_amdgcn_readlane(erow, j);
          const u32 x = __builtin_amdgcn_readlane(key, j);
          if (p.floorKeys && x <= __builtin_amdgcn_readlane(floorv, row)) continue;   // continuation round
          u32 lst = lstbuf[wv][row][lane];
          if (x < __builtin_amdgcn_readlane(lst, thrLane) && !(resweep && __builtin_amdgcn_ballot_w64(lst == x))) {
            const u32 prev = wave_shr1(lst, 0u);
            lst = (lst <= x) ? lst : (prev > x ? prev : x);
            lstbuf[wv][row][lane] = lst;
            const u32 nthr = __builtin_amdgcn_readlane(lst, thrLane);
            thrv = (lane == row) ? nthr : thrv;
            publish(row, nthr);
          }
        }
        if (qn > 64) {                                       // keep the tail (at most 63 entries)
          const u32 tail = cqbuf[wv][64 + lane];
          cqbuf[wv][lane] = tail;
        }
        qn -= nbat;
      }
    };

    // eps, filtered sweep: the same deferral.  Every passing lane is queued (never evaluated in
    // place: slot positions are handed out in queue order, which is ascending column order per
    // row), the queue is drained before a direct-form tile and at the end of the pass.  Needs the
    // column index in 27 bits; wider problems keep the in-place path.
    const bool deferEps = kEps && M::kHasLB && p.ncols < (1ll << 27);
    auto flush_eps = [&]() {
      if constexpr (kEps && M::kHasLB) {
        const int nbat = qn < 64 ? qn : 64;
        const u32 e = cqbuf[wv][lane];
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
            const long long o = (pr0 + row) * (long long)p.cap + pos;
            p.slotIdx[o] = (int)col;
            p.slotW[o] = (unsigned char)(d + p.lo);
            if constexpr (kSym) {
              if (p.slotAux) p.slotAux[o] = (int)posb;
            }
          }
          cntv = (lane == (int)row) ? cnt + (u32)__popcll(same) : cntv;
          m &= ~same;
        }
        for (int i = lane; i + 64 < qn; i += 64) cqbuf[wv][i] = cqbuf[wv][i + 64];   // keep the tail, in order
        qn -= nbat;
      }
    };

    // Direct form of one row-step: all C exact distances, one min + compare + branch.  Chunk 0 of
    // the row arrives prefetched, the remaining chunks are read here (keeping only chunk 0 in the
    // double buffer holds the kernel at 3 waves per SIMD; buffering whole rows costs a wave).
    auto row_direct = [&](const uint4 (&c)[C][Q], const uint4 &r0, int rr, u32 col0) {
      uint4 r[Q];
      r[0] = r0;
#pragma unroll
      for (int q = 1; q < Q; ++q) r[q] = rows[rr * Q + q];
      u32 d[C];
#pragma unroll
      for (int b = 0; b < C; ++b) d[b] = M::dist(r, c[b], bias);
      u32 dmin = d[0];
#pragma unroll
      for (int b = 1; b < C; ++b) dmin = dmin < d[b] ? dmin : d[b];
      const u32 bound = kEps ? p.span + 1u : ((u32)__builtin_amdgcn_readlane((int)thrv, rr) >> 24);   // (u32: readlane returns a signed int)
      if (__builtin_amdgcn_ballot_w64(dmin < bound)) {
#pragma unroll
        for (int b = 0; b < C; ++b) epilogue(d[b], col0 + b * 64, rr);
      }
    };

    // One tile against the pass's rows, direct form: the row loop is unrolled by two with two
    // static chunk-0 buffers refilled (broadcast ds_read_b128) right after their last use.  Rows
    // nr.. exist in the buffer as zeros, so trailing prefetches are harmless.
    auto sweep_direct = [&](const uint4 (&c)[C][Q], int t) {
      const u32 col0 = (u32)(t * (64 * C)) + lane;
      uint4 ra = rows[0], rb = rows[Q];
      for (int rr = 0; rr < nr; rr += 2) {
        row_direct(c, ra, rr, col0);
        ra = rows[(rr + 2) * Q];
        if (rr + 1 < nr) row_direct(c, rb, rr + 1, col0);
        rb = rows[(rr + 3) * Q];
      }
    };

    // Filtered form: rows are taken FOUR at a time.  Stage 1 of the four rows (signature lower
    // bounds seeded with minus the row's bound, see below) is straight-line code with no branch;
    // the OR of all signs is examined with one compare + one scalar branch per group, and only
    // then the rows / sub-tiles that have passing lanes are looked at.  kNN bounds come from LDS
    // (one broadcast ds_read_b128 = the four rows' current bounds, written by the slow paths), eps
    // uses the constant hi+1.  Returns the number of triggered row-steps (for the window heuristic).
    auto sweep_filtered = [&](const uint4 (&c)[C][Q], int t) -> int {
      const u32 col0 = (u32)(t * (64 * C)) + lane;
      int trig = 0;
      // running LDS pointers (one v_add each per group instead of one address per read)
      const uint4 *rp = rows;
      const uint4 *bp = bnds;
      // Two register sets (the four rows' stage-1 operands + their bounds) swap roles: while one
      // group is evaluated the next one's LDS reads are in flight.  Stage-1 operand of a row: its
      // filter signature (Hamming: four rows in one ds_read_b128) or chunk 0 of its record.
      struct RowOp {
        uint4 v[M::kSigFold ? 1 : 4];
      };
      const uint4 *sp = &sigbuf[wv][0] + opaque_zero();
      auto load_rowop = [&](RowOp &o, int ahead) {
        if constexpr (M::kSigFold) {
          o.v[0] = sp[ahead];
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) o.v[u] = rp[(4 * ahead + u) * Q];
        }
      };
      u32 csig[C];                                          // the tile's column signatures
      if constexpr (M::kSigFold) {
#pragma unroll
        for (int b = 0; b < C; ++b) csig[b] = M::fold(c[b]);
      }
      RowOp r4a, r4b;
      // bounds travel NEGATED: they seed the popcount accumulator, so stage 1 yields lb - bound and
      // "lb < bound" is the sign bit; the signs of a whole group are OR-ed with 2-cycle logic ops
      // and examined with ONE v_cmp (v_min / v_cmp / v_bcnt are 4-cycle instructions on gfx950,
      // xor / or / bitop3 take 2: tools/ubench/valu_ops.hip)
      const u32 nhi = opaque_vgpr(0u - p.hi1);   // a VGPR, or hipcc splits the seeded popcount into bcnt + sub
      uint4 bnda = make_uint4(nhi, nhi, nhi, nhi), bndb = bnda;
      load_rowop(r4a, 0);
      if constexpr (MODE == PG_MODE_KNN) bnda = bp[0];
      auto group = [&](const RowOp &r4, const uint4 &bnd, RowOp &r4n, uint4 &bndn, int rr) {
        load_rowop(r4n, 1);
        if constexpr (MODE == PG_MODE_KNN) bndn = bp[1];
        __builtin_amdgcn_sched_barrier(0);       // keep the LDS reads ahead of the arithmetic
        const u32 nb[4] = {bnd.x, bnd.y, bnd.z, bnd.w};
        u32 t[4][C], o[4];
        if constexpr (M::kSigFold) {
          const u32 rs[4] = {r4.v[0].x, r4.v[0].y, r4.v[0].z, r4.v[0].w};
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < C; ++b) t[u][b] = rs[u] ^ csig[b];
          __builtin_amdgcn_sched_barrier(0);     // run of 2-cycle ops | run of 4-cycle ops
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < C; ++b) t[u][b] = __builtin_popcount(t[u][b]) + nb[u];   // seeded v_bcnt
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < C; ++b) t[u][b] = M::lb_prep(r4.v[u], c[b][0]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < C; ++b) t[u][b] = M::lb_finish(t[u][b], r4.v[u], c[b][0], nb[u]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // OR of all 4*C signs with three-input ors (4 instructions for 8 values; every VALU
        // instruction of this mixed stream costs about the same, so fewer is better)
        u32 any = t[0][0];
        if constexpr (C == 2) {
          // (asm: hipcc would otherwise rebuild the reduction from the per-row ors of the slow path)
          u32 a1, a2;
          asm("v_or3_b32 %0, %1, %2, %3" : "=v"(a1) : "v"(t[0][0]), "v"(t[0][1]), "v"(t[1][0]));
          asm("v_or3_b32 %0, %1, %2, %3" : "=v"(a2) : "v"(t[1][1]), "v"(t[2][0]), "v"(t[2][1]));
          asm("v_or3_b32 %0, %1, %2, %3" : "=v"(a1) : "v"(a1), "v"(t[3][0]), "v"(t[3][1]));
          any = a1 | a2;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < C; ++b) any |= t[u][b];
        }
        if (__builtin_amdgcn_ballot_w64((int)any < 0)) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            o[u] = t[u][0];
#pragma unroll
            for (int b = 1; b < C; ++b) o[u] |= t[u][b];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (__builtin_amdgcn_ballot_w64((int)o[u] < 0) && rr + u < nr) {
              ++trig;
#pragma unroll
              for (int b = 0; b < C; ++b) {
                const bool hit = (int)t[u][b] < 0;
                const u64 mb = __builtin_amdgcn_ballot_w64(hit);
                if (!mb) continue;
                // two 32-bit popcounts: a 64-bit one makes hipcc compare in 64 bits on the VALU
                const int npass = __builtin_popcount((u32)mb) + __builtin_popcount((u32)(mb >> 32));
                if constexpr (MODE == PG_MODE_KNN) {
                  if (npass <= PG_PUSH_MAX) {               // few lanes: queue them
                    if (hit) cqbuf[wv][qn + mask_rank(mb)] = ((u32)(rr + u) << 24) | (col0 + b * 64);
                    qn += npass;
                    continue;
                  }
                } else if constexpr (M::kHasLB) {
                  if (deferEps) {                           // always queued: slots follow queue order
                    if (hit) cqbuf[wv][qn + mask_rank(mb)] = ((u32)(rr + u) << 27) | (col0 + b * 64);
                    qn += npass;
                    continue;
                  }
                }
                uint4 r[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) r[q] = rp[u * Q + q];
                epilogue(M::dist(r, c[b], bias), col0 + b * 64, rr + u);
              }
            }
          }
          if constexpr (MODE == PG_MODE_KNN) {
            if (qn >= 64) flush_batch();                    // at most 4*C*PUSH_MAX = 64 pushes per group
          } else if constexpr (M::kHasLB) {
            while (qn >= 64) flush_eps();                   // at most 4*C*64 pushes per group
          }
        }
        rp += 4 * Q;
        bp += 1;
        sp += 1;
      };
      for (int rr = 0; rr < nr; rr += 8) {
        group(r4a, bnda, r4b, bndb, rr);
        if (rr + 4 < nr) group(r4b, bndb, r4a, bnda, rr + 4);
      }
      return trig;
    };
    // Adaptive choice: the filter pays while, on average, fewer than ~2/3 of the row-steps need
    // stage 2; when the data are dense (mutant libraries: almost every pair is near) the direct
    // form is cheaper.  The decision uses the trigger count of a WINDOW of 8 filtered tiles (single
    // tiles are all-or-nothing when neighbouring rows share neighbouring columns); after a window
    // that fails, 120 tiles run direct before the filter is probed again (probing a dense data set
    // costs ~2 %).  Wave uniform throughout.
    int win_tiles = 0, win_trig = 0, direct_left = 0;
    auto sweep = [&](const uint4 (&c)[C][Q], int t) {
      if constexpr (M::kHasLB) {
        if (resweep) {                                        // phase 1 stays filtered: frozen rows never trigger there,
          sweep_filtered(c, t);                               // the direct form would evaluate them again
          return;
        }
        if (p.filter != 0 && direct_left == 0) {
          win_trig += sweep_filtered(c, t);
          if (++win_tiles == 8) {
            if (p.filter == 1 && win_trig * 3 > nr * 8 * 2) direct_left = 120;   // filter == 2: forced on
            win_tiles = 0;
            win_trig = 0;
          }
          return;
        }
        if (direct_left > 0) --direct_left;
        if constexpr (kEps) {
          while (qn > 0) flush_eps();                       // in-place stores must come after queued ones
        }
      }
      sweep_direct(c, t);
    };

    // checkpoints (first tile after them): after 1/32 of the tiles a row without any near column yet
    // (nearest needed rank at distance >= G0) is taken to be unclustered and loses the cap; after
    // 1/8 every row whose list is not settled below G0 does.  Phase 1 covers the larger range in use.
    auto checkpoint = [&](int tnext) {
      if constexpr (MODE == PG_MODE_KNN) {
        const int tsw1 = (ntiles + 31) >> 5, tsw2 = (ntiles + 7) >> 3;
        if (G0 && !resweep && (tnext == tsw1 || tnext == tsw2)) {
          while (qn > 0) flush_batch();
          const bool mine = lane < nr && !((failed >> lane) & 1);
          u32 dref = thrv >> 24;                               // open lists read 255
          if (tnext != tsw2) dref = mine ? lstbuf[wv][lane][p.knnFirst] >> 24 : 0u;
          const bool late = mine && dref >= G0;
          const u32 now = (u32)__builtin_amdgcn_ballot_w64(late);   // rows < 32
          if (now) {
            failed |= now;
            tredo = tnext;
            if (late) {
              capbuf[wv][lane] = 255u;
              bndw[lane] = 0u - (thrv >> 24);
            }
          }
        }
      }
    };
    int tend = ntiles;
    const int tbeg = kSym ? (int)(pr0 / (64 * C)) : 0;      // EPS_SYM: from the tile that holds the pass's first row
    for (;;) {
      load_tile(ca, tbeg);
      for (int t = tbeg; t < tend; t += 2) {
        load_tile(cb, t + 1);
        sweep(ca, t);
        checkpoint(t + 1);
        if (t + 1 < tend) {
          load_tile(ca, t + 2);
          sweep(cb, t + 1);
          checkpoint(t + 2);
        }
      }
      if constexpr (MODE == PG_MODE_KNN) {
        while (qn > 0) flush_batch();
        if (!failed || resweep) break;
        // phase 1: the rows that lost their cap see tiles [0, tredo) again; the others are frozen
        if (lane < RB + 8) bndw[lane] = ((failed >> lane) & 1) ? 0u - ((thrv >> 24) + 1u) : 0u;
        failed |= 0x80000000u;
        tend = tredo;
        win_tiles = 0; win_trig = 0; direct_left = 0;
      } else {
        break;
      }
    }

    if constexpr (MODE == PG_MODE_KNN) {
      while (qn > 0) flush_batch();
    } else {
      while (qn > 0) flush_eps();
    }
    // ---- per-row results of this pass ----
    if constexpr (kEps) {
      if (lane < nr) p.counts[pr0 + lane] = cntv;
    } else {
      for (int rr = 0; rr < nr; ++rr) {
        const u32 key = lstbuf[wv][rr][lane];
        if (lane >= p.knnFirst && lane < p.knnFirst + p.k) {
          const long long o = (pr0 + rr) * (long long)p.k + (lane - p.knnFirst);
          p.knnIdx[o] = (key == 0xFFFFFFFFu) ? -1 : (int)(key & 0x00FFFFFFu);
          p.knnDist[o] = (unsigned char)(key >> 24);
        }
        if (p.lastKeys && lane == thrLane) p.lastKeys[pr0 + rr] = key;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

#undef resweep

// ---------------------------------------------------------------------------------------
// Dense (M,N) distance matrix: hamming() operator parity (prograph/distance/hamming.py:34).
// Output bound (8 B per pair for int64), so the grid is over column tiles x row blocks and the
// workgroup shares one staged row block.
// ---------------------------------------------------------------------------------------
template <int G, int B, typename OutT>
__global__ __launch_bounds__(PG_WG_THREADS) void pg_dense_kernel(const DenseParams p) {
  constexpr int Q = Rec<G, B>::Q;
  __shared__ uint4 rowbuf[PG_RBD][Q];
  const long long col = (long long)blockIdx.x * PG_WG_THREADS + threadIdx.x;
  const long long r0 = (long long)blockIdx.y * PG_RBD;
  const int nr = (int)((p.m - r0) < PG_RBD ? (p.m - r0) : PG_RBD);
  for (int e = threadIdx.x; e < PG_RBD * Q; e += PG_WG_THREADS) {
    const int rr = e % PG_RBD, q = e / PG_RBD;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (rr < nr) v = p.yPlanes[(long long)q * p.yNpad + r0 + rr];
    rowbuf[rr][q] = v;
  }
  uint4 c[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) c[q] = p.xPlanes[(long long)q * p.xNpad + col];
  __syncthreads();
  OutT *out = reinterpret_cast<OutT *>(p.out);
  const bool ok = col < p.n;
  const uint4 *rows = &rowbuf[0][0] + opaque_zero();
  for (int rr = 0; rr < nr; ++rr) {
    uint4 r[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) r[q] = rows[rr * Q + q];
    const u32 d = mismatch<G, B>(r, c);
    if (ok) {
      OutT *o = &out[(r0 + rr) * p.ldo + col];
      *o = p.accumulate ? (OutT)(*o + (OutT)d) : (OutT)d;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Exact kNN of single rows: the rows the MFMA engine evicted from their passes (NsqParams::mmEvict) - rows without k + 1
// columns inside the optimistic cap, whose bound cannot be tight - and the rows of column pieces whose merged list the
// cap may have cut short (pg_knn_merge_kernel).  A WORKGROUP takes eight rows at a time: its four waves take every fourth
// tile of 128 columns (lane = 2 columns; one load of the column records serves the eight rows - a row alone streams the
// whole operand, 9.6 MB at cfg3: 10 000 evicted rows were 3.7 ms of L2 traffic), every distance, the k + 1 smallest
// (distance, column) keys of a row across the lanes of each wave (lane j = j-th smallest: insertion = one DPP shift); the
// waves then merge the four lists of two rows each through LDS and write ranks 1..k (prograph/prograph.py:761-763).
// ---------------------------------------------------------------------------------------
template <int G, int B>
__global__ __launch_bounds__(PG_WG_THREADS) void pg_knn_rows_kernel(const KnnRowsParams p) {
  constexpr int Q = Rec<G, B>::Q;
  constexpr int CR = 2;                                    // columns per lane and turn
  constexpr int RW = 8;                                    // rows per workgroup and turn
  __shared__ uint4 rowrec[RW][Q];
  __shared__ u32 lists[PG_WG_WAVES][RW][64];
  if (p.gate && !((p.gateMask >> __builtin_nontemporal_load(p.gate)) & 1u)) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long count = (long long)__builtin_nontemporal_load(p.count);
  const int kk = p.k;
  for (long long g0 = (long long)blockIdx.x * RW; g0 < count; g0 += (long long)gridDim.x * RW) {
    const int nrw = (int)(count - g0 < RW ? count - g0 : RW);
    __syncthreads();                                       // (the previous rows' lists and records are done with)
    for (int e = threadIdx.x; e < RW * Q; e += PG_WG_THREADS) {
      const int rr = e / Q, q = e - rr * Q;
      rowrec[rr][q] = p.rowPlanes[(long long)q * p.rowNpad + (long long)p.rows[g0 + (rr < nrw ? rr : 0)]];
    }
    __syncthreads();
    const uint4 *rrec = &rowrec[0][0] + opaque_zero();     // broadcast reads, kept "divergent"
    u32 lst[RW], thr[RW];                                  // row rr: lane j = its j-th smallest key so far; thr = lane k's
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) { lst[rr] = 0xFFFFFFFFu; thr[rr] = 0xFFFFFFFFu; }
    const long long ntiles = (p.ncols + 64 * CR - 1) / (64 * CR);
    for (long long t = wv; t < ntiles; t += PG_WG_WAVES) {
      uint4 c[CR][Q];
#pragma unroll
      for (int b = 0; b < CR; ++b) {
        const long long col = t * (64 * CR) + b * 64 + lane;   // (the plane buffer is padded to 256 sequences: in bounds)
#pragma unroll
        for (int q = 0; q < Q; ++q) c[b][q] = p.colPlanes[(long long)q * p.colNpad + col];
      }
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        if (rr >= nrw) break;
        uint4 r[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) r[q] = rrec[rr * Q + q];
#pragma unroll
        for (int b = 0; b < CR; ++b) {                      // (ascending columns: slices, then lanes)
          const long long col = t * (64 * CR) + b * 64 + lane;
          const u32 key = col < p.ncols ? (mismatch<G, B>(r, c[b]) << 24) | (u32)col : 0xFFFFFFFFu;
          u64 m = __builtin_amdgcn_ballot_w64(key < thr[rr]);
          while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const u32 x = (u32)__builtin_amdgcn_readlane((int)key, j);
            if (x < thr[rr]) {
              const u32 prev = wave_shr1(lst[rr], 0u);
              lst[rr] = (lst[rr] <= x) ? lst[rr] : (prev > x ? prev : x);
              thr[rr] = (u32)__builtin_amdgcn_readlane((int)lst[rr], kk);
            }
          }
        }
      }
    }
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) lists[wv][rr][lane] = lst[rr];
    __syncthreads();
    for (int rr = wv; rr < nrw; rr += PG_WG_WAVES) {       // wave w merges the four lists of rows w and w + 4
      u32 l = lists[0][rr][lane];
      u32 t = (u32)__builtin_amdgcn_readlane((int)l, kk);
      for (int w = 1; w < PG_WG_WAVES; ++w) {
        const u32 other = lists[w][rr][lane];
        u64 m = __builtin_amdgcn_ballot_w64(other < t && lane <= kk);
        while (m) {
          const int j = __builtin_ctzll(m);
          m &= m - 1;
          const u32 x = (u32)__builtin_amdgcn_readlane((int)other, j);
          if (x < t) {
            const u32 prev = wave_shr1(l, 0u);
            l = (l <= x) ? l : (prev > x ? prev : x);
            t = (u32)__builtin_amdgcn_readlane((int)l, kk);
          }
        }
      }
      if (lane >= 1 && lane <= kk) {
        const long long o = ((long long)p.rows[g0 + rr] - p.baseRow) * (long long)kk + (lane - 1);
        p.knnIdx[o] = l == 0xFFFFFFFFu ? -1 : (int)(l & 0x00FFFFFFu);
        p.knnDist[o] = (unsigned char)(l >> 24);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Data probe in front of a large all-pairs launch: exact distances of a few sample rows against all columns, counted
// (see ProbeParams).  Grid: nsample * wavesPerRow waves; wave w of a sample row takes every wavesPerRow-th tile of 64
// columns.  ~1e-3 of the launch's pair count.
// ---------------------------------------------------------------------------------------
template <int G, int B>
__global__ __launch_bounds__(PG_WG_THREADS) void pg_probe_kernel(const ProbeParams p) {
  constexpr int Q = Rec<G, B>::Q;
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * PG_WG_WAVES + (threadIdx.x >> 6);
  const int s = (int)(gw / p.wavesPerRow), part = (int)(gw % p.wavesPerRow);
  if (s >= p.nsample) return;
  const long long row = p.row0 + (p.nrows * (2ll * s + 1)) / (2ll * p.nsample);      // evenly spaced over the launch's rows
  uint4 r[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) r[q] = p.rowPlanes[(long long)q * p.rowNpad + row];
  u32 nearCnt = 0, epsCnt = 0;
  // every PG_PROBE_STRIDE-th tile of 64 columns (offset by the sample row: different rows, different tiles): a
  // sample of the columns is enough for the two yes/no questions asked, and S full sweeps would cost S x the matrix
  // in cache traffic.  Tile index below = index among the sampled tiles.
  const long long ntiles = ((p.ncols + 63) / 64 - (s % PG_PROBE_STRIDE) + PG_PROBE_STRIDE - 1) / PG_PROBE_STRIDE;
  // four tiles per turn: their loads are in flight together (a wave's turns are a dependent chain of L2 round trips)
  constexpr int T = 4;
  for (long long t = part; t < ntiles; t += (long long)T * p.wavesPerRow) {
    uint4 c[T][Q];
    long long col[T];
    bool in[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      const long long ti = t + (long long)i * p.wavesPerRow;
      in[i] = ti < ntiles;
      col[i] = ((in[i] ? ti : t) * PG_PROBE_STRIDE + (s % PG_PROBE_STRIDE)) * 64 + lane;
#pragma unroll
      for (int q = 0; q < Q; ++q) c[i][q] = p.colPlanes[(long long)q * p.colNpad + col[i]];
    }
#pragma unroll
    for (int i = 0; i < T; ++i) {
      const u32 d = mismatch<G, B>(r, c[i]);
      const bool ok = in[i] && col[i] < p.ncols;
      nearCnt += (u32)__popcll(__builtin_amdgcn_ballot_w64(ok && d < p.near));
      epsCnt += (u32)__popcll(__builtin_amdgcn_ballot_w64(ok && (d - p.lo) <= p.span));
    }
  }
  // one slot per wave, plain stores (atomics of thousands of waves onto a few cache lines serialise: ~88 per us and line)
  if (lane == 0) {
    p.counts[2 * gw] = nearCnt;
    p.counts[2 * gw + 1] = epsCnt;
  }
}

// ---------------------------------------------------------------------------------------
// Slots -> CSR, one wave per row.  Rows that overflowed their slot are recomputed here
// with the same range test, so the result is exact for any capacity.
// ---------------------------------------------------------------------------------------
template <int G, int B>
__global__ __launch_bounds__(PG_WG_THREADS) void pg_compact_kernel(const CompactParams p) {
  constexpr int Q = Rec<G, B>::Q;
  __shared__ u32 lcol[PG_WG_WAVES][PG_SORT_MAX];          // symmetric slots: columns of a row's back part
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * PG_WG_WAVES + (threadIdx.x >> 6);
  if (row >= p.e.nrows) return;
  // symmetric slots (pg_eps_slots_sym): `up` entries with column > row at the front of the slot, in
  // column order; `lo` entries with column < row at the back, in arrival order -> rank-sorted here
  const u32 up = p.e.counts[row];
  const u32 lo = p.e.countsLo ? p.e.countsLo[row] : 0u;
  const u32 cnt = up + lo;
  const long long dst = p.indptr[row];
  if (cnt <= p.e.cap && lo <= (u32)PG_SORT_MAX) {
    const long long src = row * (long long)p.e.cap;
    if (lo) {
      // rank sort of the back part through the wave's LDS: rank = entries with a smaller column
      u32 *lc = &lcol[threadIdx.x >> 6][0];
      for (u32 i = lane; i < lo; i += 64) lc[i] = (u32)p.e.slotIdx[src + p.e.cap - 1u - i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (u32 i = lane; i < lo; i += 64) {
        const u32 ecol = lc[i];
        u32 rank = 0;
        for (u32 j = 0; j < lo; ++j) rank += lc[j] < ecol ? 1u : 0u;
        p.indices[dst + rank] = (int)ecol;
        p.weights[dst + rank] = p.e.slotW[src + p.e.cap - 1u - i];
      }
    }
    // The front part is in ascending order of the columns' 32-column TILE, within a tile in any order (the MFMA engine
    // queues a tile's candidates lane-parallel, pg_mm.h push_signs; the VALU engine and the dense forms write in order):
    // entries 32 or more positions apart are in order already, so an entry's place is 31 fewer than its position at
    // most, plus the entries with a smaller column among its 31 neighbours on either side.
    if (up <= (u32)PG_SORT_MAX) {
      u32 *lc = &lcol[threadIdx.x >> 6][0];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                      // (the back part above is done with the buffer)
      for (u32 i = lane; i < up; i += 64) lc[i] = (u32)p.e.slotIdx[src + i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (u32 i = lane; i < up; i += 64) {
        const u32 ecol = lc[i];
        const u32 j0 = i > 31u ? i - 31u : 0u, j1 = i + 32u < up ? i + 32u : up;
        u32 rank = j0;
        for (u32 j = j0; j < j1; ++j) rank += lc[j] < ecol ? 1u : 0u;
        p.indices[dst + lo + rank] = (int)ecol;
        p.weights[dst + lo + rank] = p.e.slotW[src + i];
      }
    } else {
      for (u32 i = lane; i < up; i += 64) {
        const u32 ecol = (u32)p.e.slotIdx[src + i];
        const u32 j0 = i > 31u ? i - 31u : 0u, j1 = i + 32u < up ? i + 32u : up;
        u32 rank = j0;
        for (u32 j = j0; j < j1; ++j) rank += (u32)p.e.slotIdx[src + j] < ecol ? 1u : 0u;
        p.indices[dst + lo + rank] = (int)ecol;
        p.weights[dst + lo + rank] = p.e.slotW[src + i];
      }
    }
    return;
  }
  if (p.skipOverflow) return;                             // pg_eps_fill_rows takes these rows (all of them at engine speed)
  uint4 r[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) r[q] = p.e.rowPlanes[(long long)q * p.e.rowNpad + p.e.row0 + row];
  long long run = 0;
  const long long ntiles = (p.e.ncols + 63) / 64;
  for (long long t = 0; t < ntiles; ++t) {
    const long long col = t * 64 + lane;
    uint4 c[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) c[q] = p.e.colPlanes[(long long)q * p.e.colNpad + col];
    const u32 d = mismatch<G, B>(r, c);
    const bool hit = ((d - p.e.lo) <= p.e.span) && (col < p.e.ncols);
    const u64 m = __ballot(hit);
    if (hit) {
      const long long o = dst + run + mask_rank(m);
      p.indices[o] = (int)col;
      p.weights[o] = (unsigned char)d;
    }
    run += __popcll(m);
  }
}
