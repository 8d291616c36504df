// The all-pairs engine: pairwise Hamming + fused epsilon-threshold / kNN selection.
//
// Replaces the hot loop of Prograph.build_graph (prograph/prograph.py:731-739 and
// :756-762 of the reference): distance(X, batch) -> mask/where or sort -> gather.
//
// Mapping (CDNA4, wave64):
//   * one LANE owns C COLUMN sequences: their bit-sliced records (pg_common.h) sit in VGPRs,
//     loaded from the chunk-major layout as fully coalesced global_load_dwordx4 (1 KiB per wave
//     instruction); the next column tile is prefetched into a second register set while the
//     current one is being compared;
//   * one WAVE owns a block of up to RB=16 ROW sequences, staged once per pass into a
//     wave-private LDS region and read back as broadcast ds_read_b128 (all lanes the same
//     address): the row operand costs a few LDS cycles per 64*C pairs;
//   * the wave sweeps ALL column tiles in ascending order for its rows, therefore every row's
//     matches are produced in ascending column order by exactly one wave: the reference's
//     `torch.where` order (prograph/prograph.py:736) without any sort, and the canonical
//     (distance, index) kNN order without a merge;
//   * per row-step (64*C pairs) the epilogue is C-1 v_min + 1 v_cmp and ONE wave-uniform
//     branch on the ballot; compaction (mbcnt) and sorted insertion (DPP wave_shr + v_readlane)
//     only run in the rarely taken slow path.
// No MFMA: the inner loop is boolean bit-plane logic (xor / bitop3 / popcount), B+1 VALU ops per
// 32 tokens; the operand matrix is cache resident, see DESIGN.md for the roofline.
#pragma once
#include "pg_common.h"

template <class M, int C, int MODE>
__global__ __launch_bounds__(PG_WG_THREADS) void pg_nsq_kernel(const NsqParams p) {
  constexpr int Q = M::Q;
  __shared__ uint4 rowbuf[PG_WG_WAVES][PG_RB][Q];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long gw = (long long)blockIdx.x * PG_WG_WAVES + wv;
  const long long wr0 = gw * p.rowsPerWave;
  if (wr0 >= p.nrows) return;   // whole wave leaves; no workgroup barrier is used below
  const long long wr1 = (wr0 + p.rowsPerWave < p.nrows) ? wr0 + p.rowsPerWave : p.nrows;
  const long long ntiles = (p.ncols + 64 * C - 1) / (64 * C);
  const uint4 *__restrict__ colp = p.colPlanes;
  const u32 ncols = (u32)p.ncols;
  const uint4 *rows = &rowbuf[wv][0][0] + opaque_zero();   // broadcast reads, kept "divergent"
  // eps: the bias -lo seeds the popcount accumulator (v_bcnt's addend) from an opaque VGPR, so
  // hipcc cannot re-associate it into an extra v_sub per pair
  const u32 bias = MODE == PG_MODE_EPS ? opaque_vgpr(0u - p.lo) : 0u;

  for (long long pr0 = wr0; pr0 < wr1; pr0 += p.rowsPerPass) {
    const long long left = wr1 - pr0;
    const int nr = __builtin_amdgcn_readfirstlane((int)(left < p.rowsPerPass ? left : p.rowsPerPass));

    // ---- stage this pass's rows into the wave's LDS region (wave private: LDS operations
    // of one wave are processed in order, the fences only pin the compiler) ----
    for (int e = lane; e < PG_RB * Q; e += 64) {
      const int rr = e % PG_RB, q = e / PG_RB;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (rr < nr) v = p.rowPlanes[(long long)q * p.rowNpad + p.row0 + pr0 + rr];
      rowbuf[wv][rr][q] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    u32 cnt[PG_RB];   // eps: matches so far per row (wave uniform -> SGPRs)
    u32 thr[PG_RB];   // knn: current (k+1)-th smallest key per row (wave uniform -> SGPRs)
    u32 lst[PG_RB];   // knn: lane j holds the j-th smallest key of the row seen so far
#pragma unroll
    for (int rr = 0; rr < PG_RB; ++rr) { cnt[rr] = 0; thr[rr] = 0xFFFFFFFFu; lst[rr] = 0xFFFFFFFFu; }

    // Two register sets hold the current and the next column tile; the tile loop is unrolled by
    // two so the sets swap roles instead of being copied (no v_mov per tile), and the prefetch
    // is unconditional (the last tile re-reads itself) so that hipcc's waitcnt pass leaves it in
    // flight across the row loop with a counted vmcnt.
    uint4 ca[C][Q], cb[C][Q];
    auto load_tile = [&](uint4 (&dst)[C][Q], long long t) {
      const long long tt = t < ntiles ? t : ntiles - 1;
#pragma unroll
      for (int b = 0; b < C; ++b)
#pragma unroll
        for (int q = 0; q < Q; ++q) dst[b][q] = colp[(long long)q * p.colNpad + tt * (64 * C) + b * 64 + lane];
    };
    auto sweep_rows = [&](const uint4 (&c)[C][Q], long long t) {
      const u32 col0 = (u32)(t * (64 * C)) + lane;
      uint4 r[Q], rn[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) r[q] = rows[q];
#pragma unroll
      for (int rr = 0; rr < PG_RB; ++rr) {
        if (rr < nr) {
          // broadcast read of the next row is issued before this row's compares
          if (rr + 1 < PG_RB) {
#pragma unroll
            for (int q = 0; q < Q; ++q) rn[q] = rows[(rr + 1) * Q + q];
          }
          // eps: (comp(d, eps) & (d > 0)) is one unsigned range test lo <= d <= lo+span; the
          //      bias -lo rides in the popcount accumulator.
          // knn: keys are (distance << 24 | column); columns only grow along the sweep, so a
          //      candidate beats the current (k+1)-th key iff its distance is strictly smaller.
          u32 d[C];
#pragma unroll
          for (int b = 0; b < C; ++b) d[b] = M::dist(r, c[b], bias);
          u32 dmin = d[0];
#pragma unroll
          for (int b = 1; b < C; ++b) dmin = dmin < d[b] ? dmin : d[b];
          const u32 bound = MODE == PG_MODE_EPS ? p.span + 1u : (thr[rr] >> 24);
          if (__builtin_amdgcn_ballot_w64(dmin < bound)) {
#pragma unroll
            for (int b = 0; b < C; ++b) {
              const u32 col = col0 + b * 64;
              if constexpr (MODE == PG_MODE_EPS) {
                const bool h2 = (d[b] <= p.span) && (col < ncols);
                const u64 m2 = __builtin_amdgcn_ballot_w64(h2);
                if (m2) {
                  const u32 pos = cnt[rr] + mask_rank(m2);
                  if (h2 && pos < p.cap) {
                    const long long o = (pr0 + rr) * (long long)p.cap + pos;
                    p.slotIdx[o] = (int)col;
                    p.slotW[o] = (unsigned char)(d[b] + p.lo);
                  }
                  cnt[rr] += (u32)__popcll(m2);
                }
              } else {
                u64 m = __builtin_amdgcn_ballot_w64((d[b] < (thr[rr] >> 24)) && (col < ncols));
                if (m) {
                  const u32 key = (d[b] << 24) | col;
                  do {
                    const int j = __builtin_ctzll(m);
                    m &= m - 1;
                    const u32 x = __builtin_amdgcn_readlane(key, j);
                    if (x < thr[rr]) {
                      const u32 cur = lst[rr];
                      const u32 prev = wave_shr1(cur, 0u);
                      lst[rr] = (cur <= x) ? cur : (prev > x ? prev : x);
                      thr[rr] = __builtin_amdgcn_readlane(lst[rr], p.k);
                    }
                  } while (m);
                }
              }
            }
          }
#pragma unroll
          for (int q = 0; q < Q; ++q) r[q] = rn[q];
        }
      }
    };

    load_tile(ca, 0);
    for (long long t = 0; t < ntiles; t += 2) {
      load_tile(cb, t + 1);
      sweep_rows(ca, t);
      if (t + 1 < ntiles) {
        load_tile(ca, t + 2);
        sweep_rows(cb, t + 1);
      }
    }

    // ---- per-row results of this pass ----
#pragma unroll
    for (int rr = 0; rr < PG_RB; ++rr) {
      if (rr < nr) {
        if constexpr (MODE == PG_MODE_EPS) {
          if (lane == 0) p.counts[pr0 + rr] = cnt[rr];
        } else {
          if (lane >= 1 && lane <= p.k) {
            const u32 key = lst[rr];
            const long long o = (pr0 + rr) * (long long)p.k + (lane - 1);
            p.knnIdx[o] = (key == 0xFFFFFFFFu) ? -1 : (int)(key & 0x00FFFFFFu);
            p.knnDist[o] = (unsigned char)(key >> 24);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

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
    if (ok) out[(r0 + rr) * p.ldo + col] = (OutT)d;
  }
}

// ---------------------------------------------------------------------------------------
// Slots -> CSR, one wave per row.  Rows that overflowed their slot are recomputed here
// with the same range test, so the result is exact for any capacity.
// ---------------------------------------------------------------------------------------
template <int G, int B>
__global__ __launch_bounds__(PG_WG_THREADS) void pg_compact_kernel(const CompactParams p) {
  constexpr int Q = Rec<G, B>::Q;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * PG_WG_WAVES + (threadIdx.x >> 6);
  if (row >= p.e.nrows) return;
  const u32 cnt = p.e.counts[row];
  const long long dst = p.indptr[row];
  if (cnt <= p.e.cap) {
    const long long src = row * (long long)p.e.cap;
    for (u32 i = lane; i < cnt; i += 64) {
      p.indices[dst + i] = p.e.slotIdx[src + i];
      p.weights[dst + i] = p.e.slotW[src + i];
    }
    return;
  }
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
