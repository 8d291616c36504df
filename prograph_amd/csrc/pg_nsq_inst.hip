// Compiled once per PG_G (1..8 groups of 32 positions) so the variants build in parallel.
// Byte alphabets (8 planes) are instantiated up to G = 4 (L <= 128); beyond that the record no
// longer fits two register sets and callers take the generic path.
#include "pg_nsq.h"
#include "pg_mm.h"

#ifndef PG_G
#error "compile with -DPG_G=<1..8>"
#endif

#ifndef PG_CSEL
// measured: at Q <= 4 chunks per record (L <= 96 with 5 planes, L <= 64 with 8) two columns per lane
// keep the kernel at 4 waves per SIMD and win (C = 4 at L = 64: 167 VGPRs, 3 waves, 10-15 % slower);
// from Q = 5 on (L = 128: two register sets of 2 x 5 chunks leave 2 waves per SIMD) one column per
// lane is 22 % faster (N = 100k, L = 128: 6.43 -> 5.00 ms)
#define PG_CSEL(Q) ((Q) <= 4 ? 2 : 1)
#endif
#define PG_CAT_(a, b) a##b
#define PG_CAT(a, b) PG_CAT_(a, b)

// columns per lane: two register sets of C*Q chunks must stay well under the VGPR budget
// (override for experiments: make EXTRA='"-DPG_CSEL(Q)=4"')
template <int B>
struct Cols {
  static constexpr int Q = Rec<PG_G, B>::Q;
  static constexpr int C = PG_CSEL(Q);
  static constexpr bool kBuilt = (B == 5) || (PG_G <= 4);
};

template <int B, int MODE>
static int launch_nsq(const NsqParams &p, int grid, hipStream_t s) {
  if constexpr (Cols<B>::kBuilt) {
    pg_nsq_kernel<HammingMetric<PG_G, B>, Cols<B>::C, MODE><<<dim3(grid), dim3(PG_WG_THREADS), 0, s>>>(p);
    return (int)hipGetLastError();
  } else {
    return (int)hipErrorInvalidValue;
  }
}

template <int B, int MODE>
static int occ_nsq() {
  int n = 0;
  if constexpr (Cols<B>::kBuilt) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pg_nsq_kernel<HammingMetric<PG_G, B>, Cols<B>::C, MODE>,
                                                     PG_WG_THREADS, 0) != hipSuccess)
      n = 0;
  }
  return n;
}

int PG_CAT(pg_occ_nsq_g, PG_G)(int mode, int bits) {
  if (mode == PG_MODE_EPS_SYM) return bits == 5 ? occ_nsq<5, PG_MODE_EPS_SYM>() : occ_nsq<8, PG_MODE_EPS_SYM>();
  if (mode == PG_MODE_EPS) return bits == 5 ? occ_nsq<5, PG_MODE_EPS>() : occ_nsq<8, PG_MODE_EPS>();
  return bits == 5 ? occ_nsq<5, PG_MODE_KNN>() : occ_nsq<8, PG_MODE_KNN>();
}

int PG_CAT(pg_launch_nsq_g, PG_G)(int mode, int bits, const NsqParams &p, int grid, hipStream_t s) {
  if (mode == PG_MODE_EPS_SYM)
    return bits == 5 ? launch_nsq<5, PG_MODE_EPS_SYM>(p, grid, s) : launch_nsq<8, PG_MODE_EPS_SYM>(p, grid, s);
  if (mode == PG_MODE_EPS) return bits == 5 ? launch_nsq<5, PG_MODE_EPS>(p, grid, s) : launch_nsq<8, PG_MODE_EPS>(p, grid, s);
  return bits == 5 ? launch_nsq<5, PG_MODE_KNN>(p, grid, s) : launch_nsq<8, PG_MODE_KNN>(p, grid, s);
}

// grid < 0: no launch - the instance's resident workgroups per CU (0: unknown), for the planner
template <int B, int MODE, int KL = 64, int R = 1>
static int launch_mm(const NsqParams &p, int grid, hipStream_t s) {
  if constexpr (Cols<B>::kBuilt) {
    if (grid < 0) {
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pg_mm_kernel<HammingMetric<PG_G, B>, MODE, KL, R>, PG_WG_THREADS, 0) != hipSuccess) n = 0;
      return n;
    }
    pg_mm_kernel<HammingMetric<PG_G, B>, MODE, KL, R><<<dim3(grid), dim3(PG_WG_THREADS), 0, s>>>(p);
    return (int)hipGetLastError();
  } else {
    return (int)hipErrorInvalidValue;
  }
}

int PG_CAT(pg_launch_mm_g, PG_G)(int mode, int bits, const NsqParams &p, int grid, hipStream_t s) {
  if (mode == PG_MODE_EPS_SYM)
    return bits == 5 ? launch_mm<5, PG_MODE_EPS_SYM>(p, grid, s) : launch_mm<8, PG_MODE_EPS_SYM>(p, grid, s);
  if (mode == PG_MODE_EPS) return bits == 5 ? launch_mm<5, PG_MODE_EPS>(p, grid, s) : launch_mm<8, PG_MODE_EPS>(p, grid, s);
  // kNN with short lists (k + 1 <= PG_MM_KL, first round only): the lane-per-candidate insertion instance
  if (mode == PG_MODE_KNN_SHORT)
    return bits == 5 ? launch_mm<5, PG_MODE_KNN, PG_MM_KL>(p, grid, s) : launch_mm<8, PG_MODE_KNN, PG_MM_KL>(p, grid, s);
  if (mode == PG_MODE_KNN_SHORT2)   // ... and two row blocks per pass
    return bits == 5 ? launch_mm<5, PG_MODE_KNN, PG_MM_KL, 2>(p, grid, s) : launch_mm<8, PG_MODE_KNN, PG_MM_KL, 2>(p, grid, s);
  return bits == 5 ? launch_mm<5, PG_MODE_KNN>(p, grid, s) : launch_mm<8, PG_MODE_KNN>(p, grid, s);
}

template <int B>
static int launch_dense(const DenseParams &p, hipStream_t s) {
  if constexpr (!Cols<B>::kBuilt) return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned)((p.n + PG_WG_THREADS - 1) / PG_WG_THREADS), (unsigned)((p.m + PG_RBD - 1) / PG_RBD));
  if (p.outBytes == 8)
    pg_dense_kernel<PG_G, B, long long><<<grid, dim3(PG_WG_THREADS), 0, s>>>(p);
  else if (p.outBytes == 4)
    pg_dense_kernel<PG_G, B, int><<<grid, dim3(PG_WG_THREADS), 0, s>>>(p);
  else if (p.outBytes == 2)   // fp16 (integers up to 2048 are exact): the operand type of the fp16 selection kernels
    pg_dense_kernel<PG_G, B, _Float16><<<grid, dim3(PG_WG_THREADS), 0, s>>>(p);
  else
    pg_dense_kernel<PG_G, B, unsigned char><<<grid, dim3(PG_WG_THREADS), 0, s>>>(p);
  return (int)hipGetLastError();
}

template <int B>
static int launch_knn_rows(const KnnRowsParams &p, int grid, hipStream_t s) {
  if constexpr (!Cols<B>::kBuilt) return (int)hipErrorInvalidValue;
  pg_knn_rows_kernel<PG_G, B><<<dim3(grid), dim3(PG_WG_THREADS), 0, s>>>(p);
  return (int)hipGetLastError();
}

int PG_CAT(pg_launch_knn_rows_g, PG_G)(int bits, const KnnRowsParams &p, int grid, hipStream_t s) {
  return bits == 5 ? launch_knn_rows<5>(p, grid, s) : launch_knn_rows<8>(p, grid, s);
}

int PG_CAT(pg_launch_dense_g, PG_G)(int bits, const DenseParams &p, hipStream_t s) {
  return bits == 5 ? launch_dense<5>(p, s) : launch_dense<8>(p, s);
}

int PG_CAT(pg_launch_probe_g, PG_G)(int bits, const ProbeParams &p, hipStream_t s) {
  const unsigned grid = (unsigned)(((long long)p.nsample * p.wavesPerRow + PG_WG_WAVES - 1) / PG_WG_WAVES);
  if (bits == 5) {
    pg_probe_kernel<PG_G, 5><<<dim3(grid), dim3(PG_WG_THREADS), 0, s>>>(p);
  } else {
    if constexpr (Cols<8>::kBuilt) pg_probe_kernel<PG_G, 8><<<dim3(grid), dim3(PG_WG_THREADS), 0, s>>>(p);
    else return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}

int PG_CAT(pg_launch_compact_g, PG_G)(int bits, const CompactParams &p, hipStream_t s) {
  const unsigned grid = (unsigned)((p.e.nrows + PG_WG_WAVES - 1) / PG_WG_WAVES);
  if (bits == 5) {
    pg_compact_kernel<PG_G, 5><<<dim3(grid), dim3(PG_WG_THREADS), 0, s>>>(p);
  } else {
    if constexpr (Cols<8>::kBuilt) pg_compact_kernel<PG_G, 8><<<dim3(grid), dim3(PG_WG_THREADS), 0, s>>>(p);
    else return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}
