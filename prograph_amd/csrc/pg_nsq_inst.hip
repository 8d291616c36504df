// Compiled once per PG_Q (1..8) so the eight chunk counts build in parallel.
#include "pg_nsq.h"

#ifndef PG_Q
#error "compile with -DPG_Q=<1..8>"
#endif

#define PG_CAT_(a, b) a##b
#define PG_CAT(a, b) PG_CAT_(a, b)

// columns per lane: two register sets of B*Q chunks must stay well under the VGPR budget
static constexpr int kB = (PG_Q <= 4) ? 2 : 1;

template <int ALPHA, int MODE>
static int launch_nsq(const NsqParams &p, int grid, hipStream_t s) {
  pg_nsq_kernel<PG_Q, kB, ALPHA, MODE><<<dim3(grid), dim3(PG_WG_THREADS), 0, s>>>(p);
  return (int)hipGetLastError();
}

int PG_CAT(pg_launch_nsq_q, PG_Q)(int mode, int alpha, const NsqParams &p, int grid, hipStream_t s) {
  if (mode == PG_MODE_EPS) {
    if (alpha == 5) return launch_nsq<5, PG_MODE_EPS>(p, grid, s);
    return alpha == 7 ? launch_nsq<7, PG_MODE_EPS>(p, grid, s) : launch_nsq<8, PG_MODE_EPS>(p, grid, s);
  }
  if (alpha == 5) return launch_nsq<5, PG_MODE_KNN>(p, grid, s);
  return alpha == 7 ? launch_nsq<7, PG_MODE_KNN>(p, grid, s) : launch_nsq<8, PG_MODE_KNN>(p, grid, s);
}

template <int ALPHA>
static int launch_dense(const DenseParams &p, hipStream_t s) {
  const dim3 grid((unsigned)((p.n + PG_WG_THREADS - 1) / PG_WG_THREADS), (unsigned)((p.m + PG_RBD - 1) / PG_RBD));
  if (p.outBytes == 8)
    pg_dense_kernel<PG_Q, ALPHA, long long><<<grid, dim3(PG_WG_THREADS), 0, s>>>(p);
  else if (p.outBytes == 4)
    pg_dense_kernel<PG_Q, ALPHA, int><<<grid, dim3(PG_WG_THREADS), 0, s>>>(p);
  else
    pg_dense_kernel<PG_Q, ALPHA, unsigned char><<<grid, dim3(PG_WG_THREADS), 0, s>>>(p);
  return (int)hipGetLastError();
}

int PG_CAT(pg_launch_dense_q, PG_Q)(int alpha, const DenseParams &p, hipStream_t s) {
  return alpha == 8 ? launch_dense<8>(p, s) : launch_dense<7>(p, s);
}

int PG_CAT(pg_launch_compact_q, PG_Q)(int alpha, const CompactParams &p, hipStream_t s) {
  const unsigned grid = (unsigned)((p.e.nrows + PG_WG_WAVES - 1) / PG_WG_WAVES);
  if (alpha != 8)
    pg_compact_kernel<PG_Q, 7><<<dim3(grid), dim3(PG_WG_THREADS), 0, s>>>(p);
  else
    pg_compact_kernel<PG_Q, 8><<<dim3(grid), dim3(PG_WG_THREADS), 0, s>>>(p);
  return (int)hipGetLastError();
}
