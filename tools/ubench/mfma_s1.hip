// Micro-benchmark: stage 1 of the all-pairs engine (signature lower bound, "may this pair be below
// the row's bound?") as an int8 MFMA instead of v_xor + v_bcnt per pair.
//
//   lb(i,j) = popcount(sa_i ^ sb_j)  over 31 signature bits
//           = pa_i - sum_k a'_ik * b_jk         a' = +-1 bytes (rows), b = 0/1 bytes (columns)
//   D(i,j)  = bias_i * 1 + sum_k (-a'_ik) * b_jk = lb - pa_i + bias_i,   bias_i = pa_i - bound_i
//   D < 0  <=>  lb < bound_i.      K slot 31 carries the bias (A) against a constant 1 (B).
// One v_mfma_i32_32x32x32_i8 = 32 rows x 32 columns = 1024 pairs; the 16 result registers are
// OR-ed (8 v_or3) and ONE sign test + branch decides whether anything in the tile may match.
//
// Column operand ("E"): per 32-column tile one 1 KiB block in FRAGMENT order,
//   E[tile][h*32 + c] (uint4) = bytes k = 16h .. 16h+15 of column 32*tile + c,
// so a wave's B operand is one fully coalesced global_load_dwordx4 (or a linear LDS image).
//
// Variants timed over a full sweep (every wave: R blocks of 32 rows against all ncols columns):
//   direct<R>   every wave loads its own B fragments from global/L2 (ring of DEPTH loads in flight)
//   shared<R>   the workgroup's 4 waves share B through an LDS ring (one barrier per chunk)
// and checked against a CPU count of passing pairs on a small problem.
//
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 mfma_s1.hip -o mfma_s1 && ./mfma_s1 [ncols]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned int u32;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int or16(const v16i &d) {
  int a = d[0] | d[1] | d[2];
  int b = d[3] | d[4] | d[5];
  int c = d[6] | d[7] | d[8];
  int e = d[9] | d[10] | d[11];
  int f = d[12] | d[13] | d[14];
  a = a | b | c;
  e = e | f | d[15];
  return a | e;
}

__device__ __forceinline__ u32 count_neg(const v16i &d) {
  u32 n = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) n += (u32)__popcll(__builtin_amdgcn_ballot_w64(d[r] < 0));
  return n;
}

// ---------------------------------------------------------------------------------------
// direct: every wave streams E itself
// ---------------------------------------------------------------------------------------
template <int R, int DEPTH, int PAD>
__global__ __launch_bounds__(256) void s1_direct(const v4i *__restrict__ E, int ntiles, const v4i *__restrict__ Arows,
                                                 long long nwaves, unsigned long long *total, u32 *sink) {
  __shared__ u32 pad[PAD > 0 ? PAD : 1];
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (PAD > 0 && ntiles < 0) pad[threadIdx.x] = 1;          // keeps the padding allocated
  if (gw >= nwaves) return;
  v4i A[R];
#pragma unroll
  for (int r = 0; r < R; ++r) A[r] = Arows[(gw * R + r) * 64 + lane];
  v4i ring[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) ring[i] = E[(long long)(i < ntiles ? i : ntiles - 1) * 64 + lane];
  const v16i zero = {0};
  u32 cnt = 0;
  for (int t0 = 0; t0 < ntiles; t0 += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      const int t = t0 + i;
      const int tn = t + DEPTH < ntiles ? t + DEPTH : ntiles - 1;
      v16i d[R];
#pragma unroll
      for (int r = 0; r < R; ++r) d[r] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[r], ring[i], zero, 0, 0, 0);
      ring[i] = E[(long long)tn * 64 + lane];                // refilled once the MFMAs have read it
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (__builtin_amdgcn_ballot_w64(or16(d[r]) < 0)) cnt += count_neg(d[r]);
    }
  }
  if (lane == 0) {
    atomicAdd(total, (unsigned long long)cnt);
    if (sink && PAD > 0 && cnt == 0xFFFFFFFFu) sink[0] = pad[0];
  }
}

// direct, software pipelined: the MFMA of tile t is issued before the result of tile t-1 is reduced
template <int DEPTH, int PAD>
__global__ __launch_bounds__(256) void s1_direct_pipe(const v4i *__restrict__ E, int ntiles, const v4i *__restrict__ Arows,
                                                      long long nwaves, unsigned long long *total, u32 *sink) {
  __shared__ u32 pad[PAD > 0 ? PAD : 1];
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (PAD > 0 && ntiles < 0) pad[threadIdx.x] = 1;
  if (gw >= nwaves) return;
  const v4i A = Arows[gw * 64 + lane];
  v4i ring[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) ring[i] = E[(long long)(i < ntiles ? i : ntiles - 1) * 64 + lane];
  const v16i zero = {0};
  u32 cnt = 0;
  v16i dprev = zero;
  for (int t0 = 0; t0 < ntiles; t0 += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      const int t = t0 + i;
      const int tn = t + DEPTH < ntiles ? t + DEPTH : ntiles - 1;
      v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, ring[i], zero, 0, 0, 0);
      ring[i] = E[(long long)tn * 64 + lane];
      if (__builtin_amdgcn_ballot_w64(or16(dprev) < 0)) cnt += count_neg(dprev);   // tile t-1 (zero for t = 0)
      dprev = d;
    }
  }
  if (__builtin_amdgcn_ballot_w64(or16(dprev) < 0)) cnt += count_neg(dprev);
  if (lane == 0) {
    atomicAdd(total, (unsigned long long)cnt);
    if (sink && PAD > 0 && cnt == 0xFFFFFFFFu) sink[0] = pad[0];
  }
}

// ---------------------------------------------------------------------------------------
// shared: the workgroup stages chunks of CH tiles into an LDS ring of 3 slots; chunk c is readable
// after barrier c, its slot is refilled (chunk c+3's predecessor rule: slot of chunk c-1 is free
// after barrier c) with chunk c+2.
// ---------------------------------------------------------------------------------------
template <int R, int CH, int PAD>
__global__ __launch_bounds__(256) void s1_shared(const v4i *__restrict__ E, int ntiles, const v4i *__restrict__ Arows,
                                                 long long nwaves, unsigned long long *total, u32 *sink) {
  constexpr int SLOTS = 3;
  __shared__ v4i ring[SLOTS][CH * 64];
  __shared__ u32 pad[PAD > 0 ? PAD : 1];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  long long gw = (long long)blockIdx.x * 4 + wv;
  if (PAD > 0 && ntiles < 0) pad[threadIdx.x] = 1;
  const bool live = gw < nwaves;
  if (!live) gw = nwaves - 1;
  v4i A[R];
#pragma unroll
  for (int r = 0; r < R; ++r) A[r] = Arows[(gw * R + r) * 64 + lane];
  const int nchunks = (ntiles + CH - 1) / CH;
  const long long lastv = (long long)ntiles * 64 - 1;
  // each thread moves CH*64/256 uint4 per chunk
  constexpr int PER = CH * 64 / 256;
  auto issue = [&](int c, v4i (&st)[PER]) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      long long idx = (long long)c * CH * 64 + i * 256 + threadIdx.x;
      st[i] = E[idx < lastv ? idx : lastv];
    }
  };
  auto land = [&](int c, const v4i (&st)[PER]) {
#pragma unroll
    for (int i = 0; i < PER; ++i) ring[c % SLOTS][i * 256 + threadIdx.x] = st[i];
  };
  v4i s0[PER], s1[PER];
  issue(0, s0);
  land(0, s0);
  if (nchunks > 1) { issue(1, s1); land(1, s1); }
  const v16i zero = {0};
  u32 cnt = 0;
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();                       // chunk c (and c+1) landed; slot of chunk c-1 is free
    if (c + 2 < nchunks) issue(c + 2, s0);
    const v4i *slot = &ring[c % SLOTS][0];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const v4i b = slot[i * 64 + lane];
      if (live) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[r], b, zero, 0, 0, 0);
          if (__builtin_amdgcn_ballot_w64(or16(d) < 0)) cnt += count_neg(d);
        }
      }
    }
    if (c + 2 < nchunks) land(c + 2, s0);  // into slot (c+2)%3 == (c-1)%3, free since barrier c
  }
  if (lane == 0 && live) {
    atomicAdd(total, (unsigned long long)cnt);
    if (sink && PAD > 0 && cnt == 0xFFFFFFFFu) sink[0] = pad[0];
  }
}

// ---------------------------------------------------------------------------------------
// the VALU form for comparison: one lane = one column signature, 4 rows per step (xor + seeded bcnt)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s1_valu(const u32 *__restrict__ sigs, int ncols, const u32 *__restrict__ rowsig,
                                               const u32 *__restrict__ rownb, long long nwaves, unsigned long long *total) {
  __shared__ u32 rs[4][32], nb[4][32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long gw = (long long)blockIdx.x * 4 + wv;
  if (gw >= nwaves) return;
  if (lane < 32) { rs[wv][lane] = rowsig[gw * 32 + lane]; nb[wv][lane] = rownb[gw * 32 + lane]; }
  __builtin_amdgcn_wave_barrier();
  u32 cnt = 0;
  const int nt = (ncols + 127) / 128;
  for (int t = 0; t < nt; ++t) {
    const int c0 = t * 128 + lane, c1 = c0 + 64;
    const u32 s0 = sigs[c0 < ncols ? c0 : ncols - 1], s1 = sigs[c1 < ncols ? c1 : ncols - 1];
    for (int r = 0; r < 32; r += 4) {
      u32 any = 0, tt[8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        tt[2 * u] = __builtin_popcount(rs[wv][r + u] ^ s0) + nb[wv][r + u];
        tt[2 * u + 1] = __builtin_popcount(rs[wv][r + u] ^ s1) + nb[wv][r + u];
        any |= tt[2 * u] | tt[2 * u + 1];
      }
      if (__builtin_amdgcn_ballot_w64((int)any < 0)) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool ok = (int)tt[u] < 0 && ((u & 1) ? c1 : c0) < ncols;
          cnt += (u32)__popcll(__builtin_amdgcn_ballot_w64(ok));
        }
      }
    }
  }
  if (lane == 0) atomicAdd(total, (unsigned long long)cnt);
}

static unsigned long long rng_state = 88172645463325252ull;
static u32 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (u32)(rng_state >> 11); }

template <class F>
static float time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char **argv) {
  const int ncols = argc > 1 ? atoi(argv[1]) : 200000;
  const int nrows = argc > 2 ? atoi(argv[2]) : ncols;
  const int bound = argc > 3 ? atoi(argv[3]) : 4;
  const int ntiles = ((ncols + 31) / 32 + 7) / 8 * 8;    // whole groups of 8 tiles; padding columns are all-zero (D = 0)
  const long long nblk = (nrows + 31) / 32;              // 32-row blocks
  printf("stage-1 MFMA probe: %d rows x %d columns, bound %d, %d tiles, %lld row blocks\n", nrows, ncols, bound, ntiles, nblk);

  // signatures (31 bits)
  std::vector<u32> csig(ncols), rsig(nblk * 32);
  for (auto &s : csig) s = rnd() & 0x7FFFFFFFu;
  for (auto &s : rsig) s = rnd() & 0x7FFFFFFFu;
  // a few near pairs so that the slow branch is exercised
  for (int i = 0; i < 2000 && i < ncols; ++i) csig[(size_t)rnd() % ncols] = rsig[(size_t)rnd() % rsig.size()] ^ (1u << (rnd() % 31));

  // E in fragment order, byte 31 = 1 for real columns (padding columns stay all zero: D = 0)
  std::vector<unsigned char> E((size_t)ntiles * 1024, 0);
  for (int c = 0; c < ncols; ++c) {
    const int t = c / 32, cc = c % 32;
    for (int k = 0; k < 32; ++k) {
      const unsigned char v = k == 31 ? 1 : (csig[c] >> k) & 1;
      E[(size_t)t * 1024 + ((k / 16) * 32 + cc) * 16 + (k % 16)] = v;
    }
  }
  // A fragments per 32-row block: lane l = row l&31, bytes k = 16*(l>>5) ..; value = -(+-1) = (bit ? -1 : +1), k=31: bias
  std::vector<signed char> A((size_t)(nblk + 4) * 1024, 0);   // zero tail blocks: D = 0, never negative
  std::vector<u32> rownb(nblk * 32);
  for (long long b = 0; b < nblk; ++b)
    for (int r = 0; r < 32; ++r) {
      const u32 s = rsig[b * 32 + r];
      const int pa = __builtin_popcount(s);
      const bool real = b * 32 + r < nrows;
      const int bnd = real ? bound : 0;
      rownb[b * 32 + r] = 0u - (u32)bnd;
      for (int k = 0; k < 32; ++k) {
        int v = k == 31 ? pa - bnd : (((s >> k) & 1) ? -1 : 1);
        if (v < -128) v = -128;
        A[(size_t)b * 1024 + ((k / 16) * 32 + r) * 16 + (k % 16)] = (signed char)v;
      }
    }

  v4i *dE, *dA;
  u32 *dsig, *drsig, *drnb, *dsink;
  unsigned long long *dtot;
  CK(hipMalloc(&dE, E.size())); CK(hipMalloc(&dA, A.size()));
  CK(hipMalloc(&dsig, csig.size() * 4)); CK(hipMalloc(&drsig, rsig.size() * 4)); CK(hipMalloc(&drnb, rownb.size() * 4));
  CK(hipMalloc(&dtot, 8)); CK(hipMalloc(&dsink, 64));
  CK(hipMemcpy(dE, E.data(), E.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dsig, csig.data(), csig.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(drsig, rsig.data(), rsig.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(drnb, rownb.data(), rownb.size() * 4, hipMemcpyHostToDevice));

  // CPU reference on a sample of row blocks (exact count of passing pairs)
  unsigned long long ref = 0;
  const bool full_check = (double)nrows * ncols <= 4e9;
  if (full_check) {
    for (long long i = 0; i < nrows; ++i)
      for (int j = 0; j < ncols; ++j) ref += __builtin_popcount(rsig[i] ^ csig[j]) < bound;
    printf("cpu: %llu passing pairs\n", ref);
  }

  auto run = [&](const char *name, auto launch, long long nwaves, int reps) {
    unsigned long long got = 0;
    CK(hipMemset(dtot, 0, 8));
    launch();
    CK(hipMemcpy(&got, dtot, 8, hipMemcpyDeviceToHost));
    const float ms = time_ms(launch, reps);
    const double pairs = (double)nrows * ncols;
    printf("%-28s %9.3f ms  %.3e pairs/s  waves %lld  passing %llu%s\n", name, ms, pairs / (ms * 1e-3), nwaves, got,
           full_check ? (got == ref ? "  OK" : "  MISMATCH") : "");
    fflush(stdout);
  };

#define RUN_DIRECT(R, DEPTH, PAD)                                                                      \
  {                                                                                                     \
    const long long nw = (nblk + R - 1) / R;                                                            \
    int occ = 0;                                                                                        \
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, s1_direct<R, DEPTH, PAD>, 256, 0));           \
    char nm[64];                                                                                        \
    snprintf(nm, sizeof nm, "direct R=%d depth=%d occ=%d", R, DEPTH, occ);                              \
    run(nm, [&] { s1_direct<R, DEPTH, PAD><<<dim3((unsigned)((nw + 3) / 4)), dim3(256)>>>(dE, ntiles, dA, nw, dtot, dsink); }, nw, 5); \
  }
#define RUN_SHARED(R, CH, PAD)                                                                         \
  {                                                                                                     \
    const long long nw = (nblk + R - 1) / R;                                                            \
    int occ = 0;                                                                                        \
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, s1_shared<R, CH, PAD>, 256, 0));              \
    char nm[64];                                                                                        \
    snprintf(nm, sizeof nm, "shared R=%d chunk=%d occ=%d", R, CH, occ);                                 \
    run(nm, [&] { s1_shared<R, CH, PAD><<<dim3((unsigned)((nw + 3) / 4)), dim3(256)>>>(dE, ntiles, dA, nw, dtot, dsink); }, nw, 5); \
  }
#define RUN_PIPE(DEPTH, PAD)                                                                            \
  {                                                                                                     \
    const long long nw = nblk;                                                                          \
    int occ = 0;                                                                                        \
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, s1_direct_pipe<DEPTH, PAD>, 256, 0));         \
    char nm[64];                                                                                        \
    snprintf(nm, sizeof nm, "direct-pipe depth=%d occ=%d", DEPTH, occ);                                 \
    run(nm, [&] { s1_direct_pipe<DEPTH, PAD><<<dim3((unsigned)((nw + 3) / 4)), dim3(256)>>>(dE, ntiles, dA, nw, dtot, dsink); }, nw, 5); \
  }
  // the A buffer holds nblk blocks; waves with R > 1 read R consecutive blocks (pad the tail)
  RUN_DIRECT(1, 4, 0)
  RUN_DIRECT(1, 8, 0)
  RUN_DIRECT(1, 4, 6144)      // ~24 KB of LDS per workgroup: 6 workgroups per CU
  RUN_DIRECT(1, 4, 10240)     // ~40 KB: 4 workgroups per CU (the engine's kNN occupancy)
  RUN_DIRECT(2, 4, 0) RUN_DIRECT(2, 8, 0) RUN_DIRECT(2, 4, 10240)
  RUN_DIRECT(4, 4, 0) RUN_DIRECT(4, 4, 10240)
  RUN_PIPE(4, 0) RUN_PIPE(4, 6144) RUN_PIPE(4, 10240) RUN_PIPE(4, 20480)
  RUN_SHARED(1, 4, 0)
  RUN_SHARED(1, 8, 0)
  RUN_SHARED(1, 4, 7168)      // ring 12 KB + 28 KB pad: 4 workgroups per CU
  RUN_SHARED(2, 4, 0) RUN_SHARED(2, 4, 7168)
  {
    const long long nw = nblk;
    run("valu xor+bcnt (4 rows/step)", [&] { s1_valu<<<dim3((unsigned)((nw + 3) / 4)), dim3(256)>>>(dsig, ncols, drsig, drnb, nw, dtot); }, nw, 3);
  }
  return 0;
}
