// C ABI of the hot path (include/prograph_hip.h) + the O(N*L) helper kernels:
// plane packing, exclusive scan, flag compaction and the fused 1xN indexing pass.
#include "pg_common.h"
#include "pg_mm.h"
#include "../../include/prograph_hip.h"

#include <dlfcn.h>
#include <math.h>
#include <rccl/rccl.h>   // types only: the library is resolved at run time (dlopen), there is no link dependency
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>

static thread_local char g_err[256] = "";

void pg_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }
static int fail(int code, const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
static int hipfail(hipError_t e, const char *where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return (int)e;
}
static int launched(int rc, const char *where) {
  if (rc != 0) return hipfail((hipError_t)rc, where);
  return 0;
}

// stage-1 lower-bound filter of the engine: adaptive by default; PG_LB_FILTER=0 disables, =2 forces it on
static int lb_filter_mode() { return getenv("PG_LB_FILTER") ? atoi(getenv("PG_LB_FILTER")) : 1; }

// compute units of the CURRENT device (cached per device: a process may drive several)
#define PG_MAX_DEVICES 64
static int cu_count() {
  static std::atomic<int> cus[PG_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  const bool cached = dev >= 0 && dev < PG_MAX_DEVICES;
  if (cached && cus[dev].load(std::memory_order_relaxed) > 0) return cus[dev].load(std::memory_order_relaxed);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  if (cached) cus[dev].store(prop.multiProcessorCount, std::memory_order_relaxed);
  return prop.multiProcessorCount;
}

// waves per CU the all-pairs engine is sized for (PG_WAVES_PER_CU overrides; multiples of 4)
static int waves_per_cu() {
  const char *e = getenv("PG_WAVES_PER_CU");
  int w = e ? atoi(e) : 32;
  if (w < 4) w = 4;
  if (w > 128) w = 128;
  return (w / 4) * 4;
}

extern "C" {

int pg_version(void) { return PG_ABI_VERSION; }
const char *pg_last_error(void) { return g_err; }

int pg_device_info(int *cus, int *wave, char *arch, int arch_len) {
  int dev = 0;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hipfail(e, "hipGetDevice");
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return hipfail(e, "hipGetDeviceProperties");
  if (cus) *cus = prop.multiProcessorCount;
  if (wave) *wave = prop.warpSize;
  if (arch && arch_len > 0) snprintf(arch, (size_t)arch_len, "%s", prop.gcnArchName);
  return 0;
}

int64_t pg_npad(int64_t n) { return n <= 0 ? 256 : ((n + 255) / 256) * 256; }
int pg_ngroups(int l) { return l <= 0 ? 1 : (l + 31) / 32; }
int pg_nchunks(int l, int bits) { return (pg_ngroups(l) * bits + 3) / 4; }
int64_t pg_planes_bytes(int64_t n, int l, int bits) { return ((int64_t)pg_nchunks(l, bits) * 16 + PG_AUX_BYTES) * pg_npad(n); }
// launch-private device state of an all-pairs call, sized by the row count:
//   [0, 64)      pass counters of the engine's persistent waves (words 0 and 8: a call's main launch and its gated 32-row
//                alternative) and, word 1, the number of evicted rows                        - zeroed by the call
//   [64, 576)    (reserved; zeroed with the counters)
//   [576, 640)   the probe's decision words (gates)
//   [640, ...)   the probe's counts (8 * PG_PROBE_ROWS * PG_PROBE_WAVES bytes)
//   from PG_WS_PARTIAL on, launches of more than PG_SPLIT_MIN_ROWS rows: the lists of the column pieces - at most
//                PG_SPLIT_MAX_ROWS rows x 8 pieces (or half of them x 16) x PG_MM_KL keys; then the evicted rows (below)
#define PG_WS_FLAGS 64
#define PG_WS_GATES 576
#define PG_WS_COUNTS 640
#define PG_WS_PARTIAL (PG_WS_COUNTS + 8 * 64 * 128)
#define PG_SPLIT_MAX_ROWS 8192       // 128 row blocks of 64
#define PG_SPLIT_MIN_ROWS 65536     // (launches up to here never split)
#define PG_WS_PARTIAL_BYTES ((int64_t)PG_SPLIT_MAX_ROWS * 8 * PG_MM_KL * 4)
// ... and behind that (every launch) one word per row: the rows a kNN launch of the MFMA engine evicts (NsqParams::mmEvict;
// their number: word 1 of the counter line)
static int64_t ws_evict_offset(int64_t nrows) { return PG_WS_PARTIAL + (nrows > PG_SPLIT_MIN_ROWS ? PG_WS_PARTIAL_BYTES : 0); }
int64_t pg_workspace_bytes(int64_t nrows) { return ws_evict_offset(nrows) + 4 * (nrows > 0 ? nrows : 0) + 64; }

}  // extern "C"

// =======================================================================================
// pack: row-major tokens -> bit-sliced records in chunk-major order (one thread per sequence)
// =======================================================================================
// lut / tokOut (pg_pack_bytes): src holds the fixed-width BYTES of the sequences; token = lut[byte]
// (the reference's letter table, prograph/prograph.py:127,454-474: unknown bytes and padding -> 0), written row-major
// to tokOut as well when that is not NULL.
template <typename T, int B>
__global__ __launch_bounds__(256) void pg_pack_kernel(const T *__restrict__ src, long long n, int l, long long ld,
                                                      const long long *__restrict__ rows, u32 *__restrict__ planes,
                                                      long long npad, int ng, int nq, u32 *flags,
                                                      const unsigned char *__restrict__ lut = nullptr,
                                                      unsigned char *__restrict__ tokOut = nullptr) {
  const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
  if (s >= npad) return;
  const T *row = nullptr;
  if (s < n) row = src + (rows ? rows[s] : s) * ld;
  u32 bad = 0, fold[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sigw[2] = {0, 0};
  for (int g = 0; g < ng; ++g) {
    u32 pl[B];
#pragma unroll
    for (int p = 0; p < B; ++p) pl[p] = 0;
    if (row) {
      for (int j = 0; j < 32; ++j) {
        const int pos = g * 32 + j;
        if (pos < l) {
          long long v = (long long)row[pos];
          if (lut) {
            v = lut[(unsigned char)v];
            if (tokOut) tokOut[s * (long long)l + pos] = (unsigned char)v;
          }
          if (v < 0 || v >= (1ll << B)) bad = 1u;
#pragma unroll
          for (int p = 0; p < B; ++p) pl[p] |= (u32)((v >> p) & 1) << j;
        }
      }
    }
#pragma unroll
    for (int p = 0; p < B; ++p) fold[p] ^= pl[p];         // plane folds: XOR of each plane's group words
    sigw[g & 1] ^= pl[0];                                 // plane 0, even / odd groups: the 64 bits behind the signature
#pragma unroll
    for (int p = 0; p < B; ++p) {
      const int w = p * ng + g;               // plane-major record order
      planes[((long long)(w >> 2) * npad + s) * 4 + (w & 3)] = pl[p];
    }
  }
  for (int w = ng * B; w < nq * 4; ++w) planes[((long long)(w >> 2) * npad + s) * 4 + (w & 3)] = 0;
  // Signature section (after the nq chunk arrays, 32 * npad bytes): the column operands of the filter MFMAs
  // (pg_mm.h), v_mfma_f32_32x32x64_f8f6f4 with FP4 elements: 1.0 (0x2) per set bit of the 54-bit signature,
  // 1.0 in the ten bias slots k = 54..63 (0 for padding sequences: they never pass); per 32 sequences one
  // 1 KiB block in fragment order: uint4 [tile][h * 32 + c] = elements k = 32h .. 32h+31 of sequence 32 * tile + c.
  {
    const unsigned long long sig = pg_sig54(sigw[0], sigw[1]);
    const u32 lo = (u32)sig, hi = (u32)(sig >> 32) | (row ? 0xFFC00000u : 0u);
    uint4 *e = reinterpret_cast<uint4 *>(planes + (long long)nq * npad * 4) + (s >> 5) * 64 + (s & 31);
    e[0] = make_uint4(pg_nib8(lo) << 1, pg_nib8(lo >> 8) << 1, pg_nib8(lo >> 16) << 1, pg_nib8(lo >> 24) << 1);
    e[32] = make_uint4(pg_nib8(hi) << 1, pg_nib8(hi >> 8) << 1, pg_nib8(hi >> 16) << 1, pg_nib8(hi >> 24) << 1);
  }
  // Fold section (after the signatures, 32 * npad bytes): two uint4 arrays, planes 0..3 and 4..7 of every
  // sequence's plane folds (unused planes 0): the operands of the dense form's folded-exact bound.
  {
    uint4 *f = reinterpret_cast<uint4 *>(planes + (long long)nq * npad * 4) + npad * 2;
    f[s] = make_uint4(fold[0], fold[1], fold[2], fold[3]);
    f[npad + s] = make_uint4(fold[4], fold[5], fold[6], fold[7]);
  }
  if (bad) atomicOr(flags, bad);
}

// =======================================================================================
// exclusive scan (u32 counts or u8 flags -> i64), three small kernels
// =======================================================================================
#define PG_SCAN_TILE 2048   // elements per block (256 threads x 8)

template <typename T>
__device__ __forceinline__ long long scan_val(const T *in, long long i, long long n) {
  return i < n ? (long long)(in[i] != 0 ? (sizeof(T) == 1 ? 1 : in[i]) : 0) : 0;
}

__device__ __forceinline__ long long wave_incl_scan(long long v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    long long t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns the block total
__device__ __forceinline__ long long block_excl_scan(long long v, long long &total) {
  __shared__ long long wsum[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long inc = wave_incl_scan(v, lane);
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  long long base = 0;
  for (int w = 0; w < wv; ++w) base += wsum[w];
  total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return base + inc - v;
}

template <typename T>
__global__ __launch_bounds__(256) void pg_scan_partials(const T *__restrict__ in, long long n, long long *partials) {
  const long long b0 = (long long)blockIdx.x * PG_SCAN_TILE;
  long long s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += scan_val(in, b0 + threadIdx.x * 8 + j, n);
  long long total;
  block_excl_scan(s, total);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void pg_scan_single(long long *partials, long long nb) {
  long long carry = 0;
  for (long long c0 = 0; c0 < nb; c0 += 256) {
    const long long i = c0 + threadIdx.x;
    const long long v = i < nb ? partials[i] : 0;
    long long total;
    const long long ex = block_excl_scan(v, total);
    if (i < nb) partials[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) partials[nb] = carry;
}

// MODE 0: out[i] = exclusive prefix (and out[n] = total); MODE 1: out[prefix] = i where in[i] != 0
template <typename T, int MODE>
__global__ __launch_bounds__(256) void pg_scan_apply(const T *__restrict__ in, long long n, const long long *partials,
                                                     long long nb, long long *out, long long *count) {
  const long long b0 = (long long)blockIdx.x * PG_SCAN_TILE;
  long long v[8], s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { v[j] = scan_val(in, b0 + threadIdx.x * 8 + j, n); s += v[j]; }
  long long total;
  long long ex = block_excl_scan(s, total) + partials[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const long long i = b0 + threadIdx.x * 8 + j;
    if (i < n) {
      if (MODE == 0) out[i] = ex;
      else if (v[j]) out[ex] = i;
    }
    ex += v[j];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (MODE == 0) out[n] = partials[nb];
    if (count) *count = partials[nb];
  }
}

// =======================================================================================
// fused 1xN indexing pass (one thread per sequence; position masks are bitmasks per group)
// =======================================================================================
template <int B>
__global__ __launch_bounds__(256) void pg_index_kernel(const u32 *__restrict__ planes, long long n, long long npad, int ng,
                                                       long long ref, const u32 *__restrict__ want, int posMode,
                                                       const u32 *__restrict__ posMask, const u32 *__restrict__ notMask,
                                                       unsigned char *distOut, u64 *hist, unsigned char *flags) {
  __shared__ u32 lh[256];
  lh[threadIdx.x] = 0;
  __syncthreads();
  const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
  if (s < n) {
    u32 d = 0, anyP = 0, anyNot = 0;
    bool allP = true;
    for (int g = 0; g < ng; ++g) {
      u32 t = 0;
#pragma unroll
      for (int p = 0; p < B; ++p) {
        const int w = p * ng + g;
        const long long base = (long long)(w >> 2) * npad;
        t |= planes[(base + s) * 4 + (w & 3)] ^ planes[(base + ref) * 4 + (w & 3)];
      }
      d += __builtin_popcount(t);
      if (posMode) {
        const u32 pm = posMask[g], nm = notMask[g];
        anyP |= t & pm;
        allP = allP && ((t & pm) == pm);
        anyNot |= t & nm;
      }
    }
    if (distOut) distOut[s] = (unsigned char)d;
    if (hist) atomicAdd(&lh[d & 255u], 1u);
    if (flags) {
      bool ok = want ? ((want[(d >> 5) & 7u] >> (d & 31u)) & 1u) != 0 : true;
      if (posMode == 1) ok = ok && (anyP != 0) && (anyNot == 0);
      if (posMode == 2) ok = ok && allP && (anyNot == 0);
      flags[s] = ok ? 1 : 0;
    }
  }
  __syncthreads();
  if (hist && lh[threadIdx.x]) atomicAdd((unsigned long long *)&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

// =======================================================================================
// CSR consumers: per-row reductions the graph analytics need
//   deg[r]  = sum_j w_rj            (degree, prograph/prograph.py:797-822)
//   sf[r]   = sum_j f[col_j]        (local_variance, :924-946:  mean_j (f_r - f_j) = f_r - sf/cnt)
//   swf[r]  = sum_j w_rj f[col_j]   (dirichlet, :899-922:  f^T L f = sum_r f_r (deg_r f_r - swf_r))
// one wave per row, coalesced reads of the row's slice, DPP/shuffle tree reduction.
// =======================================================================================
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}

__global__ __launch_bounds__(256) void pg_csr_row_stats_kernel(const long long *__restrict__ indptr,
                                                               const int *__restrict__ indices,
                                                               const unsigned char *__restrict__ w8,
                                                               const float *__restrict__ wf, long long nrows,
                                                               long long row0, const double *__restrict__ f, double *deg,
                                                               double *sf, double *swf, double *selfw, double *colsum) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const long long a = indptr[row], b = indptr[row + 1];
  double d = 0, s1 = 0, s2 = 0, sw = 0;
  for (long long e = a + lane; e < b; e += 64) {
    const double w = w8 ? (double)w8[e] : (wf ? (double)wf[e] : 1.0);
    const int col = indices[e];
    const double fj = f ? f[col] : 0.0;
    d += w; s1 += fj; s2 += w * fj;
    if (col == row0 + row) sw += w;                       // the row's own node among its neighbours
    if (colsum) atomicAdd(&colsum[col], w);               // in-degree (column sums of the adjacency)
  }
  d = wave_sum(d); s1 = wave_sum(s1); s2 = wave_sum(s2); sw = wave_sum(sw);
  if (lane == 0) {
    if (deg) deg[row] = d;
    if (sf) sf[row] = s1;
    if (swf) swf[row] = s2;
    if (selfw) selfw[row] = sw;
  }
}

// =======================================================================================
// host side of the ABI
// =======================================================================================
// `l` may be the packed width rounded up to whole 32-token groups (256 for 8 groups); only
// pg_pack_planes, which sees the real tokens, enforces L <= 255 (a distance must fit uint8).
static int check_l(int l, int bits = 8, bool exact = false) {
  if (l <= 0) return fail(PG_E_BADARG, "sequence length must be positive");
  const int maxg = bits == 5 ? PG_MAX_L_5BIT / 32 + 1 : PG_MAX_L / 32;
  if ((l + 31) / 32 > maxg || (exact && l > (bits == 5 ? PG_MAX_L_5BIT : PG_MAX_L)))
    return fail(PG_E_TOOLONG, "sequence length exceeds the native limit (128 tokens, 255 with 5 bit planes)");
  return 0;
}
static int check_bits(int b) {
  if (b != PG_BITS_5 && b != PG_BITS_8) return fail(PG_E_BADARG, "bits must be 5 or 8");
  return 0;
}

// comp(d, eps) & (d > 0) as an integer interval [lo, lo+span]; an empty interval is encoded
// so that (d - lo) <= span is false for every d in 0..255
static void eps_interval(int cmp, double eps, u32 *lo, u32 *span) {
  const long long INF = 1000000;
  long long l = 1, h = INF;
  switch (cmp) {
    case PG_CMP_LE: h = (long long)floor(eps); break;
    case PG_CMP_LT: h = (long long)ceil(eps) - 1; break;
    case PG_CMP_EQ:
      if (floor(eps) == eps) { l = (long long)eps; h = (long long)eps; } else { l = 1; h = 0; }
      break;
    case PG_CMP_GE: l = (long long)ceil(eps); break;
    case PG_CMP_GT: l = (long long)floor(eps) + 1; break;
  }
  if (l < 1) l = 1;
  if (h > INF) h = INF;
  if (h < l) { *lo = 0xFFFFFF00u; *span = 0; return; }
  *lo = (u32)l;
  *span = (u32)(h - l);
}

// ---------------------------------------------------------------------------------------
// Data probe + decision (no host round trip): pg_probe_kernel counts, for PG_PROBE_ROWS sample rows of the launch, the
// columns nearer than the kNN cap and the columns inside the eps interval; pg_decide_kernel turns them into the gate
// words the alternative kernels of the launch test at their start (NsqParams::gate):
//   gate[0]  kNN engine: 0 = MFMA engine, 1 = VALU engine.  Unclustered data (fewer than half of the sample rows have
//            k + 1 columns below the cap: the signature filter never gets a useful bound, every distance is needed)
//            runs ~8 % faster on the VALU engine's direct form (profiles/r03_engine_landscape.txt: random N = 200k);
//            2 = MFMA engine with 32-row passes although 64-row passes were planned: ONE cluster (more than half of
//            all pairs below the cap) lives in the folded form, where the finer pass granularity wins (11.7 vs 12.5 ms)
//   gate[1]  whole-square eps graph: 0 = every unordered pair once (symmetric slots), 1 = rectangular VALU sweep.
//            Dense graphs (more than 4 % of all pairs match) pay more for the symmetric path's atomics and
//            unsorted back parts than it saves (one cluster, eps <= 2: 39.6 vs 29.4 ms)
// ---------------------------------------------------------------------------------------
#define PG_PROBE_ROWS 64        // sample rows
#define PG_PROBE_WAVES 128      // waves per sample row: every 8th column tile (PG_PROBE_STRIDE), one turn of four tiles each at N = 200k
                                // (a turn = one L2 / Infinity-Cache round trip, ~2 us: 1 x 1 wave 206 us, 64 x 4 67 us, 64 x 32 22 us)
#define PG_PROBE_MIN_N 65536    // launches below this are not probed: the probe (~20 us) would show, and the size rules hold
// one workgroup: 16 threads per sample row (a quarter wave) sum the row's per-wave counts, the wave of a row then the
// workgroup combine the verdicts
__global__ __launch_bounds__(1024) void pg_decide_kernel(const u32 *counts, int nsample, int wavesPerRow, u32 need, long long ncols,
                                                          int force, u32 *gate) {
  __shared__ u32 sClustered[16];
  __shared__ unsigned long long sEps[16], sNear[16];
  const int tid = threadIdx.x, s = tid >> 4, sub = tid & 15;
  u32 nearS = 0, epsS = 0;
  if (s < nsample)
    for (int w = sub; w < wavesPerRow; w += 16) {
      nearS += counts[2 * (s * wavesPerRow + w)];
      epsS += counts[2 * (s * wavesPerRow + w) + 1];
    }
  for (int o = 8; o > 0; o >>= 1) {
    nearS += (u32)__shfl_xor((int)nearS, o);
    epsS += (u32)__shfl_xor((int)epsS, o);
  }
  // (counts are over every PG_PROBE_STRIDE-th tile)
  u32 clustered = (sub == 0 && s < nsample && nearS * PG_PROBE_STRIDE >= need) ? 1u : 0u;
  unsigned long long eps = (sub == 0 && s < nsample) ? (unsigned long long)epsS * PG_PROBE_STRIDE : 0ull;
  unsigned long long nearAll = (sub == 0 && s < nsample) ? (unsigned long long)nearS * PG_PROBE_STRIDE : 0ull;
  for (int o = 32; o > 0; o >>= 1) {
    clustered += (u32)__shfl_xor((int)clustered, o);
    eps += (unsigned long long)__shfl_xor((long long)eps, o);
    nearAll += (unsigned long long)__shfl_xor((long long)nearAll, o);
  }
  if ((tid & 63) == 0) { sClustered[tid >> 6] = clustered; sEps[tid >> 6] = eps; sNear[tid >> 6] = nearAll; }
  __syncthreads();
  if (tid == 0) {
    clustered = 0; eps = 0; nearAll = 0;
    for (int i = 0; i < 16; ++i) { clustered += sClustered[i]; eps += sEps[i]; nearAll += sNear[i]; }
    const bool oneCluster = 2ull * nearAll > (unsigned long long)nsample * (unsigned long long)ncols;
    gate[0] = force >= 0 ? (u32)(force & 1) + ((force & 4) ? 2u : 0u) : (2u * clustered >= (u32)nsample ? (oneCluster ? 2u : 0u) : 1u);
    gate[1] = force >= 0 ? (u32)((force >> 1) & 1) : (eps * 25ull > (unsigned long long)nsample * (unsigned long long)ncols ? 1u : 0u);
  }
}
// kNN of rows swept in column pieces (NsqParams::mmPieces): a row's `pieces` (<= 16) sorted lists of k + 1 keys each -> its
// k + 1 smallest keys, ranks 1..k written out (the reference drops sorted rank 0, prograph/prograph.py:761-763).  Sixteen
// lanes per row, lane b holds the head of list b: k + 1 rounds of "smallest head wins and advances" (keys are
// distance << 24 | column: unique, except 0xFFFFFFFF = no entry).  Four rows per wave.
// A piece's list is exact below the optimistic cap only (NsqParams::mmPieces): a row whose merged (k+1)-th distance is
// not below `cap` joins the evicted rows (NsqParams::mmEvict: pg_knn_rows_kernel finishes them).
__global__ __launch_bounds__(PG_WG_THREADS) void pg_knn_merge_kernel(const u32 *partial, long long nrows, int pieces, int k, int *idx,
                                                                     unsigned char *dist, u32 cap, u32 *evictCount, u32 *evictRows,
                                                                     long long rowAbs0, const u32 *gate, u32 gateMask) {
  __shared__ u32 keys[PG_WG_WAVES][4][16 * PG_MM_KL];
  if (gate && !((gateMask >> __builtin_nontemporal_load(gate)) & 1u)) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, b = lane & 15;
  const long long row0 = ((long long)blockIdx.x * PG_WG_WAVES + wv) * 4;
  if (row0 >= nrows) return;                               // (whole wave; no workgroup barrier below)
  const int k1 = k + 1, n = pieces * k1;
  for (int r = 0; r < 4; ++r)
    if (row0 + r < nrows)
      for (int e = lane; e < n; e += 64) keys[wv][r][e] = partial[(row0 + r) * n + e];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const long long row = row0 + g;
  const bool mine = row < nrows && b < pieces;
  int pos = 0;
  u32 head = mine ? keys[wv][g][b * k1] : 0xFFFFFFFFu;
  for (int r = 0; r <= k; ++r) {
    u32 m = head;
    for (int o = 8; o > 0; o >>= 1) {
      const u32 x = (u32)__shfl_xor((int)m, o, 16);
      m = x < m ? x : m;
    }
    if (b == 0 && row < nrows) {
      if (r >= 1) {
        idx[row * k + r - 1] = m == 0xFFFFFFFFu ? -1 : (int)(m & 0x00FFFFFFu);
        dist[row * k + r - 1] = (unsigned char)(m >> 24);
      }
      if (r == k && (m >> 24) >= cap) evictRows[atomicAdd(evictCount, 1u)] = (u32)(rowAbs0 + row);   // (0xFFFFFFFF, no entry, reads 255)
    }
    if (mine && head == m && m != 0xFFFFFFFFu) {           // the winner (unique key) moves on in its list
      ++pos;
      head = pos < k1 ? keys[wv][g][b * k1 + pos] : 0xFFFFFFFFu;
    }
  }
}

typedef int (*probe_fn)(int, const ProbeParams &, hipStream_t);
typedef int (*knn_rows_fn)(int, const KnnRowsParams &, int, hipStream_t);
static const knn_rows_fn kKnnRows[8] = {pg_launch_knn_rows_g1, pg_launch_knn_rows_g2, pg_launch_knn_rows_g3, pg_launch_knn_rows_g4,
                                        pg_launch_knn_rows_g5, pg_launch_knn_rows_g6, pg_launch_knn_rows_g7, pg_launch_knn_rows_g8};
static const probe_fn kProbe[8] = {pg_launch_probe_g1, pg_launch_probe_g2, pg_launch_probe_g3, pg_launch_probe_g4,
                                   pg_launch_probe_g5, pg_launch_probe_g6, pg_launch_probe_g7, pg_launch_probe_g8};
// PG_PROBE=0: no probe (the size rules alone); PG_GATE_FORCE=<bits>: the decision is forced (tests): bit 0 gate[0], bit 1 gate[1]
static bool probe_enabled() {
  const char *e = getenv("PG_PROBE");
  return !(e && atoi(e) == 0) && !getenv("PG_ENGINE") && !getenv("PG_ENGINE_MIN_ROWS");
}
// zeroes and fills workspace[64, ...): returns the gate words through *gate
static int run_probe(const NsqParams &e, int l, int bits, u32 near, u32 lo, u32 span, u32 need, void *workspace, hipStream_t s,
                     const u32 **gate) {
  if (!workspace) return fail(PG_E_BADARG, "workspace required (pg_workspace_bytes)");
  u32 *gates = (u32 *)((char *)workspace + PG_WS_GATES), *counts = (u32 *)((char *)workspace + PG_WS_COUNTS);
  static_assert(PG_WS_COUNTS + 8 * PG_PROBE_ROWS * PG_PROBE_WAVES <= PG_WS_PARTIAL, "pg_workspace_bytes");
  ProbeParams pp;
  pp.rowPlanes = e.rowPlanes; pp.colPlanes = e.colPlanes; pp.rowNpad = e.rowNpad; pp.colNpad = e.colNpad;
  pp.row0 = e.row0; pp.nrows = e.nrows; pp.ncols = e.ncols;
  pp.nsample = e.nrows < PG_PROBE_ROWS ? (int)e.nrows : PG_PROBE_ROWS;
  pp.wavesPerRow = PG_PROBE_WAVES;
  if (const char *es = getenv("PG_PROBE_S")) { if (atoi(es) > 0 && atoi(es) <= PG_PROBE_ROWS) pp.nsample = atoi(es); }   // (experiments)
  if (const char *ew = getenv("PG_PROBE_W")) { if (atoi(ew) > 0 && atoi(ew) <= PG_PROBE_WAVES) pp.wavesPerRow = atoi(ew); }
  pp.near = near; pp.lo = lo; pp.span = span; pp.counts = counts;
  if (int rc = launched(kProbe[pg_ngroups(l) - 1](bits, pp, s), "pg_probe_kernel")) return rc;
  const int force = getenv("PG_GATE_FORCE") ? atoi(getenv("PG_GATE_FORCE")) : -1;
  // (counts of a short sample: pg_decide_kernel reads counts[nsample + s] - same layout as the probe wrote)
  pg_decide_kernel<<<dim3(1), dim3(1024), 0, s>>>(counts, pp.nsample, pp.wavesPerRow, need, e.ncols, force, gates);   // (PG_PROBE_ROWS <= 64)
  if (int rc = launched((int)hipGetLastError(), "pg_decide_kernel")) return rc;
  *gate = gates;
  return 0;
}

typedef int (*nsq_fn)(int, int, const NsqParams &, int, hipStream_t);
typedef int (*dense_fn)(int, const DenseParams &, hipStream_t);
typedef int (*compact_fn)(int, const CompactParams &, hipStream_t);
typedef int (*occ_fn)(int, int);
static const occ_fn kNsqOcc[8] = {pg_occ_nsq_g1, pg_occ_nsq_g2, pg_occ_nsq_g3, pg_occ_nsq_g4,
                                  pg_occ_nsq_g5, pg_occ_nsq_g6, pg_occ_nsq_g7, pg_occ_nsq_g8};
// resident waves per SIMD (= workgroups per CU) of an engine instance, asked once from the runtime
static int nsq_occupancy(int groups, int mode, int bits) {
  static std::atomic<int> cache[8][3][2];                  // (a property of the code object: the same on every gfx950)
  std::atomic<int> &c = cache[groups - 1][mode][bits == 8];
  int v = c.load(std::memory_order_relaxed);
  if (v == 0) {
    const int n = kNsqOcc[groups - 1](mode, bits);
    v = n < 1 ? 4 : (n > 8 ? 8 : n);
    c.store(v, std::memory_order_relaxed);
  }
  return v;
}
static const nsq_fn kNsq[8] = {pg_launch_nsq_g1, pg_launch_nsq_g2, pg_launch_nsq_g3, pg_launch_nsq_g4,
                               pg_launch_nsq_g5, pg_launch_nsq_g6, pg_launch_nsq_g7, pg_launch_nsq_g8};
static const dense_fn kDense[8] = {pg_launch_dense_g1, pg_launch_dense_g2, pg_launch_dense_g3, pg_launch_dense_g4,
                                   pg_launch_dense_g5, pg_launch_dense_g6, pg_launch_dense_g7, pg_launch_dense_g8};
static const compact_fn kCompact[8] = {pg_launch_compact_g1, pg_launch_compact_g2, pg_launch_compact_g3,
                                       pg_launch_compact_g4, pg_launch_compact_g5, pg_launch_compact_g6,
                                       pg_launch_compact_g7, pg_launch_compact_g8};

// Static, even split of the rows over the waves: every row costs the same (one sweep over all
// columns), so equal row counts are equal work.  The kernel is VALU-issue bound and a workgroup
// puts one wave on each SIMD of its CU, so the run time follows the busiest SIMD:
//     makespan ~ (rows_per_wave + 1) * sum over rounds of T(waves in the round),
//     n = ceil(workgroups / CUs) waves per SIMD run in rounds of `occ` residents (the instance's
//     occupancy), T(1..8) = 2.2, 3.1, 3.7, 4.3, 5.1, 6, 7, 8
// (the +1 is the per-wave cost of streaming the column tiles, ~1 row-equivalent; T(m) reflects
// that fewer than ~5 waves cannot keep the VALU issuing: one wave alone issues under half of the
// time.  Fitted to tools/sweep_rpw.py at N = 50k and 200k, within ~8 %).  rows_per_wave is chosen
// among the multiples of 4 up to one pass (PG_RB) to minimise that; ties go to the larger value
// (fewer column re-reads).  More than PG_RB rows per wave (only with the PG_WAVES_PER_CU /
// PG_ROWS_PER_WAVE overrides) are walked in passes of nearly equal size.
// Second term: every wave streams the whole column matrix (ncols * Q * 16 bytes).  While that fits
// the L2s the stream is hidden; beyond (cfg4: 48 MB) the sweep becomes bound by L2-miss traffic
// served from the Infinity Cache: measured 11.5-15 TB/s aggregate at 48 MB, ~18-20 at 24 MB
// (tools/sweep_rpw_big.py; N = 1M, 125k rows: 16 rows per wave 28.8 ms, 24 rows 20.5 ms).  The plan
// minimises max(VALU makespan, streamed bytes / bandwidth) with bandwidth = 13 TB/s * (48 MB / bytes)^0.65
// (capped at 4x) and 1.21e-11 s per makespan unit and column.
static int plan_rows(int64_t nrows, NsqParams *p, int *grid, int occ, double recBytes, int maxRows = PG_RB) {
  const int cus = cu_count();
  if (cus <= 0) return fail(PG_E_NODEV, "no HIP device");
  long long rpw = 4;
  if (getenv("PG_WAVES_PER_CU")) {
    const long long maxWaves = (long long)cus * waves_per_cu();
    rpw = (nrows + maxWaves - 1) / maxWaves;
    if (rpw < 1) rpw = 1;
    // the filtered sweep takes rows four at a time: whole groups waste no stage-1 work
    if (rpw >= 4) rpw = (rpw + 3) / 4 * 4;
  } else {
    static const long long kRound[9] = {0, 22, 31, 37, 43, 51, 60, 70, 80};   // T(m) x 10
    if (occ < 1) occ = 1;
    if (occ > 8) occ = 8;
    const double matBytes = recBytes * (double)p->ncols;
    double bw = 13e12;
    if (matBytes > 0 && matBytes < 48e6) bw *= fmin(4.0, pow(48e6 / matBytes, 0.65));
    double best = -1;
    for (long long r = 4; r <= maxRows; r += 4) {
      const long long nw = (nrows + r - 1) / r;
      const long long wgs = (nw + PG_WG_WAVES - 1) / PG_WG_WAVES;
      const long long n = (wgs + cus - 1) / cus;             // waves the busiest SIMD runs
      const double valu = 1.21e-11 * (double)(((n / occ) * kRound[occ] + kRound[n % occ]) * (r + 1));
      const double stream = (double)nw * recBytes / bw;      // both per column
      const double cost = valu > stream ? valu : stream;
      if (best < 0 || cost <= best * 1.0000001) { best = cost; rpw = r; }
    }
  }
  if (const char *e = getenv("PG_ROWS_PER_WAVE")) { if (atoi(e) > 0) rpw = atoi(e); }   // tuning sweeps
  const long long waves = (nrows + rpw - 1) / rpw;
  if (getenv("PG_DEBUG_PLAN")) fprintf(stderr, "[pg plan] rows=%lld occ=%d rows_per_wave=%lld waves=%lld\n", (long long)nrows, occ, rpw, waves);
  const long long passes = (rpw + maxRows - 1) / maxRows;
  p->rowsPerWave = (int)rpw;
  long long rpp = (rpw + passes - 1) / passes;
  if (rpp >= 4) rpp = (rpp + 3) / 4 * 4;
  if (rpp > maxRows) rpp = maxRows;
  p->rowsPerPass = (int)rpp;
  *grid = (int)((waves + PG_WG_WAVES - 1) / PG_WG_WAVES);
  return 0;
}

extern "C" {

int pg_pack_planes(const void *src, int elem_bytes, int64_t n, int l, int64_t ld, const int64_t *rows, int bits,
                   void *planes, int64_t npad, uint32_t *flags, void *stream) {
  if (!src || !planes || !flags || n < 0 || ld < l) return fail(PG_E_BADARG, "pg_pack_planes: bad argument");
  if (int rc = check_bits(bits)) return rc;
  if (int rc = check_l(l, bits, true)) return rc;
  if (npad < n || npad % 256) return fail(PG_E_BADARG, "pg_pack_planes: npad must be pg_npad(n)");
  const int ng = pg_ngroups(l), nq = pg_nchunks(l, bits);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(flags, 0, sizeof(uint32_t), s);
  if (e != hipSuccess) return hipfail(e, "hipMemsetAsync");
  const dim3 grid((unsigned)((npad + 255) / 256)), block(256);
  const long long *r = (const long long *)rows;
  u32 *pl = (u32 *)planes;
#define PG_PACK(T)                                                                                          \
  do {                                                                                                      \
    if (bits == 5) pg_pack_kernel<T, 5><<<grid, block, 0, s>>>((const T *)src, n, l, ld, r, pl, npad, ng, nq, flags); \
    else pg_pack_kernel<T, 8><<<grid, block, 0, s>>>((const T *)src, n, l, ld, r, pl, npad, ng, nq, flags); \
  } while (0)
  switch (elem_bytes) {
    case 1: PG_PACK(unsigned char); break;
    case 2: PG_PACK(short); break;
    case 4: PG_PACK(int); break;
    case 8: PG_PACK(long long); break;
    default: return fail(PG_E_BADARG, "pg_pack_planes: elem_bytes must be 1, 2, 4 or 8");
  }
#undef PG_PACK
  return launched((int)hipGetLastError(), "pg_pack_kernel");
}

int pg_pack_bytes(const uint8_t *src, int64_t n, int width, int64_t ld, const int64_t *rows, const uint8_t *lut256, int bits,
                  void *planes, int64_t npad, uint8_t *tokens_out, uint32_t *flags, void *stream) {
  if (!src || !lut256 || !planes || !flags || n < 0 || ld < width) return fail(PG_E_BADARG, "pg_pack_bytes: bad argument");
  if (int rc = check_bits(bits)) return rc;
  if (int rc = check_l(width, bits, true)) return rc;
  if (npad < n || npad % 256) return fail(PG_E_BADARG, "pg_pack_bytes: npad must be pg_npad(n)");
  const int ng = pg_ngroups(width), nq = pg_nchunks(width, bits);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(flags, 0, sizeof(uint32_t), s);
  if (e != hipSuccess) return hipfail(e, "hipMemsetAsync");
  const dim3 grid((unsigned)((npad + 255) / 256)), block(256);
  if (bits == 5)
    pg_pack_kernel<unsigned char, 5><<<grid, block, 0, s>>>(src, n, width, ld, (const long long *)rows, (u32 *)planes, npad, ng, nq,
                                                            flags, lut256, tokens_out);
  else
    pg_pack_kernel<unsigned char, 8><<<grid, block, 0, s>>>(src, n, width, ld, (const long long *)rows, (u32 *)planes, npad, ng, nq,
                                                            flags, lut256, tokens_out);
  return launched((int)hipGetLastError(), "pg_pack_kernel(bytes)");
}

int pg_hamming_dense(const void *x_planes, int64_t n, int64_t x_npad, const void *y_planes, int64_t m, int64_t y_npad,
                     int l, int bits, void *out, int out_elem_bytes, int64_t ldo, int accumulate, void *stream) {
  if (!x_planes || !y_planes || !out || n <= 0 || m <= 0 || ldo < n)
    return fail(PG_E_BADARG, "pg_hamming_dense: bad argument");
  if (int rc = check_bits(bits)) return rc;
  if (int rc = check_l(l, bits)) return rc;
  if (out_elem_bytes != 1 && out_elem_bytes != 2 && out_elem_bytes != 4 && out_elem_bytes != 8)
    return fail(PG_E_BADARG, "pg_hamming_dense: out_elem_bytes must be 1, 2, 4 or 8");
  if (x_npad < n || x_npad % 256 || y_npad < m) return fail(PG_E_BADARG, "pg_hamming_dense: bad npad");
  if ((m + PG_RBD - 1) / PG_RBD > 65535) return fail(PG_E_BADARG, "pg_hamming_dense: m too large for one launch");
  DenseParams p;
  p.xPlanes = (const uint4 *)x_planes; p.xNpad = x_npad; p.n = n;
  p.yPlanes = (const uint4 *)y_planes; p.yNpad = y_npad; p.m = m;
  p.out = out; p.ldo = ldo; p.outBytes = out_elem_bytes; p.accumulate = accumulate ? 1 : 0;
  return launched(kDense[pg_ngroups(l) - 1](bits, p, (hipStream_t)stream), "pg_dense_kernel");
}

static int fill_nsq(NsqParams *p, const void *row_planes, int64_t row_npad, int64_t row0, int64_t nrows,
                    const void *col_planes, int64_t col_npad, int64_t ncols, int l, int bits) {
  if (!row_planes || !col_planes || row0 < 0 || nrows <= 0 || ncols <= 0) return fail(PG_E_BADARG, "bad argument");
  if (int rc = check_bits(bits)) return rc;
  if (int rc = check_l(l, bits)) return rc;
  if (col_npad < ncols || col_npad % 256 || row_npad < row0 + nrows) return fail(PG_E_BADARG, "bad npad");
  if (ncols > 0x7fffffffLL) return fail(PG_E_TOOMANY, "ncols exceeds int32 indices");
  memset(p, 0, sizeof(*p));
  p->filter = lb_filter_mode();
  p->rowPlanes = (const uint4 *)row_planes; p->rowNpad = row_npad; p->row0 = row0; p->nrows = nrows;
  p->colPlanes = (const uint4 *)col_planes; p->colNpad = col_npad; p->ncols = ncols;
  p->colSig = p->colPlanes + (long long)pg_nchunks(l, bits) * col_npad;
  p->colFold = p->colSig + 2 * col_npad;
  p->rowFold = p->rowPlanes + ((long long)pg_nchunks(l, bits) + 2) * row_npad;
  return 0;
}

// Which all-pairs engine runs a Hamming launch: pg_mm.h (stage 1 on the matrix cores, passes of 32
// rows per wave) or pg_nsq.h (stage 1 on the VALU, 4..32 rows per wave).  PG_ENGINE=mfma / valu forces one.
// Auto: the MFMA engine from 8 waves per SIMD-column on, i.e. once 32-row passes fill the chip; below
// that the VALU engine's finer row split keeps more CUs busy.
static bool use_mm_engine(int64_t nrows, int l = 64, bool knn = true) {
  if (const char *e = getenv("PG_ENGINE")) {
    if (!strcmp(e, "mfma")) return true;
    if (!strcmp(e, "valu")) return false;
  }
  // Since the MFMA engine queues its candidates lane-parallel (kNN, and eps slots too: pg_mm.h push_signs) it wins from
  // ~20k rows for kNN (profiles/r03_engine_crossover.txt: N = 16k L = 64 0.24 vs 0.29 ms, L = 32 0.24 vs 0.20) and from
  // ~28k for eps (N = 24k: 0.34 vs 0.27; 32k: 0.41 vs 0.53), ~36k for sequences of one group (L <= 32: only 32 signature
  // bits; N = 50k L = 32 symmetric: 0.43 vs 0.53).  Round 2: 40k / 60k.
  const long long dflt = knn ? 20000 : (l <= 32 ? 36000 : 28000);
  const long long thr = getenv("PG_ENGINE_MIN_ROWS") ? atoll(getenv("PG_ENGINE_MIN_ROWS")) : dflt;
  return nrows >= thr;
}
#ifdef PG_MM_STATS
static unsigned long long *g_stats = nullptr;
extern "C" int pg_debug_stats(unsigned long long *out12, int reset) {   // debug builds only (tools/mm_stats.py); PG_NSTAT counters
  const size_t nst = PG_NSTAT + 2 * 65536;                 // the counters, then (start, duration) of the first 65536 passes
  if (!g_stats) { if (hipMalloc(&g_stats, nst * 8) != hipSuccess) return -1; hipMemset(g_stats, 0, nst * 8); }
  if (out12) { hipDeviceSynchronize(); hipMemcpy(out12, g_stats, (reset & 2 ? nst : PG_NSTAT) * 8, hipMemcpyDeviceToHost); }
  if (reset & 1) hipMemset(g_stats, 0, nst * 8);
  return 0;
}
#endif
static const nsq_fn kMm[8] = {pg_launch_mm_g1, pg_launch_mm_g2, pg_launch_mm_g3, pg_launch_mm_g4,
                              pg_launch_mm_g5, pg_launch_mm_g6, pg_launch_mm_g7, pg_launch_mm_g8};
// resident workgroups per CU of an MFMA-engine instance (launcher mode codes 0..4), asked once from the runtime
static int mm_occupancy(int groups, int mode, int bits) {
  static std::atomic<int> cache[8][5][2];                  // (a property of the code object: the same on every gfx950)
  std::atomic<int> &c = cache[groups - 1][mode][bits == 8];
  int v = c.load(std::memory_order_relaxed);
  if (v == 0) {
    const int n = kMm[groups - 1](mode, bits, NsqParams(), -1, nullptr);
    v = n < 1 ? 4 : (n > 4 ? 4 : n);
    c.store(v, std::memory_order_relaxed);
  }
  return v;
}
// One pass per wave.  A pass takes up to 32 rows (the M of the MFMA tile) and the CU holds 16 waves (four
// workgroups: LDS and VGPR bound), so the grid runs in rounds of `slots` passes.  When 32-row passes would not
// even fill one round, the rows are spread over ~97 % of the slots in smaller passes, down to 16 rows (N=50k L=32
// kNN 1.03 -> 0.90 ms, N=100k L=128 2.40 -> 2.16 ms; tools/rows_sweep.py, profiles/r02_rows_per_pass.txt).
// PG_MM_TAIL=1 also shrinks the passes of the LAST round of a longer grid (mmTailFrom / mmTailRows): measured
// slower (dense 200k: 12.5 -> 14.1 ms, cfg3 3.81 -> 3.90) - a round of 2154 full passes on half-empty SIMDs runs
// faster per pass than 3830 passes of 18 rows on full ones - and therefore off.
// PG_ROWS_PER_WAVE = uniform passes of that many rows (tuning sweeps).
// occ: resident workgroups per CU of the instance (= waves per SIMD), mm_occupancy().
// Single round (every pass gets a wave slot at once): what a launch takes follows the waves on its busiest SIMD, not the
// rows per wave - every wave streams all column fragments through the matrix pipe whatever its row count (measured at
// 200 000 columns, tools/dbg/balance*.py: one wave per SIMD 0.70 ms, two 0.79, three 0.94, four 1.10-1.14 with 32-row
// passes; 1.26 / 1.52 / 1.87 for two / three / four waves of 64-row passes, the same for 48..64 rows).  So: the fewest
// waves per SIMD that hold the rows, the rows spread evenly over exactly that many waves on every SIMD.  (Round 2 spread
// the rows over ~97 % of ALL slots: 65 536 rows took 1.11 ms as 4096 passes of 16 rows, 0.79 as 2048 of 32.)
static void plan_mm(int64_t nrows, NsqParams *p, int *grid, int rb = PG_MM_RB, int occ = 4, bool knn = false, int uniformRows = 0) {   // rb: rows per pass of the instance (32 or 64)
  long long rpw = rb, tailFrom = (nrows + rb - 1) / rb, tailRows = rb;
  const long long simds = (long long)(cu_count() > 0 ? cu_count() : 256) * 4;
  const long long slots = simds * (occ < 1 ? 1 : occ);
  const char *e = getenv("PG_ROWS_PER_WAVE");
  const char *t = getenv("PG_MM_TAIL");
  // eps launches keep round 2's spreading rule: their passes are not equal work (symmetric: a pass sweeps the columns
  // above its rows only; matches cost per row) - tools/dbg/plan_ab.py: N = 50k eps <= 2 symmetric 0.58 ms against 0.79
  const int planEnv = getenv("PG_MM_PLAN") ? atoi(getenv("PG_MM_PLAN")) : 0;               // (A/B: 2 = round 2's rule, 1 = the new one for eps too)
  const bool oldPlan = planEnv == 2 || (!knn && planEnv != 1);
  if ((e && atoi(e) > 0) || uniformRows > 0) {
    const int want = e && atoi(e) > 0 ? atoi(e) : uniformRows;
    rpw = want < rb ? want : rb;                            // (a wave sweeps one pass of at most rb rows at a time)
    tailFrom = (nrows + rpw - 1) / rpw; tailRows = rpw;
  } else if (!oldPlan) {
    if (tailFrom <= slots) {
      const long long w = (tailFrom + simds - 1) / simds;               // waves per SIMD
      long long r = (nrows + w * simds - 1) / (w * simds);
      r = (r + 1) / 2 * 2;
      if (r < 4) r = 4;
      if (r > rb) r = rb;
      rpw = r; tailFrom = (nrows + r - 1) / r; tailRows = r;
    }
  } else {
    const long long full = nrows / (slots * rb) * slots;                // passes of the full rounds
    const long long rest = nrows - full * rb;
    if (rest > 0 && (full == 0 || (t && atoi(t) == 1))) {
      long long r = (long long)((double)rest / (0.97 * (double)slots)) + 1;
      r = (r + 1) / 2 * 2;
      if (r < 4) r = 4;
      if (r > rb) r = rb;
      if (full == 0 && r < 16) r = 16;                                  // a single round: at least half-filled MFMA tiles
      tailFrom = full; tailRows = r;
    }
  }
  p->rowsPerWave = (int)rpw; p->rowsPerPass = rb;
  p->mmTailFrom = tailFrom; p->mmTailRows = (int)tailRows;
  const long long headRows = tailFrom * rpw < nrows ? tailFrom * rpw : nrows;
  const long long waves = tailFrom + (nrows - headRows + tailRows - 1) / tailRows;
  if (getenv("PG_DEBUG_PLAN")) fprintf(stderr, "[pg plan mm] rows=%lld: %lld passes of %lld rows + %lld of %lld\n", (long long)nrows, tailFrom, rpw, waves - tailFrom, tailRows);
  p->mmDenseL1 = getenv("PG_MM_L1") ? atoi(getenv("PG_MM_L1")) : PG_MM_DENSE_L1;
  p->mmDenseL2 = getenv("PG_MM_L2") ? atoi(getenv("PG_MM_L2")) : PG_MM_DENSE_L2;
  p->mmDirectRun = getenv("PG_MM_RUN") ? atoi(getenv("PG_MM_RUN")) : PG_MM_DIRECT_RUN;
  if (p->mmDirectRun < 1) p->mmDirectRun = 1;
#ifdef PG_MM_STATS
  if (!g_stats) pg_debug_stats(nullptr, 1);
  p->stats = g_stats;
#endif
  // persistent waves: at most one per slot of the chip (whole workgroups), the rest of the passes through the counter
  long long gridWaves = waves < slots ? waves : slots;
  if (const char *pe = getenv("PG_MM_PERSISTENT")) { if (atoi(pe) == 0) gridWaves = waves; }
  gridWaves = (gridWaves + PG_WG_WAVES - 1) / PG_WG_WAVES * PG_WG_WAVES;
  p->mmPasses = waves; p->mmGridWaves = gridWaves;
  *grid = (int)(gridWaves / PG_WG_WAVES);
}
// The pass counter of a launch lives in the caller's workspace (pg_workspace_bytes): zeroed on the launch's stream
// right before the kernel.  Launch-private by contract, so concurrent launches - other streams, other devices,
// any number of them - never share a word (the static ring of counters this replaces did after 256 launches).
static int pass_counter(NsqParams *p, void *workspace, hipStream_t s) {
  if (!workspace) return fail(PG_E_BADARG, "workspace required (pg_workspace_bytes)");
  unsigned *c = (unsigned *)workspace;
  const hipError_t e = hipMemsetAsync(c, 0, PG_WS_GATES, s);   // (the counters and the row-block flags of a launch in column pieces)
  if (e != hipSuccess) return hipfail(e, "workspace: hipMemsetAsync");
  p->mmPassCounter = c;
  return 0;
}

int pg_eps_slots(const void *row_planes, int64_t row_npad, int64_t row0, int64_t nrows, const void *col_planes,
                 int64_t col_npad, int64_t ncols, int l, int bits, int cmp, double eps, int cap, int32_t *slot_idx,
                 uint8_t *slot_w, uint32_t *counts, void *workspace, void *stream) {
  NsqParams p;
  if (int rc = fill_nsq(&p, row_planes, row_npad, row0, nrows, col_planes, col_npad, ncols, l, bits)) return rc;
  if (!slot_idx || !slot_w || !counts || cap < 0 || cmp < PG_CMP_LE || cmp > PG_CMP_GT)
    return fail(PG_E_BADARG, "pg_eps_slots: bad argument");
  eps_interval(cmp, eps, &p.lo, &p.span);
  p.hi1 = (p.lo > 0xFFFFFF00u - 1u) ? 0u : p.lo + p.span + 1u;   // empty interval: nothing can match
  p.cap = (u32)cap; p.slotIdx = slot_idx; p.slotW = slot_w; p.counts = counts;
  int grid = 0;
  p.epsOrdered = getenv("PG_EPS_ORDERED") && atoi(getenv("PG_EPS_ORDERED")) != 0;
  if (use_mm_engine(nrows, l, false)) {
    plan_mm(nrows, &p, &grid);
    if (int rc = pass_counter(&p, workspace, (hipStream_t)stream)) return rc;
    return launched(kMm[pg_ngroups(l) - 1](PG_MODE_EPS, bits, p, grid, (hipStream_t)stream), "pg_mm_kernel(eps)");
  }
  if (int rc = plan_rows(nrows, &p, &grid, nsq_occupancy(pg_ngroups(l), PG_MODE_EPS, bits), 16.0 * pg_nchunks(l, bits))) return rc;
  return launched(kNsq[pg_ngroups(l) - 1](PG_MODE_EPS, bits, p, grid, (hipStream_t)stream), "pg_nsq_kernel(eps)");
}

// Square self-graph, every unordered pair evaluated once (Hamming and the comparators are symmetric):
// the engine sweeps only columns above the row, a match (i, j) goes to the front of row i's slot
// in column order and to the back of row j's slot through an atomic counter.
int pg_eps_slots_sym(const void *planes, int64_t npad, int64_t n, int l, int bits, int cmp, double eps, int cap,
                     int32_t *slot_idx, uint8_t *slot_w, uint32_t *counts_up, uint32_t *counts_lo, void *workspace,
                     void *stream) {
  NsqParams p;
  if (int rc = fill_nsq(&p, planes, npad, 0, n, planes, npad, n, l, bits)) return rc;
  if (!slot_idx || !slot_w || !counts_up || !counts_lo || cap < 0 || cmp < PG_CMP_LE || cmp > PG_CMP_GT)
    return fail(PG_E_BADARG, "pg_eps_slots_sym: bad argument");
  if (n >= (1ll << 27)) return fail(PG_E_TOOMANY, "pg_eps_slots_sym: n must be below 2^27");
  eps_interval(cmp, eps, &p.lo, &p.span);
  p.hi1 = (p.lo > 0xFFFFFF00u - 1u) ? 0u : p.lo + p.span + 1u;
  p.cap = (u32)cap; p.slotIdx = slot_idx; p.slotW = slot_w; p.counts = counts_up; p.countsLo = counts_lo;
  hipError_t e = hipMemsetAsync(counts_lo, 0, (size_t)n * sizeof(uint32_t), (hipStream_t)stream);
  if (e != hipSuccess) return hipfail(e, "hipMemsetAsync");
  int grid = 0;
  // Large graphs: the data decide (probe above) between this path and a plain rectangular sweep on the VALU engine,
  // which writes the same slots (every match at the front of its row, counts_lo stays zero): dense graphs.
  if (probe_enabled() && n >= PG_PROBE_MIN_N) {
    const u32 *gate = nullptr;
    if (int rc = run_probe(p, l, bits, 0u, p.lo, p.span, 1u, workspace, (hipStream_t)stream, &gate)) return rc;
    NsqParams r = p;
    r.countsLo = nullptr;
    r.gate = gate + 1; r.gateMask = 1u << 1;
    int rgrid = 0;
    if (int rc = plan_rows(n, &r, &rgrid, nsq_occupancy(pg_ngroups(l), PG_MODE_EPS, bits), 16.0 * pg_nchunks(l, bits))) return rc;
    if (int rc = launched(kNsq[pg_ngroups(l) - 1](PG_MODE_EPS, bits, r, rgrid, (hipStream_t)stream), "pg_nsq_kernel(eps, gated)")) return rc;
    p.gate = gate + 1; p.gateMask = 1u << 0;
  }
  // Rows near the top sweep almost everything, rows near the bottom almost nothing; workgroups are
  // dispatched in row order, i.e. longest first, which balances by itself once there are a few
  // waves per resident slot.  Measured (tools/eps_sym_probe.py): 8 rows per wave at N = 50k, 16 at
  // N = 100k .. 200k (more rows: too few waves to balance; fewer: the per-wave column stream shows).
  p.epsOrdered = getenv("PG_EPS_ORDERED") && atoi(getenv("PG_EPS_ORDERED")) != 0;
  if (use_mm_engine(n, l, false)) {
    // the symmetric sweep's passes are not equal work (a pass sweeps the columns above its rows): about one and a half
    // rounds of passes balance best (tools/dbg/eps_plan.py: N = 50k 8 rows per pass 0.40 ms against 0.43 / 0.59 for 14 / 32;
    // N = 100k 16 rows 0.61 against 0.66 / 0.71 for 26 / 32; N = 200k 32 rows 1.12 against 1.62 for 16)
    long long rs = (n + 6399) / 6400;                       // (tools/dbg/sym_rows.py: N = 50k 8 rows 0.375 ms, 10 rows 0.40; 40k: 8; 70k: 12)
    rs = (rs + 1) / 2 * 2;
    rs = rs < 8 ? 8 : (rs > PG_MM_RB ? PG_MM_RB : rs);
    // (records of up to three chunks - four waves per SIMD; longer ones, N = 100k L = 128: 0.75 against 0.67 with the old rule)
    const bool fine = pg_nchunks(l, bits) <= 3 && !(getenv("PG_MM_PLAN") && atoi(getenv("PG_MM_PLAN")) == 2);
    plan_mm(n, &p, &grid, PG_MM_RB, 4, false, fine ? (int)rs : 0);
    if (int rc = pass_counter(&p, workspace, (hipStream_t)stream)) return rc;
    return launched(kMm[pg_ngroups(l) - 1](PG_MODE_EPS_SYM, bits, p, grid, (hipStream_t)stream), "pg_mm_kernel(eps sym)");
  }
  if (!getenv("PG_ROWS_PER_WAVE") && !getenv("PG_WAVES_PER_CU")) {
    const long long r = n >= 80000 ? 16 : 8;
    p.rowsPerWave = (int)r; p.rowsPerPass = (int)r;
    grid = (int)(((n + r - 1) / r + PG_WG_WAVES - 1) / PG_WG_WAVES);
  } else if (int rc = plan_rows(n, &p, &grid, nsq_occupancy(pg_ngroups(l), PG_MODE_EPS_SYM, bits), 16.0 * pg_nchunks(l, bits), 16)) {
    return rc;
  }
  return launched(kNsq[pg_ngroups(l) - 1](PG_MODE_EPS_SYM, bits, p, grid, (hipStream_t)stream), "pg_nsq_kernel(eps sym)");
}

int pg_eps_compact_sym(const void *planes, int64_t npad, int64_t n, int l, int bits, int cmp, double eps, int cap,
                       const int32_t *slot_idx, const uint8_t *slot_w, const uint32_t *counts_up,
                       const uint32_t *counts_lo, const int64_t *indptr, int32_t *indices, uint8_t *weights,
                       int leave_overflow, void *stream) {
  CompactParams c;
  c.skipOverflow = leave_overflow ? 1 : 0;
  if (int rc = fill_nsq(&c.e, planes, npad, 0, n, planes, npad, n, l, bits)) return rc;
  if (!slot_idx || !slot_w || !counts_up || !counts_lo || !indptr || cap < 0 || cmp < PG_CMP_LE || cmp > PG_CMP_GT)
    return fail(PG_E_BADARG, "pg_eps_compact_sym: bad argument");
  eps_interval(cmp, eps, &c.e.lo, &c.e.span);
  c.e.hi1 = c.e.lo + c.e.span + 1u;
  c.e.cap = (u32)cap;
  c.e.slotIdx = const_cast<int *>(slot_idx);
  c.e.slotW = const_cast<unsigned char *>(slot_w);
  c.e.counts = const_cast<u32 *>(counts_up);
  c.e.countsLo = const_cast<u32 *>(counts_lo);
  c.indptr = (const long long *)indptr; c.indices = indices; c.weights = weights;
  return launched(kCompact[pg_ngroups(l) - 1](bits, c, (hipStream_t)stream), "pg_compact_kernel(sym)");
}

int64_t pg_scan_scratch_bytes(int64_t n) {
  const int64_t nb = (n + PG_SCAN_TILE - 1) / PG_SCAN_TILE;
  return (nb + 2) * (int64_t)sizeof(long long);
}

int pg_exclusive_scan(const uint32_t *counts, int64_t n, int64_t *indptr, void *scratch, void *stream) {
  if (!counts || !indptr || !scratch || n <= 0) return fail(PG_E_BADARG, "pg_exclusive_scan: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const long long nb = (n + PG_SCAN_TILE - 1) / PG_SCAN_TILE;
  long long *part = (long long *)scratch;
  pg_scan_partials<u32><<<dim3((unsigned)nb), dim3(256), 0, s>>>(counts, n, part);
  pg_scan_single<<<dim3(1), dim3(256), 0, s>>>(part, nb);
  pg_scan_apply<u32, 0><<<dim3((unsigned)nb), dim3(256), 0, s>>>(counts, n, part, nb, (long long *)indptr, nullptr);
  return launched((int)hipGetLastError(), "pg_scan");
}

int pg_eps_compact(const void *row_planes, int64_t row_npad, int64_t row0, int64_t nrows, const void *col_planes,
                   int64_t col_npad, int64_t ncols, int l, int bits, int cmp, double eps, int cap,
                   const int32_t *slot_idx, const uint8_t *slot_w, const uint32_t *counts, const int64_t *indptr,
                   int32_t *indices, uint8_t *weights, int leave_overflow, void *stream) {
  CompactParams c;
  c.skipOverflow = leave_overflow ? 1 : 0;
  if (int rc = fill_nsq(&c.e, row_planes, row_npad, row0, nrows, col_planes, col_npad, ncols, l, bits)) return rc;
  if (!slot_idx || !slot_w || !counts || !indptr || cap < 0 || cmp < PG_CMP_LE || cmp > PG_CMP_GT)
    return fail(PG_E_BADARG, "pg_eps_compact: bad argument");
  eps_interval(cmp, eps, &c.e.lo, &c.e.span);
  c.e.hi1 = c.e.lo + c.e.span + 1u;
  c.e.cap = (u32)cap;
  c.e.slotIdx = const_cast<int *>(slot_idx);
  c.e.slotW = const_cast<unsigned char *>(slot_w);
  c.e.counts = const_cast<u32 *>(counts);
  c.indptr = (const long long *)indptr; c.indices = indices; c.weights = weights;
  return launched(kCompact[pg_ngroups(l) - 1](bits, c, (hipStream_t)stream), "pg_compact_kernel");
}

// Rows whose matches did not fit their slot: the all-pairs engine runs once more over just those rows
// (row_list, relative to row0) and writes their matches, in column order, straight into the CSR at
// indptr[row] - exact for any slot capacity, at engine speed however many rows overflow.
int pg_eps_fill_rows(const void *row_planes, int64_t row_npad, int64_t row0, const int64_t *row_list, int64_t n_list,
                     const void *col_planes, int64_t col_npad, int64_t ncols, int l, int bits, int cmp, double eps,
                     const int64_t *indptr, int32_t *indices, uint8_t *weights, uint32_t *scratch_counts, void *workspace,
                     void *stream) {
  NsqParams p;
  if (!row_list || n_list <= 0) return fail(PG_E_BADARG, "pg_eps_fill_rows: bad argument");
  if (int rc = fill_nsq(&p, row_planes, row_npad, row0, n_list, col_planes, col_npad, ncols, l, bits)) return rc;
  if (!indptr || !indices || !weights || !scratch_counts || cmp < PG_CMP_LE || cmp > PG_CMP_GT)
    return fail(PG_E_BADARG, "pg_eps_fill_rows: bad argument");
  eps_interval(cmp, eps, &p.lo, &p.span);
  p.hi1 = (p.lo > 0xFFFFFF00u - 1u) ? 0u : p.lo + p.span + 1u;
  p.cap = 0xFFFFFFFFu;                                     // a row's place in the CSR holds all of its matches
  p.rowList = (const long long *)row_list; p.fillIndptr = (const long long *)indptr;
  p.slotIdx = indices; p.slotW = weights; p.counts = scratch_counts;
  int grid = 0;
  plan_mm(n_list, &p, &grid);
  if (int rc = pass_counter(&p, workspace, (hipStream_t)stream)) return rc;
  return launched(kMm[pg_ngroups(l) - 1](PG_MODE_EPS, bits, p, grid, (hipStream_t)stream), "pg_mm_kernel(eps fill)");
}

// What a kNN launch of `nrows` rows x `ncols` columns takes (ms) on the 32-row (rb = 32) or the 64-row (rb = 64) short-list
// instance, and which rows go into plain passes (*mainRows; the rest: column pieces, knn_launch).  A round of waves costs
//     T(w, ncols) = S[w] + H[w] * ncols / 100 000,     w = waves on the busiest SIMD
// (it follows the waves per SIMD, not the rows per wave: profiles/r03_pass_plan.txt).  S is what does not grow with the
// columns - the candidates of a row's cluster mates, pass set-up, the tail - H the sweep itself; calibrated at 300 000
// and 900 000 columns of cfg3-like data (tools/dbg/calib.py).  The 64-row instance sweeps twice the rows per fragment
// (H per row 0.6 x) but pays more per pass: 131 072 rows x 200 000 columns 1.07 ms (32-row passes) against 1.27, x
// 1 000 000 columns 4.9 against 3.6 (BASELINE configs[3]: one GPU's 125 000 x 1M block 5.5 -> 3.6 ms).
//   one round   : T(w); with w >= 3 and at most 128 row blocks beyond w - 1 full waves: T(w - 1) + the pieces
//   more rounds : whole rounds of full occupancy + (a last round: T(its waves) | at most 128 row blocks: the pieces)
static double knn_plan_cost(long long nrows, long long ncols, int rb, int occ, bool canSplit, long long *mainRows) {
  static const double S1[4] = {0.43, 0.42, 0.37, 0.12}, H1[4] = {0.148, 0.200, 0.294, 0.477};
  static const double S2[4] = {0.64, 0.69, 0.64, 0.54}, H2[4] = {0.229, 0.292, 0.404, 0.557};
  const bool big = rb > PG_MM_RB;
  const double c = (double)ncols / 1e5, kPieces = 0.04 * c;
  auto T = [&](long long w) { return (big ? S2 : S1)[w - 1] + (big ? H2 : H1)[w - 1] * c; };
  const long long simds = (long long)(cu_count() > 0 ? cu_count() : 256) * 4;
  occ = occ < 1 ? 1 : (occ > 4 ? 4 : occ);
  const long long passes = (nrows + rb - 1) / rb, slots = simds * occ, maxRem = 128ll * rb;
  *mainRows = nrows;
  if (passes <= slots) {
    const long long w = (passes + simds - 1) / simds;
    if (canSplit && w >= 3 && nrows - (w - 1) * simds * rb <= maxRem) {
      *mainRows = (w - 1) * simds * rb;
      return T(w - 1) + kPieces;
    }
    return T(w);
  }
  const long long full = nrows / (slots * rb), rem = nrows - full * slots * rb;
  double t = (double)full * T(occ);
  if (rem > 0) {
    if (canSplit && rem <= maxRem) {
      *mainRows = full * slots * rb;
      t += kPieces;
    } else {
      t += T(((rem + rb - 1) / rb + simds - 1) / simds);
    }
  }
  return t;
}

static int knn_launch(const void *row_planes, int64_t row_npad, int64_t row0, int64_t nrows, const void *col_planes,
                      int64_t col_npad, int64_t ncols, int l, int bits, int k, int first, const uint32_t *floor_keys,
                      uint32_t *last_keys, int32_t *idx_out, uint8_t *dist_out, void *workspace, void *stream) {
  NsqParams p;
  if (int rc = fill_nsq(&p, row_planes, row_npad, row0, nrows, col_planes, col_npad, ncols, l, bits)) return rc;
  if (!idx_out || !dist_out) return fail(PG_E_BADARG, "pg_knn_hamming: bad argument");
  if (k < 1 || first < 0 || first > 1 || first + k > 64) return fail(PG_E_BADARG, "pg_knn_hamming: k out of range");
  if (ncols > PG_MAX_N_KNN) return fail(PG_E_TOOMANY, "pg_knn_hamming: ncols exceeds 2^24");
  p.k = k; p.knnFirst = first; p.floorKeys = floor_keys; p.lastKeys = last_keys;
  // optimistic stage-1 cap.  pg_nsq.h: 8 = half of what unrelated sequences show in its 32-bit plane-0 bound.
  // pg_mm.h: 10 with the 54-bit signature (L > 32) - unrelated pairs below it are one in 3e6, kNN time is flat from
  // 7 to 12 on every shape tried (tools/guess_sweep.py) and rows whose k-th distance reaches 9 keep their cap;
  // 7 where the signature is the 32 plane-0 bits of a single group (L <= 32: one in 4e3 at 7, one in 400 at 10)
  const bool mm = use_mm_engine(nrows);
  const u32 guessMm = l > 32 ? 10u : 7u, guessValu = 8u;
  p.knnGuess = getenv("PG_KNN_GUESS") ? (u32)atoi(getenv("PG_KNN_GUESS")) : (mm ? guessMm : guessValu);
  if (p.filter == 0) p.knnGuess = 0;                       // no stage 1, nothing to cap
  p.knnIdx = idx_out; p.knnDist = dist_out;
  int grid = 0;
  auto launch_valu = [&](NsqParams &q) -> int {
    int g = 0;
    if (int rc = plan_rows(nrows, &q, &g, nsq_occupancy(pg_ngroups(l), PG_MODE_KNN, bits), 16.0 * pg_nchunks(l, bits), PG_RB_KNN)) return rc;
    return launched(kNsq[pg_ngroups(l) - 1](PG_MODE_KNN, bits, q, g, (hipStream_t)stream), "pg_nsq_kernel(knn)");
  };
  if (mm) {
    // Large launches: the data decide between the engines (probe above): unclustered data -> the VALU engine
    if (probe_enabled() && ncols >= PG_PROBE_MIN_N && nrows >= PG_PROBE_MIN_N / 2 && p.filter != 0 && !getenv("PG_KNN_GUESS")) {
      const u32 *gate = nullptr;
      if (int rc = run_probe(p, l, bits, p.knnGuess, 1u, 0u, (u32)(first + k), workspace, (hipStream_t)stream, &gate)) return rc;
      NsqParams v = p;
      v.knnGuess = guessValu;
      v.gate = gate; v.gateMask = 1u << 1;
      if (int rc = launch_valu(v)) return rc;
      p.gate = gate; p.gateMask = (1u << 0) | (1u << 2);   // (2 = one cluster: the 32-row alternative below takes it over where it exists)
    }
    // short lists (the usual k): the instance that inserts a whole batch of candidates at once (pg_mm.h, KL).
    // PG_MM_SHORT=0 keeps the 64-lane lists (A/B runs)
    const bool shortList = first == 1 && k + 1 <= PG_MM_KL && !floor_keys && !last_keys &&
                           !(getenv("PG_MM_SHORT") && atoi(getenv("PG_MM_SHORT")) == 0);
    // ... with 64 rows per pass (every column fragment feeds two MFMAs: half the vector-memory traffic per pair) where
    // that is the cheaper plan (knn_plan_cost; PG_MM_R=1 / 2 forces either).  5-bit L = 64: 64-row passes from 131 073
    // rows on (131 072 rows 1.14 ms against 1.26; 147 456: 1.55 / 1.47; 196 608: 1.84 / 1.57); L = 128 (three waves per
    // SIMD) N = 100k: 32-row passes 0.75 against 0.97; byte alphabets (64-row instance: two waves per SIMD): 32-row passes.
    // One wave more on every SIMD for a few rows more costs the launch as much as a full round of them (plan_mm), and
    // so does a last round of the persistent waves that only a few passes are left for.  The rows that whole rounds
    // hold go into plain passes; the few beyond - at most 128 row blocks - are swept in PIECES of the columns (as many
    // pieces as give every SIMD about one wave: short passes, a sixteenth or an eighth of a sweep each), and
    // pg_knn_merge_kernel makes the rows' lists of the pieces'.
    // cfg3 (200 000 rows = 3 x 65 536 + 3 392): three waves per SIMD + 53 row blocks x 16 pieces.  PG_MM_SPLIT=0: off
    const int ng = pg_ngroups(l);
    const int occ1 = mm_occupancy(ng, shortList ? PG_MODE_KNN_SHORT : PG_MODE_KNN, bits);
    const int occ2 = shortList ? mm_occupancy(ng, PG_MODE_KNN_SHORT2, bits) : 0;
    const bool evictOn = !(getenv("PG_MM_EVICT") && atoi(getenv("PG_MM_EVICT")) == 0);
    const bool canSplit = shortList && workspace && evictOn && p.knnGuess > 0 && ncols >= 65536 && nrows > PG_SPLIT_MIN_ROWS &&
                          !getenv("PG_ROWS_PER_WAVE") && !(getenv("PG_MM_SPLIT") && atoi(getenv("PG_MM_SPLIT")) == 0);
    long long main1 = nrows, main2 = nrows;
    const double t1 = knn_plan_cost(nrows, ncols, PG_MM_RB, occ1, canSplit, &main1);
    const double t2 = shortList ? knn_plan_cost(nrows, ncols, 2 * PG_MM_RB, occ2, canSplit, &main2) : 1e30;
    bool two = shortList && t2 < t1;
    if (const char *e = getenv("PG_MM_R")) two = shortList && atoi(e) == 2;
    const int rbm = two ? 2 * PG_MM_RB : PG_MM_RB, occm = two ? occ2 : occ1;
    const int modeM = two ? PG_MODE_KNN_SHORT2 : (shortList ? PG_MODE_KNN_SHORT : PG_MODE_KNN);
    const long long mainRows = two ? main2 : main1;
    // the probe's "one cluster" (gate value 2) goes to the 32-row alternative where one is launched (below); elsewhere
    // the main launch takes it as well
    const bool alt32 = two && p.gate && !getenv("PG_MM_R");
    if (p.gate) p.gateMask = alt32 ? (1u << 0) : ((1u << 0) | (1u << 2));
    plan_mm(mainRows, &p, &grid, rbm, occm, true);
    p.nrows = mainRows;
    if (int rc = pass_counter(&p, workspace, (hipStream_t)stream)) return rc;
    // rows that lose their optimistic cap are evicted from their passes and finished by pg_knn_rows_kernel behind the
    // launch (NsqParams::mmEvict; PG_MM_EVICT=0: the second phase inside the pass, as before)
    const bool evict = first == 1 && !floor_keys && !last_keys && p.knnGuess > 0 && evictOn;
    if (evict) {
      p.mmEvict = (u32 *)workspace + 1;                     // (word 1 of the counter line: zeroed by pass_counter)
      p.mmEvictRows = (u32 *)((char *)workspace + ws_evict_offset(nrows));
    }
    auto finish_evicted = [&]() -> int {
      if (!evict) return 0;
      KnnRowsParams kr;
      kr.rowPlanes = p.rowPlanes; kr.colPlanes = p.colPlanes; kr.rowNpad = p.rowNpad; kr.colNpad = p.colNpad; kr.ncols = p.ncols;
      kr.count = p.mmEvict; kr.rows = p.mmEvictRows; kr.baseRow = row0; kr.k = k;
      kr.knnIdx = idx_out; kr.knnDist = dist_out; kr.gate = p.gate; kr.gateMask = p.gateMask;
      const long long cap = (long long)(cu_count() > 0 ? cu_count() : 256) * 8;   // (one workgroup per row, the rows in turns)
      return launched(kKnnRows[ng - 1](bits, kr, (int)(nrows < cap ? nrows : cap), (hipStream_t)stream), "pg_knn_rows_kernel");
    };
    if (mainRows < nrows) {
      const long long rem = nrows - mainRows, nb = (rem + rbm - 1) / rbm;
      const long long simds = (long long)(cu_count() > 0 ? cu_count() : 256) * 4;
      int pieces = (int)(simds / nb);                       // (nb <= 128: at least 8; nb * pieces <= 1024 passes = the workspace's share)
      pieces = pieces > 16 ? 16 : (pieces < 2 ? 2 : pieces);
      if (const char *e = getenv("PG_MM_PIECES")) { if (atoi(e) >= 2 && atoi(e) <= (nb <= 64 ? 16 : 8)) pieces = atoi(e); }   // (experiments)
      // the pieces ride in the main launch: its passes beyond the plain ones, fetched through the counter by the waves
      // that finish first - they run while the slower passes are still at work
      const long long mainPasses = p.mmPasses;              // (plain passes of rbm rows: plan_mm gave whole rounds)
      p.nrows = nrows;
      p.mmPieces = pieces; p.mmPieceFrom = mainPasses; p.mmPartial = (u32 *)((char *)workspace + PG_WS_PARTIAL);
      p.mmPasses = mainPasses + nb * pieces;
      p.mmTailFrom = p.mmPasses; p.mmTailRows = rbm;
      // ... or, where the chip has slots left beside the plain passes (one round: a wave per SIMD is free), as waves of
      // their own from the start: short guests at four waves per SIMD instead of a tail of 886 waves on an empty chip
      // (PG_MM_PIECES_AHEAD=0: through the counter)
      if (mainPasses + nb * pieces <= simds * occm && !(getenv("PG_MM_PIECES_AHEAD") && atoi(getenv("PG_MM_PIECES_AHEAD")) == 0)) {
        p.mmGridWaves = (p.mmPasses + PG_WG_WAVES - 1) / PG_WG_WAVES * PG_WG_WAVES;
        grid = (int)(p.mmGridWaves / PG_WG_WAVES);
      }
      if (int rc = launched(kMm[ng - 1](modeM, bits, p, grid, (hipStream_t)stream), "pg_mm_kernel(knn + column pieces)")) return rc;
      pg_knn_merge_kernel<<<dim3((unsigned)((rem + 4 * PG_WG_WAVES - 1) / (4 * PG_WG_WAVES))), dim3(PG_WG_THREADS), 0, (hipStream_t)stream>>>(
          p.mmPartial, rem, pieces, k, idx_out + mainRows * k, dist_out + mainRows * k, p.knnGuess, p.mmEvict, p.mmEvictRows,
          (long long)row0 + mainRows, p.gate, p.gateMask);
      if (int rc = launched((int)hipGetLastError(), "pg_knn_merge_kernel")) return rc;
      p.nrows = nrows;                                      // (the gated 32-row alternative below covers all rows)
    }
    if (alt32) {
      // the probe may say "one cluster" (gate 2): the same engine with 32-row passes, launched as a third alternative
      NsqParams q = p;
      int qgrid = 0;
      q.mmPieces = 0; q.mmPieceFrom = 0; q.mmPartial = nullptr;
      q.mmEvict = nullptr; q.mmEvictRows = nullptr;         // (one cluster: every row has its neighbours; the second phase if not)
      plan_mm(nrows, &q, &qgrid, PG_MM_RB, occ1, true);
      q.mmPassCounter = p.mmPassCounter + 8;
      q.gateMask = 1u << 2;
      if (int rc = launched(kMm[pg_ngroups(l) - 1](PG_MODE_KNN_SHORT, bits, q, qgrid, (hipStream_t)stream), "pg_mm_kernel(knn, 32-row passes, gated)")) return rc;
    }
    if (mainRows < nrows) return finish_evicted();          // (the main launch and the pieces are out already)
    if (int rc = launched(kMm[pg_ngroups(l) - 1](two ? PG_MODE_KNN_SHORT2 : (shortList ? PG_MODE_KNN_SHORT : PG_MODE_KNN), bits, p, grid,
                                                 (hipStream_t)stream), "pg_mm_kernel(knn)")) return rc;
    return finish_evicted();
  }
  return launch_valu(p);
}

int pg_knn_hamming(const void *row_planes, int64_t row_npad, int64_t row0, int64_t nrows, const void *col_planes,
                   int64_t col_npad, int64_t ncols, int l, int bits, int k, int32_t *idx_out, uint8_t *dist_out,
                   void *workspace, void *stream) {
  if (k < 1 || k > PG_MAX_K) return fail(PG_E_BADARG, "pg_knn_hamming: k must be in 1..63");
  return knn_launch(row_planes, row_npad, row0, nrows, col_planes, col_npad, ncols, l, bits, k, 1, nullptr, nullptr,
                    idx_out, dist_out, workspace, stream);
}

int pg_knn_hamming_round(const void *row_planes, int64_t row_npad, int64_t row0, int64_t nrows, const void *col_planes,
                         int64_t col_npad, int64_t ncols, int l, int bits, int k, int first_round,
                         const uint32_t *floor_keys, uint32_t *last_keys, int32_t *idx_out, uint8_t *dist_out,
                         void *workspace, void *stream) {
  if (!last_keys || (!first_round && !floor_keys)) return fail(PG_E_BADARG, "pg_knn_hamming_round: key arrays required");
  return knn_launch(row_planes, row_npad, row0, nrows, col_planes, col_npad, ncols, l, bits, k, first_round ? 1 : 0,
                    first_round ? nullptr : floor_keys, last_keys, idx_out, dist_out, workspace, stream);
}

int pg_index_flags(const void *planes, int64_t n, int64_t npad, int l, int bits, int64_t ref,
                   const uint32_t *want_dist, int pos_mode, const uint32_t *pos_mask, const uint32_t *not_mask,
                   uint8_t *dist_out, uint64_t *hist, uint8_t *flags, void *stream) {
  if (!planes || n <= 0 || npad < n || ref < 0 || ref >= n) return fail(PG_E_BADARG, "pg_index_flags: bad argument");
  if (int rc = check_bits(bits)) return rc;
  if (int rc = check_l(l, bits)) return rc;
  if (pos_mode < 0 || pos_mode > 2 || (pos_mode && (!pos_mask || !not_mask)))
    return fail(PG_E_BADARG, "pg_index_flags: bad position mode / masks");
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (bits == 5)
    pg_index_kernel<5><<<grid, block, 0, s>>>((const u32 *)planes, n, npad, pg_ngroups(l), ref, want_dist, pos_mode,
                                              pos_mask, not_mask, dist_out, (u64 *)hist, flags);
  else
    pg_index_kernel<8><<<grid, block, 0, s>>>((const u32 *)planes, n, npad, pg_ngroups(l), ref, want_dist, pos_mode,
                                              pos_mask, not_mask, dist_out, (u64 *)hist, flags);
  return launched((int)hipGetLastError(), "pg_index_kernel");
}

// ---------------------------------------------------------------------------------------
// banded Levenshtein kNN (build defined; see pg_lev.hip)
// ---------------------------------------------------------------------------------------
int pg_lev_profile(const uint8_t *tokens, int64_t n, int l, int64_t ld, void *profiles, int64_t npad, int32_t *lens,
                   uint32_t *flags, void *stream) {
  if (!tokens || !profiles || !lens || !flags || n <= 0 || ld < l) return fail(PG_E_BADARG, "pg_lev_profile: bad argument");
  if (int rc = check_l(l)) return rc;
  if (npad < n || npad % 256) return fail(PG_E_BADARG, "pg_lev_profile: npad must be pg_npad(n)");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(flags, 0, sizeof(uint32_t), s);
  if (e != hipSuccess) return hipfail(e, "hipMemsetAsync");
  return launched(pg_launch_lev_profile(tokens, n, l, ld, (u32 *)profiles, npad, lens, flags, s), "pg_lev_profile_kernel");
}

int pg_lev_candidates(const void *profiles, int64_t npad, int64_t n, int64_t row0, int64_t nrows, int band, int cap,
                      int32_t *slot_idx, uint8_t *slot_w, uint32_t *counts, void *stream) {
  if (!profiles || !slot_idx || !slot_w || !counts || n <= 0 || nrows <= 0 || row0 < 0 || row0 + nrows > n || cap < 0)
    return fail(PG_E_BADARG, "pg_lev_candidates: bad argument");
  if (band < 0 || band > PG_LEV_MAX_BAND) return fail(PG_E_BADARG, "pg_lev_candidates: band must be in 0..8");
  if (npad < n || npad % 256) return fail(PG_E_BADARG, "pg_lev_candidates: bad npad");
  if (n > 0x7fffffffLL) return fail(PG_E_TOOMANY, "n exceeds int32 indices");
  NsqParams p;
  memset(&p, 0, sizeof(p));
  p.rowPlanes = (const uint4 *)profiles; p.rowNpad = npad; p.row0 = row0; p.nrows = nrows;
  p.colPlanes = (const uint4 *)profiles; p.colNpad = npad; p.ncols = n;
  p.lo = 0; p.span = 2u * (u32)band; p.hi1 = p.span + 1u;
  p.filter = lb_filter_mode();             // keep pairs with max(SAD, 2*|dlen|) <= 2*band, self included
  p.cap = (u32)cap; p.slotIdx = slot_idx; p.slotW = slot_w; p.counts = counts;
  int grid = 0;
  static int bag_occ = 0;
  if (!bag_occ) { bag_occ = pg_occ_nsq_bag(); if (bag_occ < 1) bag_occ = 4; }
  if (int rc = plan_rows(nrows, &p, &grid, bag_occ, 48.0)) return rc;
  return launched(pg_launch_nsq_bag(p, grid, (hipStream_t)stream), "pg_nsq_kernel(bag)");
}

// Symmetric candidate generation (all rows against the same profiles): every unordered pair once,
// the row itself excluded; slots as in pg_eps_slots_sym.  pg_lev_knn takes both counters.
int pg_lev_candidates_sym(const void *profiles, int64_t npad, int64_t n, int band, int cap, int32_t *slot_idx,
                          uint8_t *slot_w, int32_t *slot_aux, uint32_t *counts_up, uint32_t *counts_lo, void *stream) {
  if (!profiles || !slot_idx || !slot_w || !counts_up || !counts_lo || n <= 0 || cap < 0)
    return fail(PG_E_BADARG, "pg_lev_candidates_sym: bad argument");
  if (band < 0 || band > PG_LEV_MAX_BAND) return fail(PG_E_BADARG, "pg_lev_candidates_sym: band must be in 0..8");
  if (npad < n || npad % 256) return fail(PG_E_BADARG, "pg_lev_candidates_sym: bad npad");
  if (n >= (1ll << 27)) return fail(PG_E_TOOMANY, "pg_lev_candidates_sym: n must be below 2^27");
  NsqParams p;
  memset(&p, 0, sizeof(p));
  p.rowPlanes = (const uint4 *)profiles; p.rowNpad = npad; p.row0 = 0; p.nrows = n;
  p.colPlanes = (const uint4 *)profiles; p.colNpad = npad; p.ncols = n;
  p.lo = 0; p.span = 2u * (u32)band; p.hi1 = p.span + 1u;
  p.filter = lb_filter_mode();
  p.cap = (u32)cap; p.slotIdx = slot_idx; p.slotW = slot_w; p.slotAux = slot_aux; p.counts = counts_up; p.countsLo = counts_lo;
  hipError_t e = hipMemsetAsync(counts_lo, 0, (size_t)n * sizeof(uint32_t), (hipStream_t)stream);
  if (e != hipSuccess) return hipfail(e, "hipMemsetAsync");
  long long r = n >= 80000 ? 16 : 8;                        // as pg_eps_slots_sym
  if (const char *ev = getenv("PG_ROWS_PER_WAVE")) { if (atoi(ev) > 0) r = atoi(ev); }
  p.rowsPerWave = (int)r; p.rowsPerPass = (int)(r > PG_RB ? PG_RB : r);
  const int grid = (int)(((n + r - 1) / r + PG_WG_WAVES - 1) / PG_WG_WAVES);
  return launched(pg_launch_nsq_bag_sym(p, grid, (hipStream_t)stream), "pg_nsq_kernel(bag sym)");
}

int pg_lev_knn(const uint8_t *tokens, int64_t n, int l, int64_t ld, const void *planes128, int64_t npad,
               const int32_t *lens, int64_t row0, int64_t nrows,
               int band, int k, int cap, const int32_t *slot_idx, uint8_t *slot_w, const int32_t *slot_aux,
               const uint32_t *counts, const uint32_t *counts_lo, int32_t *idx_out, uint8_t *dist_out, void *stream) {
  if (!tokens || !planes128 || npad < n || npad % 256 || !lens || !slot_idx || !counts || !idx_out || !dist_out || n <= 0 || nrows <= 0 || row0 < 0 ||
      row0 + nrows > n || ld < l || cap < 0)
    return fail(PG_E_BADARG, "pg_lev_knn: bad argument");
  if (int rc = check_l(l)) return rc;
  if (band < 0 || band > PG_LEV_MAX_BAND) return fail(PG_E_BADARG, "pg_lev_knn: band must be in 0..8");
  if (k < 1 || k > PG_MAX_K) return fail(PG_E_BADARG, "pg_lev_knn: k must be in 1..63");
  if (n > PG_MAX_N_KNN) return fail(PG_E_TOOMANY, "pg_lev_knn: n exceeds 2^24");
  if (counts_lo && (row0 != 0 || nrows != n)) return fail(PG_E_BADARG, "pg_lev_knn: symmetric slots cover all rows");
  return launched(pg_launch_lev_select(tokens, n, l, ld, (const uint4 *)planes128, npad, lens, row0, nrows, band, k, (u32)cap, slot_idx, slot_w,
                                       slot_aux, counts, counts_lo, idx_out, dist_out, (hipStream_t)stream), "pg_lev_select_kernel");
}

int pg_csr_row_stats(const int64_t *indptr, const int32_t *indices, const uint8_t *weights_u8, const float *weights_f32,
                     int64_t nrows, int64_t row0, const double *f, double *deg, double *sum_f, double *sum_wf,
                     double *self_w, double *col_sum, void *stream) {
  if (!indptr || !indices || nrows <= 0 || row0 < 0) return fail(PG_E_BADARG, "pg_csr_row_stats: bad argument");
  if ((sum_f || sum_wf) && !f) return fail(PG_E_BADARG, "pg_csr_row_stats: node values required");
  pg_csr_row_stats_kernel<<<dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
      (const long long *)indptr, indices, weights_u8, weights_f32, nrows, row0, f, deg, sum_f, sum_wf, self_w, col_sum);
  return launched((int)hipGetLastError(), "pg_csr_row_stats_kernel");
}

int pg_compact_flags(const uint8_t *flags, int64_t n, int64_t *out_idx, int64_t *out_count, void *scratch,
                     void *stream) {
  if (!flags || !out_idx || !out_count || !scratch || n <= 0) return fail(PG_E_BADARG, "pg_compact_flags: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const long long nb = (n + PG_SCAN_TILE - 1) / PG_SCAN_TILE;
  long long *part = (long long *)scratch;
  pg_scan_partials<unsigned char><<<dim3((unsigned)nb), dim3(256), 0, s>>>(flags, n, part);
  pg_scan_single<<<dim3(1), dim3(256), 0, s>>>(part, nb);
  pg_scan_apply<unsigned char, 1><<<dim3((unsigned)nb), dim3(256), 0, s>>>(flags, n, part, nb, (long long *)out_idx,
                                                                          (long long *)out_count);
  return launched((int)hipGetLastError(), "pg_compact_flags");
}

// ---------------------------------------------------------------------------------------
// The path's one collective: all-gather of the row shards of the token matrix (e).
// RCCL is bound at run time: the copy the host process already carries (torch ships one as
// "librccl.so") or the ROCm installation's.  One communicator = one rank = one GPU (the current HIP
// device at pg_comm_init); the 128-byte id travels between the ranks by whatever channel the host has.
// ---------------------------------------------------------------------------------------
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  const char *(*GetErrorString)(ncclResult_t);
};
static RcclApi *rccl() {
  static RcclApi api;
  static int state = 0;   // 0 untried, 1 ok, -1 missing
  static std::once_flag once;
  std::call_once(once, [&]() {
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);           // the host's copy, if it has one loaded
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) {
      api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
      api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    }
    state = (h && api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather) ? 1 : -1;
  });
  return state == 1 ? &api : nullptr;
}
static int rcclfail(RcclApi *r, ncclResult_t e, const char *where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, r->GetErrorString ? r->GetErrorString(e) : "RCCL error");
  return PG_E_COMM;
}

int pg_comm_available(void) { return rccl() ? 1 : 0; }   // local, not a collective: can this process bind RCCL at all?

int pg_comm_unique_id(void *id128) {
  RcclApi *r = rccl();
  if (!r) return fail(PG_E_COMM, "pg_comm_unique_id: librccl.so not found");
  if (!id128) return fail(PG_E_BADARG, "pg_comm_unique_id: bad argument");
  ncclUniqueId id;
  ncclResult_t e = r->GetUniqueId(&id);
  if (e != ncclSuccess) return rcclfail(r, e, "ncclGetUniqueId");
  static_assert(sizeof(id) == PG_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int pg_comm_init(void **comm, int nranks, int rank, const void *id128) {
  RcclApi *r = rccl();
  if (!r) return fail(PG_E_COMM, "pg_comm_init: librccl.so not found");
  if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(PG_E_BADARG, "pg_comm_init: bad argument");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  ncclResult_t e = r->CommInitRank(&c, nranks, id, rank);
  if (e != ncclSuccess) return rcclfail(r, e, "ncclCommInitRank");
  *comm = (void *)c;
  return 0;
}

int pg_comm_destroy(void *comm) {
  RcclApi *r = rccl();
  if (!r || !comm) return fail(PG_E_BADARG, "pg_comm_destroy: bad argument");
  ncclResult_t e = r->CommDestroy((ncclComm_t)comm);
  return e == ncclSuccess ? 0 : rcclfail(r, e, "ncclCommDestroy");
}

int pg_allgather_tokens(void *comm, const void *shard, int64_t rows_per_rank, int l, void *full, void *stream) {
  RcclApi *r = rccl();
  if (!r) return fail(PG_E_COMM, "pg_allgather_tokens: librccl.so not found");
  if (!comm || !shard || !full || rows_per_rank <= 0 || l <= 0) return fail(PG_E_BADARG, "pg_allgather_tokens: bad argument");
  ncclResult_t e = r->AllGather(shard, full, (size_t)rows_per_rank * (size_t)l, ncclUint8, (ncclComm_t)comm, (hipStream_t)stream);
  return e == ncclSuccess ? 0 : rcclfail(r, e, "ncclAllGather");
}

}  // extern "C"
