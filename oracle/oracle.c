/*
 * ORACLE (C leg) — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain C restatement of the reference's hot path (acmater/prograph; citations into
 * /root/reference) for sizes where the Python oracle (oracle/prograph_oracle.py) is too slow.
 * Pinned: tests/test_oracle.py checks every function here against the Python oracle, which is
 * itself pinned to golden vectors generated from the real reference.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 *   orc_hamming   d[m][n] = #{j : Y[m][j] != X[n][j]}            prograph/distance/hamming.py:34
 *   orc_eps       (comp(d,eps) & (d>0)) per row, columns ascending prograph/prograph.py:731-753
 *   orc_knn       stable ascending sort by distance, ranks 1..k    prograph/prograph.py:756-764
 *   orc_synth     the deterministic generator of prograph_amd/synth.py
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int ham(const uint8_t *a, const uint8_t *b, int l) {
  int d = 0;
  for (int j = 0; j < l; ++j) d += a[j] != b[j];
  return d;
}

void orc_hamming(const uint8_t *X, int64_t n, const uint8_t *Y, int64_t m, int l, int64_t *out) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < m; ++r)
    for (int64_t c = 0; c < n; ++c) out[r * n + c] = ham(Y + r * l, X + c * l, l);
}

static inline int cmp_ok(int cmp, double d, double eps) {
  switch (cmp) {
    case 0: return d <= eps;
    case 1: return d < eps;
    case 2: return d == eps;
    case 3: return d >= eps;
    default: return d > eps;
  }
}

/* pass 1 (indices == NULL): counts[r]; pass 2: fill indices/weights at indptr[r] */
void orc_eps(const uint8_t *T, int64_t n, int l, int64_t row0, int64_t nrows, int cmp, double eps,
             int64_t *counts, const int64_t *indptr, int32_t *indices, uint8_t *weights) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t r = 0; r < nrows; ++r) {
    const uint8_t *a = T + (row0 + r) * l;
    int64_t k = 0, base = indices ? indptr[r] : 0;
    for (int64_t c = 0; c < n; ++c) {
      int d = ham(a, T + c * l, l);
      if (d > 0 && cmp_ok(cmp, (double)d, eps)) {
        if (indices) { indices[base + k] = (int32_t)c; weights[base + k] = (uint8_t)d; }
        ++k;
      }
    }
    if (!indices) counts[r] = k;
  }
}

/* canonical kNN: k+1 smallest (d, column), rank 0 dropped; missing ranks -> -1 / 255 */
void orc_knn(const uint8_t *T, int64_t n, int l, int64_t row0, int64_t nrows, int k, int32_t *idx, uint8_t *dist) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t r = 0; r < nrows; ++r) {
    const uint8_t *a = T + (row0 + r) * l;
    int64_t best[1026];                      /* k <= 1024 */
    int nb = 0;
    for (int64_t c = 0; c < n; ++c) {
      int64_t key = ((int64_t)ham(a, T + c * l, l) << 32) | c;
      if (nb == k + 1 && key >= best[nb - 1]) continue;
      int p = nb < k + 1 ? nb++ : nb - 1;
      while (p > 0 && best[p - 1] > key) { best[p] = best[p - 1]; --p; }
      best[p] = key;
    }
    for (int j = 0; j < k; ++j) {
      if (j + 1 < nb) { idx[r * k + j] = (int32_t)(best[j + 1] & 0xffffffff); dist[r * k + j] = (uint8_t)(best[j + 1] >> 32); }
      else { idx[r * k + j] = -1; dist[r * k + j] = 255; }
    }
  }
}

/* ---------------------------------------------------------------------------------------
 * FAST LEG of orc_eps / orc_knn for full-size checks (N = 200 000: 4e10 pairs).  The same definitions -
 * d = number of differing token bytes; (comp(d, eps) & d > 0), columns ascending; the k + 1 smallest
 * (d, column) keys, rank 0 dropped - with the byte compare done 32 bytes at a time (AVX2 vpcmpeqb +
 * movemask + popcount; plain 8-byte words where AVX2 is missing).  Pinned: tests/test_oracle.py checks it
 * against the scalar functions above on ragged lengths.
 * ------------------------------------------------------------------------------------- */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2,popcnt"))) static int ham_avx2(const uint8_t *a, const uint8_t *b, int l) {
  int eq = 0, j = 0;
  for (; j + 32 <= l; j += 32) {
    const __m256i x = _mm256_loadu_si256((const __m256i *)(a + j)), y = _mm256_loadu_si256((const __m256i *)(b + j));
    eq += __builtin_popcount((unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, y)));
  }
  int d = j - eq;
  for (; j < l; ++j) d += a[j] != b[j];
  return d;
}
#endif
static int ham_words(const uint8_t *a, const uint8_t *b, int l) {
  int d = 0, j = 0;
  for (; j + 8 <= l; j += 8) {
    uint64_t x, y;
    memcpy(&x, a + j, 8); memcpy(&y, b + j, 8);
    x ^= y;
    x |= x >> 4; x |= x >> 2; x |= x >> 1;                /* bit 0 of every byte: the byte differs */
    d += __builtin_popcountll(x & 0x0101010101010101ULL);
  }
  for (; j < l; ++j) d += a[j] != b[j];
  return d;
}
typedef int (*ham_fn)(const uint8_t *, const uint8_t *, int);
static ham_fn pick_ham(void) {
#if defined(__x86_64__)
  if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt")) return ham_avx2;
#endif
  return ham_words;
}

void orc_eps_fast(const uint8_t *T, int64_t n, int l, int64_t row0, int64_t nrows, int cmp, double eps,
                  int64_t *counts, const int64_t *indptr, int32_t *indices, uint8_t *weights) {
  const ham_fn hf = pick_ham();
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t r = 0; r < nrows; ++r) {
    const uint8_t *a = T + (row0 + r) * l;
    int64_t k = 0, base = indices ? indptr[r] : 0;
    for (int64_t c = 0; c < n; ++c) {
      const int d = hf(a, T + c * l, l);
      if (d > 0 && cmp_ok(cmp, (double)d, eps)) {
        if (indices) { indices[base + k] = (int32_t)c; weights[base + k] = (uint8_t)d; }
        ++k;
      }
    }
    if (!indices) counts[r] = k;
  }
}

void orc_knn_fast(const uint8_t *T, int64_t n, int l, int64_t row0, int64_t nrows, int k, int32_t *idx, uint8_t *dist) {
  const ham_fn hf = pick_ham();
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t r = 0; r < nrows; ++r) {
    const uint8_t *a = T + (row0 + r) * l;
    int64_t best[1026];                      /* k <= 1024 */
    int nb = 0;
    for (int64_t c = 0; c < n; ++c) {
      const int64_t key = ((int64_t)hf(a, T + c * l, l) << 32) | c;
      if (nb == k + 1 && key >= best[nb - 1]) continue;
      int p = nb < k + 1 ? nb++ : nb - 1;
      while (p > 0 && best[p - 1] > key) { best[p] = best[p - 1]; --p; }
      best[p] = key;
    }
    for (int j = 0; j < k; ++j) {
      if (j + 1 < nb) { idx[r * k + j] = (int32_t)(best[j + 1] & 0xffffffff); dist[r * k + j] = (uint8_t)(best[j + 1] >> 32); }
      else { idx[r * k + j] = -1; dist[r * k + j] = 255; }
    }
  }
}

static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline uint64_t hh(uint64_t seed, uint64_t stream, uint64_t i) {
  return mix64(seed + 0x9E3779B97F4A7C15ULL * (4 * i + stream + 1));
}

void orc_synth(int64_t n, int l, uint64_t seed, int64_t members, uint8_t *out) {
  int64_t nc = n / members;
  if (nc < 1) nc = 1;
  for (int64_t i = 0; i < n; ++i) {
    int64_t c = i % nc;
    uint8_t *row = out + i * l;
    for (int j = 0; j < l; ++j) row[j] = (uint8_t)(1 + hh(seed, 0, (uint64_t)(c * l + j)) % 20);
    int m = 1 + (int)(hh(seed, 1, (uint64_t)i) % 3);
    /* positions and steps are drawn from the UNMUTATED row index stream; substitutions apply in
       order t = 0,1,2 and each reads the token as left by the previous ones */
    for (int t = 0; t < m; ++t) {
      int pos = (int)(hh(seed, 2, (uint64_t)(8 * i + t)) % (uint64_t)l);
      int step = (int)(hh(seed, 3, (uint64_t)(8 * i + t)) % 19);
      row[pos] = (uint8_t)(1 + ((row[pos] - 1 + 1 + step) % 20));
    }
  }
}

/* ---------------------------------------------------------------------------------------
 * Banded Levenshtein kNN — BUILD DEFINED (no reference counterpart; parity unpinned).
 * d(a,b) = min(edit_distance(a,b), band+1) on the non-zero prefixes; plain Wagner–Fischer
 * restricted to |i-j| <= band (exact whenever the true distance is <= band, > band otherwise),
 * then the canonical (d, column) order, rank 0 dropped, ranks 1..k.
 * ------------------------------------------------------------------------------------- */
static int seq_len(const uint8_t *a, int l) {
  int n = 0;
  while (n < l && a[n] != 0) ++n;
  return n;
}

int orc_lev_pair(const uint8_t *a, int la, const uint8_t *b, int lb, int band) {
  const int cap = band + 1, INF = 1 << 20;
  if (la - lb > band || lb - la > band) return cap;
  int prev[130], cur[130];
  for (int j = 0; j <= lb; ++j) prev[j] = j <= band ? j : INF;
  for (int i = 1; i <= la; ++i) {
    int lo = i - band < 0 ? 0 : i - band, hi = i + band > lb ? lb : i + band;
    for (int j = 0; j <= lb; ++j) cur[j] = INF;
    if (lo == 0) cur[0] = i;
    for (int j = lo < 1 ? 1 : lo; j <= hi; ++j) {
      int v = prev[j - 1] + (a[i - 1] != b[j - 1]);
      if (prev[j] + 1 < v) v = prev[j] + 1;
      if (cur[j - 1] + 1 < v) v = cur[j - 1] + 1;
      cur[j] = v;
    }
    memcpy(prev, cur, sizeof(int) * (size_t)(lb + 1));
  }
  return prev[lb] < cap ? prev[lb] : cap;
}

void orc_lev_knn(const uint8_t *T, int64_t n, int l, int64_t row0, int64_t nrows, int band, int k,
                 int32_t *idx, uint8_t *dist) {
  int *lens = (int *)malloc(sizeof(int) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) lens[i] = seq_len(T + i * l, l);
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t r = 0; r < nrows; ++r) {
    const uint8_t *a = T + (row0 + r) * l;
    int64_t best[65];
    int nb = 0;
    for (int64_t c = 0; c < n; ++c) {
      int64_t key = ((int64_t)orc_lev_pair(a, lens[row0 + r], T + c * l, lens[c], band) << 32) | c;
      if (nb == k + 1 && key >= best[nb - 1]) continue;
      int p = nb < k + 1 ? nb++ : nb - 1;
      while (p > 0 && best[p - 1] > key) { best[p] = best[p - 1]; --p; }
      best[p] = key;
    }
    for (int j = 0; j < k; ++j) {
      if (j + 1 < nb) { idx[r * k + j] = (int32_t)(best[j + 1] & 0xffffffff); dist[r * k + j] = (uint8_t)(best[j + 1] >> 32); }
      else { idx[r * k + j] = -1; dist[r * k + j] = 255; }
    }
  }
  free(lens);
}
