// Minkowski (p = 2) distances of fp16 embeddings and the epsilon / kNN selection on them
// (SURVEY.md §8 f2): the reference's `build_graph(representation="Embedded", distance=minkowski)`
// (prograph/distance/minkowski.py:8-41 through prograph/prograph.py:726-764).
//
// The reference stages the embedding as fp16 (`torch.as_tensor(..., dtype=float16)`, :726) and then
// evaluates  pow(sum(pow(X - Y[:,None,:], 2), axis=2), 1/2)  with fp16 tensors: EVERY elementwise
// step rounds to fp16 (the difference, its square, the sum, the root - and 1/(1+d) twice for
// similarities); only the sum itself accumulates wider.  Its own tests pin that rounding
// (tests/tests.py:164-167: sqrt(0.625) -> 0.79052734).  The kernel reproduces the same sequence:
//     diff = v_pk_add_f16(x, -y)        rounded to fp16, two elements per instruction
//     sq   = v_pk_mul_f16(diff, diff)   rounded to fp16
//     acc += sq.lo + sq.hi              v_dot2c_f32_f16 against (1, 1): exact products, fp32 accumulation in
//                                       element order like the reference's float accumulator (whose own
//                                       order is implementation defined: torch vectorises it)
//     d    = fp16(sqrt(float(fp16(acc))))
// so distances are bit-identical to the reference on the goldens up to D = 64 and differ by one fp16 ulp
// on < 0.05 % of the pairs at D = 1280 (measured against reference-generated goldens, tests/golden/minkowski_f16.npz).
// This is why the ||x||^2 + ||y||^2 - 2 x.y form on the matrix cores is NOT used: it is ~10x cheaper at
// large D but does not round like the reference.
//
// Layout: embeddings are packed chunk-major like the token planes: chunk q (8 halfs, 16 bytes) of
// vector n at byte (q * Npad + n) * 16, so 64 consecutive vectors load one chunk each as a coalesced
// 1 KiB global_load_dwordx4.
#include "pg_common.h"
#include "../../include/prograph_hip.h"

#include <hip/hip_fp16.h>
#include <stdio.h>

typedef _Float16 pg_h2 __attribute__((ext_vector_type(2)));

#define MK_ROWS 16          // rows (Y vectors) per workgroup of the dense kernel
#define MK_SEG 16           // chunks staged per segment: 128 halfs of every row in LDS

__global__ __launch_bounds__(256) void pg_pack_f16_kernel(const __half *__restrict__ src, long long n, int d, long long ld,
                                                          const long long *__restrict__ rows, uint4 *__restrict__ out,
                                                          long long npad, int nq) {
  const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
  if (s >= npad) return;
  const __half *row = s < n ? src + (rows ? rows[s] : s) * ld : nullptr;
  for (int q = 0; q < nq; ++q) {
    union { uint4 v; __half h[8]; } u;
    u.v = make_uint4(0, 0, 0, 0);
    if (row) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int pos = q * 8 + j;
        if (pos < d) u.h[j] = row[pos];
      }
    }
    out[(long long)q * npad + s] = u.v;
  }
}

// (M, N) fp16 distance (or similarity) matrix: out[m * ldo + n] = minkowski2(Y[m], X[n])
__global__ __launch_bounds__(256) void pg_mink_dense_kernel(const uint4 *__restrict__ xp, long long n, long long xnpad,
                                                            const uint4 *__restrict__ yp, long long m, long long ynpad,
                                                            int nq, int similarity, __half *__restrict__ out, long long ldo) {
  __shared__ uint4 ybuf[MK_ROWS][MK_SEG];
  const long long col = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long r0 = (long long)blockIdx.y * MK_ROWS;
  const int nr = (int)((m - r0) < MK_ROWS ? (m - r0) : MK_ROWS);
  const long long c = col < xnpad ? col : xnpad - 1;
  float acc[MK_ROWS];
#pragma unroll
  for (int r = 0; r < MK_ROWS; ++r) acc[r] = 0.0f;
  const pg_h2 ones = {(_Float16)1.0f, (_Float16)1.0f};
  for (int q0 = 0; q0 < nq; q0 += MK_SEG) {
    __syncthreads();
    {
      const int rr = threadIdx.x / MK_SEG, qq = threadIdx.x % MK_SEG;     // 16 x 16 = 256 chunks per segment
      uint4 v = make_uint4(0, 0, 0, 0);
      if (rr < nr && q0 + qq < nq) v = yp[(long long)(q0 + qq) * ynpad + r0 + rr];
      ybuf[rr][qq] = v;
    }
    __syncthreads();
    const int qn = nq - q0 < MK_SEG ? nq - q0 : MK_SEG;
    for (int qq = 0; qq < qn; ++qq) {
      const uint4 xv = xp[(long long)(q0 + qq) * xnpad + c];
      const pg_h2 x0 = __builtin_bit_cast(pg_h2, xv.x), x1 = __builtin_bit_cast(pg_h2, xv.y);
      const pg_h2 x2 = __builtin_bit_cast(pg_h2, xv.z), x3 = __builtin_bit_cast(pg_h2, xv.w);
#pragma unroll
      for (int r = 0; r < MK_ROWS; ++r) {
        const uint4 yv = ybuf[r][qq];
        const pg_h2 d0 = x0 - __builtin_bit_cast(pg_h2, yv.x), d1 = x1 - __builtin_bit_cast(pg_h2, yv.y);
        const pg_h2 d2 = x2 - __builtin_bit_cast(pg_h2, yv.z), d3 = x3 - __builtin_bit_cast(pg_h2, yv.w);
        float p = __builtin_amdgcn_fdot2(d0 * d0, ones, acc[r], false);
        p = __builtin_amdgcn_fdot2(d1 * d1, ones, p, false);
        p = __builtin_amdgcn_fdot2(d2 * d2, ones, p, false);
        acc[r] = __builtin_amdgcn_fdot2(d3 * d3, ones, p, false);
      }
    }
  }
  if (col < n) {
#pragma unroll
    for (int r = 0; r < MK_ROWS; ++r) {
      if (r < nr) {
        const _Float16 s16 = (_Float16)acc[r];                          // the float sum as fp16
        __half d16 = __float2half_rn(sqrtf((float)s16));                  // pow(., 1/2) on the fp16 value
        if (similarity) {                                               // 1 / (1 + d): two fp16 roundings (minkowski.py:40)
          const __half t = __float2half_rn(1.0f + __half2float(d16));
          d16 = __float2half_rn(1.0f / __half2float(t));
        }
        out[(r0 + r) * ldo + col] = d16;
      }
    }
  }
}

// sortable 16-bit key of a non-negative fp16 value: ascending distance, or descending similarity
__device__ __forceinline__ u32 mk_key(unsigned short bits, int descending) {
  return descending ? (0xFFFFu - (u32)bits) : (u32)bits;
}

// ranks first .. first+k-1 of every row's (key, column) order (the stable sort of :758-760), one wave per row
__global__ __launch_bounds__(256) void pg_f16_knn_kernel(const unsigned short *__restrict__ dist, long long m, long long n,
                                                         long long ld, int k, int first, int descending, int *__restrict__ idx,
                                                         unsigned short *__restrict__ w) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const unsigned short *d = dist + row * ld;
  u32 lk = 0xFFFFFFFFu, lc = 0xFFFFFFFFu;                 // lane j = j-th smallest (key, column)
  const int last = first + k - 1;
  u32 tk = 0xFFFFFFFFu, tc = 0xFFFFFFFFu;                 // current entry of lane `last`
  for (long long c0 = 0; c0 < n; c0 += 64) {
    const long long c = c0 + lane;
    const u32 key = c < n ? mk_key(d[c], descending) : 0xFFFFFFFFu;
    bool cand = c < n && (key < tk || (key == tk && (u32)c < tc));
    u64 mask = __builtin_amdgcn_ballot_w64(cand);
    while (mask) {
      const int j = __builtin_ctzll(mask);
      mask &= mask - 1;
      const u32 xk = __builtin_amdgcn_readlane(key, j), xc = (u32)(c0 + j);
      if (xk < tk || (xk == tk && xc < tc)) {
        const bool keep = lk < xk || (lk == xk && lc <= xc);            // entries not after x stay
        const u32 pk = wave_shr1(lk, 0u), pc = wave_shr1(lc, 0u);
        const bool prev_after = pk > xk || (pk == xk && pc > xc);        // lane-1's entry also moves: take it, else x lands here
        lk = keep ? lk : (prev_after ? pk : xk);
        lc = keep ? lc : (prev_after ? pc : xc);
        tk = __builtin_amdgcn_readlane(lk, last);
        tc = __builtin_amdgcn_readlane(lc, last);
      }
    }
  }
  if (lane >= first && lane <= last) {
    const long long o = row * (long long)k + (lane - first);
    idx[o] = lc == 0xFFFFFFFFu ? -1 : (int)lc;
    w[o] = lc == 0xFFFFFFFFu ? 0 : d[lc];
  }
}

// epsilon selection on a distance / similarity block: count, or fill at indptr
//   distances:    comp(d, eps) & (d > 0)      (prograph.py:736)
//   similarities: comp(eps, s) & (s < 1)      (:734), eps already 1/(1+eps) rounded to fp16
__device__ __forceinline__ bool mk_match(float v, float eps, int cmp, int similarity) {
  const float a = similarity ? eps : v, b = similarity ? v : eps;
  bool ok;
  switch (cmp) {
    case PG_CMP_LE: ok = a <= b; break;
    case PG_CMP_LT: ok = a < b; break;
    case PG_CMP_EQ: ok = a == b; break;
    case PG_CMP_GE: ok = a >= b; break;
    default: ok = a > b; break;
  }
  return ok && (similarity ? v < 1.0f : v > 0.0f);
}

__global__ __launch_bounds__(256) void pg_f16_eps_kernel(const __half *__restrict__ dist, long long m, long long n, long long ld,
                                                         int cmp, float eps, int similarity, u32 *__restrict__ counts,
                                                         const long long *__restrict__ indptr, int *__restrict__ indices,
                                                         __half *__restrict__ weights) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const __half *d = dist + row * ld;
  long long run = indptr ? indptr[row] : 0;
  u32 cnt = 0;
  for (long long c0 = 0; c0 < n; c0 += 64) {
    const long long c = c0 + lane;
    const __half v = c < n ? d[c] : __float2half(0.0f);
    const bool hit = c < n && mk_match(__half2float(v), eps, cmp, similarity);
    const u64 mask = __builtin_amdgcn_ballot_w64(hit);
    if (indptr && hit) {
      const long long o = run + mask_rank(mask);
      indices[o] = (int)c;
      weights[o] = v;
    }
    run += __popcll(mask);
    cnt += (u32)__popcll(mask);
  }
  if (!indptr && lane == 0) counts[row] = cnt;
}

static int mfail(int code, const char *msg) {
  pg_set_error(msg);
  return code;
}
static int mlaunched(const char *where) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[200];
    snprintf(buf, sizeof(buf), "%s: %s", where, hipGetErrorString(e));
    pg_set_error(buf);
    return (int)e;
  }
  return 0;
}

extern "C" {

int pg_f16_nchunks(int d) { return d <= 0 ? 1 : (d + 7) / 8; }

int pg_pack_f16(const void *src, int64_t n, int d, int64_t ld, const int64_t *rows, void *packed, int64_t npad, void *stream) {
  if (!src || !packed || n < 0 || d <= 0 || ld < d) return mfail(PG_E_BADARG, "pg_pack_f16: bad argument");
  if (npad < n || npad % 256) return mfail(PG_E_BADARG, "pg_pack_f16: npad must be pg_npad(n)");
  pg_pack_f16_kernel<<<dim3((unsigned)(npad / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      (const __half *)src, n, d, ld, (const long long *)rows, (uint4 *)packed, npad, pg_f16_nchunks(d));
  return mlaunched("pg_pack_f16_kernel");
}

int pg_minkowski_dense(const void *x_packed, int64_t n, int64_t x_npad, const void *y_packed, int64_t m, int64_t y_npad,
                       int d, int similarity, void *out_f16, int64_t ldo, void *stream) {
  if (!x_packed || !y_packed || !out_f16 || n <= 0 || m <= 0 || d <= 0 || ldo < n)
    return mfail(PG_E_BADARG, "pg_minkowski_dense: bad argument");
  if (x_npad < n || x_npad % 256 || y_npad < m) return mfail(PG_E_BADARG, "pg_minkowski_dense: bad npad");
  if ((m + MK_ROWS - 1) / MK_ROWS > 65535) return mfail(PG_E_BADARG, "pg_minkowski_dense: m too large for one launch");
  const dim3 grid((unsigned)((n + 255) / 256), (unsigned)((m + MK_ROWS - 1) / MK_ROWS));
  pg_mink_dense_kernel<<<grid, dim3(256), 0, (hipStream_t)stream>>>((const uint4 *)x_packed, n, x_npad, (const uint4 *)y_packed, m,
                                                                    y_npad, pg_f16_nchunks(d), similarity ? 1 : 0,
                                                                    (__half *)out_f16, ldo);
  return mlaunched("pg_mink_dense_kernel");
}

int pg_f16_knn(const void *dist_f16, int64_t m, int64_t n, int64_t ld, int k, int first, int descending, int32_t *idx_out,
               void *w_out_f16, void *stream) {
  if (!dist_f16 || !idx_out || !w_out_f16 || m <= 0 || n <= 0 || ld < n) return mfail(PG_E_BADARG, "pg_f16_knn: bad argument");
  if (k < 1 || first < 0 || first + k > 64) return mfail(PG_E_BADARG, "pg_f16_knn: first + k must be at most 64");
  pg_f16_knn_kernel<<<dim3((unsigned)((m + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
      (const unsigned short *)dist_f16, m, n, ld, k, first, descending ? 1 : 0, idx_out, (unsigned short *)w_out_f16);
  return mlaunched("pg_f16_knn_kernel");
}

int pg_f16_eps_count(const void *dist_f16, int64_t m, int64_t n, int64_t ld, int cmp, float eps_f16, int similarity,
                     uint32_t *counts, void *stream) {
  if (!dist_f16 || !counts || m <= 0 || n <= 0 || ld < n || cmp < PG_CMP_LE || cmp > PG_CMP_GT)
    return mfail(PG_E_BADARG, "pg_f16_eps_count: bad argument");
  pg_f16_eps_kernel<<<dim3((unsigned)((m + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
      (const __half *)dist_f16, m, n, ld, cmp, eps_f16, similarity ? 1 : 0, counts, nullptr, nullptr, nullptr);
  return mlaunched("pg_f16_eps_kernel(count)");
}

int pg_f16_eps_fill(const void *dist_f16, int64_t m, int64_t n, int64_t ld, int cmp, float eps_f16, int similarity,
                    const int64_t *indptr, int32_t *indices, void *weights_f16, void *stream) {
  if (!dist_f16 || !indptr || !indices || !weights_f16 || m <= 0 || n <= 0 || ld < n || cmp < PG_CMP_LE || cmp > PG_CMP_GT)
    return mfail(PG_E_BADARG, "pg_f16_eps_fill: bad argument");
  pg_f16_eps_kernel<<<dim3((unsigned)((m + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
      (const __half *)dist_f16, m, n, ld, cmp, eps_f16, similarity ? 1 : 0, nullptr, (const long long *)indptr, indices,
      (__half *)weights_f16);
  return mlaunched("pg_f16_eps_kernel(fill)");
}

}  // extern "C"
