// Test infrastructure: the C ABI of libprograph_hip.so used WITHOUT Python or torch — plain
// hipMalloc'ed buffers, the declarations of include/prograph_hip.h, the default stream — checked
// against the C oracle (oracle/oracle.c, linked as liboracle.so).  This is what a non-Python host
// (or the cgo / JNI / ctypes stub of INTEGRATION.md) does.  Built by tests/capi/Makefile, run by
// tests/test_gpu_native.py::test_c_abi_without_python.
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "prograph_hip.h"

extern "C" {
void orc_synth(int64_t n, int l, uint64_t seed, int64_t members, uint8_t *out);
void orc_knn(const uint8_t *T, int64_t n, int l, int64_t row0, int64_t nrows, int k, int32_t *idx, uint8_t *dist);
void orc_eps(const uint8_t *T, int64_t n, int l, int64_t row0, int64_t nrows, int cmp, double eps, int64_t *counts,
             const int64_t *indptr, int32_t *indices, uint8_t *weights);
}

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define PG(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, pg_last_error()); return 3; } } while (0)

template <typename T> static T *dmalloc(size_t n) { void *p = nullptr; return hipMalloc(&p, (n ? n : 1) * sizeof(T)) == hipSuccess ? (T *)p : nullptr; }

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 40000;
  const int l = argc > 2 ? atoi(argv[2]) : 64, k = 16, cap = 128;
  const double eps = 2;
  printf("libprograph_hip ABI %d, n=%lld l=%d\n", pg_version(), (long long)n, l);
  std::vector<uint8_t> tok((size_t)n * l);
  orc_synth(n, l, 20260104ull, 256, tok.data());

  // ---- device side: tokens -> planes
  const int64_t npad = pg_npad(n);
  uint8_t *d_tok = dmalloc<uint8_t>((size_t)n * l);
  uint8_t *d_planes = dmalloc<uint8_t>((size_t)pg_planes_bytes(n, l, PG_BITS_5));
  uint32_t *d_flag = dmalloc<uint32_t>(1);
  HIP(hipMemcpy(d_tok, tok.data(), (size_t)n * l, hipMemcpyHostToDevice));
  PG(pg_pack_planes(d_tok, 1, n, l, l, nullptr, PG_BITS_5, d_planes, npad, d_flag, nullptr));
  uint32_t flag = 1;
  HIP(hipMemcpy(&flag, d_flag, 4, hipMemcpyDeviceToHost));
  if (flag) { fprintf(stderr, "pack flagged a token\n"); return 4; }

  // ---- the path's collective through the C ABI: a one-rank RCCL communicator gathers the shard onto itself
  {
    unsigned char id[PG_COMM_ID_BYTES];
    void *comm = nullptr;
    PG(pg_comm_unique_id(id));
    PG(pg_comm_init(&comm, 1, 0, id));
    uint8_t *d_full = dmalloc<uint8_t>((size_t)n * l);
    PG(pg_allgather_tokens(comm, d_tok, n, l, d_full, nullptr));
    HIP(hipDeviceSynchronize());
    std::vector<uint8_t> back((size_t)n * l);
    HIP(hipMemcpy(back.data(), d_full, (size_t)n * l, hipMemcpyDeviceToHost));
    if (memcmp(back.data(), tok.data(), (size_t)n * l)) { fprintf(stderr, "pg_allgather_tokens: gathered matrix differs\n"); return 9; }
    PG(pg_comm_destroy(comm));
    printf("pg_allgather_tokens (RCCL, 1 rank): %lld x %d tokens gathered, identical\n", (long long)n, l);
  }

  // ---- kNN
  int32_t *d_idx = dmalloc<int32_t>((size_t)n * k);
  uint8_t *d_dist = dmalloc<uint8_t>((size_t)n * k);
  // launch-private workspace of the all-pairs calls (one per launch in flight; these launches are serialised on one stream)
  void *d_ws = dmalloc<uint8_t>((size_t)pg_workspace_bytes(n));
  PG(pg_knn_hamming(d_planes, npad, 0, n, d_planes, npad, n, l, PG_BITS_5, k, d_idx, d_dist, d_ws, nullptr));
  std::vector<int32_t> idx((size_t)n * k), ridx((size_t)n * k);
  std::vector<uint8_t> dist((size_t)n * k), rdist((size_t)n * k);
  HIP(hipMemcpy(idx.data(), d_idx, idx.size() * 4, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(dist.data(), d_dist, dist.size(), hipMemcpyDeviceToHost));
  orc_knn(tok.data(), n, l, 0, n, k, ridx.data(), rdist.data());
  if (memcmp(idx.data(), ridx.data(), idx.size() * 4) || memcmp(dist.data(), rdist.data(), dist.size())) { fprintf(stderr, "kNN differs from the oracle\n"); return 5; }
  printf("kNN k=%d: identical to the oracle\n", k);

  // ---- eps CSR, rectangular and symmetric entry points
  std::vector<int64_t> rcnt(n), rptr(n + 1, 0);
  orc_eps(tok.data(), n, l, 0, n, PG_CMP_LE, eps, rcnt.data(), nullptr, nullptr, nullptr);
  for (int64_t i = 0; i < n; ++i) rptr[i + 1] = rptr[i] + rcnt[i];
  const int64_t rnnz = rptr[n];
  std::vector<int32_t> rind(rnnz ? rnnz : 1);
  std::vector<uint8_t> rw(rnnz ? rnnz : 1);
  orc_eps(tok.data(), n, l, 0, n, PG_CMP_LE, eps, nullptr, rptr.data(), rind.data(), rw.data());

  int32_t *d_sidx = dmalloc<int32_t>((size_t)n * cap);
  uint8_t *d_sw = dmalloc<uint8_t>((size_t)n * cap);
  uint32_t *d_cnt = dmalloc<uint32_t>(n), *d_lo = dmalloc<uint32_t>(n), *d_tot = dmalloc<uint32_t>(n);
  int64_t *d_ptr = dmalloc<int64_t>(n + 1);
  void *d_scratch = dmalloc<uint8_t>((size_t)pg_scan_scratch_bytes(n));
  for (int sym = 0; sym < 2; ++sym) {
    std::vector<uint32_t> cnt(n), lo(n, 0);
    if (sym) {
      PG(pg_eps_slots_sym(d_planes, npad, n, l, PG_BITS_5, PG_CMP_LE, eps, cap, d_sidx, d_sw, d_cnt, d_lo, d_ws, nullptr));
      HIP(hipMemcpy(lo.data(), d_lo, n * 4, hipMemcpyDeviceToHost));
    } else {
      PG(pg_eps_slots(d_planes, npad, 0, n, d_planes, npad, n, l, PG_BITS_5, PG_CMP_LE, eps, cap, d_sidx, d_sw, d_cnt, d_ws, nullptr));
    }
    HIP(hipMemcpy(cnt.data(), d_cnt, n * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) cnt[i] += lo[i];           // totals (a device add in a real host)
    HIP(hipMemcpy(d_tot, cnt.data(), n * 4, hipMemcpyHostToDevice));
    PG(pg_exclusive_scan(d_tot, n, d_ptr, d_scratch, nullptr));
    std::vector<int64_t> ptr(n + 1);
    HIP(hipMemcpy(ptr.data(), d_ptr, (n + 1) * 8, hipMemcpyDeviceToHost));
    if (memcmp(ptr.data(), rptr.data(), (n + 1) * 8)) { fprintf(stderr, "eps indptr differs (sym=%d)\n", sym); return 6; }
    int32_t *d_ind = dmalloc<int32_t>(rnnz);
    uint8_t *d_w = dmalloc<uint8_t>(rnnz);
    if (sym) PG(pg_eps_compact_sym(d_planes, npad, n, l, PG_BITS_5, PG_CMP_LE, eps, cap, d_sidx, d_sw, d_cnt, d_lo, d_ptr, d_ind, d_w, 0, nullptr));
    else PG(pg_eps_compact(d_planes, npad, 0, n, d_planes, npad, n, l, PG_BITS_5, PG_CMP_LE, eps, cap, d_sidx, d_sw, d_cnt, d_ptr, d_ind, d_w, 0, nullptr));
    std::vector<int32_t> ind(rnnz ? rnnz : 1);
    std::vector<uint8_t> w(rnnz ? rnnz : 1);
    HIP(hipMemcpy(ind.data(), d_ind, rnnz * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(w.data(), d_w, rnnz, hipMemcpyDeviceToHost));
    if (memcmp(ind.data(), rind.data(), rnnz * 4) || memcmp(w.data(), rw.data(), rnnz)) { fprintf(stderr, "eps CSR differs (sym=%d)\n", sym); return 7; }
    printf("eps<=%g CSR (%s entry points): nnz=%lld, identical to the oracle\n", eps, sym ? "symmetric" : "rectangular", (long long)rnnz);
    (void)hipFree(d_ind); (void)hipFree(d_w);
  }
  printf("C ABI OK\n");
  return 0;
}
