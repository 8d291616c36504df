// Probe of v_mfma_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands on gfx950: operand layout and exactness for the
// small integers the signature filter needs (+-1, 0, and bias elements up to +-6).
// Hypothesis checked against a CPU product: lane l holds row/column l & 31, K elements 32 * (l >> 5) .. + 31,
// element j in nibble j of the lane's 128 bits (VGPR j / 8, bits 4 * (j % 8)).
//   hipcc --offload-arch=gfx950 -O3 mfma_fp4.hip -o mfma_fp4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const unsigned *a, const unsigned *b, float *d) {
  const int l = threadIdx.x;
  v8i A = {0, 0, 0, 0, 0, 0, 0, 0}, B = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) { A[i] = (int)a[l * 4 + i]; B[i] = (int)b[l * 4 + i]; }
  v16f c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[l * 16 + r] = c[r];
}

static const float kVal[16] = {0, 0.5f, 1, 1.5f, 2, 3, 4, 6, -0.0f, -0.5f, -1, -1.5f, -2, -3, -4, -6};
int main() {
  srand(3);
  std::vector<unsigned char> An(32 * 64), Bn(64 * 32);                 // nibble codes
  const unsigned char pool[] = {0x0, 0x2, 0xA, 0x2, 0xA, 0x7, 0xF, 0x4, 0x5, 0x6};   // 0, +-1, +-6, 2, 3, 4
  for (auto &x : An) x = pool[rand() % 10];
  for (auto &x : Bn) x = (rand() & 1) ? 0x2 : 0x0;
  std::vector<unsigned> a(64 * 4, 0), b(64 * 4, 0);
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 32; ++j) {
      const int kk = 32 * (l >> 5) + j;
      a[l * 4 + j / 8] |= (unsigned)An[(l & 31) * 64 + kk] << (4 * (j % 8));
      b[l * 4 + j / 8] |= (unsigned)Bn[kk * 32 + (l & 31)] << (4 * (j % 8));
    }
  unsigned *da, *db; float *dd;
  hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dd, 64 * 16 * 4);
  hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(da, db, dd);
  std::vector<float> d(64 * 16);
  hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
      float ref = 0;
      for (int kk = 0; kk < 64; ++kk) ref += kVal[An[row * 64 + kk]] * kVal[Bn[kk * 32 + col]];
      if (ref != d[l * 16 + r]) { if (bad < 8) printf("mismatch lane %d reg %d (row %d col %d): got %g want %g\n", l, r, row, col, d[l * 16 + r], ref); ++bad; }
    }
  printf("%s: %d mismatches of 1024\n", bad ? "LAYOUT WRONG" : "layout confirmed, products exact", bad);
  return bad != 0;
}
