// Micro-benchmark: one row group of the folded-exact bound (pg_mm.h, tile_folded) - 4 rows x 4 column slices of
// (xor + 4 bitop3) + 16 popcounts + 16 alignbit - with the row folds as SGPR operands (A) or VGPR operands (B),
// and the logic ops alone (C: SGPR, D: VGPR).  Prints SIMD cycles per group at 1/2/4/8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 fold_group.hip -o fold_group
#include <hip/hip_runtime.h>
#include <cstdio>

#define ROW_S(t, c, r) "v_xor_b32 " t ", " r "0, " c "0\n v_bitop3_b32 " t ", " r "1, " c "1, " t " bitop3:0xbe\n v_bitop3_b32 " t ", " r "2, " c "2, " t " bitop3:0xbe\n v_bitop3_b32 " t ", " r "3, " c "3, " t " bitop3:0xbe\n v_bitop3_b32 " t ", " r "4, " c "4, " t " bitop3:0xbe\n"

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed, int iters) {
  unsigned c[4][5], t[16], acc[4] = {0, 0, 0, 0}, rv[5];
  for (int b = 0; b < 4; ++b) for (int p = 0; p < 5; ++p) c[b][p] = threadIdx.x * (7 + b * 5 + p) + seed;
  for (int p = 0; p < 5; ++p) rv[p] = threadIdx.x * 3 + p + seed;
  for (int i = 0; i < 16; ++i) t[i] = 0;
  unsigned s0 = seed, s1 = seed * 3, s2 = seed * 5, s3 = seed * 7, s4 = seed * 11, nb = 0u - 3u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (MODE == 0 || MODE == 2)
          asm volatile("v_xor_b32 %0, %1, %6\n v_bitop3_b32 %0, %2, %7, %0 bitop3:0xbe\n v_bitop3_b32 %0, %3, %8, %0 bitop3:0xbe\n v_bitop3_b32 %0, %4, %9, %0 bitop3:0xbe\n v_bitop3_b32 %0, %5, %10, %0 bitop3:0xbe\n"
                       : "+v"(t[u * 4 + b]) : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "v"(c[b][0]), "v"(c[b][1]), "v"(c[b][2]), "v"(c[b][3]), "v"(c[b][4]));
        else
          asm volatile("v_xor_b32 %0, %1, %6\n v_bitop3_b32 %0, %2, %7, %0 bitop3:0xbe\n v_bitop3_b32 %0, %3, %8, %0 bitop3:0xbe\n v_bitop3_b32 %0, %4, %9, %0 bitop3:0xbe\n v_bitop3_b32 %0, %5, %10, %0 bitop3:0xbe\n"
                       : "+v"(t[u * 4 + b]) : "v"(rv[0]), "v"(rv[1]), "v"(rv[2]), "v"(rv[3]), "v"(rv[4]), "v"(c[b][0]), "v"(c[b][1]), "v"(c[b][2]), "v"(c[b][3]), "v"(c[b][4]));
      }
    }
    if (MODE < 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_bcnt_u32_b32 %0, %0, %2\n v_alignbit_b32 %1, %1, %0, 31\n" : "+v"(t[i]), "+v"(acc[i & 3]) : "s"(nb));
    }
    if (MODE == 4 || MODE == 5) {                           // E / F: popcounts, then the sign shifts (F: VGPR addend)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == 4) asm volatile("v_bcnt_u32_b32 %0, %0, %1\n" : "+v"(t[i]) : "s"(nb));
        else asm volatile("v_bcnt_u32_b32 %0, %0, %1\n" : "+v"(t[i]) : "v"(rv[0]));
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_alignbit_b32 %0, %0, %1, 31\n" : "+v"(acc[i & 3]) : "v"(t[i]));
    }
    if (MODE == 6) {                                        // G: popcounts + v_min3 tree + one sign shift a row
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_bcnt_u32_b32 %0, %0, 0\n" : "+v"(t[i]));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile("v_min3_u32 %0, %0, %1, %2\n v_min_u32 %0, %0, %3\n" : "+v"(t[u * 4]) : "v"(t[u * 4 + 1]), "v"(t[u * 4 + 2]), "v"(t[u * 4 + 3]));
        asm volatile("v_sub_u32 %0, %0, %2\n v_alignbit_b32 %1, %1, %0, 31\n" : "+v"(t[u * 4]), "+v"(acc[0]) : "v"(rv[1]));
      }
    }
  }
  unsigned r = acc[0] + acc[1] + acc[2] + acc[3];
  for (int i = 0; i < 16; ++i) r += t[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

typedef void (*kern_t)(unsigned *, unsigned, int);
static void run(const char *name, kern_t kern, unsigned *out) {
  const int iters = 2000, cus = 256;
  printf("%-44s", name);
  for (int w : {1, 2, 4, 8}) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<cus * w, 256>>>(out, 0x12345u, 20);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<cus * w, 256>>>(out, 0x12345u, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  w%d: %6.1f", w, ms * 1e-3 * 2.4e9 / ((double)w * iters));
  }
  printf("   SIMD cycles per group @2.4GHz\n");
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned *out;
  (void)hipMalloc(&out, 256 * 16 * 256 * sizeof(unsigned));
  run("A  80 logic (SGPR rows) + 16 bcnt + 16 alignbit", k<0>, out);
  run("B  80 logic (VGPR rows) + 16 bcnt + 16 alignbit", k<1>, out);
  run("C  80 logic (SGPR rows) only", k<2>, out);
  run("D  80 logic (VGPR rows) only", k<3>, out);
  run("E  VGPR rows: 80 logic, 16 bcnt(s), 16 alignbit", k<4>, out);
  run("F  VGPR rows: 80 logic, 16 bcnt(v), 16 alignbit", k<5>, out);
  run("G  VGPR rows: 80 logic, 16 bcnt, min3 tree, 4 shifts", k<6>, out);
  return 0;
}
