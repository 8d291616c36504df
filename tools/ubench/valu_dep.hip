// Micro-benchmark: VALU issue rate of dependent vs independent instruction chains on gfx950,
// as a function of resident waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_dep.hip -o valu_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

__global__ __launch_bounds__(256) void k_dep(unsigned *out, unsigned r, int iters) {
  unsigned a = threadIdx.x, acc = 0;
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP16("v_xor_b32 %0, %0, %2\n v_bcnt_u32_b32 %1, %0, %1\n v_xor_b32 %0, %1, %2\n v_bcnt_u32_b32 %1, %0, %1\n")
                 : "+v"(a), "+v"(acc) : "v"(r));
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc + a;
}

__global__ __launch_bounds__(256) void k_ind2(unsigned *out, unsigned r, int iters) {
  unsigned a = threadIdx.x, b = a * 3, acc = 0, acc2 = 1;
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP16("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_bcnt_u32_b32 %2, %0, %2\n v_bcnt_u32_b32 %3, %1, %3\n")
                 : "+v"(a), "+v"(b), "+v"(acc), "+v"(acc2) : "v"(r));
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc + a + b + acc2;
}

__global__ __launch_bounds__(256) void k_ind4(unsigned *out, unsigned r, int iters) {
  unsigned a = threadIdx.x, b = a * 3, c = a * 5, d = a * 7;
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP16("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(r));
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}

// dependent chain through v_cmp -> SGPR -> s_or (the stage-1 tail)
__global__ __launch_bounds__(256) void k_cmp(unsigned *out, unsigned r, int iters) {
  unsigned a = threadIdx.x, b = a * 3;
  unsigned long long m = 0;
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP16("v_min_u32 %0, %0, %1\n v_cmp_lt_u32 vcc, %0, %3\n s_or_b64 %2, %2, vcc\n v_xor_b32 %1, %1, %3\n")
                 : "+v"(a), "+v"(b), "+s"(m) : "v"(r) : "vcc");
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + (unsigned)m;
}

template <typename K>
static void run(const char *name, K kern, int wavesPerSimd, unsigned *out, int instrPerIter) {
  const int iters = 20000, cus = 256;
  const int grid = cus * wavesPerSimd;       // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, 256>>>(out, 0x12345u, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<grid, 256>>>(out, 0x12345u, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)grid * 4 * iters * instrPerIter;      // wave-instructions
  const double rate = instr / (ms * 1e-3);
  printf("%-8s waves/SIMD=%d  %.3f ms  %.3e wave-instr/s  = %.2f cycles/instr/SIMD @2.4GHz\n", name, wavesPerSimd, ms, rate,
         2.4e9 * 1024 / rate);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned *out;
  hipMalloc(&out, 256 * 16 * 256 * sizeof(unsigned));
  for (int w : {1, 2, 3, 4, 6, 8}) {
    run("dep", k_dep, w, out, 64);
    run("ind2", k_ind2, w, out, 64);
    run("ind4", k_ind4, w, out, 64);
    run("cmp", k_cmp, w, out, 64);
  }
  return 0;
}
