#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  const unsigned lane = threadIdx.x;
  unsigned v = 1000 + lane;           // lane-indexed value
  unsigned idx = (lane * 7) & 31;     // gather index
  unsigned g = (unsigned)__builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)v);
  out[lane] = g;
}
int main() {
  unsigned *d, h[64];
  (void)hipMalloc(&d, 256);
  k<<<1, 64>>>(d);
  (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) if (h[i] != 1000u + ((i * 7) & 31)) ++bad;
  printf("bpermute mismatches: %d (lane1 got %u want %u)\n", bad, h[1], 1000u + 7u);
  return 0;
}
