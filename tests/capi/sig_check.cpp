// Host-side check of the MFMA engine's signature helpers (prograph_amd/csrc/pg_common.h: pg_sig54, pg_nib8,
// pg_bias_nibbles) - plain arithmetic, no GPU needed.  Test infrastructure.
#include <hip/hip_runtime.h>
#include "../../prograph_amd/csrc/pg_common.h"
#include <cstdio>
#include <cstdlib>

static const float kFp4[16] = {0, 0.5f, 1, 1.5f, 2, 3, 4, 6, -0.0f, -0.5f, -1, -1.5f, -2, -3, -4, -6};

int main() {
  int bad = 0;
  // pg_nib8: bit i of the byte -> bit 0 of nibble i, nothing else
  for (u32 x = 0; x < 256; ++x) {
    u32 want = 0;
    for (int i = 0; i < 8; ++i) want |= ((x >> i) & 1u) << (4 * i);
    if (pg_nib8(x) != want || pg_nib8(x | 0xABCDEF00u) != want) { ++bad; printf("nib8(%u)\n", x); }
  }
  // pg_bias_nibbles: ten FP4 elements, the largest representable sum <= clamp(b, -60, 58)
  for (int b = -400; b <= 400; ++b) {
    const unsigned long long s = pg_bias_nibbles(b);
    if (s >> 40) { ++bad; printf("bias(%d): more than ten nibbles\n", b); }
    float sum = 0;
    for (int i = 0; i < 10; ++i) sum += kFp4[(s >> (4 * i)) & 15u];
    const int c = b < -60 ? -60 : (b > 58 ? 58 : b);
    const int want = c == -59 ? -60 : c;
    // never above the (clamped) bias: a smaller bias only passes more
    if ((int)sum != want || sum > (float)c) { ++bad; printf("bias(%d): sum %g want %d\n", b, sum, want); }
  }
  // pg_sig54: XOR-linear, 54 bits, popcount(sig(x) ^ sig(y)) <= popcount(x ^ y)
  srand(7);
  for (int it = 0; it < 20000; ++it) {
    const u32 a0 = (u32)rand() * 2654435761u, a1 = (u32)rand() * 40503u + (u32)rand(), b0 = (u32)rand() * 97u, b1 = (u32)rand() * 31337u;
    const unsigned long long sa = pg_sig54(a0, a1), sb = pg_sig54(b0, b1), sx = pg_sig54(a0 ^ b0, a1 ^ b1);
    if ((sa ^ sb) != sx || (sa >> 54) || __builtin_popcountll(sa ^ sb) > __builtin_popcount(a0 ^ b0) + __builtin_popcount(a1 ^ b1)) { ++bad; printf("sig54\n"); }
  }
  printf(bad ? "SIGNATURE HELPERS WRONG: %d\n" : "signature helpers OK\n", bad);
  return bad != 0;
}
