// Developer check: the three-operation quotient by a kernel-constant divisor (evomg_scalar.h: const_div) against the
// compiler's f64 division, bit for bit, on 2^32 pseudo-random numerators per divisor (wide exponent range) and on
// numerators near the rescaling threshold of v_div_scale.  build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "../evostencils_amd/csrc/evomg_scalar.h"

__device__ inline uint64_t mix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31);
}
__global__ void check(double b, int exp_lo, int exp_span, uint64_t seed, unsigned long long* bad, double* example) {
  const ConstDiv<double> cd = const_div_make(b);
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  for (int i = 0; i < 4096; ++i) {
    const uint64_t h = mix(seed + t * 4096 + i);
    const uint64_t mant = h & 0xfffffffffffffull;
    const uint64_t e = (uint64_t)(1023 + exp_lo + (int)((h >> 52) % (uint64_t)exp_span));
    const uint64_t bits = (h & 0x8000000000000000ull) | (e << 52) | mant;
    double a; memcpy(&a, &bits, 8);
    const double q0 = a / b, q1 = const_div(a, cd);
    uint64_t u0, u1; memcpy(&u0, &q0, 8); memcpy(&u1, &q1, 8);
    if (u0 != u1) { if (atomicAdd(bad, 1ull) == 0) { example[0] = a; example[1] = q0; example[2] = q1; } }
  }
}
int main() {
  unsigned long long* bad; double* ex;
  hipMalloc(&bad, 8); hipMalloc(&ex, 24);
  const double s = 262144.0;   // 1 / h^2 at level 9
  const double divisors[] = {6.0 * s, 6.0 * s - (s / (6.0 * s)) * s, 4.0 * s, 3.0, 7.0, 1.0 / 3.0, 0.1, 1.7976931348623157e10, 2.2250738585072014e-10,
                             1.9999999999999998, 1.0000000000000002, 6.0 * 65536.0, 6.0 * 16384.0, 35.0 / 6.0 * 65536.0, 4.0 * 16777216.0 - 1.0};
  int rc = 0;
  for (double b : divisors) {
    for (int range = 0; range < 2; ++range) {
      const int lo = range == 0 ? -60 : -900, span = range == 0 ? 120 : 1800;
      unsigned long long h = 0; double e[3] = {0, 0, 0};
      hipMemset(bad, 0, 8);
      hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, b, lo, span, (uint64_t)(b * 1e3) + range, bad, ex);
      hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(e, ex, 24, hipMemcpyDeviceToHost);
      printf("b=%.17g exponents [%d,%d): %llu mismatches of %llu", b, lo, lo + span, h, 4096ull * 256 * 4096);
      if (h) { printf("  e.g. a=%.17g  a/b=%.17g  const_div=%.17g", e[0], e[1], e[2]); rc = 1; }
      printf("\n");
    }
  }
  return rc;
}
