// Developer check: cg1_stop_threshold (csrc/evomg_cg1.h) — the stopping rule sqrt(g) <= T decided as g <= g* — against the square
// root itself ON THE DEVICE: for 2^22 pseudo-random T over a wide exponent range (and the edge cases: 0, subnormal T, T whose
// square under- or overflows, infinity, NaN, negative) every g within 8 units in the last place of g* (and a few far away) must
// get the same decision both ways; where the search reports "not exact" the rule keeps the square root and is not checked.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ievostencils_amd/csrc -o tools/check_stop_threshold tools/check_stop_threshold.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include "evomg_cg1.h"

__global__ void check(const double* t, int n, unsigned long long* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double T = t[i];
  const Cg1Stop s = cg1_stop_threshold(T);
  if (!s.exact) { atomicAdd(&out[1], 1ull); return; }
  unsigned long long bad = 0;
  const long long bits = __double_as_longlong(s.gstar);
  for (long long d = -8; d <= 8; ++d) {
    const long long b = bits + d;
    if (b < 0 || b >= 0x7ff0000000000000ll) continue;
    const double g = __longlong_as_double(b);
    if (cg1_stops(s, g) != !(sqrt(g) > T)) ++bad;
  }
  const double far[6] = {0.0, s.gstar * 0.5, s.gstar * 2.0, 1e-300, 1e300, __builtin_nan("")};
  for (int k = 0; k < 6; ++k)
    if (cg1_stops(s, far[k]) != !(sqrt(far[k]) > T)) ++bad;
  if (bad) atomicAdd(&out[0], bad);
}

int main() {
  const int n = 1 << 22;
  double* h = new double[n];
  uint64_t x = 0x9e3779b97f4a7c15ull;
  for (int i = 0; i < n; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const double m = 1.0 + (double)(x >> 12) / 4503599627370496.0;        // [1, 2)
    const int e = (int)((x >> 3) % 1400) - 700;                            // T from 2^-700 to 2^699 (T^2 under- and overflows at the ends)
    h[i] = ldexp(m, e);
  }
  const double edge[] = {0.0, -0.0, 4.9e-324, 1e-310, 1e-200, 1.4e-162, 1.5e-154, 1.3e154, 1.4e154, 1e200, 1.7e308, INFINITY, -1.0, NAN, 1e-12, 1.0};
  for (size_t k = 0; k < sizeof edge / sizeof edge[0]; ++k) h[k] = edge[k];
  double* d;
  unsigned long long *dout, out[2] = {0, 0};
  hipMalloc(&d, n * sizeof(double));
  hipMalloc(&dout, sizeof out);
  hipMemcpy(d, h, n * sizeof(double), hipMemcpyHostToDevice);
  hipMemcpy(dout, out, sizeof out, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, d, n, dout);
  hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
  printf("cg1_stop_threshold: %d thresholds, %llu decisions differ from the square root's, %llu thresholds kept the square root (search not exact)\n", n, out[0], out[1]);
  return out[0] ? 1 : 0;
}
