// Developer tool: what one FP64 instruction of a wave64 costs on this device — the unit the VALU-bound estimates of
// profiles/r05_cross_kernel.md are written in. Independent chains of v_fma_f64 / v_add_f64 / v_mul_f64 / v_mov_b32_dpp in one
// wave per SIMD (grid = 4 x compute units waves of 64 threads) and in two; wall time by HIP events, shader clock by wall_clock64
// (100 MHz) against clock64.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bench_valu tools/bench_valu.hip ; run: tools/bench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(64) chains(double* out, int iters, double a, double b, long long* ticks) {
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (double)threadIdx.x + k;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (OP == 0) v[k] = __builtin_fma(v[k], a, b);
      if (OP == 1) v[k] = v[k] + a;
      if (OP == 2) v[k] = v[k] * a;
      if (OP == 3) v[k] = __builtin_fma(v[(k + 1) & 7], a, v[k]);
      if (OP == 4) {
        int lo = __double2loint(v[k]), hi = __double2hiint(v[k]);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);   // wave_shr:1
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
        v[k] = __hiloint2double(hi, lo);
      }
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = w1 - w0; }
}

template <int OP> void run(const char* name, int waves, double* out, long long* ticks) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(chains<OP>, dim3(waves), dim3(64), 0, 0, out, 100, 1.0000001, 1e-9, ticks);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(chains<OP>, dim3(waves), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9, ticks);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; (void)hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
  const double n_inst = (double)iters * 8 * (OP == 4 ? 2 : 1);
  printf("%-34s %5d waves: %.3f ms, clock64 %lld, wall_clock64 %lld (100 MHz: %.3f ms) -> shader clock %.2f GHz, %.2f clock64 ticks per instruction of a wave\n",
         name, waves, ms, h[0], h[1], h[1] / 1e5, h[0] / (h[1] * 10.0), h[0] / n_inst);
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s: %d compute units, clockRate %d kHz\n", p.name, cus, p.clockRate);
  double* out; long long* ticks;
  (void)hipMalloc(&out, sizeof(double) * 64 * cus * 16); (void)hipMalloc(&ticks, 16);
  for (int per_simd = 1; per_simd <= 2; ++per_simd) {
    const int waves = cus * 4 * per_simd;
    run<0>("v_fma_f64, 8 independent chains", waves, out, ticks);
    run<1>("v_add_f64, 8 independent chains", waves, out, ticks);
    run<2>("v_mul_f64, 8 independent chains", waves, out, ticks);
    run<3>("v_fma_f64, reading a neighbour", waves, out, ticks);
    run<4>("v_mov_b32_dpp pairs", waves, out, ticks);
  }
  run<0>("v_fma_f64, ONE wave on the device", 1, out, ticks);
  return 0;
}
