// Developer tool: cost of one grid-wide rendezvous (G co-resident workgroups exchange one double each and all of them form
// the sum in the same order) — the building block of a multi-workgroup Krylov solve (csrc/evomg_krylov.h: grid_publish /
// grid_collect). Variants:
//   0  release store / acquire polls at agent scope (what evomg_cg_multi_wg3d does)
//   1  relaxed agent-scope atomics only; the data store is waited for (s_waitcnt vmcnt(0)) before the flag store
//   2  as 1, value and sequence number packed into ONE 16-byte slot written by one store? (not atomic on this ISA) -> two
//      8-byte relaxed stores to the same 16-byte-aligned slot, flag second
//   3  "LL" words: the value travels as two 8-byte words {32 data bits, 32-bit sequence number}, each written by ONE relaxed
//      agent-scope 64-bit atomic store (single-copy atomic); the consumer polls the two words themselves until both carry
//      this round's number — no flag after the data, no data load after the flag: one store trip + one poll trip
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bench_rendezvous tools/bench_rendezvous.hip ; run: tools/bench_rendezvous
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// loads that bypass the compute unit's L1 and are served by the XCD's L2 (scope bit sc0 = "workgroup"): coherent among the
// compute units of ONE XCD, which is all variant 2 needs when every participating workgroup sits on the same XCD
__device__ __forceinline__ unsigned load_l2_u32(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ double load_l2_f64(const double* p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store_l2_u32(unsigned* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_l2_f64(double* p, double v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VARIANT>
__global__ void __launch_bounds__(512) rendezvous(int G, int rounds, int stride, unsigned* flags, double* part, double* out, int* fail) {
  __shared__ double lds_part[64];
  __shared__ int lds_flag;
  const int w = blockIdx.x / stride;                 // only every stride-th block takes part (stride 8: one XCD)
  if (blockIdx.x % stride != 0 || w >= G) return;
  double acc = 1.0 + w;
  for (int k = 1; k <= rounds; ++k) {
    // publish
    if (threadIdx.x == 0) {
      if (VARIANT == 2) {
        store_l2_f64(part + w, acc);
        store_l2_u32(flags + w, (unsigned)k);
      } else
      __hip_atomic_store(part + w, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (VARIANT == 2) {
      } else if (VARIANT == 0) {
        __hip_atomic_store(flags + w, (unsigned)k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __builtin_amdgcn_s_waitcnt(0);               // the data store has been acknowledged
        __hip_atomic_store(flags + w, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      lds_flag = 0;
    }
    __syncthreads();
    // collect
    if ((int)threadIdx.x < G) {
      int budget = 1 << 22;
      if (VARIANT == 2) {
        while (load_l2_u32(flags + threadIdx.x) < (unsigned)k) {
          if (--budget == 0) { lds_flag = 1; break; }
        }
        lds_part[threadIdx.x] = load_l2_f64(part + threadIdx.x);
      } else if (VARIANT == 0) {
        while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)k) {
          if (--budget == 0) { lds_flag = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      } else {
        while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)k) {
          if (--budget == 0) { lds_flag = 1; break; }
        }
      }
      if (VARIANT != 2) lds_part[threadIdx.x] = __hip_atomic_load(part + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (lds_flag) { if (threadIdx.x == 0) *fail = 1; return; }
    double s = lds_part[0];
    for (int q = 1; q < G; ++q) s += lds_part[q];
    // peers may only overwrite part[] after everybody has read it: in a Krylov iteration the two exchanges alternate
    // buffers; here: double-buffer by round parity
    acc = s * 0.125 + w;
    part += (k & 1) ? 64 : -64;
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[w] = acc; out[32 + w] = (double)xcc_id(); }
}

__global__ void __launch_bounds__(512) rendezvous_ll(int G, int rounds, int stride, unsigned long long* ll, double* out, int* fail) {
  __shared__ double lds_part[64];
  __shared__ int lds_flag;
  const int w = blockIdx.x / stride;
  if (blockIdx.x % stride != 0 || w >= G) return;
  double acc = 1.0 + w;
  for (int k = 1; k <= rounds; ++k) {
    unsigned long long* slot = ll + (k & 1) * 128;       // [parity][group][2]
    if (threadIdx.x < 2) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
      const unsigned half = threadIdx.x == 0 ? (unsigned)bits : (unsigned)(bits >> 32);
      __hip_atomic_store(slot + 2 * w + threadIdx.x, ((unsigned long long)(unsigned)k << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) lds_flag = 0;
    __syncthreads();
    if ((int)threadIdx.x < G) {
      int budget = 1 << 22;
      unsigned long long a, b;
      for (;;) {
        a = __hip_atomic_load(slot + 2 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b = __hip_atomic_load(slot + 2 * threadIdx.x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(a >> 32) == (unsigned)k && (unsigned)(b >> 32) == (unsigned)k) break;
        if (--budget == 0) { lds_flag = 1; break; }
      }
      lds_part[threadIdx.x] = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
    }
    __syncthreads();
    if (lds_flag) { if (threadIdx.x == 0) *fail = 1; return; }
    double s = lds_part[0];
    for (int q = 1; q < G; ++q) s += lds_part[q];
    acc = s * 0.125 + w;
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[w] = acc; out[32 + w] = (double)xcc_id(); }
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned* flags; double *part, *out; int* fail;
  CHECK(hipMalloc(&flags, 64 * sizeof(unsigned)));
  CHECK(hipMalloc(&part, 128 * sizeof(double)));
  CHECK(hipMalloc(&out, 64 * sizeof(double)));
  CHECK(hipMalloc(&fail, sizeof(int)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  unsigned long long* ll;
  CHECK(hipMalloc(&ll, 256 * sizeof(unsigned long long)));
  for (int variant = 0; variant < 4; ++variant)
    for (int stride : {1, 8})
      for (int G : {2, 4, 8}) {
        double res[64] = {};
        float best = 1e30f;
        int failed = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CHECK(hipMemset(flags, 0, 64 * sizeof(unsigned)));
          CHECK(hipMemset(part, 0, 128 * sizeof(double)));
          CHECK(hipMemset(fail, 0, sizeof(int)));
          CHECK(hipMemset(ll, 0, 256 * sizeof(unsigned long long)));
          CHECK(hipEventRecord(e0));
          if (variant == 3) hipLaunchKernelGGL(rendezvous_ll, dim3(G * stride), dim3(512), 0, 0, G, rounds, stride, ll, out, fail);
          else if (variant == 0) hipLaunchKernelGGL(rendezvous<0>, dim3(G * stride), dim3(512), 0, 0, G, rounds, stride, flags, part, out, fail);
          else if (variant == 2) hipLaunchKernelGGL(rendezvous<2>, dim3(G * stride), dim3(512), 0, 0, G, rounds, stride, flags, part, out, fail);
          else hipLaunchKernelGGL(rendezvous<1>, dim3(G * stride), dim3(512), 0, 0, G, rounds, stride, flags, part, out, fail);
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
          CHECK(hipMemcpy(&failed, fail, sizeof(int), hipMemcpyDeviceToHost));
          CHECK(hipMemcpy(res, out, 64 * sizeof(double), hipMemcpyDeviceToHost));
        }
        printf("variant %d  %s  G=%d: %.3f us per rendezvous  (failed=%d) xcc ids:", variant, stride == 1 ? "one WG per XCD" : "all on one XCD  ",
               G, 1e3 * best / rounds, failed);
        for (int q = 0; q < G; ++q) printf(" %d", (int)res[32 + q]);
        printf("\n");
      }
  return 0;
}
