// Launch cadence of a chain of tiny dependent kernels on one stream: plain launches vs one captured hipGraph replayed.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/bench_graph tools/bench_graph.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(double* a, int n, double s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = a[i] * s + 1.0;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int n = 65 * 65, chain = 88, reps = 200;
  double* a;
  CK(hipMalloc(&a, n * sizeof(double)));
  CK(hipMemset(a, 0, n * sizeof(double)));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int blocks : {8, 17}) {
    // plain
    for (int w = 0; w < 2; ++w) {
      const double t0 = now_us();
      for (int r = 0; r < reps; ++r)
        for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, a, n, 0.5);
      const double t1 = now_us();
      CK(hipStreamSynchronize(s));
      const double t2 = now_us();
      if (w) printf("blocks %d plain : host %.2f us/launch, total %.2f us/launch\n", blocks, (t1 - t0) / (reps * chain), (t2 - t0) / (reps * chain));
    }
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, a, n, 0.5);
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int w = 0; w < 2; ++w) {
      const double t0 = now_us();
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(exec, s));
      const double t1 = now_us();
      CK(hipStreamSynchronize(s));
      const double t2 = now_us();
      if (w) printf("blocks %d graph : host %.2f us/kernel, total %.2f us/kernel\n", blocks, (t1 - t0) / (reps * chain), (t2 - t0) / (reps * chain));
    }
    // graph replay with a host synchronisation after every replay (the outer solver loop reads a norm per iteration)
    {
      const double t0 = now_us();
      for (int r = 0; r < reps; ++r) { CK(hipGraphLaunch(exec, s)); CK(hipStreamSynchronize(s)); }
      const double t2 = now_us();
      printf("blocks %d graph + sync per replay: total %.2f us/kernel\n", blocks, (t2 - t0) / (reps * chain));
      const double t3 = now_us();
      for (int r = 0; r < reps; ++r) {
        for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, a, n, 0.5);
        CK(hipStreamSynchronize(s));
      }
      const double t4 = now_us();
      printf("blocks %d plain + sync per chain : total %.2f us/kernel\n", blocks, (t4 - t3) / (reps * chain));
    }
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
  }
  return 0;
}
