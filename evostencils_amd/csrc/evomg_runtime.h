// evomg_runtime.h — what every implementation class of the backend shares: the interface behind the C-ABI handle,
// error plumbing, constants. (Split out of evomg.hip in round 4; see impl_*.hip for the implementations.)
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/evomg.h"
#include <cstdint>
#include <set>

namespace evomg_detail {

extern thread_local char g_create_error[512];
// evaluations of this process under way right now ON THE DEVICE OF THE CALLING THREAD'S CURRENT C-ABI CALL (defined in
// evomg.hip): more than one means candidates in flight on that device; 0 outside an evaluation
int evaluations_in_flight();
int next_cg_phase();       // a process-wide round-robin counter (which XCD the next multi-workgroup solve in flight sits on)

// Host waits. HIP's own waits spin: a candidate alone on the device wants that (its solver loop ends every iteration with one,
// a wake-up would cost more than the wait), but W candidates in flight kept W cores busy, and a GPU box's 16-core share throttles
// the process beyond sixteen of them (tools/run_throughput.py: 20 in flight 3 s, 24 in flight 7 s of throttled time per 0.8 s of
// wall — profiles/r05_throughput.md). With other evaluations under way ON THE SAME DEVICE a wait therefore polls and sleeps in between
// (the device has the other candidates' work to do while this thread wakes up).
hipError_t wait_event(hipEvent_t e);
hipError_t wait_stream(hipStream_t s);

// Read-only device arrays shared between the handles of a process (round 6): the right-hand side of the finest level is the same
// array for every candidate of a problem, and with candidates in flight sixteen private copies of it (134 MB each at 4097^2)
// stream from HBM where ONE copy is read sixteen times and partly stays in the 256 MB Infinity Cache (sixteen C2 solves in
// flight: 421 -> 439 solves/s). Keyed by device, size and a 64-bit hash of the host data it was uploaded from.
// shared_array_acquire: `candidate` is the caller's freshly uploaded device copy; returns the array to use — `candidate`
// itself (now registered: the registry owns it) or an equal one registered earlier (the caller frees its candidate).
void* shared_array_acquire(int device, size_t device_bytes, uint64_t hash, void* candidate);
void shared_array_release(void* array);       // the last release frees it
uint64_t hash_words(const void* data, size_t bytes);

inline double wall_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

struct ImplBase {
  virtual ~ImplBase() {}
  virtual int init() = 0;
  virtual int set_initial_field(int level, int field, const double* host) = 0;
  virtual int get_field(int level, int field, double* host) = 0;
  virtual int set_field(int level, int field, const double* host) = 0;
  virtual int set_operators(const evomg_problem_t* p) = 0;
  virtual int evaluate(const evomg_op_t* ops, int n_ops, int samples, double wall_limit_ms, evomg_result_t* out,
                       double* hist, int cap) = 0;
  virtual int run_ops(const evomg_op_t* ops, int n_ops) = 0;
  virtual int norm(int level, int field, double* out) = 0;
  virtual int bench_op(const evomg_op_t* op, int warmup, int reps, evomg_bench_t* out) = 0;
  virtual int kernel_name(const evomg_op_t* op, char* buf, int len) = 0;
  virtual int profile_begin(int opcode, int level) = 0;
  virtual int profile_end(evomg_bench_t* out) = 0;
  virtual long long coarse_iterations() { return 0; }   // Krylov iterations of all coarse solves of the last evaluate (last sample)
  // false: coarse_iterations() is always 0 (systems of PDEs, FAS): the time model must not cap the coarse solver then —
  // nothing would add its cost back (evomg_time_model)
  virtual bool counts_coarse_iterations() const { return false; }
  virtual evomg_problem_t& problem() = 0;      // the handle's descriptor (evomg_time_iterations re-targets the loop's stopping rule)
  char err[512] = "";
};

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      snprintf(err, sizeof err, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return EVOMG_ERR_DEVICE;                                                                         \
    }                                                                                                  \
  } while (0)

constexpr int kDotBlocks = 1024;
constexpr int kTmpVectors = 8;      // BiCGStab: six work vectors + the second buffers of p and nu (kb2_ps)

// factories of the four implementations (impl_real.hip, impl_complex.hip, impl_system.hip, impl_fas.hip)
ImplBase* make_impl_real(const evomg_problem_t& p, int device);
ImplBase* make_impl_complex(const evomg_problem_t& p, int device);
ImplBase* make_impl_system(const evomg_problem_t& p, int device);
ImplBase* make_impl_fas(const evomg_problem_t& p, int device);

}  // namespace evomg_detail
