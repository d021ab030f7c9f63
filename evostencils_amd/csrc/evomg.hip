#include <atomic>
#include <sys/prctl.h>
#include <time.h>
// evomg.hip — the C-ABI of include/evomg.h: descriptor checks, the handle, dispatch to one of four implementation classes
// (impl_real.hip / impl_complex.hip: evomg_impl.h, the tuned scalar path; impl_system.hip; impl_fas.hip).
//
// Replaces, for one candidate, everything the reference does between lowering and parsed result
// (evostencils/code_generation/exastencils.py:485-537): no JVM, no g++, no subprocess — the op program is
// interpreted as a train of pre-compiled HIP kernel launches on one stream of one GPU.
#include "evomg_runtime.h"
#include <mutex>

namespace evomg_detail {

thread_local char g_create_error[512] = "";

// evaluations under way PER DEVICE (one process driving several GPUs with one candidate each is alone on every one
// of them). The entry points of the C-ABI note the handle's device for the calling thread (InFlight); the waits and the
// coarse-solve placement of that thread consult that device's count.
constexpr int kMaxDevices = 64;
std::atomic<int> g_evaluations_in_flight[kMaxDevices];
thread_local int tl_device = -1;
int evaluations_in_flight() {
  return tl_device >= 0 ? g_evaluations_in_flight[tl_device].load(std::memory_order_relaxed) : 0;
}
int next_cg_phase() {
  static std::atomic<int> counter{0};
  return counter.fetch_add(1, std::memory_order_relaxed) & 0x7fff;
}
namespace {
constexpr long g_wait_sleep_ns = 20000L;             // 5, 20 and 50 us measured alike (profiles/r05_throughput.md)
template <typename Q> hipError_t sleepy_wait(Q&& query) {
  // a sleeping thread wakes `timer slack` late (50 us by default, two to three times the sleep asked for): 1 us while this
  // wait sleeps, the caller's own value again afterwards (it is the caller's thread — Python's, under ctypes)
  long saved_slack = -1;
  hipError_t q;
  for (int polls = 0;; ++polls) {
    q = query();
    if (q != hipErrorNotReady) {
      // "not ready" is an answer, not an error: should the runtime have recorded it as this thread's last error, take it back
      // (the launch checks of the solver loops read that slot); any other recorded error stays where it is
      if (polls > 0 && hipPeekAtLastError() == hipErrorNotReady) (void)hipGetLastError();
      break;
    }
    if (polls < 4) continue;                                   // what is nearly there: no sleep
    if (saved_slack < 0) {
      saved_slack = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
      if (saved_slack < 0) saved_slack = 0;                    // (0 = "the default" when set)
      prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
    }
    const timespec ts{0, g_wait_sleep_ns};
    nanosleep(&ts, nullptr);
  }
  if (saved_slack >= 0) prctl(PR_SET_TIMERSLACK, (unsigned long)saved_slack, 0, 0, 0);
  return q;
}
}  // namespace
hipError_t wait_event(hipEvent_t e) {
  if (evaluations_in_flight() <= 1) return hipEventSynchronize(e);
  return sleepy_wait([&] { return hipEventQuery(e); });
}
hipError_t wait_stream(hipStream_t s) {
  if (evaluations_in_flight() <= 1) return hipStreamSynchronize(s);
  return sleepy_wait([&] { return hipStreamQuery(s); });
}
namespace {
struct SharedArray { void* array; int device; size_t bytes; uint64_t hash; int refs; };
std::mutex g_shared_mutex;
std::vector<SharedArray> g_shared_arrays;
}  // namespace
uint64_t hash_words(const void* data, size_t bytes) {
  // FNV-1a over 8-byte words (the tail byte by byte): ~20 ms for 134 MB, once per handle and upload
  uint64_t h = 1469598103934665603ull;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  size_t i = 0;
  for (; i + 8 <= bytes; i += 8) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = (h ^ w) * 1099511628211ull;
  }
  for (; i < bytes; ++i) h = (h ^ p[i]) * 1099511628211ull;
  return h;
}
void* shared_array_acquire(int device, size_t device_bytes, uint64_t hash, void* candidate) {
  std::lock_guard<std::mutex> lock(g_shared_mutex);
  for (auto& e : g_shared_arrays)
    if (e.device == device && e.bytes == device_bytes && e.hash == hash) {
      ++e.refs;
      return e.array;
    }
  g_shared_arrays.push_back(SharedArray{candidate, device, device_bytes, hash, 1});
  return candidate;
}
void shared_array_release(void* array) {
  std::lock_guard<std::mutex> lock(g_shared_mutex);
  for (size_t i = 0; i < g_shared_arrays.size(); ++i)
    if (g_shared_arrays[i].array == array) {
      if (--g_shared_arrays[i].refs == 0) {
        (void)hipFree(array);
        g_shared_arrays.erase(g_shared_arrays.begin() + (long)i);
      }
      return;
    }
}
struct InFlight {
  int device, previous;
  explicit InFlight(int d) : device(d >= 0 && d < kMaxDevices ? d : 0), previous(tl_device) {
    tl_device = device;
    g_evaluations_in_flight[device].fetch_add(1, std::memory_order_relaxed);
  }
  ~InFlight() {
    g_evaluations_in_flight[device].fetch_sub(1, std::memory_order_relaxed);
    tl_device = previous;
  }
};

int check_problem(const evomg_problem_t* p, char* err, size_t len) {
  if (!p) { snprintf(err, len, "null problem"); return EVOMG_ERR_INVALID; }
  if (p->abi_version != EVOMG_ABI_VERSION) { snprintf(err, len, "abi version %d != %d", p->abi_version, EVOMG_ABI_VERSION); return EVOMG_ERR_INVALID; }
  if (p->dimension != 2 && p->dimension != 3) { snprintf(err, len, "dimension must be 2 or 3"); return EVOMG_ERR_INVALID; }
  if (p->min_level < 1 || p->max_level >= EVOMG_MAX_LEVELS || p->min_level > p->max_level) { snprintf(err, len, "bad level range"); return EVOMG_ERR_INVALID; }
  if (p->scalar_kind != EVOMG_REAL_F64 && p->scalar_kind != EVOMG_COMPLEX_F64) { snprintf(err, len, "bad scalar kind"); return EVOMG_ERR_INVALID; }
  if (p->scalar_kind == EVOMG_COMPLEX_F64 && p->dimension != 2) { snprintf(err, len, "complex problems are 2D only"); return EVOMG_ERR_UNSUPPORTED; }
  if (p->bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN && p->scalar_kind != EVOMG_COMPLEX_F64) { snprintf(err, len, "Robin BC needs complex fields"); return EVOMG_ERR_INVALID; }
  if (p->outer_kind == EVOMG_OUTER_PBICGSTAB && p->dimension != 2) { snprintf(err, len, "PBiCGStab outer solver is 2D only"); return EVOMG_ERR_UNSUPPORTED; }
  if (p->max_iterations < 1 || p->cgs_max_iterations < 1) { snprintf(err, len, "iteration limits must be positive"); return EVOMG_ERR_INVALID; }
  if (p->jacobi_semantics != EVOMG_JACOBI_TEXTBOOK && p->jacobi_semantics != EVOMG_JACOBI_EXASTENCILS_V11) { snprintf(err, len, "jacobi_semantics must be EVOMG_JACOBI_TEXTBOOK or EVOMG_JACOBI_EXASTENCILS_V11"); return EVOMG_ERR_INVALID; }
  if (p->cgs_kind != EVOMG_CGS_CG && p->cgs_kind != EVOMG_CGS_BICGSTAB && p->cgs_kind != EVOMG_CGS_FAS_SMOOTHER && p->cgs_kind != EVOMG_CGS_CG_CLASSICAL) { snprintf(err, len, "bad coarse-grid solver kind"); return EVOMG_ERR_INVALID; }
  if (p->nonlinear_kind != EVOMG_NONLINEAR_NONE) {
    if (p->nonlinear_kind == EVOMG_NONLINEAR_PROGRAM) {
      // both programs well-formed: known opcodes, operands there when an operator wants them, at most EVOMG_NL_MAX_STACK
      // values at a time, exactly one left at the end
      const evomg_nl_program_t& q = p->nonlinear_program;
      for (int which = 0; which < 2; ++which) {
        const int n = which ? q.n_jacobian : q.n_sigma;
        const uint8_t* code = which ? q.jacobian : q.sigma;
        int depth = 0;
        bool ok = n >= 1 && n <= EVOMG_NL_MAX_OPS;
        for (int k = 0; ok && k < n; ++k) {
          const int op = code[k];
          if (op == EVOMG_NL_PUSH_U || (op >= EVOMG_NL_PUSH_CONST && op < EVOMG_NL_PUSH_CONST + EVOMG_NL_MAX_CONSTS)) ++depth;
          else if (op == EVOMG_NL_DUP) { ok = depth >= 1; ++depth; }
          else if (op >= EVOMG_NL_ADD && op <= EVOMG_NL_DIV) { ok = depth >= 2; --depth; }
          else if (op >= EVOMG_NL_NEG && op <= EVOMG_NL_COSH) ok = depth >= 1;
          else ok = false;
          ok = ok && depth <= EVOMG_NL_MAX_STACK;
        }
        if (!ok || depth != 1) { snprintf(err, len, "non-linear FAS: malformed %s program", which ? "Newton-term" : "stencil-entry"); return EVOMG_ERR_INVALID; }
      }
    }
    if ((p->nonlinear_kind != EVOMG_NONLINEAR_GAM_U_EXP_U && p->nonlinear_kind != EVOMG_NONLINEAR_PROGRAM) || p->dimension != 2 || p->scalar_kind != EVOMG_REAL_F64 ||
        p->bc_kind != EVOMG_BC_DIRICHLET || p->outer_kind != EVOMG_OUTER_MG_ITERATION || p->cgs_kind != EVOMG_CGS_FAS_SMOOTHER ||
        p->n_unknowns > 1) {
      snprintf(err, len, "non-linear FAS: 2D real Dirichlet scalar problems, N(u) = sigma(u) u (gam exp(u), or a program), smoother as coarse solver");
      return EVOMG_ERR_UNSUPPORTED;
    }
  } else if (p->cgs_kind == EVOMG_CGS_FAS_SMOOTHER) { snprintf(err, len, "EVOMG_CGS_FAS_SMOOTHER needs a non-linear problem"); return EVOMG_ERR_INVALID; }
  if (p->stencil_kind != EVOMG_STENCIL_STAR && p->stencil_kind != EVOMG_STENCIL_COMPACT) { snprintf(err, len, "bad stencil kind"); return EVOMG_ERR_INVALID; }
  if (p->n_unknowns > 1 || p->stencil_kind == EVOMG_STENCIL_COMPACT) {
    if (p->n_unknowns > EVOMG_MAX_UNKNOWNS) { snprintf(err, len, "at most %d unknowns per node", EVOMG_MAX_UNKNOWNS); return EVOMG_ERR_UNSUPPORTED; }
    if (p->dimension != 2 || p->scalar_kind != EVOMG_REAL_F64 || p->bc_kind != EVOMG_BC_DIRICHLET ||
        p->outer_kind != EVOMG_OUTER_MG_ITERATION || (p->cgs_kind != EVOMG_CGS_CG && p->cgs_kind != EVOMG_CGS_CG_CLASSICAL)) {
      snprintf(err, len, "systems of PDEs: 2D real Dirichlet problems with the generated solver loop and CG only");
      return EVOMG_ERR_UNSUPPORTED;
    }
  }
  return EVOMG_OK;
}

}  // namespace evomg_detail

using namespace evomg_detail;

struct evomg_handle {
  ImplBase* impl;
  int device;
};

extern "C" {

#pragma GCC visibility push(default)      // the library is built with -fvisibility=hidden: the C-ABI is ALL it exports

int evomg_abi_version(void) { return EVOMG_ABI_VERSION; }
int evomg_device_count(void) {
  int count = 0;
  return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
}

int evomg_create(const evomg_problem_t* problem, int device_id, evomg_handle_t** out) {
  if (!out) return EVOMG_ERR_INVALID;
  *out = nullptr;
  int rc = check_problem(problem, g_create_error, sizeof g_create_error);
  if (rc != EVOMG_OK) return rc;
  ImplBase* impl = nullptr;
  if (problem->nonlinear_kind != EVOMG_NONLINEAR_NONE) impl = make_impl_fas(*problem, device_id);
  else if (problem->n_unknowns > 1 || problem->stencil_kind == EVOMG_STENCIL_COMPACT) impl = make_impl_system(*problem, device_id);
  else if (problem->scalar_kind == EVOMG_COMPLEX_F64) impl = make_impl_complex(*problem, device_id);
  else impl = make_impl_real(*problem, device_id);
  rc = impl->init();
  if (rc != EVOMG_OK) {
    snprintf(g_create_error, sizeof g_create_error, "%s", impl->err);
    delete impl;
    return rc;
  }
  *out = new evomg_handle{impl, device_id};
  return EVOMG_OK;
}

void evomg_destroy(evomg_handle_t* h) {
  if (!h) return;
  delete h->impl;
  delete h;
}

const char* evomg_last_error(const evomg_handle_t* h) { return h ? h->impl->err : g_create_error; }

int evomg_set_initial_field(evomg_handle_t* h, int level, int field, const double* host) {
  return h ? h->impl->set_initial_field(level, field, host) : EVOMG_ERR_INVALID;
}
int evomg_get_field(evomg_handle_t* h, int level, int field, double* host) {
  return h ? h->impl->get_field(level, field, host) : EVOMG_ERR_INVALID;
}
int evomg_set_field(evomg_handle_t* h, int level, int field, const double* host) {
  return h ? h->impl->set_field(level, field, host) : EVOMG_ERR_INVALID;
}
int evomg_set_operators(evomg_handle_t* h, const evomg_problem_t* problem) {
  return h ? h->impl->set_operators(problem) : EVOMG_ERR_INVALID;
}
int evomg_evaluate(evomg_handle_t* h, const evomg_op_t* ops, int n_ops, int evaluation_samples, double wall_limit_ms,
                   evomg_result_t* out, double* res_history, int history_capacity) {
  if (!h) return EVOMG_ERR_INVALID;
  const InFlight in_flight(h->device);
  return h->impl->evaluate(ops, n_ops, evaluation_samples, wall_limit_ms, out, res_history, history_capacity);
}
int evomg_time_iterations(evomg_handle_t* h, const evomg_op_t* ops, int n_ops, int n_iterations, double* ms_per_iteration, int* n_done) {
  if (!h || !ms_per_iteration || n_iterations < 1 || n_ops < 0 || (!ops && n_ops > 0)) return EVOMG_ERR_INVALID;
  evomg_problem_t& p = h->impl->problem();
  const int saved_max = p.max_iterations;
  const double saved_target = p.target_res_reduction;
  p.max_iterations = n_iterations;
  p.target_res_reduction = -1.0;          // a norm is never below a negative bound: the loop runs all n_iterations
  evomg_result_t r;
  const InFlight in_flight(h->device);    // the probes of the time model count as evaluations under way (they are meant to run alone: then they spin)
  const int rc = h->impl->evaluate(ops, n_ops, 1, 0.0, &r, nullptr, 0);
  p.max_iterations = saved_max;
  p.target_res_reduction = saved_target;
  if (rc != EVOMG_OK) return rc;
  const double its = r.n_iterations >= 1.0 ? r.n_iterations : 1.0;
  *ms_per_iteration = r.time_ms / its;
  if (n_done) *n_done = (int)r.n_iterations;
  return EVOMG_OK;
}
int evomg_coarse_iterations(evomg_handle_t* h, long long* total) {
  if (!h || !total) return EVOMG_ERR_INVALID;
  *total = h->impl->coarse_iterations();
  return EVOMG_OK;
}
// the probes of the time model: `count` runs of evomg_time_iterations, the first two with the coarse solver capped at one
// iteration — where the implementation counts its coarse-solver iterations (K adds their cost back); where it does not
// (systems of PDEs, FAS: coarse_iterations() is 0) every probe runs the coarse solver as it is and the model degenerates to
// fixed + per_iteration * its. The two cheap probes run twice, the faster run counts: a probe is a few milliseconds, and one
// disturbed run would tilt the whole model.
static int time_model_probes(evomg_handle_t* h, const evomg_op_t* ops, int n_ops, const int* n, int count, double* t, long long* k, int* done) {
  evomg_problem_t& p = h->impl->problem();
  const int saved_cgs = p.cgs_max_iterations;
  const bool cap = h->impl->counts_coarse_iterations();
  int rc = EVOMG_OK;
  for (int q = 0; q < count && rc == EVOMG_OK; ++q) {
    p.cgs_max_iterations = (q < 2 && cap) ? 1 : saved_cgs;
    const int repeats = (q < 2 && cap) ? 2 : 1;
    for (int rep = 0; rep < repeats && rc == EVOMG_OK; ++rep) {
      double ms = 0.0;
      int d = 0;
      rc = evomg_time_iterations(h, ops, n_ops, n[q], &ms, &d);
      if (rep == 0 || (d == done[q] && ms * d < t[q])) {
        done[q] = d;
        t[q] = ms * d;
        k[q] = h->impl->coarse_iterations();
      }
    }
  }
  p.cgs_max_iterations = saved_cgs;
  return rc;
}
int evomg_time_model(evomg_handle_t* h, const evomg_op_t* ops, int n_ops, int n_short, int n_long, double* ms_fixed,
                     double* ms_per_iteration, double* ms_per_coarse_iteration) {
  if (!h || !ms_fixed || !ms_per_iteration || !ms_per_coarse_iteration || n_short < 1 || n_long <= n_short) return EVOMG_ERR_INVALID;
  double t[3] = {0, 0, 0};
  long long k[3] = {0, 0, 0};
  int done[3] = {0, 0, 0};
  const int n[3] = {n_short, n_long, n_long};
  const int rc = time_model_probes(h, ops, n_ops, n, 3, t, k, done);
  if (rc != EVOMG_OK) return rc;
  if (done[0] != n_short || done[1] != n_long || done[2] != n_long) {
    // a probe blew up before its last iteration: the plain average of the longest one is all there is
    const int q = done[2] > 0 ? 2 : (done[1] > 0 ? 1 : 0);
    *ms_fixed = 0.0;
    *ms_per_iteration = done[q] > 0 ? t[q] / done[q] : 0.0;
    *ms_per_coarse_iteration = 0.0;
    return EVOMG_OK;
  }
  const double c_cg = k[2] > k[1] ? (t[2] - t[1]) / (double)(k[2] - k[1]) : 0.0;
  const double c_it = ((t[1] - t[0]) - c_cg * (double)(k[1] - k[0])) / (double)(n_long - n_short);
  *ms_per_coarse_iteration = c_cg > 0.0 ? c_cg : 0.0;
  *ms_per_iteration = c_it;
  *ms_fixed = t[0] - c_it * n_short - *ms_per_coarse_iteration * (double)k[0];
  return EVOMG_OK;
}
int evomg_time_model_known_coarse(evomg_handle_t* h, const evomg_op_t* ops, int n_ops, int n_short, int n_long,
                                  double ms_per_coarse_iteration, double* ms_fixed, double* ms_per_iteration) {
  if (!h || !ms_fixed || !ms_per_iteration || n_short < 1 || n_long <= n_short || !(ms_per_coarse_iteration >= 0.0)) return EVOMG_ERR_INVALID;
  double t[2] = {0, 0};
  long long k[2] = {0, 0};
  int done[2] = {0, 0};
  const int n[2] = {n_short, n_long};
  const int rc = time_model_probes(h, ops, n_ops, n, 2, t, k, done);
  if (rc != EVOMG_OK) return rc;
  if (done[0] != n_short || done[1] != n_long) {
    const int q = done[1] > 0 ? 1 : 0;
    *ms_fixed = 0.0;
    *ms_per_iteration = done[q] > 0 ? (t[q] - ms_per_coarse_iteration * (double)k[q]) / done[q] : 0.0;
    return EVOMG_OK;
  }
  const double c_it = ((t[1] - t[0]) - ms_per_coarse_iteration * (double)(k[1] - k[0])) / (double)(n_long - n_short);
  *ms_per_iteration = c_it;
  *ms_fixed = t[0] - c_it * n_short - ms_per_coarse_iteration * (double)k[0];
  return EVOMG_OK;
}
int evomg_run_ops(evomg_handle_t* h, const evomg_op_t* ops, int n_ops) {
  if (!h || n_ops < 0 || (!ops && n_ops > 0)) return EVOMG_ERR_INVALID;
  const InFlight in_flight(h->device);
  return h->impl->run_ops(ops, n_ops);
}
int evomg_norm(evomg_handle_t* h, int level, int field, double* out) {
  return h ? h->impl->norm(level, field, out) : EVOMG_ERR_INVALID;
}
int evomg_bench_op(evomg_handle_t* h, const evomg_op_t* op, int warmup, int reps, evomg_bench_t* out) {
  return h ? h->impl->bench_op(op, warmup, reps, out) : EVOMG_ERR_INVALID;
}
int evomg_profile_begin(evomg_handle_t* h, int opcode, int level) {
  return h ? h->impl->profile_begin(opcode, level) : EVOMG_ERR_INVALID;
}
int evomg_profile_end(evomg_handle_t* h, evomg_bench_t* out) {
  return h ? h->impl->profile_end(out) : EVOMG_ERR_INVALID;
}
int evomg_kernel_name(evomg_handle_t* h, const evomg_op_t* op, char* buf, int buf_len) {
  return h ? h->impl->kernel_name(op, buf, buf_len) : EVOMG_ERR_INVALID;
}

#pragma GCC visibility pop

}  // extern "C"
