// evomg.hip — runtime of the MI355X multigrid fitness-evaluation backend: device field storage, the op-program
// interpreter, the solver loops and the C-ABI of include/evomg.h.
//
// Replaces, for one candidate, everything the reference does between lowering and parsed result
// (evostencils/code_generation/exastencils.py:485-537): no JVM, no g++, no subprocess — the op program is
// interpreted as a train of pre-compiled HIP kernel launches on one stream of one GPU.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/evomg.h"
#include <cstdint>
#include <set>
#include "evomg_kernels.h"
#include "evomg_krylov.h"
#include "evomg_strip.h"
#include "evomg_small2d.h"
#include "evomg_small3d.h"
#include "evomg_tile3d.h"
#include "evomg_batch.h"
#include "evomg_strip_block.h"
#include "evomg_system.h"
#include "evomg_fas.h"
#ifndef EVOMG_FUSED_PF
#define EVOMG_FUSED_PF 2   // prefetch depth of the fused sweeps (fewer waves per SIMD than the plain ones)
#endif

namespace {

thread_local char g_create_error[512] = "";

double wall_ms() {
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

template <typename T> struct type_name;
template <> struct type_name<double> { static const char* get() { return "double"; } };
template <> struct type_name<cplx> { static const char* get() { return "cplx"; } };

constexpr int kDotBlocks = 1024;
constexpr int kTmpVectors = 6;

template <typename T> struct Level {
  Geo g{};
  size_t elems = 0;             // allocated elements (pitch * rows)
  T* f[EVOMG_F_COUNT] = {};
  T* init[EVOMG_F_COUNT] = {};  // device copies of the initial contents (nullptr = zero)
  T* alt = nullptr;             // ping-pong twin for out-of-place smoothers
  const T* alt_mirror = nullptr;   // array whose boundary values `alt` is known to share (nullptr: unknown)
  T* tmp[kTmpVectors] = {};     // Krylov work vectors, allocated on demand
  Stencil<T> st{};
};

template <typename T> class Impl : public ImplBase {
 public:
  Impl(const evomg_problem_t& p, int device) : prob_(p), device_(device), dim_(p.dimension) {}

  ~Impl() override {
    (void)hipSetDevice(device_);
    for (int L = prob_.min_level; L <= prob_.max_level; ++L) {
      Level<T>& lv = lv_[L];
      for (int f = 0; f < EVOMG_F_COUNT; ++f) {
        if (lv.f[f]) (void)hipFree(lv.f[f]);
        if (lv.init[f]) (void)hipFree(lv.init[f]);
      }
      if (lv.alt) (void)hipFree(lv.alt);
      for (auto*& t : lv.tmp)
        if (t) (void)hipFree(t);
    }
    for (auto* v : outer_) if (v) (void)hipFree(v);
    if (partials_) (void)hipFree(partials_);
    if (dscalar_) (void)hipFree(dscalar_);
    if (dinfo_) (void)hipFree(dinfo_);
    if (norm_event_) (void)hipEventDestroy(norm_event_);
    if (outer_event_) (void)hipEventDestroy(outer_event_);
    if (norm_partials_) (void)hipFree(norm_partials_);
    if (cg_xpart_) (void)hipFree(cg_xpart_);
    if (cg_xface_) (void)hipFree(cg_xface_);
    if (cg_flags_) (void)hipFree(cg_flags_);
    if (kcg_state_) (void)hipFree(kcg_state_);
    if (kb_state_) (void)hipFree(kb_state_);
    if (kb_scal_) (void)hipFree(kb_scal_);
    if (kb_host_) (void)hipHostFree(kb_host_);
    if (kcg_host_) (void)hipHostFree(kcg_host_);
    if (hscalar_) (void)hipHostFree(hscalar_);
    for (auto e : prof_events_) (void)hipEventDestroy(e);
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    if (stream_) (void)hipStreamDestroy(stream_);
  }

  int init() override {
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device_ < 0 || device_ >= count) {
      snprintf(err, sizeof err, "device %d not available (%d devices)", device_, count);
      return EVOMG_ERR_DEVICE;
    }
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    if (hipDeviceGetAttribute(&n_cu_, hipDeviceAttributeMultiprocessorCount, device_) != hipSuccess || n_cu_ <= 0) n_cu_ = 256;
    HIP_TRY(hipEventCreate(&ev0_));
    HIP_TRY(hipEventCreate(&ev1_));
    HIP_TRY(hipMalloc(&partials_, sizeof(T) * 5 * kDotBlocks));   // two dot products per fused pass; five sets in kb_*
    HIP_TRY(hipMalloc(&dscalar_, sizeof(T) * 8));
    HIP_TRY(hipMalloc(&dinfo_, sizeof(int) * 4));
    HIP_TRY(hipMemset(dinfo_, 0, sizeof(int) * 4));
    // multi-workgroup CG exchange area: [2][G] partial sums, [G][2][1024] face planes, [2][G] sequence flags
    HIP_TRY(hipMalloc(&cg_xpart_, sizeof(T) * 2 * EVOMG_CG_GROUPS));
    HIP_TRY(hipMalloc(&cg_xface_, sizeof(T) * EVOMG_CG_GROUPS * 2048));
    HIP_TRY(hipMalloc(&cg_flags_, sizeof(unsigned) * 2 * EVOMG_CG_GROUPS));
    HIP_TRY(hipHostMalloc(&hscalar_, sizeof(T) * 8));
    const int align = 128 / (int)sizeof(T);
    for (int L = prob_.min_level; L <= prob_.max_level; ++L) {
      Level<T>& lv = lv_[L];
      const int n = (1 << L) + 1;
      lv.g.n = n;
      lv.g.pitch = ((n + align - 1) / align) * align;
      lv.g.plane = (dim_ == 3) ? (long)lv.g.pitch * n : 0;
      lv.elems = (size_t)lv.g.pitch * n * (dim_ == 3 ? n : 1);
      for (int f = 0; f < EVOMG_F_COUNT; ++f) bnd_zero_[L][f] = true;
      for (int f : {EVOMG_F_SOL, EVOMG_F_RHS, EVOMG_F_RES}) HIP_TRY(alloc_zero(&lv.f[f], lv.elems));
      if (L == prob_.max_level) HIP_TRY(alloc_zero(&lv.f[EVOMG_F_CORR], lv.elems));
      if (L == prob_.min_level) HIP_TRY(alloc_zero(&lv.f[EVOMG_F_CGS_RHS], lv.elems));
      HIP_TRY(alloc_zero(&lv.alt, lv.elems));
    }
    load_stencils(prob_);
    HIP_TRY(hipStreamSynchronize(stream_));
    return EVOMG_OK;
  }

  int set_initial_field(int level, int field, const double* host) override {
    if (!level_ok(level) || !host) return invalid("set_initial_field: bad level/pointer");
    HIP_TRY(hipSetDevice(device_));
    field = canon(level, field);
    Level<T>& lv = lv_[level];
    if (field < 0 || !lv.f[field]) return invalid("set_initial_field: no such field on this level");
    if (!lv.init[field]) HIP_TRY(alloc_zero(&lv.init[field], lv.elems));
    bnd_zero_[level][field] = false;
    lv.alt_mirror = nullptr;
    HIP_TRY(upload(lv, lv.init[field], host));
    HIP_TRY(hipMemcpyAsync(lv.f[field], lv.init[field], lv.elems * sizeof(T), hipMemcpyDeviceToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return EVOMG_OK;
  }

  int set_field(int level, int field, const double* host) override {
    if (!level_ok(level) || !host) return invalid("set_field: bad level/pointer");
    HIP_TRY(hipSetDevice(device_));
    T* d = fieldptr(level, field);
    if (!d) return invalid("set_field: no such field on this level");
    bnd_zero_[level][canon(level, field)] = false;
    lv_[level].alt_mirror = nullptr;
    HIP_TRY(upload(lv_[level], d, host));
    HIP_TRY(hipStreamSynchronize(stream_));
    return EVOMG_OK;
  }

  int get_field(int level, int field, double* host) override {
    if (!level_ok(level) || !host) return invalid("get_field: bad level/pointer");
    HIP_TRY(hipSetDevice(device_));
    const T* d = fieldptr(level, field);
    if (!d) return invalid("get_field: no such field on this level");
    const Level<T>& lv = lv_[level];
    const size_t rows = (size_t)lv.g.n * (dim_ == 3 ? lv.g.n : 1);
    HIP_TRY(hipStreamSynchronize(stream_));
    HIP_TRY(hipMemcpy2D(host, lv.g.n * sizeof(T), d, lv.g.pitch * sizeof(T), lv.g.n * sizeof(T), rows,
                        hipMemcpyDeviceToHost));
    return EVOMG_OK;
  }

  int set_operators(const evomg_problem_t* p) override {
    if (!p || p->min_level != prob_.min_level || p->max_level != prob_.max_level ||
        p->dimension != prob_.dimension || p->scalar_kind != prob_.scalar_kind)
      return invalid("set_operators: descriptor does not match the handle");
    memcpy(prob_.mg_stencil, p->mg_stencil, sizeof p->mg_stencil);
    memcpy(prob_.outer_stencil, p->outer_stencil, sizeof p->outer_stencil);
    prob_.wavenumber = p->wavenumber;
    load_stencils(prob_);
    return EVOMG_OK;
  }

  int run_ops(const evomg_op_t* ops, int n_ops) override {
    HIP_TRY(hipSetDevice(device_));
    int rc = validate(ops, n_ops);
    if (rc != EVOMG_OK) return rc;
    if (cg_multi_used_) HIP_TRY(hipMemsetAsync(dinfo_, 0, sizeof(int) * 4, stream_));
    rc = exec_program(ops, n_ops, /*cyclic=*/false);
    if (rc != EVOMG_OK) return rc;
    HIP_TRY(hipStreamSynchronize(stream_));
    return coarse_solver_health();
  }

  // the multi-workgroup coarse solver reports a missed rendezvous in info[1] instead of hanging
  int coarse_solver_health() {
    if (!cg_multi_used_) return EVOMG_OK;
    int info[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(info, dinfo_, sizeof info, hipMemcpyDeviceToHost));
    if (info[1] != 0) {
      snprintf(err, sizeof err, "coarse-grid CG: the %d workgroups did not rendezvous (GPU oversubscribed?)", EVOMG_CG_GROUPS);
      return EVOMG_ERR_DEVICE;
    }
    return EVOMG_OK;
  }

  int norm(int level, int field, double* out) override {
    if (!level_ok(level) || !out) return invalid("norm: bad level/pointer");
    HIP_TRY(hipSetDevice(device_));
    const T* d = fieldptr(level, field);
    if (!d) return invalid("norm: no such field on this level");
    T v;
    int rc = dot_to_host(level, d, d, &v);
    if (rc != EVOMG_OK) return rc;
    *out = sc_norm_from_dot(v);
    return EVOMG_OK;
  }

  int evaluate(const evomg_op_t* ops, int n_ops, int samples, double wall_limit_ms, evomg_result_t* out,
               double* hist, int cap) override {
    HIP_TRY(hipSetDevice(device_));
    if (!out || samples < 1 || n_ops < 0 || (!ops && n_ops > 0)) return invalid("evaluate: bad arguments");
    memset(out, 0, sizeof *out);
    int rc = validate(ops, n_ops);
    if (rc != EVOMG_OK) return rc;
    // The eight-workgroup coarse solver (evomg_cg_multi_wg3d) needs its workgroups co-resident; with several candidates
    // in flight on one GPU (population.ConcurrentEvaluator) or another tenant on the device they can miss their
    // rendezvous. That is a property of the moment, not of the candidate: quiesce the device and evaluate once more,
    // alone, before reporting a device error (the solve is deterministic, a clean second attempt is the answer).
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (cg_multi_used_ || attempt > 0) HIP_TRY(hipMemsetAsync(dinfo_, 0, sizeof(int) * 4, stream_));
      rc = evaluate_attempt(ops, n_ops, samples, wall_limit_ms, out, hist, cap);
      if (rc != EVOMG_OK) return rc;
      rc = coarse_solver_health();
      if (rc == EVOMG_OK) return rc;
      if (attempt == 0) HIP_TRY(hipDeviceSynchronize());
    }
    return rc;
  }
  int evaluate_attempt(const evomg_op_t* ops, int n_ops, int samples, double wall_limit_ms, evomg_result_t* out,
                       double* hist, int cap) {
    double tsum = 0.0;
    evomg_result_t r;
    int done = 0;
    for (int s = 0; s < samples; ++s) {
      memset(&r, 0, sizeof r);
      int rc = reset_fields();
      if (rc != EVOMG_OK) return rc;
      if (prob_.outer_kind == EVOMG_OUTER_PBICGSTAB) rc = solve_pbicgstab(ops, n_ops, wall_limit_ms, &r, hist, cap);
      else rc = solve_mg(ops, n_ops, wall_limit_ms, &r, hist, cap);
      if (rc != EVOMG_OK) return rc;
      tsum += r.time_ms;
      ++done;
      if (r.status != EVOMG_SOLVE_CONVERGED) break;
    }
    *out = r;
    out->time_ms = tsum / (double)done;
    return EVOMG_OK;
  }

  int bench_op(const evomg_op_t* op, int warmup, int reps, evomg_bench_t* out) override {
    HIP_TRY(hipSetDevice(device_));
    if (!op || !out || reps < 1 || warmup < 0) return invalid("bench_op: bad arguments");
    int rc = validate(op, 1);
    if (rc != EVOMG_OK) return rc;
    for (int i = 0; i < warmup; ++i) {
      rc = exec(*op);
      if (rc != EVOMG_OK) return rc;
    }
    HIP_TRY(hipEventRecord(ev0_, stream_));
    for (int i = 0; i < reps; ++i) {
      rc = exec(*op);
      if (rc != EVOMG_OK) return rc;
    }
    HIP_TRY(hipEventRecord(ev1_, stream_));
    HIP_TRY(hipEventSynchronize(ev1_));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev0_, ev1_));
    out->ms = (double)ms / reps;
    out->bytes = algorithmic_bytes(*op);
    out->gbps = out->bytes / (out->ms * 1e-3) * 1e-9;
    out->launches = reps;
    return EVOMG_OK;
  }

  int kernel_name(const evomg_op_t* op, char* buf, int len) override {
    if (!op || !buf || len < 8) return invalid("kernel_name: bad arguments");
    const char* base = "?";
    switch (op->opcode) {
      case EVOMG_OP_SMOOTH: {
        const int cells = op->block[0] * op->block[1] * (dim_ == 3 ? op->block[2] : 1);
        base = cells > 1 ? (block_fast_ ? "evomg_block_jacobi_fast" : "evomg_block_jacobi") : op->partitioning == EVOMG_PART_RED_BLACK ? "evomg_rbgs_colour" : "evomg_jacobi";
        if (cells == 1 && use_strip(lv_[op->level].g) && small2d_ok(lv_[op->level].g)) {
          snprintf(buf, len, "evomg_small2d<%s, %d, ", type_name<T>::get(), op->partitioning == EVOMG_PART_RED_BLACK ? STRIP_RBGS : STRIP_JACOBI);
          return EVOMG_OK;
        }
        if (cells == 1 && use_strip(lv_[op->level].g)) {
          snprintf(buf, len, "evomg_strip2d<%s, %d, 2, %s>", type_name<T>::get(), op->partitioning == EVOMG_PART_RED_BLACK ? STRIP_RBGS : STRIP_JACOBI, strip_streams(lv_[op->level].g) ? "true" : "false");
          return EVOMG_OK;
        }
        if (cells > 1 && dim_ == 2 && strip_block_ && use_strip(lv_[op->level].g) && op->block[0] <= 2 && op->block[1] <= 2) {
          snprintf(buf, len, "evomg_strip2d_block<%s, %d, %d, %s>", type_name<T>::get(), op->block[0], op->block[1], strip_streams(lv_[op->level].g) ? "true" : "false");
          return EVOMG_OK;
        }
        if (cells == 1 && use_tile3d(lv_[op->level].g) && (op->partitioning != EVOMG_PART_RED_BLACK || tile3d_rbgs_)) {
          if (op->partitioning == EVOMG_PART_RED_BLACK) snprintf(buf, len, "evomg_tile3d_rbgs<%s, ", type_name<T>::get());
          else snprintf(buf, len, "evomg_tile3d<%s, %d, ", type_name<T>::get(), STRIP_JACOBI);
          return EVOMG_OK;
        }
        if (cells == 2 && dim_ == 3 && op->block[2] == 1 && block_fast_ && use_tile3d_blocks(lv_[op->level].g) && !scalar_traits<T>::is_complex) {
          snprintf(buf, len, "evomg_tile3d<%s, %d, ", type_name<T>::get(), op->block[0] == 2 ? TILE_BLOCK211 : TILE_BLOCK121);
          return EVOMG_OK;
        }
        if (cells == 1 && use_strip3d(lv_[op->level].g, op->partitioning == EVOMG_PART_RED_BLACK)) {
          if (op->partitioning == EVOMG_PART_RED_BLACK) snprintf(buf, len, "evomg_strip3d_rbgs<%s, 4, %s>", type_name<T>::get(), strip_streams3d(lv_[op->level].g) ? "true" : "false");
          else snprintf(buf, len, "evomg_strip3d<%s, %d, 4, %s>", type_name<T>::get(), STRIP_JACOBI, strip_streams3d(lv_[op->level].g) ? "true" : "false");
          return EVOMG_OK;
        }
        break;
      }
      case EVOMG_OP_RESIDUAL:
        base = "evomg_apply";
        if (use_strip(lv_[op->level].g)) { snprintf(buf, len, "evomg_strip2d<%s, %d, 2, %s>", type_name<T>::get(), STRIP_RESIDUAL, strip_streams(lv_[op->level].g) ? "true" : "false"); return EVOMG_OK; }
        if (use_tile3d(lv_[op->level].g)) { snprintf(buf, len, "evomg_tile3d<%s, %d, ", type_name<T>::get(), STRIP_RESIDUAL); return EVOMG_OK; }
        if (use_strip3d(lv_[op->level].g, false)) { snprintf(buf, len, "evomg_strip3d<%s, %d, 4, %s>", type_name<T>::get(), STRIP_RESIDUAL, strip_streams3d(lv_[op->level].g) ? "true" : "false"); return EVOMG_OK; }
        break;
      case EVOMG_OP_RICHARDSON: base = "evomg_richardson"; break;
      case EVOMG_OP_RESTRICT: base = "evomg_restrict"; break;
      case EVOMG_OP_PROLONG_ADD: case EVOMG_OP_PROLONG_SET:
        base = "evomg_prolong";
        if (use_strip(lv_[op->level].g)) { snprintf(buf, len, "evomg_strip_prolong<%s, %s>", type_name<T>::get(), op->opcode == EVOMG_OP_PROLONG_ADD ? "true" : "false"); return EVOMG_OK; }
        if (dim_ == 3 && strip_enabled_) { snprintf(buf, len, "evomg_prolong_cell3d<%s, %s>", type_name<T>::get(), op->opcode == EVOMG_OP_PROLONG_ADD ? "true" : "false"); return EVOMG_OK; }
        break;
      case EVOMG_OP_SET_ZERO: base = "evomg_zero_inner"; break;
      case EVOMG_OP_COPY: base = "hipMemcpyAsync"; break;
      case EVOMG_OP_COARSE_SOLVE: {
        // the kernel coarse_solve() launches for this problem's coarsest grid
        const int n = lv_[prob_.min_level].g.n;
        const bool real_cg = prob_.cgs_kind != EVOMG_CGS_BICGSTAB && !scalar_traits<T>::is_complex;
        if (prob_.cgs_kind == EVOMG_CGS_BICGSTAB) {
          base = (dim_ == 2 && n <= 9 && use_reg_krylov_) ? "evomg_bicgstab_wave2d"
                 : (dim_ == 2 && kcg_enabled_ && n >= kb_min_n_) ? "kb_init / kb_p / kb_spmv / kb_hs / kb_xr / kb_check" : "evomg_bicgstab_single_wg";
        } else if (real_cg && dim_ == 3 && krylov_cg_groups<3>(n) > 1) base = "evomg_cg_multi_wg3d";
        else if (real_cg && kcg_enabled_ && (dim_ == 2 ? krylov_cg_per_launch<2>(n) : krylov_cg_per_launch<3>(n))) base = "kcg_init / kcg_apply / kcg_update / kcg_direct";
        else if (real_cg && use_reg_krylov_ && dim_ == 2 && krylov_own<2>(n, true).wave) base = "evomg_cg_wave2d";
        else if (real_cg && use_reg_krylov_ && dim_ == 2 && krylov_own<2>(n, true).tile && krylov_own<2>(n, true).lanes == 512) base = "evomg_cg_reg2d";
        else if (real_cg && use_reg_krylov_ && dim_ == 3 && krylov_own<3>(n, true).column && krylov_own<3>(n, true).m <= EVOMG_CG_WAVE3D_MAX_M) base = "evomg_cg_wave3d";
        else base = "evomg_cg_single_wg";
        break;
      }
    }
    snprintf(buf, len, "%s<%s, %d", base, type_name<T>::get(), dim_);
    return EVOMG_OK;
  }

  int profile_begin(int opcode, int level) override {
    if (!level_ok(level)) return invalid("profile_begin: bad level");
    prof_opcode_ = opcode;
    prof_level_ = level;
    prof_used_ = 0;
    prof_bytes_ = 0.0;
    return EVOMG_OK;
  }

  int profile_end(evomg_bench_t* out) override {
    if (!out) return invalid("profile_end: null result");
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(hipStreamSynchronize(stream_));
    double total = 0.0;
    for (size_t i = 0; i + 1 < prof_used_; i += 2) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, prof_events_[i], prof_events_[i + 1]));
      total += ms;
    }
    const int launches = (int)(prof_used_ / 2);
    out->launches = launches;
    out->ms = launches ? total / launches : 0.0;
    out->bytes = launches ? prof_bytes_ / launches : 0.0;      // mean over the launches (fused ones move more)
    out->gbps = (launches && total > 0) ? prof_bytes_ / (total * 1e-3) * 1e-9 : 0.0;
    prof_opcode_ = -1;
    return EVOMG_OK;
  }

 private:
  // ---- storage ---------------------------------------------------------------------------------------------------
  hipError_t alloc_zero(T** p, size_t elems) {
    hipError_t e = hipMalloc(p, elems * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemsetAsync(*p, 0, elems * sizeof(T), stream_);
  }
  hipError_t upload(const Level<T>& lv, T* dst, const double* host) {
    const size_t rows = (size_t)lv.g.n * (dim_ == 3 ? lv.g.n : 1);
    hipError_t e = hipStreamSynchronize(stream_);
    if (e != hipSuccess) return e;
    return hipMemcpy2D(dst, lv.g.pitch * sizeof(T), host, lv.g.n * sizeof(T), lv.g.n * sizeof(T), rows,
                       hipMemcpyHostToDevice);
  }
  bool level_ok(int L) const { return L >= prob_.min_level && L <= prob_.max_level; }
  int canon(int level, int field) const {
    if (field < 0 || field >= EVOMG_F_COUNT) return -1;
    // gen_error_u@L and the solution of level L < max are the same array (exastencils.py:296-301)
    if (field == EVOMG_F_CORR && level < prob_.max_level) return EVOMG_F_SOL;
    return field;
  }
  T* fieldptr(int level, int field) {
    field = canon(level, field);
    return field < 0 ? nullptr : lv_[level].f[field];
  }
  void set_fieldptr(int level, int field, T* p) { lv_[level].f[canon(level, field)] = p; }
  int invalid(const char* msg) {
    snprintf(err, sizeof err, "%s", msg);
    return EVOMG_ERR_INVALID;
  }
  void load_stencils(const evomg_problem_t& p) {
    for (int L = p.min_level; L <= p.max_level; ++L)
      for (int e = 0; e < EVOMG_STENCIL_ENTRIES; ++e)
        lv_[L].st.c[e] = scalar_traits<T>::from_c64(p.mg_stencil[L][e].re, p.mg_stencil[L][e].im);
    for (int e = 0; e < EVOMG_STENCIL_ENTRIES; ++e)
      outer_st_.c[e] = scalar_traits<T>::from_c64(p.outer_stencil[e].re, p.outer_stencil[e].im);
  }
  int tmpvec(int level, int which, T** out) {
    Level<T>& lv = lv_[level];
    if (!lv.tmp[which]) HIP_TRY(alloc_zero(&lv.tmp[which], lv.elems));
    *out = lv.tmp[which];
    return EVOMG_OK;
  }

  // ---- launch geometry -------------------------------------------------------------------------------------------
  static dim3 node_block() { return dim3(64, 4, 1); }
  dim3 node_grid(const Geo& g) const { return dim3((g.n + 63) / 64, (g.n + 3) / 4, dim_ == 3 ? g.n : 1); }
  double kh(int level) const { return prob_.wavenumber / (double)(lv_[level].g.n - 1); }

  int apply_bc(int level, T* x) {
    if constexpr (scalar_traits<T>::is_complex) {
      if (prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN) {
        const Geo& g = lv_[level].g;
        hipLaunchKernelGGL(evomg_robin_bc, dim3((g.n + 255) / 256), dim3(256), 0, stream_, g, x, kh(level));
      }
    }
    return EVOMG_OK;
  }

  // ---- validation (mirrors oracle FN(check_op)) ---------------------------------------------------------------------
  int validate(const evomg_op_t* ops, int n_ops) {
    for (int i = 0; i < n_ops; ++i) {
      const evomg_op_t& op = ops[i];
      const int lo = prob_.min_level, hi = prob_.max_level;
      bool ok = op.level >= lo && op.level <= hi && op.dst >= 0 && op.dst < EVOMG_F_COUNT;
      auto fld = [&](int f, int level) { return f >= 0 && f < EVOMG_F_COUNT && level >= lo && level <= hi && fieldptr(level, f) != nullptr; };
      int rc = EVOMG_ERR_INVALID;
      if (ok) {
        ok = fld(op.dst, op.level);
        switch (op.opcode) {
          case EVOMG_OP_SMOOTH: {
            ok = ok && fld(op.src1, op.level);
            int cells = 1;
            for (int d = 0; d < dim_; ++d) {
              if (op.block[d] < 1) ok = false;
              cells *= op.block[d] > 0 ? op.block[d] : 1;
            }
            if (cells > EVOMG_MAX_BLOCK_CELLS) ok = false;
            if (ok && cells > 1 && op.partitioning != EVOMG_PART_SINGLE) { ok = false; rc = EVOMG_ERR_UNSUPPORTED; }
            if (op.partitioning != EVOMG_PART_SINGLE && op.partitioning != EVOMG_PART_RED_BLACK) ok = false;
            break;
          }
          case EVOMG_OP_RESIDUAL: ok = ok && fld(op.src0, op.level) && fld(op.src1, op.level); break;
          case EVOMG_OP_RICHARDSON: ok = ok && fld(op.src1, op.level); break;
          case EVOMG_OP_RESTRICT: ok = ok && op.level < hi && fld(op.src0, op.level + 1); break;
          case EVOMG_OP_PROLONG_ADD: case EVOMG_OP_PROLONG_SET: ok = ok && op.level > lo && fld(op.src0, op.level - 1); break;
          case EVOMG_OP_SET_ZERO: break;
          case EVOMG_OP_COPY: ok = ok && fld(op.src0, op.level); break;
          case EVOMG_OP_COARSE_SOLVE: ok = ok && op.level == lo; break;
          default: ok = false;
        }
      }
      if (!ok) {
        snprintf(err, sizeof err, "op %d (opcode %d, level %d) rejected", i, op.opcode, op.level);
        return rc;
      }
    }
    return EVOMG_OK;
  }

  // ---- op interpreter ----------------------------------------------------------------------------------------------
  // ---- op program execution with peephole fusion -------------------------------------------------------------------
  // RESIDUAL@L immediately followed by RESTRICT@(L-1) of that residual runs as ONE kernel (evomg_strip_residual_restrict).
  // `cyclic`: the program is the body of a solver loop, so "never read again before it is overwritten" can be decided
  // by scanning around the loop; then the residual field is not even written.
  bool touches(const evomg_op_t& op, int level, int field, bool* reads) const {
    auto same = [&](int f, int l) { return l == level && canon(l, f) == canon(level, field); };
    bool r = false, w = false;
    switch (op.opcode) {
      case EVOMG_OP_SMOOTH: r = same(op.dst, op.level) || same(op.src1, op.level); w = same(op.dst, op.level); break;
      case EVOMG_OP_RESIDUAL: r = same(op.src0, op.level) || same(op.src1, op.level); w = same(op.dst, op.level); break;
      case EVOMG_OP_RICHARDSON: r = same(op.dst, op.level) || same(op.src1, op.level); w = same(op.dst, op.level); break;
      case EVOMG_OP_RESTRICT: r = same(op.src0, op.level + 1); w = same(op.dst, op.level); break;
      case EVOMG_OP_PROLONG_ADD: r = same(op.src0, op.level - 1) || same(op.dst, op.level); w = same(op.dst, op.level); break;
      case EVOMG_OP_PROLONG_SET: r = same(op.src0, op.level - 1); w = same(op.dst, op.level); break;
      case EVOMG_OP_SET_ZERO: w = same(op.dst, op.level); break;
      case EVOMG_OP_COPY: r = same(op.src0, op.level); w = same(op.dst, op.level); break;
      case EVOMG_OP_COARSE_SOLVE:
        r = same(EVOMG_F_CGS_RHS, op.level) || same(EVOMG_F_CORR, op.level);
        w = same(EVOMG_F_CORR, op.level) || same(EVOMG_F_RES, op.level);
        break;
    }
    *reads = r;
    return r || w;
  }
  bool residual_dead_after(const evomg_op_t* ops, int n_ops, int i) const {
    const int level = ops[i].level, field = ops[i].dst;
    for (int k = i + 2; k < n_ops; ++k) {
      bool reads;
      if (touches(ops[k], level, field, &reads)) return !reads;
    }
    // end of the loop body: the outer loops recompute (overwrite) RES@max before reading it and never read any other
    // level's fields; afterwards the body starts over
    if (level == prob_.max_level && canon(level, field) == EVOMG_F_RES) return true;
    for (int k = 0; k < i; ++k) {
      bool reads;
      if (touches(ops[k], level, field, &reads)) return !reads;
    }
    return true;   // nothing else touches it: the same pair overwrites it in the next pass
  }
  // COPY gen_rhs@min = ..., SET_ZERO gen_error@min, COARSE_SOLVE@min (what every cycle emits around gen_mgCycle@min,
  // exastencils.py:890-896) on a grid the one-wave CG serves: ONE launch instead of three
  // One visit of a small 3D level in one launch (evomg_small3d.h): [SET_ZERO u | PROLONG_ADD u] + 1..4 identical red-black
  // point sweeps on u + [RESIDUAL + RESTRICT of it], as many of these as the program has at position i. Returns the
  // number of ops consumed (0: not this shape), -1 on a launch error.
  int small3d_chain(const evomg_op_t* ops, int n_ops, int i) {
    if constexpr (scalar_traits<T>::is_complex) return 0;
    else {
      if (!small3d_enabled_ || !fuse_enabled_ || dim_ != 3 || prob_.bc_kind != EVOMG_BC_DIRICHLET) return 0;
      const int L = ops[i].level;
      if (L < prob_.min_level || L > prob_.max_level || lv_[L].g.n > 17) return 0;
      int p = i, zero_first = 0;
      const T* coarse_in = nullptr;
      double om_p = 0.0;
      int head_field = -1;
      if (ops[p].opcode == EVOMG_OP_SET_ZERO) {
        zero_first = 1;
        head_field = canon(L, ops[p].dst);
        ++p;
      } else if (ops[p].opcode == EVOMG_OP_PROLONG_ADD && L > prob_.min_level) {
        head_field = canon(L, ops[p].dst);
        coarse_in = fieldptr(L - 1, ops[p].src0);
        om_p = ops[p].omega;
        ++p;
      }
      if (p < n_ops && prof_matches(ops[i])) return 0;
      if (p >= n_ops || ops[p].opcode != EVOMG_OP_SMOOTH || ops[p].level != L) return 0;
      const evomg_op_t& sm = ops[p];
      if (sm.partitioning != EVOMG_PART_RED_BLACK || sm.block[0] * sm.block[1] * sm.block[2] != 1) return 0;
      const int xf = canon(L, sm.dst), ff = canon(L, sm.src1);
      if (xf == ff || (head_field >= 0 && head_field != xf)) return 0;
      if (coarse_in && coarse_in == fieldptr(L, sm.dst)) return 0;
      int nsweeps = 0;
      while (p < n_ops && nsweeps < 4 && ops[p].opcode == EVOMG_OP_SMOOTH && ops[p].level == L && ops[p].partitioning == EVOMG_PART_RED_BLACK &&
             ops[p].block[0] * ops[p].block[1] * ops[p].block[2] == 1 && canon(L, ops[p].dst) == xf && canon(L, ops[p].src1) == ff &&
             ops[p].omega == sm.omega && !prof_matches(ops[p])) {
        ++nsweeps;
        ++p;
      }
      if (nsweeps == 0) return 0;
      T* res_out = nullptr;
      T* coarse_out = nullptr;
      if (p + 1 < n_ops && L > prob_.min_level && ops[p].opcode == EVOMG_OP_RESIDUAL && ops[p].level == L && canon(L, ops[p].src0) == xf &&
          canon(L, ops[p].src1) == ff && canon(L, ops[p].dst) != xf && canon(L, ops[p].dst) != ff && ops[p + 1].opcode == EVOMG_OP_RESTRICT &&
          ops[p + 1].level == L - 1 && canon(L, ops[p + 1].src0) == canon(L, ops[p].dst) && !prof_matches(ops[p]) && !prof_matches(ops[p + 1])) {
        res_out = fieldptr(L, ops[p].dst);
        coarse_out = fieldptr(L - 1, ops[p + 1].dst);
        // the coarse field written must not be one the launch reads
        if (coarse_out == coarse_in) { res_out = nullptr; coarse_out = nullptr; }
        else p += 2;
      }
      Level<T>& lv = lv_[L];
      const Geo& g = lv.g;
      const size_t nn = (size_t)g.n * g.n * g.n;
      const size_t lds_bytes = (coarse_out ? 3 : 2) * nn * sizeof(T);
      if (lds_bytes > 48 * 1024 && !small3d_attr_set_) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(evomg_small3d_chain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        small3d_attr_set_ = true;
      }
      const Geo gc = L > prob_.min_level ? lv_[L - 1].g : g;
      hipLaunchKernelGGL(evomg_small3d_chain, dim3(1), dim3(1024), lds_bytes, stream_, g, lv.st, fieldptr(L, sm.dst), (const T*)fieldptr(L, sm.src1),
                         sm.omega, 1.0 - sm.omega, nsweeps, zero_first, coarse_in, gc, om_p, res_out, coarse_out);
      if (hipGetLastError() != hipSuccess) { snprintf(err, sizeof err, "evomg_small3d_chain: launch failed"); return -1; }
      return p - i;
    }
  }
  bool fusable_copy_zero_solve(const evomg_op_t* ops, int n_ops, int i) const {
    if (!fuse_enabled_ || !use_reg_krylov_ || i + 2 >= n_ops || dim_ != 2) return false;
    const int L = prob_.min_level;
    // the one-wave CG (real, Dirichlet) or the one-wave BiCGStab (up to 9^2, real or complex with its Robin columns)
    const bool cg_case = !scalar_traits<T>::is_complex && prob_.cgs_kind == EVOMG_CGS_CG && prob_.bc_kind == EVOMG_BC_DIRICHLET;
    const bool bicg_case = prob_.cgs_kind == EVOMG_CGS_BICGSTAB && lv_[L].g.n <= 9 &&
                           (prob_.bc_kind == EVOMG_BC_DIRICHLET || (prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN && scalar_traits<T>::is_complex));
    if (!cg_case && !bicg_case) return false;
    const evomg_op_t &cp = ops[i], &z = ops[i + 1], &cs = ops[i + 2];
    if (cp.opcode != EVOMG_OP_COPY || z.opcode != EVOMG_OP_SET_ZERO || cs.opcode != EVOMG_OP_COARSE_SOLVE) return false;
    if (cp.level != L || z.level != L || cs.level != L) return false;
    if (canon(L, cp.dst) != EVOMG_F_CGS_RHS || canon(L, cp.src0) == EVOMG_F_CGS_RHS) return false;
    if (canon(L, z.dst) != canon(L, EVOMG_F_CORR) || canon(L, cp.src0) == canon(L, EVOMG_F_CORR) || canon(L, cp.src0) == EVOMG_F_RES) return false;
    if (prof_matches(cp) || prof_matches(z) || prof_matches(cs)) return false;
    // "x = 0, so r = b" needs the boundary values of x to be zero as well (SET_ZERO leaves them alone); true for every
    // error field of a cycle, not for a field a test filled with random numbers
    if (!bnd_zero_[L][canon(L, EVOMG_F_CORR)]) return false;
    return bicg_case || krylov_own<2>(lv_[L].g.n, true).wave != 0;
  }
  bool fusable_residual_restrict(const evomg_op_t* ops, int n_ops, int i) const {
    if (!fuse_enabled_ || i + 1 >= n_ops) return false;
    const evomg_op_t &a = ops[i], &b = ops[i + 1];
    if (a.opcode != EVOMG_OP_RESIDUAL || b.opcode != EVOMG_OP_RESTRICT) return false;
    if (b.level != a.level - 1 || canon(a.level, b.src0) != canon(a.level, a.dst)) return false;
    if (canon(a.level, a.dst) == canon(a.level, a.src0) || canon(a.level, a.dst) == canon(a.level, a.src1)) return false;
    if (!use_strip(lv_[a.level].g)) return false;
    if ((a.opcode == prof_opcode_ && a.level == prof_level_) || (b.opcode == prof_opcode_ && b.level == prof_level_)) return false;
    return true;
  }
  // tail_norm (solver loop only): set to true when the last sweep of the program also produced the canonical partial
  // sums of the squared residual norm (norm_partials_)
  int exec_program(const evomg_op_t* ops, int n_ops, bool cyclic, int first = 0, bool* tail_norm = nullptr) {
    if (tail_norm) *tail_norm = false;
    for (int i = first; i < n_ops; ++i) {
      if (batchable(ops[i])) {              // small level: the op joins the list one workgroup runs in one launch
        const int rc = exec_batched(ops[i]);
        if (rc != EVOMG_OK) return rc;
        continue;
      }
      flush_batch();
      if (elidable_zero(ops, n_ops, i)) {
        x_zero_next_ = true;          // consumed by the sweep launched for ops[i + 1] (plain or fused)
        continue;
      }
      {
        const int len = small3d_chain(ops, n_ops, i);
        if (len < 0) return EVOMG_ERR_DEVICE;
        if (len > 0) {
          i += len - 1;
          continue;
        }
      }
      if (fusable_copy_zero_solve(ops, n_ops, i)) {
        const int rc = coarse_solve(&ops[i]);
        if (rc != EVOMG_OK) return rc;
        i += 2;
        continue;
      }
      if (fusable_prolong_twice(ops, n_ops, i)) {
        const int rc = exec_twice(ops[i + 1], &ops[i]);
        if (rc != EVOMG_OK) return rc;
        i += 2;
        continue;
      }
      if (fusable_twice(ops, n_ops, i)) {
        const int rc = exec_twice(ops[i]);
        if (rc != EVOMG_OK) return rc;
        ++i;
        continue;
      }
      if (fusable_two_sweeps_residual_restrict(ops, n_ops, i, cyclic)) {
        // pre-smoothing with two identical red-black sweeps on a level that is ONE LDS tile: both sweeps, the residual and
        // the restriction in one launch (evomg_small2d with nsweeps = 2)
        const evomg_op_t &sm = ops[i], &rs = ops[i + 3];
        const int L = sm.level;
        Level<T>& lv = lv_[L];
        T* x = fieldptr(L, sm.dst);
        T* y = lv.alt;
        launch_strip_restrict_tail<STRIP_RBGS>(lv.g, lv_[L - 1].g, lv.st, x, fieldptr(L, sm.src1), y, sm.omega, 1.0 - sm.omega, fieldptr(L - 1, rs.dst), 2);
        lv.alt = x;
        set_fieldptr(L, sm.dst, y);
        lv.alt_mirror = y;
        HIP_TRY(hipGetLastError());
        i += 3;
        continue;
      }
      if (fusable_smooth_residual_restrict(ops, n_ops, i, cyclic)) {
        const evomg_op_t &sm = ops[i], &rs = ops[i + 2];
        const int L = sm.level;
        Level<T>& lv = lv_[L];
        T* x = fieldptr(L, sm.dst);
        T* y = lv.alt;
        const double om = sm.omega, omm = 1.0 - sm.omega;
        const bool timed = prof_matches(sm);
        if (timed) { const int rc = prof_open(); if (rc != EVOMG_OK) return rc; }
        if (sm.partitioning == EVOMG_PART_RED_BLACK)
          launch_strip_restrict_tail<STRIP_RBGS>(lv.g, lv_[L - 1].g, lv.st, x, fieldptr(L, sm.src1), y, om, omm, fieldptr(L - 1, rs.dst));
        else
          launch_strip_restrict_tail<STRIP_JACOBI>(lv.g, lv_[L - 1].g, lv.st, x, fieldptr(L, sm.src1), y, om, omm, fieldptr(L - 1, rs.dst));
        // x, f read + y written + the coarse right-hand side written
        if (timed) { const int rc = prof_close(algorithmic_bytes(sm) + (double)sizeof(T) * lv_[L - 1].g.n * lv_[L - 1].g.n); if (rc != EVOMG_OK) return rc; }
        lv.alt = x;
        set_fieldptr(L, sm.dst, y);
        lv.alt_mirror = y;             // the old array (now the spare one) has the same boundary values
        HIP_TRY(hipGetLastError());
        i += 2;
        continue;
      }
      if (fusable_residual_restrict3d(ops, n_ops, i, cyclic)) {
        const int L = ops[i].level;
        launch_tile3d_residual_restrict(L, fieldptr(L, ops[i].src0), fieldptr(L, ops[i].src1), fieldptr(L - 1, ops[i + 1].dst));
        apply_bc(L - 1, fieldptr(L - 1, ops[i + 1].dst));
        HIP_TRY(hipGetLastError());
        ++i;
        continue;
      }
      if (fusable_residual_restrict(ops, n_ops, i)) {
        const bool write_r = !(cyclic && residual_dead_after(ops, n_ops, i));
        const int L = ops[i].level;
        launch_residual_restrict(L, fieldptr(L, ops[i].src0), fieldptr(L, ops[i].src1), fieldptr(L, ops[i].dst),
                                 fieldptr(L - 1, ops[i + 1].dst), write_r);
        if (!strip_bc_) {                                   // otherwise the kernel wrote the Robin columns itself
          if (write_r) apply_bc(L, fieldptr(L, ops[i].dst));
          apply_bc(L - 1, fieldptr(L - 1, ops[i + 1].dst));
        }
        HIP_TRY(hipGetLastError());
        ++i;
        continue;
      }
      if (fusable_prolong_smooth(ops, n_ops, i)) {
        const evomg_op_t &pr = ops[i], &sm = ops[i + 1];
        const int L = pr.level;
        Level<T>& lv = lv_[L];
        T* x = fieldptr(L, sm.dst);
        T* y = lv.alt;
        const double om = sm.omega, omm = 1.0 - sm.omega;
        const bool rb = sm.partitioning == EVOMG_PART_RED_BLACK;
        const bool tail = tail_norm && tail_norm_candidate(ops, n_ops, i + 1);
        const T* f = fieldptr(L, sm.src1);
        const T* cz = fieldptr(L - 1, pr.src0);
        const bool timed = prof_matches(sm);
        if (timed) { const int rc0 = prof_open(); if (rc0 != EVOMG_OK) return rc0; }
        int rc;
        if (tail) rc = rb ? launch_strip_fused<STRIP_RBGS, true, true>(lv.g, lv_[L - 1].g, lv.st, x, f, y, om, omm, cz, pr.omega)
                          : launch_strip_fused<STRIP_JACOBI, true, true>(lv.g, lv_[L - 1].g, lv.st, x, f, y, om, omm, cz, pr.omega);
        else rc = rb ? launch_strip_fused<STRIP_RBGS, true, false>(lv.g, lv_[L - 1].g, lv.st, x, f, y, om, omm, cz, pr.omega)
                     : launch_strip_fused<STRIP_JACOBI, true, false>(lv.g, lv_[L - 1].g, lv.st, x, f, y, om, omm, cz, pr.omega);
        if (rc != EVOMG_OK) return rc;
        // x, f read + y written + the coarse correction read
        if (timed) { const int rc2 = prof_close(algorithmic_bytes(sm) + (double)sizeof(T) * lv_[L - 1].g.n * lv_[L - 1].g.n); if (rc2 != EVOMG_OK) return rc2; }
        if (tail) *tail_norm = true;
        lv.alt = x;
        set_fieldptr(L, sm.dst, y);
        lv.alt_mirror = y;             // the old array (now the spare one) has the same boundary values
        HIP_TRY(hipGetLastError());
        ++i;
        continue;
      }
      if (tail_norm && tail_norm_candidate(ops, n_ops, i)) {
        const evomg_op_t& sm = ops[i];
        const int L = sm.level;
        Level<T>& lv = lv_[L];
        T* x = fieldptr(L, sm.dst);
        T* y = lv.alt;
        const double om = sm.omega, omm = 1.0 - sm.omega;
        const bool timed = prof_matches(sm);
        if (timed) { const int rc0 = prof_open(); if (rc0 != EVOMG_OK) return rc0; }
        const int rc = sm.partitioning == EVOMG_PART_RED_BLACK
                         ? launch_strip_fused<STRIP_RBGS, false, true>(lv.g, lv.g, lv.st, x, fieldptr(L, sm.src1), y, om, omm, nullptr, 0.0)
                         : launch_strip_fused<STRIP_JACOBI, false, true>(lv.g, lv.g, lv.st, x, fieldptr(L, sm.src1), y, om, omm, nullptr, 0.0);
        if (rc != EVOMG_OK) return rc;
        if (timed) { const int rc2 = prof_close(algorithmic_bytes(sm)); if (rc2 != EVOMG_OK) return rc2; }
        *tail_norm = true;
        lv.alt = x;
        set_fieldptr(L, sm.dst, y);
        lv.alt_mirror = y;             // the old array (now the spare one) has the same boundary values
        HIP_TRY(hipGetLastError());
        continue;
      }
      int rc = exec(ops[i]);
      if (rc != EVOMG_OK) return rc;
    }
    flush_batch();
    return EVOMG_OK;
  }

  // ---- coarse-level op batches (evomg_batch.h) --------------------------------------------------------------------
  bool batch_level(int L) const { return L >= 0 && L < EVOMG_BATCH_LEVELS && lv_[L].g.n <= batch_max_n_; }
  bool batchable(const evomg_op_t& op) const {
    if (!batch_enabled_ || dim_ != 2 || scalar_traits<T>::is_complex || prob_.bc_kind != EVOMG_BC_DIRICHLET) return false;
    if (prof_matches(op)) return false;
    const int L = op.level;
    switch (op.opcode) {
      case EVOMG_OP_SMOOTH: return batch_level(L) && op.block[0] * op.block[1] == 1 && canon(L, op.src1) != canon(L, op.dst);
      case EVOMG_OP_RESIDUAL: return batch_level(L) && canon(L, op.dst) != canon(L, op.src0) && canon(L, op.dst) != canon(L, op.src1);
      case EVOMG_OP_RESTRICT: return batch_level(L) && batch_level(L + 1);
      case EVOMG_OP_PROLONG_ADD: case EVOMG_OP_PROLONG_SET: return batch_level(L) && L - 1 >= prob_.min_level;
      case EVOMG_OP_SET_ZERO: case EVOMG_OP_COPY: return batch_level(L);
      default: return false;
    }
  }
  void push_micro(int kind, int level, const void* a, const void* b, void* c, double om) {
    if (batch_.n == EVOMG_BATCH_MAX) flush_batch();
    batch_.op[batch_.n++] = MicroOp{kind, level, a, b, c, om};
  }
  void flush_batch() {
    if (batch_.n == 0) return;
    if constexpr (!scalar_traits<T>::is_complex) {
      BatchLevels<T> lv;
      memset(&lv, 0, sizeof lv);
      for (int L = prob_.min_level; L <= prob_.max_level && L < EVOMG_BATCH_LEVELS; ++L) { lv.g[L] = lv_[L].g; lv.st[L] = lv_[L].st; }
      hipLaunchKernelGGL((evomg_coarse_batch<T>), dim3(1), dim3(1024), 0, stream_, lv, batch_);
    }
    batch_.n = 0;
  }
  // the bookkeeping of exec_impl (ping-pong arrays, `with jacobi` slots), a micro-op instead of each launch
  int exec_batched(const evomg_op_t& op) {
    const int L = op.level;
    Level<T>& lv = lv_[L];
    switch (op.opcode) {
      case EVOMG_OP_SMOOTH: {
        T* x = fieldptr(L, op.dst);
        const T* f = fieldptr(L, op.src1);
        if (op.partitioning == EVOMG_PART_RED_BLACK) {
          push_micro(MB_RBGS, L, x, f, nullptr, op.omega);          // two colour passes in place
        } else {
          T* y = lv.alt;
          push_micro(MB_JACOBI, L, x, f, y, op.omega);              // writes all of y, boundary included
          advance(lv, L, op, x, y);
        }
        break;
      }
      case EVOMG_OP_RESIDUAL: push_micro(MB_RESIDUAL, L, fieldptr(L, op.src0), fieldptr(L, op.src1), fieldptr(L, op.dst), 0.0); break;
      case EVOMG_OP_RESTRICT: push_micro(MB_RESTRICT, L, fieldptr(L + 1, op.src0), nullptr, fieldptr(L, op.dst), 0.0); break;
      case EVOMG_OP_PROLONG_ADD: push_micro(MB_PROLONG_ADD, L, fieldptr(L - 1, op.src0), nullptr, fieldptr(L, op.dst), op.omega); break;
      case EVOMG_OP_PROLONG_SET: push_micro(MB_PROLONG_SET, L, fieldptr(L - 1, op.src0), nullptr, fieldptr(L, op.dst), 1.0); break;
      case EVOMG_OP_SET_ZERO: push_micro(MB_ZERO, L, nullptr, nullptr, fieldptr(L, op.dst), 0.0); break;
      case EVOMG_OP_COPY: {
        T* d = fieldptr(L, op.dst);
        const T* s = fieldptr(L, op.src0);
        if (d != s) push_micro(MB_COPY, L, s, nullptr, d, 0.0);
        bnd_zero_[L][canon(L, op.dst)] = bnd_zero_[L][canon(L, op.src0)];
        lv.alt_mirror = nullptr;
        break;
      }
      default: return invalid("exec_batched: unsupported opcode");
    }
    return EVOMG_OK;
  }
  void launch_residual_restrict(int L, const T* x, const T* f, T* r, T* coarse, bool write_r) {
    const Geo& g = lv_[L].g;
    const Geo& gc = lv_[L - 1].g;
    const int strips = (g.n + EVOMG_STRIP_COLS - 1) / EVOMG_STRIP_COLS;
    int rows = strip_rows_ > 0 ? strip_rows_ : 32;
    while (rows > 8 && (long)strips * ((g.n + rows - 1) / rows) < 2048) rows /= 2;
    rows = (rows + 1) & ~1;
    if (strip_rows_ <= 0 && balance_rows_) rows = balanced_rows(g.n, strips, 2, 8, 96);
    const int chunks = (g.n + rows - 1) / rows;
    const int waves = strips * chunks;
    const dim3 grid((((waves + 3) / 4 + 7) / 8) * 8), block(256);
    const bool nt = strip_streams(g);
    const bool robin = strip_bc_ && prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN;
    const double rkh = robin ? kh(L) : 0.0, rkhc = robin ? kh(L - 1) : 0.0;
    if (write_r) {
      if (nt) hipLaunchKernelGGL((evomg_strip_residual_restrict<T, true, true>), grid, block, 0, stream_, g, gc, lv_[L].st, x, f, r, coarse, rows, strips, waves, rkh, rkhc);
      else hipLaunchKernelGGL((evomg_strip_residual_restrict<T, true, false>), grid, block, 0, stream_, g, gc, lv_[L].st, x, f, r, coarse, rows, strips, waves, rkh, rkhc);
    } else {
      if (nt) hipLaunchKernelGGL((evomg_strip_residual_restrict<T, false, true>), grid, block, 0, stream_, g, gc, lv_[L].st, x, f, r, coarse, rows, strips, waves, rkh, rkhc);
      else hipLaunchKernelGGL((evomg_strip_residual_restrict<T, false, false>), grid, block, 0, stream_, g, gc, lv_[L].st, x, f, r, coarse, rows, strips, waves, rkh, rkhc);
    }
  }
  void launch_strip_prolong(int L, const T* coarse, T* fine, bool acc, double om) {
    const Geo& g = lv_[L].g;
    const Geo& gc = lv_[L - 1].g;
    const int strips = (g.n + EVOMG_STRIP_COLS - 1) / EVOMG_STRIP_COLS;
    int rows = strip_rows_ > 0 ? strip_rows_ : 32;
    while (rows > 8 && (long)strips * ((g.n + rows - 1) / rows) < 2048) rows /= 2;
    rows = (rows + 1) & ~1;
    if (strip_rows_ <= 0 && balance_rows_) rows = balanced_rows(g.n, strips, 2, 8, 96);
    const int chunks = (g.n + rows - 1) / rows;
    const int waves = strips * chunks;
    const dim3 grid((((waves + 3) / 4 + 7) / 8) * 8), block(256);
    const double rkh = (strip_bc_ && prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN) ? kh(L) : 0.0;
    if (acc) hipLaunchKernelGGL((evomg_strip_prolong<T, true>), grid, block, 0, stream_, g, gc, coarse, fine, om, rows, strips, waves, rkh);
    else hipLaunchKernelGGL((evomg_strip_prolong<T, false>), grid, block, 0, stream_, g, gc, coarse, fine, 1.0, rows, strips, waves, rkh);
  }

  // live profiling (evomg_profile_begin/end): HIP events around every launch that executes an op of the selected
  // (opcode, level), fused launches included; bytes = algorithmic bytes of what that launch does
  bool prof_matches(const evomg_op_t& op) const { return op.opcode == prof_opcode_ && op.level == prof_level_; }
  int prof_open() {
    while (prof_events_.size() < prof_used_ + 2) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      prof_events_.push_back(e);
    }
    HIP_TRY(hipEventRecord(prof_events_[prof_used_], stream_));
    return EVOMG_OK;
  }
  int prof_close(double bytes) {
    HIP_TRY(hipEventRecord(prof_events_[prof_used_ + 1], stream_));
    prof_used_ += 2;
    prof_bytes_ += bytes;
    return EVOMG_OK;
  }
  int exec(const evomg_op_t& op) {
    const bool timed = prof_matches(op);
    if (timed) {
      const int rc = prof_open();
      if (rc != EVOMG_OK) return rc;
    }
    const int rc = exec_impl(op);
    if (timed) {
      const int rc2 = prof_close(algorithmic_bytes(op));
      if (rc2 != EVOMG_OK) return rc2;
    }
    return rc;
  }

  // `with jacobi` slot handling after an out-of-place sweep x -> y (EVOMG_JACOBI_* in include/evomg.h): the sweep of
  // a Single-partitioned smoother wrote the next slot; under EVOMG_JACOBI_EXASTENCILS_V11 that slot never becomes the
  // active one (the reference's pipeline never advances it: notebooks/tutorial.ipynb:3373 is reproduced only this
  // way), so the launch has happened — and cost its time, as in the reference — but the field keeps x.
  bool jacobi_lost(const evomg_op_t& sm) const {
    return prob_.jacobi_semantics == EVOMG_JACOBI_EXASTENCILS_V11 && sm.partitioning == EVOMG_PART_SINGLE;
  }
  void advance(Level<T>& lv, int L, const evomg_op_t& sm, T* x, T* y) {
    if (jacobi_lost(sm)) {
      lv.alt_mirror = x;             // the spare array now carries x's boundary values
      return;
    }
    lv.alt = x;
    set_fieldptr(L, sm.dst, y);
    lv.alt_mirror = y;               // the old array (now the spare one) has the same boundary values
  }

  int exec_impl(const evomg_op_t& op) {
    const int L = op.level;
    Level<T>& lv = lv_[L];
    const Geo& g = lv.g;
    const dim3 blk = node_block(), grd = node_grid(g);
    switch (op.opcode) {
      case EVOMG_OP_SMOOTH: {
        const double om = op.omega, omm = 1.0 - op.omega;
        T* x = fieldptr(L, op.dst);
        const T* f = fieldptr(L, op.src1);
        const int cells = op.block[0] * op.block[1] * (dim_ == 3 ? op.block[2] : 1);
        if (cells == 2 && dim_ == 3 && op.block[2] == 1 && block_fast_ && use_tile3d_blocks(g) && tile_two_cell(lv, x, f, op)) {
          advance(lv, L, op, x, lv.alt);
        } else if (cells > 1) {
          T* y = lv.alt;
          const int bx = op.block[0], by = op.block[1], bz = dim_ == 3 ? op.block[2] : 1;
          const int inner = g.n - 2;
          const int tx = (inner + bx - 1) / bx, ty = (inner + by - 1) / by, tz = dim_ == 3 ? (inner + bz - 1) / bz : 1;
          const dim3 tgrid((tx + 63) / 64, (ty + 3) / 4, dim_ == 3 ? tz : 1);
          const bool mir = lv.alt_mirror == x;
          if (dim_ == 2) block_smooth<2>(cells, tgrid, g, lv.st, x, f, y, om, omm, bx, by, bz, tx, ty, tz, mir);
          else block_smooth<3>(cells, tgrid, g, lv.st, x, f, y, om, omm, bx, by, bz, tx, ty, tz, mir);
          apply_bc(L, y);
          advance(lv, L, op, x, y);
        } else if (use_strip(g)) {
          T* y = lv.alt;
          if (op.partitioning == EVOMG_PART_RED_BLACK) launch_strip<STRIP_RBGS>(g, lv.st, x, f, y, om, omm);
          else launch_strip<STRIP_JACOBI>(g, lv.st, x, f, y, om, omm);
          if (!strip_bc_) apply_bc(L, y);                   // the strip kernels write the Robin columns themselves
          advance(lv, L, op, x, y);
        } else if (use_tile3d(g) && (op.partitioning != EVOMG_PART_RED_BLACK || tile3d_rbgs_)) {
          T* y = lv.alt;
          ensure_faces(lv, x, y);
          if (op.partitioning == EVOMG_PART_RED_BLACK) launch_tile3d_rbgs(g, lv.st, x, f, y, om, omm);
          else launch_tile3d<STRIP_JACOBI>(g, lv.st, x, f, y, om, omm);
          advance(lv, L, op, x, y);
        } else if (use_strip3d(g, op.partitioning == EVOMG_PART_RED_BLACK)) {
          T* y = lv.alt;
          if (op.partitioning == EVOMG_PART_RED_BLACK) launch_strip3d_rbgs(g, lv.st, x, f, y, om, omm);
          else launch_strip3d<STRIP_JACOBI>(g, lv.st, x, f, y, om, omm);
          apply_bc(L, y);
          advance(lv, L, op, x, y);
        } else if (op.partitioning == EVOMG_PART_RED_BLACK) {
          const dim3 hgrd((g.n / 2 + 1 + 63) / 64, grd.y, grd.z);
          for (int colour = 0; colour < 2; ++colour) {
            if (dim_ == 2) hipLaunchKernelGGL((evomg_rbgs_colour<T, 2>), hgrd, blk, 0, stream_, g, lv.st, x, f, om, omm, colour);
            else hipLaunchKernelGGL((evomg_rbgs_colour<T, 3>), hgrd, blk, 0, stream_, g, lv.st, x, f, om, omm, colour);
            apply_bc(L, x);
          }
        } else {
          T* y = lv.alt;
          if (dim_ == 2) hipLaunchKernelGGL((evomg_jacobi<T, 2>), grd, blk, 0, stream_, g, lv.st, x, f, y, om, omm);
          else hipLaunchKernelGGL((evomg_jacobi<T, 3>), grd, blk, 0, stream_, g, lv.st, x, f, y, om, omm);
          apply_bc(L, y);
          advance(lv, L, op, x, y);
        }
        break;
      }
      case EVOMG_OP_RESIDUAL:
        residual(L, lv.st, fieldptr(L, op.src0), fieldptr(L, op.src1), fieldptr(L, op.dst), /*bc_out=*/true);
        if (!(strip_bc_ && use_strip(g))) apply_bc(L, fieldptr(L, op.dst));
        break;
      case EVOMG_OP_RICHARDSON: {
        T* x = fieldptr(L, op.dst);
        T* y = lv.alt;
        if (dim_ == 2) hipLaunchKernelGGL((evomg_richardson<T, 2>), grd, blk, 0, stream_, g, lv.st, x, fieldptr(L, op.src1), y, op.omega);
        else hipLaunchKernelGGL((evomg_richardson<T, 3>), grd, blk, 0, stream_, g, lv.st, x, fieldptr(L, op.src1), y, op.omega);
        apply_bc(L, y);
        lv.alt = x;
        set_fieldptr(L, op.dst, y);
        lv.alt_mirror = y;             // the old array (now the spare one) has the same boundary values
        break;
      }
      case EVOMG_OP_RESTRICT: {
        const Geo& gf = lv_[L + 1].g;
        if (dim_ == 2) hipLaunchKernelGGL((evomg_restrict<T, 2>), grd, blk, 0, stream_, g, gf, fieldptr(L + 1, op.src0), fieldptr(L, op.dst));
        else hipLaunchKernelGGL((evomg_restrict<T, 3>), grd, blk, 0, stream_, g, gf, fieldptr(L + 1, op.src0), fieldptr(L, op.dst));
        apply_bc(L, fieldptr(L, op.dst));
        break;
      }
      case EVOMG_OP_PROLONG_ADD:
      case EVOMG_OP_PROLONG_SET: {
        const Geo& gc = lv_[L - 1].g;
        const T* cz = fieldptr(L - 1, op.src0);
        T* fz = fieldptr(L, op.dst);
        if (use_strip(g) && cz != fz) {
          launch_strip_prolong(L, cz, fz, op.opcode == EVOMG_OP_PROLONG_ADD, op.omega);
          if (strip_bc_) break;                             // Robin columns written by the kernel
        } else if (dim_ == 3 && strip_enabled_) {
          const int cells = gc.n - 1;
          const dim3 cgrid((cells + 63) / 64, (cells + 3) / 4, cells);
          if (op.opcode == EVOMG_OP_PROLONG_ADD) hipLaunchKernelGGL((evomg_prolong_cell3d<T, true>), cgrid, blk, 0, stream_, g, gc, cz, fz, op.omega);
          else hipLaunchKernelGGL((evomg_prolong_cell3d<T, false>), cgrid, blk, 0, stream_, g, gc, cz, fz, 1.0);
        } else if (op.opcode == EVOMG_OP_PROLONG_ADD) {
          if (dim_ == 2) hipLaunchKernelGGL((evomg_prolong<T, 2, true>), grd, blk, 0, stream_, g, gc, cz, fz, op.omega);
          else hipLaunchKernelGGL((evomg_prolong<T, 3, true>), grd, blk, 0, stream_, g, gc, cz, fz, op.omega);
        } else {
          if (dim_ == 2) hipLaunchKernelGGL((evomg_prolong<T, 2, false>), grd, blk, 0, stream_, g, gc, cz, fz, 1.0);
          else hipLaunchKernelGGL((evomg_prolong<T, 3, false>), grd, blk, 0, stream_, g, gc, cz, fz, 1.0);
        }
        apply_bc(L, fz);
        break;
      }
      case EVOMG_OP_SET_ZERO: {
        T* x = fieldptr(L, op.dst);
        // Robin of a zeroed interior = zero side columns: one launch instead of two
        const int side = (scalar_traits<T>::is_complex && prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN) ? 1 : 0;
        if (dim_ == 2) hipLaunchKernelGGL((evomg_zero_inner<T, 2>), grd, blk, 0, stream_, g, x, side);
        else hipLaunchKernelGGL((evomg_zero_inner<T, 3>), grd, blk, 0, stream_, g, x, 0);
        if (!side) apply_bc(L, x);
        break;
      }
      case EVOMG_OP_COPY: {
        T* d = fieldptr(L, op.dst);
        const T* s = fieldptr(L, op.src0);
        if (d != s) HIP_TRY(hipMemcpyAsync(d, s, lv.elems * sizeof(T), hipMemcpyDeviceToDevice, stream_));
        bnd_zero_[L][canon(L, op.dst)] = bnd_zero_[L][canon(L, op.src0)];
        lv.alt_mirror = nullptr;
        break;
      }
      case EVOMG_OP_COARSE_SOLVE: return coarse_solve();
      default: return invalid("exec: unknown opcode");
    }
    HIP_TRY(hipGetLastError());
    return EVOMG_OK;
  }

  // block-Jacobi sweep = boundary faces copied to y + full tiles (fast kernel) + the layers of tiles cut by the far
  // boundary (generic kernel, pinned to the last tile layer of each truncated dimension)
  template <int DIM> void block_smooth(int cells, dim3 tgrid, const Geo& g, const Stencil<T>& st, const T* x, const T* f, T* y,
                                       double om, double omm, int bx, int by, int bz, int tx, int ty, int tz, bool y_mirrors_x) {
    const dim3 blk = node_block();
    // the block kernels write inner nodes only; y needs x's boundary values — unless y is the array this field lived
    // in before its last out-of-place sweep (ping-pong), whose boundary is the same by construction
    if (!(mirror_enabled_ && prob_.bc_kind == EVOMG_BC_DIRICHLET && y_mirrors_x)) {
      const dim3 fgrid((g.n + 63) / 64, DIM == 3 ? (g.n + 3) / 4 : 1, 2 * DIM);
      hipLaunchKernelGGL((evomg_copy_faces<T, DIM>), fgrid, blk, 0, stream_, g, x, y);
    }
    if (!block_fast_) {
      hipLaunchKernelGGL((evomg_block_jacobi<T, DIM>), tgrid, blk, 0, stream_, g, st, x, f, y, om, omm, bx, by, bz, tx, ty, tz, 0, -1);
      return;
    }
    const int inner = g.n - 2;
    const int b[3] = {bx, by, bz};
    const int t[3] = {tx, ty, tz};
    bool strip_blocks = false;
    if constexpr (DIM == 2) {
      // (2,1), (1,2), (2,2) tiles on the strip layout (evomg_strip_block.h), cut tiles included
      if (strip_block_ && use_strip(g) && bx <= 2 && by <= 2 &&
          !make_block_plan<T, 2>(st, 1, 2, 1).skip[1][0] && !make_block_plan<T, 2>(st, 2, 1, 1).skip[1][0]) {
        launch_strip_block(g, st, x, f, y, om, omm, bx, by);
        strip_blocks = true;
      }
    }
    if (strip_blocks) return;                      // the strip kernel's extra waves solve the cut tiles
    launch_block_fast<DIM>(cells, tgrid, blk, g, st, x, f, y, om, omm, bx, by, bz, tx, ty, tz);
    // two-cell shapes: the fast kernel relaxes the single leftover node of a cut tile itself
    if (bx * by * (DIM == 3 ? bz : 1) == 2) return;
    for (int d = 0; d < DIM; ++d) {
      if (inner % b[d] == 0) continue;                 // no tile is cut in this dimension
      const int ta = (d == 0) ? t[1] : t[0];            // the two free tile coordinates
      const int tb = (DIM == 3) ? ((d == 2) ? t[1] : t[2]) : 1;
      if (d == 2) {
        hipLaunchKernelGGL((evomg_block_jacobi<T, DIM>), dim3((tx + 63) / 64, (ty + 3) / 4, 1), blk, 0, stream_, g, st, x, f, y, om, omm, bx, by, bz, tx, ty, tz, 1, 2);
      } else {
        hipLaunchKernelGGL((evomg_block_jacobi<T, DIM>), dim3((ta + 63) / 64, (tb + 3) / 4, 1), blk, 0, stream_, g, st, x, f, y, om, omm, bx, by, bz, tx, ty, tz, 1, d);
      }
    }
  }

  void launch_strip_block(const Geo& g, const Stencil<T>& st, const T* x, const T* f, T* y, double om, double omm, int bx, int by) {
    {
      const BlockPlan<T> pl = make_block_plan<T, 2>(st, bx, by, 1);
      const int strips = (g.n + EVOMG_STRIP_COLS - 1) / EVOMG_STRIP_COLS;
      int rows = strip_rows_ > 0 ? ((strip_rows_ + 1) & ~1) : 64;
      if (strip_rows_ <= 0) {
        while (rows > 8 && (long)strips * ((g.n + rows - 1) / rows) < 2048) rows /= 2;
        if (balance_rows_) rows = balanced_rows(g.n, strips, 2, 8, 96);
      }
      const int chunks = (g.n - 2 + rows - 1) / rows;
      const int waves = strips * chunks;
      // two-cell shapes: extra waves behind the strip waves relax the partner-less last column / row, 64 nodes each
      // (2,2): one thread per cut tile of the last column and row + the corner node
      const int extra = (bx * by == 2) ? (g.n - 2 + 63) / 64 : (2 * ((g.n - 2) / 2) + 1 + 63) / 64;
      const BlockPlan<T> pc = make_block_plan<T, 2>(st, 1, 2, 1), pr = make_block_plan<T, 2>(st, 2, 1, 1);
      const TwoCell2D<T> cut_col{pc.fac[1][0], pc.upper[0][1], pc.pivot[0], pc.pivot[1]}, cut_row{pr.fac[1][0], pr.upper[0][1], pr.pivot[0], pr.pivot[1]};
      const dim3 grid((((waves + extra + 3) / 4 + 7) / 8) * 8), block(256);
      const bool nt = strip_streams(g);
#define EVOMG_SB(BX, BY)                                                                                                              \
      if (bx == BX && by == BY) {                                                                                                      \
        if (nt) hipLaunchKernelGGL((evomg_strip2d_block<T, BX, BY, true>), grid, block, 0, stream_, g, st, pl, x, f, y, om, omm, rows, strips, waves, cut_col, cut_row);  \
        else hipLaunchKernelGGL((evomg_strip2d_block<T, BX, BY, false>), grid, block, 0, stream_, g, st, pl, x, f, y, om, omm, rows, strips, waves, cut_col, cut_row);    \
      }
      EVOMG_SB(2, 1)
      EVOMG_SB(1, 2)
      EVOMG_SB(2, 2)
#undef EVOMG_SB
    }
  }
  // full tiles: replay of the host-side tile factorisation, one instantiation per number of cells
  template <int DIM> void launch_block_fast(int cells, dim3 grid, dim3 block, const Geo& g, const Stencil<T>& st, const T* x,
                                            const T* f, T* y, double om, double omm, int bx, int by, int bz, int tx, int ty, int tz) {
    const BlockPlan<T> pl = make_block_plan<T, DIM>(st, bx, by, bz);
    // the shapes with one or two cells per dimension get their own instantiation (compile-time shape)
#define EVOMG_BJ_STATIC(BX, BY, BZ)                                                                                        \
    if (bx == BX && by == BY && (DIM == 2 || bz == BZ)) {                                                                  \
      hipLaunchKernelGGL((evomg_block_jacobi_fast<T, DIM, BX * BY * BZ, BX, BY, BZ>), grid, block, 0, stream_, g, st, pl,  \
                         x, f, y, om, omm, tx, ty, tz);                                                                    \
      return;                                                                                                              \
    }
    EVOMG_BJ_STATIC(2, 1, 1)
    EVOMG_BJ_STATIC(1, 2, 1)
    EVOMG_BJ_STATIC(2, 2, 1)
    if (DIM == 3) {
      EVOMG_BJ_STATIC(1, 1, 2)
      EVOMG_BJ_STATIC(2, 1, 2)
      EVOMG_BJ_STATIC(1, 2, 2)
      EVOMG_BJ_STATIC(2, 2, 2)
    }
#undef EVOMG_BJ_STATIC
    switch (cells) {
      case 2: hipLaunchKernelGGL((evomg_block_jacobi_fast<T, DIM, 2>), grid, block, 0, stream_, g, st, pl, x, f, y, om, omm, tx, ty, tz); break;
      case 3: hipLaunchKernelGGL((evomg_block_jacobi_fast<T, DIM, 3>), grid, block, 0, stream_, g, st, pl, x, f, y, om, omm, tx, ty, tz); break;
      case 4: hipLaunchKernelGGL((evomg_block_jacobi_fast<T, DIM, 4>), grid, block, 0, stream_, g, st, pl, x, f, y, om, omm, tx, ty, tz); break;
      case 5: hipLaunchKernelGGL((evomg_block_jacobi_fast<T, DIM, 5>), grid, block, 0, stream_, g, st, pl, x, f, y, om, omm, tx, ty, tz); break;
      case 6: hipLaunchKernelGGL((evomg_block_jacobi_fast<T, DIM, 6>), grid, block, 0, stream_, g, st, pl, x, f, y, om, omm, tx, ty, tz); break;
      case 7: hipLaunchKernelGGL((evomg_block_jacobi_fast<T, DIM, 7>), grid, block, 0, stream_, g, st, pl, x, f, y, om, omm, tx, ty, tz); break;
      default: hipLaunchKernelGGL((evomg_block_jacobi_fast<T, DIM, 8>), grid, block, 0, stream_, g, st, pl, x, f, y, om, omm, tx, ty, tz); break;
    }
  }

  // ---- 2D marching-strip kernels (evomg_strip.h) ------------------------------------------------------------------
  // exact reciprocal of the centre coefficient when that is a (normal) power of two, else 0 (kernels then divide)
  static double inv_c0(const Stencil<double>& st) {
    int e = 0;
    const double m = frexp(st.c[0], &e);
    if (!(m == 0.5 || m == -0.5) || e < -900 || e > 900 || !pow2_reciprocal_) return 0.0;
    return 1.0 / st.c[0];
  }
  static double inv_c0(const Stencil<cplx>&) { return 0.0; }
  // SET_ZERO elision: set by the interpreter right before the sweep that consumes it
  int take_x_zero() { const int z = x_zero_next_ ? 1 : 0; x_zero_next_ = false; return z; }
  // SET_ZERO(L, X) immediately followed by a strip point smoother on X whose boundary is known to be zero: the zero
  // launch is skipped and the sweep reads no x at all (same arithmetic on literal zeros => same bits)
  bool elidable_zero(const evomg_op_t* ops, int n_ops, int i) const {
    if (!fuse_enabled_ || !zero_elision_ || dim_ != 2 || i + 1 >= n_ops) return false;
    const evomg_op_t &z = ops[i], &sm = ops[i + 1];
    if (z.opcode != EVOMG_OP_SET_ZERO || sm.opcode != EVOMG_OP_SMOOTH || z.level != sm.level) return false;
    if (canon(z.level, z.dst) != canon(sm.level, sm.dst)) return false;
    if (!bnd_zero_[z.level][canon(z.level, z.dst)]) return false;
    if (prof_matches(z)) return false;
    return zero_input_sweep(ops, n_ops, i + 1);
  }
  // ops[i] is a sweep that can run as the zero-input variant of its strip kernel (its field is zero everywhere, boundary
  // included; the variant loads nothing of it and writes every node of the output)
  bool zero_input_sweep(const evomg_op_t* ops, int n_ops, int i) const {
    if (!fuse_enabled_ || !zero_elision_ || dim_ != 2 || i >= n_ops) return false;
    const evomg_op_t& sm = ops[i];
    if (sm.opcode != EVOMG_OP_SMOOTH || sm.block[0] * sm.block[1] != 1) return false;
    if (jacobi_lost(sm)) return false;        // the sweep's output is discarded: the field has to be zeroed for real
    if (canon(sm.level, sm.src1) == canon(sm.level, sm.dst)) return false;
    if (prof_matches(sm)) return false;
    // the zero-input variants exist for the plain sweep and the sweep + residual + restriction; a sweep that is the
    // tail of the iteration (norm) would go another way
    if (tail_norm_candidate(ops, n_ops, i)) return false;
    return use_strip(lv_[sm.level].g);
  }
  bool use_strip(const Geo& g) const { return dim_ == 2 && strip_enabled_ && g.n >= strip_min_n_; }
  // Non-temporal placement of the 2D strip kernels (template flag NT_XY): on a level whose three arrays (x, y, f) exceed
  // the 256 MiB Infinity Cache while f alone fits with room to spare (<= 192 MiB), x loads and y stores carry the hint and
  // f stays cacheable. When not even f fits (8193^2), no hint at all is fastest (profiles/r03_nt_placement.md).
  bool strip_streams(const Geo& g) const {
    const double field = (double)g.pitch * g.n * sizeof(T);
    return strip_nt_ && 3.0 * field > nt_lo_mib_ * 1024 * 1024 && field <= nt_hi_mib_ * 1024 * 1024;
  }
  // Chunk height of the 2D smoother strips. A block is four waves = four chunks and every block is resident at once, so a
  // launch lasts as long as the compute unit that got the most blocks: ceil(blocks / CUs) * (rows + halo and prologue).
  // Measured at 4097^2 (tools/sweep_rows.sh): 64 rows -> 553 blocks, three on some units and two on most, fused sweeps
  // 91.2 us, plain 72.4 us; 48 rows -> 731 blocks, three on nearly every unit: 83.5 us / 67.4 us; 40: 84.4, 56: 87.5,
  // 80: 88.2. `quantum`: rows must be a multiple of it (2: static row parity; EVOMG_NORM_ROWS: tail-norm chunks).
  // Only heights that leave at least three blocks (twelve waves) per unit compete: below that a unit cannot keep enough
  // loads in flight and the model no longer holds; if none does, the smallest height wins (as many waves as possible).
  int balanced_rows(int n, int strips, int quantum, int lo, int hi) const {
    long best_cost = -1;
    int best = lo;
    for (int r = lo; r <= hi; r += quantum) {
      const long blocks = ((long)strips * ((n + r - 1) / r) + 3) / 4;
      const long rounds = (blocks + n_cu_ - 1) / n_cu_;
      if (rounds < 3) break;                                                      // blocks only shrink from here on
      const long cost = rounds * (r + 4);
      if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best = r; }     // ties: the taller chunk (less halo)
    }
    // small grids (less than one block of four waves per compute unit even with the shortest chunks): a launch lasts as
    // long as one wave's march — a chain of dependent row steps — so the shortest chunks win (513^2 and below: 4 rows,
    // tutorial-size solve 1.87 -> 1.77 ms); the halo they re-read comes from L2
    if (best_cost < 0 && lo == 8 && quantum == 2 && ((long)strips * ((n + 7) / 8) + 3) / 4 < n_cu_) return 4;
    return best;
  }
  // point smoothers on small and medium grids: tiles of the field in LDS (evomg_small2d.h); same results as the strip form
  bool small2d_ok(const Geo& g) const { return small2d_enabled_ && dim_ == 2 && g.n <= small2d_max_n_; }
  template <int MODE, bool PRO, bool TAILR, bool XZ>
  void launch_small2d(const Geo& g, const Stencil<T>& st, const T* x, const T* f, T* y, double om, double omm, double rkh, int rout,
                      const ProlongIn<T>& pro, int nsweeps = 1) {
    auto kern = evomg_small2d<T, MODE, PRO, TAILR, XZ>;
    const size_t lds_bytes = (size_t)((MODE == STRIP_JACOBI || TAILR) ? 2 : 1) * small2d_image_elems(g.n) * sizeof(T);
    const int tiles = small2d_tiles(g.n);
    const void* key = reinterpret_cast<const void*>(kern);
    if (lds_bytes > 48 * 1024 && !small2d_attr_set_.count(key)) {
      (void)hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
      small2d_attr_set_.insert(key);
    }
    hipLaunchKernelGGL(kern, dim3(tiles * tiles), dim3(1024), lds_bytes, stream_, g, st, x, f, y, om, omm, rkh, rout, pro, nsweeps);
  }
  template <int MODE> void launch_strip(const Geo& g, const Stencil<T>& st, const T* x, const T* f, T* y, double om, double omm, bool bc_out = true) {
    const int strips = (g.n + EVOMG_STRIP_COLS - 1) / EVOMG_STRIP_COLS;
    // enough waves to fill 256 CUs several times over, but chunks tall enough to amortise the vertical halo
    // measured on MI355X at 4097^2 (tools/tune_strip.py, interleaved A/B): smoothers are fastest with tall chunks
    // (vertical halo 4/64), the residual with short ones
    int rows = strip_rows_ > 0 ? strip_rows_ : (MODE == STRIP_RESIDUAL ? 16 : 64);
    while (rows > 8 && (long)strips * ((g.n + rows - 1) / rows) < 2048) rows /= 2;
    rows = (rows + 1) & ~1;   // even chunk heights: the kernel relies on y0 being even (static row parity)
    if (strip_rows_ <= 0 && balance_rows_) rows = balanced_rows(g.n, strips, 2, 8, 96);   // residual too: 71.0 -> 65.4 us at 4097^2
    const int chunks = (g.n + rows - 1) / rows;
    const int waves = strips * chunks;
    const dim3 grid((((waves + 3) / 4 + 7) / 8) * 8), block(256);   // multiple of 8: XCD remap in the kernel
    const double robin_kh = (prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN) ? prob_.wavenumber / (double)(g.n - 1) : 0.0;
    // stream the right-hand side past the 256 MiB Infinity Cache only where x, y and f together exceed it; on
    // smaller levels all three stay cache-resident and a streaming hint would only evict f
    const bool nt = strip_streams(g);
    const ProlongIn<T> pro{nullptr, g, 0.0, nullptr, inv_c0(st), 0.0};
    const int rout = (bc_out && strip_bc_) ? 1 : 0;
    if constexpr (MODE == STRIP_JACOBI || MODE == STRIP_RBGS) {
      // small and medium grids: tiles of the field in LDS (evomg_small2d.h); tiles read their halo from x while others store y
      if (small2d_ok(g) && (x != y || small2d_tiles(g.n) == 1)) {
        if (take_x_zero()) launch_small2d<MODE, false, false, true>(g, st, x, f, y, om, omm, robin_kh, rout, pro);
        else launch_small2d<MODE, false, false, false>(g, st, x, f, y, om, omm, robin_kh, rout, pro);
        return;
      }
      if (take_x_zero()) {      // SET_ZERO elided: zero-input variant (small levels: never streams)
        hipLaunchKernelGGL((evomg_strip2d<T, MODE, 2, false, false, false, false, true>), grid, block, 0, stream_, g, st, x, f, y, om, omm, rows, strips, waves, robin_kh, (T*)nullptr, rout, pro);
        return;
      }
    }
    if (nt) hipLaunchKernelGGL((evomg_strip2d<T, MODE, 2, true>), grid, block, 0, stream_, g, st, x, f, y, om, omm, rows, strips, waves, robin_kh, (T*)nullptr, rout, pro);
    else hipLaunchKernelGGL((evomg_strip2d<T, MODE, 2, false>), grid, block, 0, stream_, g, st, x, f, y, om, omm, rows, strips, waves, robin_kh, (T*)nullptr, rout, pro);
  }
  // PROLONG_ADD@L immediately followed by a point smoother on the same field: one sweep that forms the corrected
  // input on the fly (evomg_strip2d<.., PRO>)
  template <int MODE> void launch_strip_prolonged(const Geo& g, const Geo& gc, const Stencil<T>& st, const T* x, const T* f, T* y,
                                                   double om, double omm, const T* coarse, double om_p) {
    launch_strip_fused<MODE, true, false>(g, gc, st, x, f, y, om, omm, coarse, om_p);
  }
  // PRO: input corrected by the prolongation of `coarse`; TAIL: also emit the canonical partial sums of the squared
  // residual of the sweep's output into norm_partials_ (chunk heights are then multiples of EVOMG_NORM_ROWS)
  template <int MODE, bool PRO, bool TAIL> int launch_strip_fused(const Geo& g, const Geo& gc, const Stencil<T>& st, const T* x, const T* f,
                                                                    T* y, double om, double omm, const T* coarse, double om_p) {
    constexpr int cols = TAIL ? EVOMG_NORM_STRIP_COLS : EVOMG_STRIP_COLS;
    const int strips = (g.n + cols - 1) / cols;
    int rows = strip_rows_ > 0 ? strip_rows_ : 64;
    if (TAIL && rows > 4 * EVOMG_NORM_ROWS) rows = 4 * EVOMG_NORM_ROWS;   // the kernel parks at most four chunk sums
    const int min_rows = TAIL ? EVOMG_NORM_ROWS : 8;
    while (rows > min_rows && (long)strips * ((g.n + rows - 1) / rows) < 2048) rows /= 2;
    rows = TAIL ? ((rows + EVOMG_NORM_ROWS - 1) / EVOMG_NORM_ROWS) * EVOMG_NORM_ROWS : ((rows + 1) & ~1);
    if (strip_rows_ <= 0 && balance_rows_) rows = TAIL ? balanced_rows(g.n, strips, EVOMG_NORM_ROWS, EVOMG_NORM_ROWS, 4 * EVOMG_NORM_ROWS)
                                                        : balanced_rows(g.n, strips, 2, 8, 96);
    const int chunks = (g.n + rows - 1) / rows;
    const int waves = strips * chunks;
    if (TAIL) {
      const int rc = ensure_norm_partials(strips * ((g.n + EVOMG_NORM_ROWS - 1) / EVOMG_NORM_ROWS));
      if (rc != EVOMG_OK) return rc;
    }
    const dim3 grid((((waves + 3) / 4 + 7) / 8) * 8), block(256);
    const ProlongIn<T> pro{coarse, gc, om_p, nullptr, inv_c0(st), 0.0};
    T* parts = TAIL ? norm_partials_ : (T*)nullptr;
    const bool robin = prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN;
    const double rkh = robin ? prob_.wavenumber / (double)(g.n - 1) : 0.0;
    const int rout = (robin && strip_bc_) ? 1 : 0;
    if constexpr (PRO && !TAIL) {
      if (small2d_ok(g) && (x != y || small2d_tiles(g.n) == 1)) {   // tiles read their halo from x while others store y
        launch_small2d<MODE, true, false, false>(g, st, x, f, y, om, omm, rkh, rout, pro);
        return EVOMG_OK;
      }
    }
    if (strip_streams(g)) hipLaunchKernelGGL((evomg_strip2d<T, MODE, EVOMG_FUSED_PF, true, PRO, TAIL>), grid, block, 0, stream_, g, st, x, f, y, om, omm, rows, strips, waves, rkh, parts, rout, pro);
    else hipLaunchKernelGGL((evomg_strip2d<T, MODE, EVOMG_FUSED_PF, false, PRO, TAIL>), grid, block, 0, stream_, g, st, x, f, y, om, omm, rows, strips, waves, rkh, parts, rout, pro);
    return EVOMG_OK;
  }
  // SMOOTH@L + RESIDUAL@L + RESTRICT@(L-1) with a residual nobody else reads: one sweep (evomg_strip2d<.., TAIL_RESTRICT>)
  template <int MODE> void launch_strip_restrict_tail(const Geo& g, const Geo& gc, const Stencil<T>& st, const T* x, const T* f, T* y,
                                                       double om, double omm, T* coarse_out, int nsweeps = 1) {
    const int strips = (g.n + EVOMG_NORM_STRIP_COLS - 1) / EVOMG_NORM_STRIP_COLS;     // two halo lanes per side
    int rows = strip_rows_ > 0 ? strip_rows_ : 64;
    while (rows > 8 && (long)strips * ((g.n + rows - 1) / rows) < 2048) rows /= 2;
    rows = (rows + 1) & ~1;
    if (strip_rows_ <= 0 && balance_rows_) rows = balanced_rows(g.n, strips, 2, 8, 96);
    const int chunks = (g.n + rows - 1) / rows;
    const int waves = strips * chunks;
    const dim3 grid((((waves + 3) / 4 + 7) / 8) * 8), block(256);
    const bool robin = prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN;
    const double rkh = robin ? prob_.wavenumber / (double)(g.n - 1) : 0.0;
    const double rkhc = robin ? prob_.wavenumber / (double)(gc.n - 1) : 0.0;
    const int rout = (robin && strip_bc_) ? 1 : 0;
    const ProlongIn<T> pro{nullptr, gc, 0.0, coarse_out, inv_c0(st), rkhc};
    if (small2d_ok(g) && (x != y || small2d_tiles(g.n) == 1)) {   // tiles read their halo from x while others store y
      if (take_x_zero()) launch_small2d<MODE, false, true, true>(g, st, x, f, y, om, omm, rkh, rout, pro, nsweeps);
      else launch_small2d<MODE, false, true, false>(g, st, x, f, y, om, omm, rkh, rout, pro, nsweeps);
      return;
    }
    if (take_x_zero()) {
      hipLaunchKernelGGL((evomg_strip2d<T, MODE, EVOMG_FUSED_PF, false, false, false, true, true>), grid, block, 0, stream_, g, st, x, f, y, om, omm, rows, strips, waves, rkh, (T*)nullptr, rout, pro);
      return;
    }
    if (strip_streams(g)) hipLaunchKernelGGL((evomg_strip2d<T, MODE, EVOMG_FUSED_PF, true, false, false, true>), grid, block, 0, stream_, g, st, x, f, y, om, omm, rows, strips, waves, rkh, (T*)nullptr, rout, pro);
    else hipLaunchKernelGGL((evomg_strip2d<T, MODE, EVOMG_FUSED_PF, false, false, false, true>), grid, block, 0, stream_, g, st, x, f, y, om, omm, rows, strips, waves, rkh, (T*)nullptr, rout, pro);
  }
  bool fusable_two_sweeps_residual_restrict(const evomg_op_t* ops, int n_ops, int i, bool cyclic) const {
    if (i + 3 >= n_ops || !fusable_smooth_residual_restrict(ops, n_ops, i + 1, cyclic)) return false;
    const evomg_op_t &a = ops[i], &b = ops[i + 1];
    if (a.opcode != EVOMG_OP_SMOOTH || a.partitioning != EVOMG_PART_RED_BLACK || b.partitioning != EVOMG_PART_RED_BLACK) return false;
    if (a.level != b.level || a.omega != b.omega || a.block[0] * a.block[1] != 1) return false;
    if (canon(a.level, a.dst) != canon(b.level, b.dst) || canon(a.level, a.src1) != canon(b.level, b.src1)) return false;
    if (prof_matches(a) || prof_matches(b) || prof_matches(ops[i + 2]) || prof_matches(ops[i + 3])) return false;
    const Geo& g = lv_[a.level].g;
    return small2d_ok(g) && small2d_tiles(g.n) == 1 && use_strip(g);
  }
  bool fusable_smooth_residual_restrict(const evomg_op_t* ops, int n_ops, int i, bool cyclic) const {
    if (!fuse_enabled_ || !tail_restrict_enabled_ || !cyclic || i + 2 >= n_ops || dim_ != 2) return false;
    if (prob_.bc_kind != EVOMG_BC_DIRICHLET && !(prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN && strip_bc_)) return false;
    const evomg_op_t &sm = ops[i], &a = ops[i + 1];
    if (sm.opcode != EVOMG_OP_SMOOTH || sm.block[0] * sm.block[1] != 1 || jacobi_lost(sm)) return false;
    if (!fusable_residual_restrict(ops, n_ops, i + 1) || !residual_dead_after(ops, n_ops, i + 1)) return false;
    if (a.level != sm.level) return false;
    // the residual is taken of what the sweep produces, with the sweep's right-hand side
    if (canon(a.level, a.src0) != canon(sm.level, sm.dst) || canon(a.level, a.src1) != canon(sm.level, sm.src1)) return false;
    if (canon(sm.level, sm.src1) == canon(sm.level, sm.dst)) return false;
    return use_strip(lv_[sm.level].g);
  }
  int ensure_norm_partials(size_t count) {
    if (count > norm_partials_cap_) {
      if (norm_partials_) (void)hipFree(norm_partials_);
      norm_partials_ = nullptr;
      HIP_TRY(hipMalloc(&norm_partials_, sizeof(T) * count));
      norm_partials_cap_ = count;
    }
    return EVOMG_OK;
  }
  // the solver loop's residual field is dead when the cycle overwrites RES@max before reading it
  bool loop_residual_is_dead(const evomg_op_t* ops, int n_ops) const {
    for (int k = 0; k < n_ops; ++k) {
      bool reads;
      if (touches(ops[k], prob_.max_level, EVOMG_F_RES, &reads)) return !reads;
    }
    return true;
  }
  // Is the op at index i (a point smoother on SOL@max with RHS@max, strip-capable, real Dirichlet problem) the last
  // thing a solver-loop iteration does, so that the residual norm can ride on it?
  bool tail_norm_candidate(const evomg_op_t* ops, int n_ops, int i) const {
    if (!tail_norm_enabled_ || scalar_traits<T>::is_complex || dim_ != 2 || prob_.bc_kind != EVOMG_BC_DIRICHLET) return false;
    if (i != n_ops - 1) return false;
    const evomg_op_t& b = ops[i];
    if (b.opcode != EVOMG_OP_SMOOTH || b.level != prob_.max_level || b.block[0] * b.block[1] != 1 || jacobi_lost(b)) return false;
    if (canon(b.level, b.dst) != EVOMG_F_SOL || canon(b.level, b.src1) != EVOMG_F_RHS) return false;
    const Geo& g = lv_[b.level].g;
    return use_strip(g) && norm_fused(g) && loop_residual_is_dead(ops, n_ops);
  }
  bool fusable_prolong_smooth(const evomg_op_t* ops, int n_ops, int i) const {
    if (!fuse_enabled_ || i + 1 >= n_ops || dim_ != 2) return false;
    // Robin problems: only when the strip kernels own the boundary columns (strip_bc_)
    if (prob_.bc_kind != EVOMG_BC_DIRICHLET && !(prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN && strip_bc_)) return false;
    const evomg_op_t &a = ops[i], &b = ops[i + 1];
    if (a.opcode != EVOMG_OP_PROLONG_ADD || b.opcode != EVOMG_OP_SMOOTH || a.level != b.level || jacobi_lost(b)) return false;
    if (canon(a.level, a.dst) != canon(b.level, b.dst)) return false;
    if (b.block[0] * b.block[1] != 1) return false;                        // point smoothers only
    if (canon(b.level, b.src1) == canon(b.level, b.dst)) return false;
    return use_strip(lv_[a.level].g);
  }
  // Fused residual + squared norm (the solver loop's residual check): the residual field is still written, the norm
  // costs no second pass. The summation order — per (strip, 16-row chunk) wave, lane = column pair, rows ascending —
  // is canonical for 2D levels of at least 33 nodes (the oracle's FN(norm_strip2d)) and does NOT depend on
  // EVOMG_NO_STRIP or the tuning knobs.
  bool norm_fused(const Geo& g) const { return dim_ == 2 && g.n >= 33; }
  // 3D from 129^3 up: residual + norm in ONE tile pass (evomg_tile3d<.., TILE_RESNORM>): canonical order of the solver
  // loop's norm there, whatever kernels the ops themselves run on (oracle: FN(norm_tile3d))
  bool norm_fused3d(const Geo& g) const { return dim_ == 3 && !scalar_traits<T>::is_complex && g.n >= 129 && prob_.bc_kind == EVOMG_BC_DIRICHLET; }
  int residual_norm_launch3d(int L, const Stencil<T>& st, const T* x, const T* f, T* r, bool write_r) {
    const Geo& g = lv_[L].g;
    const int strips = (g.n + EVOMG_STRIP_COLS - 1) / EVOMG_STRIP_COLS;
    const long count = (long)(g.n - 2) * (g.n - 2) * strips;
    {
      const int rc = ensure_norm_partials((size_t)count + kDotBlocks);
      if (rc != EVOMG_OK) return rc;
    }
    if (write_r) launch_tile3d<TILE_RESNORM_W>(g, st, x, f, r, 0.0, 0.0, TwoCellPlan<T>{}, norm_partials_);
    else launch_tile3d<TILE_RESNORM>(g, st, x, f, r, 0.0, 0.0, TwoCellPlan<T>{}, norm_partials_);
    hipLaunchKernelGGL((evomg_sum_stage1<T>), dim3(kDotBlocks), dim3(256), 0, stream_, norm_partials_, count, norm_partials_ + count);
    hipLaunchKernelGGL((evomg_dot_final<T>), dim3(1), dim3(256), 0, stream_, norm_partials_ + count, kDotBlocks, dscalar_);
    return EVOMG_OK;
  }
  int residual_norm_launch(int L, const Stencil<T>& st, const T* x, const T* f, T* r) {
    const Geo& g = lv_[L].g;
    const int strips = (g.n + EVOMG_NORM_STRIP_COLS - 1) / EVOMG_NORM_STRIP_COLS;
    const int chunks = (g.n + EVOMG_NORM_ROWS - 1) / EVOMG_NORM_ROWS;
    const int waves = strips * chunks;
    {
      const int rc = ensure_norm_partials(waves);
      if (rc != EVOMG_OK) return rc;
    }
    const dim3 grid((((waves + 3) / 4 + 7) / 8) * 8), block(256);
    if (strip_streams(g)) hipLaunchKernelGGL((evomg_strip2d<T, STRIP_RESNORM, 2, true>), grid, block, 0, stream_, g, st, x, f, r, 0.0, 0.0, EVOMG_NORM_ROWS, strips, waves, 0.0, norm_partials_, 0, ProlongIn<T>{nullptr, g, 0.0, nullptr, inv_c0(st), 0.0});
    else hipLaunchKernelGGL((evomg_strip2d<T, STRIP_RESNORM, 2, false>), grid, block, 0, stream_, g, st, x, f, r, 0.0, 0.0, EVOMG_NORM_ROWS, strips, waves, 0.0, norm_partials_, 0, ProlongIn<T>{nullptr, g, 0.0, nullptr, inv_c0(st), 0.0});
    hipLaunchKernelGGL((evomg_dot_final<T>), dim3(1), dim3(256), 0, stream_, norm_partials_, waves, dscalar_);
    return EVOMG_OK;
  }

  // ---- 3D marching tiles (evomg_strip.h) ---------------------------------------------------------------------------
  // Measured on MI355X (tools/bench_ops.py poisson3d 7|8|9): the fused two-colour RB-GS tiles beat two colour passes
  // from 129^3 up (23.6 vs 40 us, 112 vs 220 us, 1.06 vs 1.64 ms). The Jacobi / residual tiles beat the
  // thread-per-node kernels only where the grid fills the chip AND the halo rows re-read by neighbouring tiles come
  // out of the 256 MiB Infinity Cache (257^3: 82 vs 105 us); at 513^3 those re-reads cost HBM traffic (rocprofv3
  // FETCH_SIZE 1.30x) and the plain kernel wins (772 vs 895 us). EVOMG_STRIP3D_MIN_N overrides both rules (tests).
  bool use_strip3d(const Geo& g, bool red_black) const {
    if (dim_ != 3 || !strip_enabled_) return false;
    if (strip3d_min_n_ > 0) return g.n >= strip3d_min_n_;
    if (red_black) return g.n >= 129;
    return g.n >= 193 && (double)g.plane * g.n * sizeof(T) <= 256.0 * 1024 * 1024;
  }
  template <int MODE> void launch_strip3d(const Geo& g, const Stencil<T>& st, const T* x, const T* f, T* y, double om, double omm) {
    constexpr int TY = 4;
    const int strips = (g.n + EVOMG_STRIP_COLS - 1) / EVOMG_STRIP_COLS;
    const int ygroups = ((g.n + TY - 1) / TY + 3) / 4;          // a block = 4 y-adjacent tiles
    // z chunks: long enough to amortise the two halo planes, short enough for several thousand waves; multiples
    // of 4 (the kernel's plane ring is unrolled four times)
    int planes = strip_planes_ > 0 ? strip_planes_ : 64;
    while (planes > 8 && (long)strips * ygroups * ((g.n + planes - 1) / planes) < 1024) planes /= 2;
    planes = (planes + 3) & ~3;
    const int chunks = (g.n + planes - 1) / planes;
    const int blocks = strips * ygroups * chunks;
    const dim3 grid(((blocks + 7) / 8) * 8), block(256);
    const bool nt = strip_streams3d(g);
    if (nt) hipLaunchKernelGGL((evomg_strip3d<T, MODE, TY, true>), grid, block, 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks);
    else hipLaunchKernelGGL((evomg_strip3d<T, MODE, TY, false>), grid, block, 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks);
  }
  void launch_strip3d_rbgs(const Geo& g, const Stencil<T>& st, const T* x, const T* f, T* y, double om, double omm) {
    constexpr int TY = 4;
    const int strips = (g.n + EVOMG_STRIP_COLS - 1) / EVOMG_STRIP_COLS;
    const int ygroups = ((g.n + TY - 1) / TY + 3) / 4;
    int planes = strip_planes_ > 0 ? strip_planes_ : 64;       // halo: 3 planes per chunk
    while (planes > 8 && (long)strips * ygroups * ((g.n + planes - 1) / planes) < 1024) planes /= 2;
    planes = (planes + 3) & ~3;
    const int chunks = (g.n + planes - 1) / planes;
    const int blocks = strips * ygroups * chunks;
    const dim3 grid(((blocks + 7) / 8) * 8), block(256);
    if (strip_streams3d(g)) hipLaunchKernelGGL((evomg_strip3d_rbgs<T, TY, true>), grid, block, 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks);
    else hipLaunchKernelGGL((evomg_strip3d_rbgs<T, TY, false>), grid, block, 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks);
  }
  // ---- 3D tiles with LDS-shared y halo (evomg_tile3d.h): Jacobi and residual ------------------------------------------
  bool use_tile3d(const Geo& g) const {
    if (dim_ != 3 || !strip_enabled_ || !tile3d_enabled_ || scalar_traits<T>::is_complex) return false;
    // measured (profiles/r02_pmc_3d.md): from 129^3 up the tiles match or beat every other 3D kernel of this library;
    // below, a tile grid has too few workgroups and the thread-per-node kernels win on latency
    return g.n >= (strip3d_min_n_ > 0 ? strip3d_min_n_ : 129);
  }
  // the two-cell block modes pay from 257^3 (at 129^3 one thread per tile is 30 % faster)
  bool use_tile3d_blocks(const Geo& g) const { return use_tile3d(g) && g.n >= (strip3d_min_n_ > 0 ? strip3d_min_n_ : 257); }
  // block-Jacobi with (2,1,1) or (1,2,1) tiles as a tile-kernel mode; false: not applicable (caller falls back)
  bool tile_two_cell(Level<T>& lv, const T* x, const T* f, const evomg_op_t& op) {
    if constexpr (scalar_traits<T>::is_complex) return false;
    else {
      const BlockPlan<T> pl = make_block_plan<T, 3>(lv.st, op.block[0], op.block[1], 1);
      if (pl.skip[1][0]) return false;
      const TwoCellPlan<T> two{pl.fac[1][0], pl.upper[0][1], pl.pivot[0], pl.pivot[1]};
      T* y = lv.alt;
      ensure_faces(lv, x, y);
      const double om = op.omega, omm = 1.0 - op.omega;
      if (op.block[0] == 2) launch_tile3d<TILE_BLOCK211>(lv.g, lv.st, x, f, y, om, omm, two);
      else launch_tile3d<TILE_BLOCK121>(lv.g, lv.st, x, f, y, om, omm, two);
      return true;
    }
  }
  // the tiles write inner nodes only: y's boundary faces must equal x's (ping-pong twin, or one thin copy)
  void ensure_faces(const Level<T>& lv, const T* x, T* y) {
    if (mirror_enabled_ && prob_.bc_kind == EVOMG_BC_DIRICHLET && lv.alt_mirror == x) return;
    const Geo& g = lv.g;
    const dim3 fgrid((g.n + 63) / 64, (g.n + 3) / 4, 6);
    hipLaunchKernelGGL((evomg_copy_faces<T, 3>), fgrid, node_block(), 0, stream_, g, x, y);
  }
  template <int MODE> void launch_tile3d(const Geo& g, const Stencil<T>& st, const T* x, const T* f, T* y, double om, double omm,
                                         TwoCellPlan<T> two = TwoCellPlan<T>{}, T* aux = nullptr) {
    constexpr int TY = EVOMG_TILE3D_TY;
    const int inner = g.n - 2;
    const int strips = (g.n + EVOMG_STRIP_COLS - 1) / EVOMG_STRIP_COLS;
    // eight waves per workgroup (y halo 34/32) where that still leaves a few thousand workgroups, else four (18/16)
    int nw = tile3d_waves_ > 0 ? tile3d_waves_ : 8;
    if (tile3d_waves_ == 0 && (long)strips * ((inner + 8 * TY - 1) / (8 * TY)) * ((inner + 15) / 16) < 1024) nw = 4;
    const int ygroups = (inner + nw * TY - 1) / (nw * TY);
    // z halo: 2 planes per chunk. Measured at 513^3 (profiles/r02_pmc_3d.md): 16 planes (2560 workgroups) beat 32 and 64
    int planes = strip_planes_ > 0 ? strip_planes_ : 16;
    if (strip_planes_ <= 0) while (planes > 8 && (long)strips * ygroups * ((inner + planes - 1) / planes) < 2048) planes /= 2;
    planes = (planes + 3) & ~3;
    const int chunks = (inner + planes - 1) / planes;
    const int blocks = strips * ygroups * chunks;
    const dim3 grid(((blocks + 7) / 8) * 8);
    const int nt = strip_streams3d(g) ? tile3d_nt_ : 0;
    if constexpr (!scalar_traits<T>::is_complex) {
#define EVOMG_TILE3D_LAUNCH(NW, NT) hipLaunchKernelGGL((evomg_tile3d<T, MODE, NW, NT>), grid, dim3(64 * NW), 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks, two, tile3d_order_, aux)
#define EVOMG_TILE3D_NT(NW) if (nt & 1) EVOMG_TILE3D_LAUNCH(NW, 1); else EVOMG_TILE3D_LAUNCH(NW, 0);
      if (nw == 8) { EVOMG_TILE3D_NT(8) } else { EVOMG_TILE3D_NT(4) }
#undef EVOMG_TILE3D_NT
#undef EVOMG_TILE3D_LAUNCH
    }
  }
  // two identical Jacobi-type sweeps in a row on one field (evomg_tile3d_twice): same geometry as the red-black tiles
  template <int MODE> void launch_tile3d_twice(const Geo& g, const Stencil<T>& st, const T* x, const T* f, T* y, double om, double omm,
                                               TwoCellPlan<T> two, const T* coarse = nullptr, const Geo* gc = nullptr, double om_p = 0.0) {
    constexpr int TY = EVOMG_TILE3D_TY;
    constexpr int cols = (MODE == TILE_BLOCK211) ? 120 : EVOMG_STRIP_COLS;      // two halo lanes per side for the blocks
    const int strips = (g.n + cols - 1) / cols;
    const int nw = 8;
    const int stored = nw * TY - 2;
    const int ygroups = (g.n - 3 + stored - 1) / stored > 0 ? (g.n - 3 + stored - 1) / stored : 1;
    int planes = strip_planes_ > 0 ? strip_planes_ : twice_planes_;
    planes = (planes + 3) & ~3;
    const int chunks = (g.n - 2 + planes - 1) / planes;
    const int blocks = strips * ygroups * chunks;
    const dim3 grid(((blocks + 7) / 8) * 8);
    if constexpr (!scalar_traits<T>::is_complex) {
      if (coarse) hipLaunchKernelGGL((evomg_tile3d_twice<T, MODE, 8, 0, true>), grid, dim3(512), 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks, two,
                                     TwiceProlong<T>{coarse, *gc, om_p});
      else hipLaunchKernelGGL((evomg_tile3d_twice<T, MODE, 8, 0, false>), grid, dim3(512), 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks, two,
                              TwiceProlong<T>{nullptr, g, 0.0});
    }
  }
  // PROLONG_ADD@L, SMOOTH, SMOOTH: the coarse-grid correction folded into the double sweep's loads (evomg_tile3d_twice<PRO>)
  bool fusable_prolong_twice(const evomg_op_t* ops, int n_ops, int i) const {
    if (!tile3d_pro_ || i + 2 >= n_ops || ops[i].opcode != EVOMG_OP_PROLONG_ADD || !fusable_twice(ops, n_ops, i + 1)) return false;
    const evomg_op_t &pr = ops[i], &sm = ops[i + 1];
    if (pr.level != sm.level || pr.level <= prob_.min_level || prob_.bc_kind != EVOMG_BC_DIRICHLET) return false;
    if (canon(pr.level, pr.dst) != canon(sm.level, sm.dst)) return false;
    if (prof_matches(pr)) return false;
    return canon(pr.level - 1, pr.src0) >= 0 && lv_[pr.level - 1].f[canon(pr.level - 1, pr.src0)] != nullptr;
  }
  // SMOOTH, SMOOTH: the same Single-partitioned point or (2,1,1) block smoother twice in a row on the same field, on a 3D
  // level the tiles serve (>= 257^3: below, two launches are as fast)
  bool fusable_twice(const evomg_op_t* ops, int n_ops, int i) const {
    if (!fuse_enabled_ || !tile3d_twice_ || dim_ != 3 || scalar_traits<T>::is_complex || i + 1 >= n_ops) return false;
    const evomg_op_t &a = ops[i], &b = ops[i + 1];
    if (a.opcode != EVOMG_OP_SMOOTH || b.opcode != EVOMG_OP_SMOOTH || a.level != b.level) return false;
    if (a.partitioning != EVOMG_PART_SINGLE || b.partitioning != EVOMG_PART_SINGLE || a.omega != b.omega) return false;
    if (canon(a.level, a.dst) != canon(b.level, b.dst) || canon(a.level, a.src1) != canon(b.level, b.src1)) return false;
    if (canon(a.level, a.src1) == canon(a.level, a.dst)) return false;
    for (int d = 0; d < 3; ++d) if (a.block[d] != b.block[d]) return false;
    const bool point = a.block[0] == 1 && a.block[1] == 1 && a.block[2] == 1;
    const bool b211 = a.block[0] == 2 && a.block[1] == 1 && a.block[2] == 1 && block_fast_;
    if (!point && !b211) return false;
    // `with jacobi` sweeps whose slot is never advanced: each one is launched on its own and its output discarded (the
    // contract of EVOMG_JACOBI_EXASTENCILS_V11: the sweep runs and costs its time); a double sweep would also swallow a
    // prolongation folded into it, because its only output is the discarded array
    if (jacobi_lost(a) || jacobi_lost(b)) return false;
    if (prof_matches(a) || prof_matches(b)) return false;
    const Geo& g = lv_[a.level].g;
    return use_tile3d_blocks(g);
  }
  int exec_twice(const evomg_op_t& sm, const evomg_op_t* pr = nullptr) {
    const int L = sm.level;
    Level<T>& lv = lv_[L];
    T* x = fieldptr(L, sm.dst);
    const T* f = fieldptr(L, sm.src1);
    T* y = lv.alt;
    const double om = sm.omega, omm = 1.0 - sm.omega;
    ensure_faces(lv, x, y);
    if constexpr (!scalar_traits<T>::is_complex) {
      if (sm.block[0] == 2) {
        const BlockPlan<T> pl = make_block_plan<T, 3>(lv.st, 2, 1, 1);
        const T* cz = pr ? fieldptr(L - 1, pr->src0) : nullptr;
        launch_tile3d_twice<TILE_BLOCK211>(lv.g, lv.st, x, f, y, om, omm, TwoCellPlan<T>{pl.fac[1][0], pl.upper[0][1], pl.pivot[0], pl.pivot[1]},
                                           cz, &lv_[L - 1 >= prob_.min_level ? L - 1 : L].g, pr ? pr->omega : 0.0);
      } else {
        const T* cz = pr ? fieldptr(L - 1, pr->src0) : nullptr;
        launch_tile3d_twice<STRIP_JACOBI>(lv.g, lv.st, x, f, y, om, omm, TwoCellPlan<T>{}, cz, &lv_[L - 1 >= prob_.min_level ? L - 1 : L].g,
                                          pr ? pr->omega : 0.0);
      }
    }
    advance(lv, L, sm, x, y);      // two sweeps, one pass: the result lives in the spare array -> ONE swap
    HIP_TRY(hipGetLastError());
    return EVOMG_OK;
  }
  void launch_tile3d_rbgs(const Geo& g, const Stencil<T>& st, const T* x, const T* f, T* y, double om, double omm) {
    constexpr int TY = EVOMG_TILE3D_TY;
    const int strips = (g.n + EVOMG_STRIP_COLS - 1) / EVOMG_STRIP_COLS;
    int nw = tile3d_waves_ > 0 ? tile3d_waves_ : 8;
    if (tile3d_waves_ == 0 && (long)strips * ((g.n + 8 * TY - 3) / (8 * TY - 2)) * ((g.n + 15) / 16) < 1024) nw = 4;
    const int stored = nw * TY - 2;                              // rows a workgroup finishes (the top one: one more)
    const int ygroups = (g.n - 3 + stored - 1) / stored > 0 ? (g.n - 3 + stored - 1) / stored : 1;
    int planes = strip_planes_ > 0 ? strip_planes_ : 64;       // z halo: 3 planes per chunk
    if (strip_planes_ <= 0) while (planes > 8 && (long)strips * ygroups * ((g.n - 2 + planes - 1) / planes) < 2048) planes /= 2;
    planes = (planes + 3) & ~3;
    const int chunks = (g.n - 2 + planes - 1) / planes;
    const int blocks = strips * ygroups * chunks;
    const dim3 grid(((blocks + 7) / 8) * 8);
    if constexpr (!scalar_traits<T>::is_complex) {
      if (nw == 8) hipLaunchKernelGGL((evomg_tile3d_rbgs<T, 8, 0>), grid, dim3(512), 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks);
      else hipLaunchKernelGGL((evomg_tile3d_rbgs<T, 4, 0>), grid, dim3(256), 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks);
    }
  }
  // RESIDUAL@L, RESTRICT@(L-1) of that residual on a 3D level the tiles serve, the residual field dead afterwards
  // (evomg_tile3d_residual_restrict never writes it)
  bool fusable_residual_restrict3d(const evomg_op_t* ops, int n_ops, int i, bool cyclic) const {
    if (!fuse_enabled_ || !tile3d_rr_ || dim_ != 3 || scalar_traits<T>::is_complex || !cyclic || i + 1 >= n_ops) return false;
    const evomg_op_t &a = ops[i], &b = ops[i + 1];
    if (a.opcode != EVOMG_OP_RESIDUAL || b.opcode != EVOMG_OP_RESTRICT) return false;
    if (b.level != a.level - 1 || canon(a.level, b.src0) != canon(a.level, a.dst)) return false;
    if (canon(a.level, a.dst) == canon(a.level, a.src0) || canon(a.level, a.dst) == canon(a.level, a.src1)) return false;
    if (prob_.bc_kind != EVOMG_BC_DIRICHLET || !use_tile3d(lv_[a.level].g)) return false;
    if ((a.opcode == prof_opcode_ && a.level == prof_level_) || (b.opcode == prof_opcode_ && b.level == prof_level_)) return false;
    return residual_dead_after(ops, n_ops, i);
  }
  void launch_tile3d_residual_restrict(int L, const T* x, const T* f, T* coarse) {
    constexpr int TY = EVOMG_TILE3D_TY;
    const Geo& g = lv_[L].g;
    const Geo& gc = lv_[L - 1].g;
    const int strips = (g.n + EVOMG_STRIP_COLS - 1) / EVOMG_STRIP_COLS;
    const int rows_c = gc.n - 2;                                  // coarse rows / planes to produce
    int nw = tile3d_waves_ > 0 ? tile3d_waves_ : 8;
    if (tile3d_waves_ == 0 && (long)strips * ((rows_c + 14) / 15) * ((rows_c + 7) / 8) < 1024) nw = 4;
    const int per_group = (nw * TY - 2) / 2;                      // coarse rows a workgroup finishes: 15 (7 with four waves)
    const int ygroups = (rows_c + per_group - 1) / per_group;
    int cplanes = strip_planes_ > 0 ? (strip_planes_ + 1) / 2 : rr_cplanes_;   // fine planes per chunk: 2 * cplanes + 1
    const int chunks = (rows_c + cplanes - 1) / cplanes;
    const int blocks = strips * ygroups * chunks;
    const dim3 grid(((blocks + 7) / 8) * 8);
    if constexpr (!scalar_traits<T>::is_complex) {
      if (nw == 8) hipLaunchKernelGGL((evomg_tile3d_residual_restrict<T, 8, 0>), grid, dim3(512), 0, stream_, g, gc, lv_[L].st, x, f, coarse, cplanes, strips, ygroups, blocks);
      else hipLaunchKernelGGL((evomg_tile3d_residual_restrict<T, 4, 0>), grid, dim3(256), 0, stream_, g, gc, lv_[L].st, x, f, coarse, cplanes, strips, ygroups, blocks);
    }
  }
  bool strip_streams3d(const Geo& g) const { return strip_nt_ && 3.0 * (double)g.plane * g.n * sizeof(T) > 256.0 * 1024 * 1024; }

  // bc_out: the caller applies the boundary condition to r afterwards (the strip kernel then does it itself)
  void residual(int L, const Stencil<T>& st, const T* x, const T* f, T* r, bool bc_out = false) {
    const Geo& g = lv_[L].g;
    if (use_strip(g)) { launch_strip<STRIP_RESIDUAL>(g, st, x, f, r, 0.0, 0.0, bc_out); return; }
    if (use_tile3d(g)) { launch_tile3d<STRIP_RESIDUAL>(g, st, x, f, r, 0.0, 0.0); return; }
    if (use_strip3d(g, false)) { launch_strip3d<STRIP_RESIDUAL>(g, st, x, f, r, 0.0, 0.0); return; }
    if (dim_ == 2) hipLaunchKernelGGL((evomg_apply<T, 2, true>), node_grid(g), node_block(), 0, stream_, g, st, x, f, r);
    else hipLaunchKernelGGL((evomg_apply<T, 3, true>), node_grid(g), node_block(), 0, stream_, g, st, x, f, r);
  }
  void apply_operator(int L, const Stencil<T>& st, const T* x, T* y) {
    const Geo& g = lv_[L].g;
    if (dim_ == 2) hipLaunchKernelGGL((evomg_apply<T, 2, false>), node_grid(g), node_block(), 0, stream_, g, st, x, (const T*)nullptr, y);
    else hipLaunchKernelGGL((evomg_apply<T, 3, false>), node_grid(g), node_block(), 0, stream_, g, st, x, (const T*)nullptr, y);
  }
  void axpy(int L, T* z, const T* x, T a, const T* y) {
    const Geo& g = lv_[L].g;
    if (dim_ == 2) hipLaunchKernelGGL((evomg_axpy<T, 2>), node_grid(g), node_block(), 0, stream_, g, z, x, a, y);
    else hipLaunchKernelGGL((evomg_axpy<T, 3>), node_grid(g), node_block(), 0, stream_, g, z, x, a, y);
  }

  // copy_op: the COPY `gen_rhs = ...` and the SET_ZERO of the solution that precede the solve are folded into its launch
  // (fusable_copy_zero_solve): the kernel reads its right-hand side from the copy's source, writes the copy itself and
  // starts from x = 0 without reading x
  int coarse_solve(const evomg_op_t* copy_op = nullptr) {
    const int L = prob_.min_level;
    Level<T>& lv = lv_[L];
    T* x = fieldptr(L, EVOMG_F_CORR);
    const T* b = fieldptr(L, EVOMG_F_CGS_RHS);
    T* r = fieldptr(L, EVOMG_F_RES);
    if (copy_op && prob_.cgs_kind == EVOMG_CGS_BICGSTAB) {
      const T* src = fieldptr(L, copy_op->src0);
      T* cgs = fieldptr(L, EVOMG_F_CGS_RHS);
      if (prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN)
        hipLaunchKernelGGL((evomg_bicgstab_wave2d<T, true, true>), dim3(1), dim3(64), 0, stream_, lv.g, lv.st, x, src, r, prob_.cgs_max_iterations,
                           prob_.cgs_target_res_reduction, kh(L), dinfo_, cgs);
      else
        hipLaunchKernelGGL((evomg_bicgstab_wave2d<T, false, true>), dim3(1), dim3(64), 0, stream_, lv.g, lv.st, x, src, r, prob_.cgs_max_iterations,
                           prob_.cgs_target_res_reduction, 0.0, dinfo_, cgs);
      bnd_zero_[L][canon(L, copy_op->dst)] = bnd_zero_[L][canon(L, copy_op->src0)];     // the COPY's bookkeeping
      lv.alt_mirror = nullptr;
      HIP_TRY(hipGetLastError());
      return EVOMG_OK;
    }
    if (copy_op) {
      if constexpr (!scalar_traits<T>::is_complex) {
        const T* src = fieldptr(L, copy_op->src0);
        T* cgs = fieldptr(L, EVOMG_F_CGS_RHS);
        launch_cg_wave2d<true>(lv, x, src, r, cgs);
        bnd_zero_[L][canon(L, copy_op->dst)] = bnd_zero_[L][canon(L, copy_op->src0)];     // the COPY's bookkeeping
        lv.alt_mirror = nullptr;
        apply_bc(L, x);
        HIP_TRY(hipGetLastError());
        return EVOMG_OK;
      }
      return invalid("coarse_solve: folded copy on a complex problem");
    }
    T* w[kTmpVectors];
    const int need = prob_.cgs_kind == EVOMG_CGS_BICGSTAB ? 6 : 2;
    for (int i = 0; i < need; ++i) {
      int rc = tmpvec(L, i, &w[i]);
      if (rc != EVOMG_OK) return rc;
    }
    const bool real_cg = prob_.cgs_kind != EVOMG_CGS_BICGSTAB && !scalar_traits<T>::is_complex;   // ownership of the real-valued CG
    const int lanes = (dim_ == 2 ? krylov_own<2>(lv.g.n, real_cg) : krylov_own<3>(lv.g.n, real_cg)).lanes;
    const dim3 one(1), thr(lanes);
    if (prob_.cgs_kind == EVOMG_CGS_BICGSTAB && dim_ == 2 && lv.g.n <= 9 && use_reg_krylov_) {
      // up to 9^2: the whole solve in one wave, one node per lane (evomg_bicgstab_wave2d; canonical order unchanged)
      const bool robin = prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN;
      if (robin) hipLaunchKernelGGL((evomg_bicgstab_wave2d<T, true>), dim3(1), dim3(64), 0, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations,
                                    prob_.cgs_target_res_reduction, kh(L), dinfo_);
      else hipLaunchKernelGGL((evomg_bicgstab_wave2d<T, false>), dim3(1), dim3(64), 0, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations,
                              prob_.cgs_target_res_reduction, 0.0, dinfo_);
    } else if (prob_.cgs_kind == EVOMG_CGS_BICGSTAB && dim_ == 2 && kcg_enabled_ && lv.g.n >= kb_min_n_) {
      // 129^2 and up (C3's coarsest grid): six many-unit launches per iteration (evomg_krylov.h: kb_*)
      const int rc = prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN ? coarse_bicgstab_per_launch<true>(L, lv, x, b, r, w)
                                                               : coarse_bicgstab_per_launch<false>(L, lv, x, b, r, w);
      if (rc != EVOMG_OK) return rc;
    } else if (prob_.cgs_kind == EVOMG_CGS_BICGSTAB) {
      const bool robin = prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN;
      if constexpr (scalar_traits<T>::is_complex) {
        if (dim_ != 2) return invalid("complex problems are 2D only");
        if (robin) hipLaunchKernelGGL((evomg_bicgstab_single_wg<T, 2, true>), one, thr, 0, stream_, lv.g, lv.st, x, b, r, w[0], w[1], w[2], w[3], w[4], w[5], prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, kh(L), dinfo_);
        else hipLaunchKernelGGL((evomg_bicgstab_single_wg<T, 2, false>), one, thr, 0, stream_, lv.g, lv.st, x, b, r, w[0], w[1], w[2], w[3], w[4], w[5], prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, 0.0, dinfo_);
      } else {
        if (dim_ == 2) hipLaunchKernelGGL((evomg_bicgstab_single_wg<T, 2, false>), one, thr, 0, stream_, lv.g, lv.st, x, b, r, w[0], w[1], w[2], w[3], w[4], w[5], prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, 0.0, dinfo_);
        else hipLaunchKernelGGL((evomg_bicgstab_single_wg<T, 3, false>), one, thr, 0, stream_, lv.g, lv.st, x, b, r, w[0], w[1], w[2], w[3], w[4], w[5], prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, 0.0, dinfo_);
      }
    } else if (!scalar_traits<T>::is_complex && dim_ == 3 && krylov_cg_groups<3>(lv.g.n) > 1) {
      // 33^3: eight co-resident workgroups (evomg_cg_multi_wg3d); the canonical reduction order of this grid is the
      // grouped one, so there is no single-workgroup alternative
      const KrylovOwn own = krylov_own<3>(lv.g.n);
      const size_t lds_bytes = sizeof(T) * (34 + EVOMG_CG_GROUPS + (size_t)EVOMG_CG_SLAB * (own.m + 2) * (own.m + 2));
      auto kern = evomg_cg_multi_wg3d<double>;   // complex problems are 2D only
      if (!cg_multi_attr_set_) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        cg_multi_attr_set_ = true;
      }
      cg_multi_used_ = true;
      HIP_TRY(hipMemsetAsync(cg_flags_, 0, sizeof(unsigned) * 2 * EVOMG_CG_GROUPS, stream_));
      const int blocks = EVOMG_CG_GROUPS;
      if constexpr (!scalar_traits<T>::is_complex)
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds_bytes, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations,
                           prob_.cgs_target_res_reduction, dinfo_, cg_xpart_, cg_xface_, cg_flags_);
      apply_bc(L, x);
    } else if (!scalar_traits<T>::is_complex && kcg_enabled_ &&
               (dim_ == 2 ? krylov_cg_per_launch<2>(lv.g.n) : krylov_cg_per_launch<3>(lv.g.n))) {
      // a LARGE coarsest grid (257^2 / 65^3 and up): three many-unit launches per iteration (evomg_krylov.h: kcg_*)
      const int rc = (dim_ == 2) ? coarse_cg_per_launch<2>(lv, x, b, r, w[0], w[1]) : coarse_cg_per_launch<3>(lv, x, b, r, w[0], w[1]);
      if (rc != EVOMG_OK) return rc;
      apply_bc(L, x);
    } else {
      bool launched = false;
      int rc = (dim_ == 2) ? launch_cg_reg<2>(lv, x, b, r, &launched) : launch_cg_reg<3>(lv, x, b, r, &launched);
      if (rc != EVOMG_OK) return rc;
      if (!launched) {
        if (dim_ == 2) hipLaunchKernelGGL((evomg_cg_single_wg<T, 2>), one, thr, 0, stream_, lv.g, lv.st, x, b, r, w[0], w[1], prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_);
        else hipLaunchKernelGGL((evomg_cg_single_wg<T, 3>), one, thr, 0, stream_, lv.g, lv.st, x, b, r, w[0], w[1], prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_);
      }
      apply_bc(L, x);
    }
    HIP_TRY(hipGetLastError());
    return EVOMG_OK;
  }

  // CG as a sequence of launches (large coarsest grids): batches of 16 iterations are queued, the done flag is read between
  // batches; kernels of iterations past convergence return at once
  template <int DIM> int coarse_cg_per_launch(const Level<T>& lv, T* x, const T* b, T* r, T* p, T* ap) {
    if constexpr (!scalar_traits<T>::is_complex) {
      if (!kcg_state_) {
        HIP_TRY(hipMalloc(&kcg_state_, 2 * sizeof(KcgState)));
        HIP_TRY(hipHostMalloc(&kcg_host_, sizeof(KcgState)));
      }
      const Geo& g = lv.g;
      const int m = g.n - 2;
      const long rows = DIM == 2 ? m : (long)m * m;
      const int blocks = (int)(rows < kDotBlocks ? rows : kDotBlocks);
      T* part_a = partials_;
      T* part_b = partials_ + kDotBlocks;
      const dim3 grid(blocks), block(256);
      hipLaunchKernelGGL((kcg_init<T, DIM>), grid, block, 0, stream_, g, lv.st, (const T*)x, b, r, p, part_a);
      hipLaunchKernelGGL((kcg_start<T>), dim3(1), block, 0, stream_, (const T*)part_a, blocks, kcg_state_);
      const int maxit = prob_.cgs_max_iterations;
      int k = 0;
      kcg_host_->done = 0;
      kcg_host_->it = 0;
      while (k < maxit) {
        const int batch = maxit - k < 16 ? maxit - k : 16;
        for (int j = 0; j < batch; ++j, ++k) {
          const int cur = k & 1;
          hipLaunchKernelGGL((kcg_apply<T, DIM>), grid, block, 0, stream_, g, lv.st, (const T*)p, ap, part_a, (const KcgState*)kcg_state_, cur);
          hipLaunchKernelGGL((kcg_update<T, DIM>), grid, block, 0, stream_, g, x, r, (const T*)p, (const T*)ap, (const T*)part_a, blocks, part_b,
                             (const KcgState*)kcg_state_, cur);
          hipLaunchKernelGGL((kcg_direct<T, DIM>), grid, block, 0, stream_, g, (const T*)r, p, (const T*)part_b, blocks, kcg_state_, cur,
                             prob_.cgs_target_res_reduction);
        }
        HIP_TRY(hipMemcpyAsync(kcg_host_, kcg_state_ + (k & 1), sizeof(KcgState), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        if (kcg_host_->done) break;
      }
      HIP_TRY(hipMemcpyAsync(dinfo_, &kcg_state_[k & 1].it, sizeof(int), hipMemcpyDeviceToDevice, stream_));
      HIP_TRY(hipGetLastError());
      return EVOMG_OK;
    }
    return invalid("coarse_cg_per_launch: real problems only");
  }

  // BiCGStab as a sequence of launches (2D, 129^2 and up). The batches between two reads of the done flag start small
  // (C3 converges in two iterations) and double up to 16.
  template <bool ROBIN> int coarse_bicgstab_per_launch(int L, const Level<T>& lv, T* x, const T* b, T* r, T* const* w) {
    if constexpr (ROBIN && !scalar_traits<T>::is_complex) return invalid("Robin columns on a real problem");
    else {
      if (!kb_state_) {
        HIP_TRY(hipMalloc(&kb_state_, 2 * sizeof(KbState<T>)));
        HIP_TRY(hipMalloc(&kb_scal_, 4 * sizeof(T)));
        HIP_TRY(hipHostMalloc(&kb_host_, sizeof(KbState<T>)));
      }
      T *p = w[0], *nu = w[1], *hh = w[2], *s = w[3], *t = w[4], *rhat = w[5];
      const Geo& g = lv.g;
      const int m = g.n - 2;
      const int blocks = m < kDotBlocks ? m : kDotBlocks;
      T* part_r = partials_;                     // [0] r . r, [1] rhat . r
      T* part_nu = partials_ + 2 * kDotBlocks;   // rhat . nu
      T* part_t = partials_ + 3 * kDotBlocks;    // [0] t . s, [1] t . t
      const dim3 grid(blocks), block(256);
      const double khv = ROBIN ? kh(L) : 0.0;
      const double tol = prob_.cgs_target_res_reduction;
      hipLaunchKernelGGL((kb_init<T, ROBIN>), grid, block, 0, stream_, g, lv.st, (const T*)x, b, r, rhat, nu, p, part_r, khv);
      hipLaunchKernelGGL((kb_start<T>), dim3(1), block, 0, stream_, (const T*)part_r, blocks, kb_state_, kb_scal_);
      const int maxit = prob_.cgs_max_iterations;
      int k = 0, batch = 2;
      kb_host_->done = 0;
      while (k < maxit) {
        const int todo = maxit - k < batch ? maxit - k : batch;
        for (int j = 0; j < todo; ++j, ++k) {
          const int cur = k & 1;
          const KbState<T>* st = kb_state_;
          hipLaunchKernelGGL((kb_p<T, ROBIN>), grid, block, 0, stream_, g, (const T*)r, (const T*)nu, p, (const T*)(part_r + blocks), blocks, st, cur,
                             kb_scal_, khv);
          hipLaunchKernelGGL((kb_spmv<T, 1>), grid, block, 0, stream_, g, lv.st, (const T*)p, nu, (const T*)rhat, part_nu, st, cur);
          hipLaunchKernelGGL((kb_hs<T, ROBIN>), grid, block, 0, stream_, g, (const T*)x, (const T*)r, (const T*)p, (const T*)nu, hh, s,
                             (const T*)part_nu, blocks, st, cur, kb_scal_, khv);
          hipLaunchKernelGGL((kb_spmv<T, 2>), grid, block, 0, stream_, g, lv.st, (const T*)s, t, (const T*)s, part_t, st, cur);
          hipLaunchKernelGGL((kb_xr<T, ROBIN>), grid, block, 0, stream_, g, x, r, (const T*)s, (const T*)hh, (const T*)t, (const T*)rhat,
                             (const T*)part_t, blocks, part_r, st, cur, kb_scal_, khv);
          hipLaunchKernelGGL((kb_check<T>), dim3(1), block, 0, stream_, (const T*)part_r, blocks, kb_state_, cur, (const T*)kb_scal_, tol);
        }
        HIP_TRY(hipMemcpyAsync(kb_host_, kb_state_ + (k & 1), sizeof(KbState<T>), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        if (kb_host_->done) break;
        if (batch < 16) batch *= 2;
      }
      HIP_TRY(hipMemcpyAsync(dinfo_, &kb_state_[k & 1].it, sizeof(int), hipMemcpyDeviceToDevice, stream_));
      HIP_TRY(hipGetLastError());
      return EVOMG_OK;
    }
  }

  // register-resident CG when a thread's share of the coarsest grid fits its registers (2D, up to 129^2); otherwise the
  // caller falls back to the global-memory kernel
  template <typename K> int set_lds_limit(K kern, int slot, size_t lds_bytes) {
    if (!cg_lds_attr_set_[slot]) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
      cg_lds_attr_set_[slot] = true;
    }
    (void)lds_bytes;
    return EVOMG_OK;
  }
  template <int RT, int E> int launch_cg_reg2d(const Level<T>& lv, T* x, const T* b, T* r, const KrylovOwn& own, int slot, bool* launched) {
    const size_t lds_bytes = (size_t)(cg_reg2d_lds_elems(RT, E) + 32) * sizeof(T);
    // the kernel assumes one band per wave of the 512-thread workgroup and exactly RT rows per band
    if (lds_bytes > 160 * 1024 || own.lanes != 512 || own.bands != 8 || own.R != RT || E != (own.pair ? 2 : 1)) return EVOMG_OK;
    // rows every wave is guaranteed to own: all of own.R except in the last band(s)
    const int last_rows = own.m - (own.bands - 1) * own.R;
    constexpr int kMaskLast = RT > 1 ? RT - 1 : 0;
    const bool nearly_full = last_rows >= kMaskLast;
    const dim3 one(1), thr(own.lanes);
    if (nearly_full) {
      auto kern = evomg_cg_reg2d<T, RT, E, kMaskLast>;
      int rc = set_lds_limit(kern, 2 * slot, lds_bytes);
      if (rc != EVOMG_OK) return rc;
      hipLaunchKernelGGL(kern, one, thr, lds_bytes, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_);
    } else {
      auto kern = evomg_cg_reg2d<T, RT, E, 0>;
      int rc = set_lds_limit(kern, 2 * slot + 1, lds_bytes);
      if (rc != EVOMG_OK) return rc;
      hipLaunchKernelGGL(kern, one, thr, lds_bytes, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_);
    }
    *launched = true;
    return EVOMG_OK;
  }
  template <bool XZERO> void launch_cg_wave2d(const Level<T>& lv, T* x, const T* b, T* r, T* b_copy) {
    if constexpr (!scalar_traits<T>::is_complex) {
      // rows every lane is guaranteed to own: the upper half's share (33^2: 15 of 16)
      const int m = lv.g.n - 2, upper = m - EVOMG_CG_WAVE_ROWS;
      if (upper >= EVOMG_CG_WAVE_ROWS - 1)
        hipLaunchKernelGGL((evomg_cg_wave2d<T, XZERO, EVOMG_CG_WAVE_ROWS - 1>), dim3(1), dim3(64), 0, stream_, lv.g, lv.st, x, b, r, b_copy,
                           prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_);
      else
        hipLaunchKernelGGL((evomg_cg_wave2d<T, XZERO, 0>), dim3(1), dim3(64), 0, stream_, lv.g, lv.st, x, b, r, b_copy,
                           prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_);
    }
  }
  template <int DIM> int launch_cg_reg(const Level<T>& lv, T* x, const T* b, T* r, bool* launched) {
    *launched = false;
    if (!use_reg_krylov_) return EVOMG_OK;
    const KrylovOwn own = krylov_own<DIM>(lv.g.n, !scalar_traits<T>::is_complex);
    if (DIM == 2) {
      if (own.wave) {             // up to 33^2: the whole solve in one wave
        if constexpr (!scalar_traits<T>::is_complex) {
          launch_cg_wave2d<false>(lv, x, b, r, nullptr);
          *launched = true;
        }
        return EVOMG_OK;
      }
      if (!own.tile || own.lanes != 512) return EVOMG_OK;
      // 512 threads = 8 waves = 8 bands; up to 65^2: one column per lane, 129^2: a column pair per lane
      if (!own.pair) {
        if (own.R == 1) return launch_cg_reg2d<1, 1>(lv, x, b, r, own, 0, launched);
        if (own.R == 2) return launch_cg_reg2d<2, 1>(lv, x, b, r, own, 1, launched);
        if (own.R == 4) return launch_cg_reg2d<4, 1>(lv, x, b, r, own, 2, launched);
        if (own.R == 8) return launch_cg_reg2d<8, 1>(lv, x, b, r, own, 3, launched);
        return EVOMG_OK;
      }
      // complex 127^2: 48 values of 16 B per thread do not fit the register file
      if constexpr (!scalar_traits<T>::is_complex) {
        if (own.R == 16) return launch_cg_reg2d<16, 2>(lv, x, b, r, own, 4, launched);
      }
      return EVOMG_OK;
    }
    // 3D: up to 9^3 the whole solve in one wave (column ownership: nothing changes for the oracle); else
    // evomg_cg_multi_wg3d (33^3) or the global-memory kernel with column ownership
    if constexpr (!scalar_traits<T>::is_complex) {
      if (own.column && own.m <= EVOMG_CG_WAVE3D_MAX_M && own.m * own.m <= 64) {
        hipLaunchKernelGGL((evomg_cg_wave3d<T>), dim3(1), dim3(64), 0, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations,
                           prob_.cgs_target_res_reduction, dinfo_);
        *launched = true;
      }
    }
    return EVOMG_OK;
  }

  // ---- reductions ----------------------------------------------------------------------------------------------------
  void dot_async(int L, const T* a, const T* b, T* dev_out) {
    const Geo& g = lv_[L].g;
    const int m = g.n - 2;
    const long rows = dim_ == 2 ? m : (long)m * m;
    const int blocks = (int)(rows < kDotBlocks ? rows : kDotBlocks);
    if (dim_ == 2) hipLaunchKernelGGL((evomg_dot_partial<T, 2>), dim3(blocks), dim3(256), 0, stream_, g, a, b, partials_);
    else hipLaunchKernelGGL((evomg_dot_partial<T, 3>), dim3(blocks), dim3(256), 0, stream_, g, a, b, partials_);
    hipLaunchKernelGGL((evomg_dot_final<T>), dim3(1), dim3(256), 0, stream_, partials_, blocks, dev_out);
  }
  int dot_to_host(int L, const T* a, const T* b, T* out) {
    dot_async(L, a, b, dscalar_);
    HIP_TRY(hipMemcpyAsync(hscalar_, dscalar_, sizeof(T), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    *out = hscalar_[0];
    return EVOMG_OK;
  }

  int reset_fields() {
    for (int L = prob_.min_level; L <= prob_.max_level; ++L) {
      Level<T>& lv = lv_[L];
      lv.alt_mirror = nullptr;
      for (int f = 0; f < EVOMG_F_COUNT; ++f) {
        if (!lv.f[f]) continue;
        if (lv.init[f]) HIP_TRY(hipMemcpyAsync(lv.f[f], lv.init[f], lv.elems * sizeof(T), hipMemcpyDeviceToDevice, stream_));
        else {
          HIP_TRY(hipMemsetAsync(lv.f[f], 0, lv.elems * sizeof(T), stream_));
          bnd_zero_[L][f] = true;
        }
      }
    }
    HIP_TRY(hipStreamSynchronize(stream_));
    return EVOMG_OK;
  }

  // ---- outer loop A: generated MG iteration (oracle FN(solve_mg)) ----------------------------------------------------------
  // an out-of-place smoother reads its field and writes the level's spare array; only the host-side pointer swap makes
  // the result current, so launching it speculatively is free of side effects
  bool smoother_is_out_of_place(const evomg_op_t& op) const {
    if (op.opcode != EVOMG_OP_SMOOTH) return false;
    const Geo& g = lv_[op.level].g;
    const int cells = op.block[0] * op.block[1] * (dim_ == 3 ? op.block[2] : 1);
    const bool rb = op.partitioning == EVOMG_PART_RED_BLACK;
    return cells > 1 || use_strip(g) || use_strip3d(g, rb) || (rb && use_tile3d(g) && tile3d_rbgs_) || !rb;
  }
  int solve_mg(const evomg_op_t* ops, int n_ops, double wall_limit_ms, evomg_result_t* out, double* hist, int cap) {
    const int L = prob_.max_level;
    Level<T>& lv = lv_[L];
    int nh = 0;
    const double t0 = wall_ms();
    T d;
    if (!norm_event_) HIP_TRY(hipEventCreateWithFlags(&norm_event_, hipEventDisableTiming));
    bool ahead = false, res_stale = false;
    // residual + norm of the current iterate: launched, the scalar copied back asynchronously, an event recorded
    auto residual_norm_launch_all = [&]() -> int {
      int rc = EVOMG_OK;
      if (norm_fused(lv.g)) rc = residual_norm_launch(L, lv.st, fieldptr(L, EVOMG_F_SOL), fieldptr(L, EVOMG_F_RHS), fieldptr(L, EVOMG_F_RES));
      else if (norm_fused3d(lv.g)) {
        // the residual field itself is written only if the cycle reads it before overwriting it
        const bool dead = fuse_enabled_ && loop_residual_is_dead(ops, n_ops);
        if (dead) res_stale = true;
        rc = residual_norm_launch3d(L, lv.st, fieldptr(L, EVOMG_F_SOL), fieldptr(L, EVOMG_F_RHS), fieldptr(L, EVOMG_F_RES), !dead);
      } else {
        residual(L, lv.st, fieldptr(L, EVOMG_F_SOL), fieldptr(L, EVOMG_F_RHS), fieldptr(L, EVOMG_F_RES));
        dot_async(L, fieldptr(L, EVOMG_F_RES), fieldptr(L, EVOMG_F_RES), dscalar_);
      }
      if (rc != EVOMG_OK) return rc;
      HIP_TRY(hipMemcpyAsync(hscalar_, dscalar_, sizeof(T), hipMemcpyDeviceToHost, stream_));
      HIP_TRY(hipEventRecord(norm_event_, stream_));
      return EVOMG_OK;
    };
    // While the host waits for that scalar the GPU would idle. If the cycle starts with an out-of-place smoother, its
    // launch is queued behind the norm right away (speculatively): should the norm say "converged", the pointer swap
    // is simply undone and the sweep's output discarded; otherwise the cycle continues from its second op.
    T *saved_alt = nullptr, *saved_x = nullptr;
    const T* saved_mirror = nullptr;
    auto speculate = [&](bool* launched) -> int {
      *launched = false;
      if (!speculate_ || n_ops < 1 || !smoother_is_out_of_place(ops[0])) return EVOMG_OK;
      // a sweep that fuses with the residual + restriction behind it is worth more than the overlap (and the fused
      // kernel writes the coarse right-hand side, which a speculative launch must not)
      if (fusable_smooth_residual_restrict(ops, n_ops, 0, true)) return EVOMG_OK;
      if (fusable_twice(ops, n_ops, 0)) return EVOMG_OK;          // two sweeps in one pass beat the overlap
      const int l0 = ops[0].level;
      saved_alt = lv_[l0].alt;
      saved_mirror = lv_[l0].alt_mirror;
      saved_x = fieldptr(l0, ops[0].dst);
      const int rc = exec(ops[0]);
      *launched = rc == EVOMG_OK;
      return rc;
    };
    auto undo_speculation = [&]() {
      const int l0 = ops[0].level;
      lv_[l0].alt = saved_alt;
      lv_[l0].alt_mirror = saved_mirror;
      set_fieldptr(l0, ops[0].dst, saved_x);
    };
    auto wait_norm = [&]() -> int {
      HIP_TRY(hipEventSynchronize(norm_event_));
      d = hscalar_[0];
      return EVOMG_OK;
    };
    int rc = residual_norm_launch_all();
    if (rc != EVOMG_OK) return rc;
    if ((rc = speculate(&ahead)) != EVOMG_OK) return rc;
    if ((rc = wait_norm()) != EVOMG_OK) return rc;
    const double res0 = sc_norm_from_dot(d);
    if (hist && nh < cap) hist[nh] = res0;
    ++nh;
    double res = res0;
    int it = 0, status = EVOMG_SOLVE_MAXITER;
    if (res0 == 0.0) status = EVOMG_SOLVE_CONVERGED;
    else
      for (it = 1; it <= prob_.max_iterations; ++it) {
        bool norm_rode_along = false;
        rc = exec_program(ops, n_ops, /*cyclic=*/true, ahead ? 1 : 0, &norm_rode_along);
        ahead = false;
        if (rc != EVOMG_OK) return rc;
        if (norm_rode_along) {
          // the last sweep left the canonical partial sums behind: reduce them, no residual sweep
          res_stale = true;
          const Geo& g = lv.g;
          const int parts = ((g.n + EVOMG_NORM_STRIP_COLS - 1) / EVOMG_NORM_STRIP_COLS) * ((g.n + EVOMG_NORM_ROWS - 1) / EVOMG_NORM_ROWS);
          hipLaunchKernelGGL((evomg_dot_final<T>), dim3(1), dim3(256), 0, stream_, norm_partials_, parts, dscalar_);
          HIP_TRY(hipMemcpyAsync(hscalar_, dscalar_, sizeof(T), hipMemcpyDeviceToHost, stream_));
          HIP_TRY(hipEventRecord(norm_event_, stream_));
        } else if ((rc = residual_norm_launch_all()) != EVOMG_OK) return rc;
        if (it < prob_.max_iterations && (rc = speculate(&ahead)) != EVOMG_OK) return rc;
        if ((rc = wait_norm()) != EVOMG_OK) return rc;
        res = sc_norm_from_dot(d);
        if (hist && nh < cap) hist[nh] = res;
        ++nh;
        if (!(res == res) || res > 1.7e308) { status = EVOMG_SOLVE_DIVERGED; break; }
        if (res <= prob_.target_res_reduction * res0) { status = EVOMG_SOLVE_CONVERGED; break; }
        if (wall_limit_ms > 0 && wall_ms() - t0 > wall_limit_ms) { status = EVOMG_SOLVE_WALL_LIMIT; break; }
      }
    if (ahead) undo_speculation();                 // the sweep queued for an iteration that is not going to happen
    HIP_TRY(hipStreamSynchronize(stream_));
    const double t_end = wall_ms();
    if (res_stale) {
      // leave RES@max as the reference's loop would (the residual of the returned iterate); outside the solve timer
      residual(L, lv.st, fieldptr(L, EVOMG_F_SOL), fieldptr(L, EVOMG_F_RHS), fieldptr(L, EVOMG_F_RES));
      HIP_TRY(hipStreamSynchronize(stream_));
    }
    if (it > prob_.max_iterations) it = prob_.max_iterations;
    out->time_ms = t_end - t0;
    out->n_iterations = (double)it;
    out->init_res_norm = res0;
    out->final_res_norm = res;
    out->status = status;
    out->n_history = (nh < cap || !hist) ? nh : cap;
    out->conv_factor = (it > 0 && res0 > 0) ? pow(res / res0, 1.0 / (double)it) : 0.0;
    return EVOMG_OK;
  }

  // ---- outer loop B: MG-preconditioned BiCGStab (oracle FN(solve_pbicgstab); Helmholtz exa3:144-200) ------------------------
  int solve_pbicgstab(const evomg_op_t* ops, int n_ops, double wall_limit_ms, evomg_result_t* out, double* hist, int cap) {
    if (dim_ != 2) return invalid("preconditioned BiCGStab outer solver is 2D only");
    const int L = prob_.max_level;
    Level<T>& lv = lv_[L];
    const size_t bytes = lv.elems * sizeof(T);
    for (auto*& v : outer_)
      if (!v) HIP_TRY(alloc_zero(&v, lv.elems));
    T *Sol = outer_[0], *Res = outer_[1], *p = outer_[2], *ap = outer_[3], *s = outer_[4], *t = outer_[5],
      *hh = outer_[6], *rhat = outer_[7];
    for (auto* v : outer_) HIP_TRY(hipMemsetAsync(v, 0, bytes, stream_));
    const T* RHS = lv.init[EVOMG_F_RHS];
    if (!RHS) return invalid("no right-hand side set");
    int nh = 0;
    HIP_TRY(hipStreamSynchronize(stream_));
    const double t0 = wall_ms();
    if (lv.init[EVOMG_F_SOL]) HIP_TRY(hipMemcpyAsync(Sol, lv.init[EVOMG_F_SOL], bytes, hipMemcpyDeviceToDevice, stream_));
    apply_bc(L, Sol);
    residual(L, outer_st_, Sol, RHS, Res);
    T d;
    int rc = dot_to_host(L, Res, Res, &d);
    if (rc != EVOMG_OK) return rc;
    const double res0 = sc_norm_from_dot(d);
    if (hist && nh < cap) hist[nh] = res0;
    ++nh;
    double res = res0;
    int it = 0, status = EVOMG_SOLVE_MAXITER;
    const T one = scalar_traits<T>::from_real(1.0);
    T alpha = one, beta, rho, rho_new = one, omega = one, rho_next = one;
    HIP_TRY(hipMemcpyAsync(rhat, Res, bytes, hipMemcpyDeviceToDevice, stream_));
    // Fused passes (EVOMG_NO_OUTER_FUSE switches them off; same bits either way): the preconditioner's right-hand side is
    // read where it is — the cycle never writes RHS@max, checked below — instead of being copied; the two updates of p in
    // one pass; A u with the dot product(s) of its result; the last update of the residual with its norm and the next
    // iteration's rho. Per iteration: 670 MB of 2049^2 complex traffic and two of five host round trips less.
    const bool fuse = outer_fuse_;
    const bool dev_steps = fuse && outer_dev_steps_;
    bool rhs_written = false;
    for (int i = 0; i < n_ops; ++i) {
      bool reads;
      if (touches(ops[i], L, EVOMG_F_RHS, &reads)) {
        bool w = false;
        switch (ops[i].opcode) {
          case EVOMG_OP_SMOOTH: case EVOMG_OP_RESIDUAL: case EVOMG_OP_RICHARDSON: case EVOMG_OP_PROLONG_ADD: case EVOMG_OP_PROLONG_SET:
          case EVOMG_OP_SET_ZERO: case EVOMG_OP_COPY: case EVOMG_OP_RESTRICT:
            w = ops[i].level == L && canon(L, ops[i].dst) == EVOMG_F_RHS; break;
          default: break;
        }
        rhs_written = rhs_written || w;
      }
    }
    // `u = 0` in front of a cycle that starts with a point sweep on u: the sweep's zero-input variant instead of a memset
    // (not with the coarse-level batches: a batched first op would never consume the flag)
    const bool zero_by_sweep = fuse && !batch_enabled_ && n_ops > 0 && ops[0].level == L && canon(L, ops[0].dst) == EVOMG_F_SOL && zero_input_sweep(ops, n_ops, 0);
    auto precondition = [&](const T* rhs_src) -> int {   // u = 0 ; f = rhs ; gen_mgCycle()
      if (zero_by_sweep) x_zero_next_ = true;
      else HIP_TRY(hipMemsetAsync(fieldptr(L, EVOMG_F_SOL), 0, bytes, stream_));
      if (fuse && !rhs_written) {
        T* own = lv.f[EVOMG_F_RHS];
        lv.f[EVOMG_F_RHS] = const_cast<T*>(rhs_src);
        const int rc2 = exec_program(ops, n_ops, /*cyclic=*/true);
        lv.f[EVOMG_F_RHS] = own;
        return rc2;
      }
      HIP_TRY(hipMemcpyAsync(fieldptr(L, EVOMG_F_RHS), rhs_src, bytes, hipMemcpyDeviceToDevice, stream_));
      return exec_program(ops, n_ops, /*cyclic=*/true);
    };
    const Geo& g = lv.g;
    const int dot_blocks = (g.n - 2) < kDotBlocks ? (g.n - 2) : kDotBlocks;
    // results of a fused pass: `ndot` dot products to the host in one round trip
    auto fetch_dots = [&](int ndot, T* out0, T* out1) -> int {
      hipLaunchKernelGGL((evomg_dot_final2<T>), dim3(1), dim3(256), 0, stream_, partials_, dot_blocks, ndot, dscalar_);
      HIP_TRY(hipMemcpyAsync(hscalar_, dscalar_, sizeof(T) * ndot, hipMemcpyDeviceToHost, stream_));
      HIP_TRY(hipStreamSynchronize(stream_));
      *out0 = hscalar_[0];
      if (ndot > 1) *out1 = hscalar_[1];
      return EVOMG_OK;
    };
    bool have_rho = false;        // rho_new of the next iteration came out of the residual update
    if (res0 == 0.0) status = EVOMG_SOLVE_CONVERGED;
    else if (dev_steps) {
      // Every scalar of the recurrence lives on the device (dscalar_: [0] r.r, [1] rho', [2] alpha, [3] omega, [4] rho,
      // [5] beta), so the first half of the NEXT iteration — beta, the update of p, the preconditioner on p — is queued before
      // the host waits for this iteration's norm: the round trip hides behind it. Speculative only in what it touches: p and
      // the multigrid fields, none of which the result (Sol, Res) reads once the loop has stopped. The host still decides,
      // on the same norm with the same arithmetic.
      if ((rc = dot_to_host(L, rhat, Res, &rho_new)) != EVOMG_OK) return rc;
      // staging: [0] r.r (unused), [1] rho' = rhat . r, [2] alpha = 1, [3] omega = 1, [4] rho = 1
      const T zero_t = scalar_traits<T>::zero();
      T init[5] = {zero_t, rho_new, one, one, one};
      HIP_TRY(hipMemcpyAsync(dscalar_, init, sizeof init, hipMemcpyHostToDevice, stream_));
      HIP_TRY(hipStreamSynchronize(stream_));         // `init` is a stack array
      if (!outer_event_) HIP_TRY(hipEventCreateWithFlags(&outer_event_, hipEventDisableTiming));
      auto front = [&]() -> int {                     // beta ; p = Res + beta (p - omega ap) ; u = M^-1 p
        hipLaunchKernelGGL((evomg_bicg_beta<T>), dim3(1), dim3(64), 0, stream_, dscalar_);
        hipLaunchKernelGGL((evomg_axpy2_dev<T>), node_grid(g), node_block(), 0, stream_, g, p, (const T*)p, (const T*)ap, (const T*)Res, (const T*)dscalar_);
        return precondition(p);
      };
      if ((rc = front()) != EVOMG_OK) return rc;
      for (it = 1; it <= prob_.max_iterations; ++it) {
        T* u = fieldptr(L, EVOMG_F_SOL);
        hipLaunchKernelGGL((evomg_apply_dots<T, 1>), dim3(dot_blocks), dim3(256), 0, stream_, g, outer_st_, (const T*)u, ap, (const T*)rhat, partials_);
        hipLaunchKernelGGL((evomg_dot_final_quot<T>), dim3(1), dim3(256), 0, stream_, (const T*)partials_, dot_blocks, 1, one, dscalar_, 2, 4);
        hipLaunchKernelGGL((evomg_axpy_pair_dev<T>), node_grid(g), node_block(), 0, stream_, g, hh, (const T*)Sol, (const T*)u, s, (const T*)Res, (const T*)ap,
                           (const T*)(dscalar_ + 2));
        if ((rc = precondition(s)) != EVOMG_OK) return rc;
        u = fieldptr(L, EVOMG_F_SOL);
        hipLaunchKernelGGL((evomg_apply_dots<T, 2>), dim3(dot_blocks), dim3(256), 0, stream_, g, outer_st_, (const T*)u, t, (const T*)s, partials_);
        hipLaunchKernelGGL((evomg_dot_final_quot<T>), dim3(1), dim3(256), 0, stream_, (const T*)partials_, dot_blocks, 2, one, dscalar_, 3, -1);
        hipLaunchKernelGGL((evomg_axpy_dev<T>), node_grid(g), node_block(), 0, stream_, g, Sol, (const T*)hh, (const T*)(dscalar_ + 3), (const T*)u);
        apply_bc(L, Sol);
        hipLaunchKernelGGL((evomg_axpy_dots_dev<T>), dim3(dot_blocks), dim3(256), 0, stream_, g, Res, (const T*)s, (const T*)(dscalar_ + 3), (const T*)t, (const T*)rhat, partials_);
        hipLaunchKernelGGL((evomg_dot_final2<T>), dim3(1), dim3(256), 0, stream_, partials_, dot_blocks, 2, dscalar_);
        HIP_TRY(hipMemcpyAsync(hscalar_, dscalar_, sizeof(T), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipEventRecord(outer_event_, stream_));
        if (it < prob_.max_iterations && (rc = front()) != EVOMG_OK) return rc;      // the next iteration's first half
        HIP_TRY(hipEventSynchronize(outer_event_));
        d = hscalar_[0];
        res = sc_norm_from_dot(d);
        if (hist && nh < cap) hist[nh] = res;
        ++nh;
        if (!(res == res) || res > 1.7e308) { status = EVOMG_SOLVE_DIVERGED; break; }
        if (res < prob_.target_res_reduction * res0) { status = EVOMG_SOLVE_CONVERGED; break; }
        if (wall_limit_ms > 0 && wall_ms() - t0 > wall_limit_ms) { status = EVOMG_SOLVE_WALL_LIMIT; break; }
      }
    } else
      for (it = 1; it <= prob_.max_iterations; ++it) {
        rho = rho_new;
        if (have_rho) rho_new = rho_next;
        else if ((rc = dot_to_host(L, rhat, Res, &rho_new)) != EVOMG_OK) return rc;
        beta = sc_mul(sc_div(rho_new, rho), sc_div(alpha, omega));
        if (fuse) {
          hipLaunchKernelGGL((evomg_axpy2<T>), node_grid(g), node_block(), 0, stream_, g, p, (const T*)p, sc_neg(omega), (const T*)ap, (const T*)Res, beta);
        } else {
          axpy(L, p, p, sc_neg(omega), ap);
          axpy(L, p, Res, beta, p);
        }
        if ((rc = precondition(p)) != EVOMG_OK) return rc;
        T* u = fieldptr(L, EVOMG_F_SOL);
        T ra;
        if (fuse) {
          hipLaunchKernelGGL((evomg_apply_dots<T, 1>), dim3(dot_blocks), dim3(256), 0, stream_, g, outer_st_, (const T*)u, ap, (const T*)rhat, partials_);
          if ((rc = fetch_dots(1, &ra, &ra)) != EVOMG_OK) return rc;
        } else {
          apply_operator(L, outer_st_, u, ap);
          if ((rc = dot_to_host(L, rhat, ap, &ra)) != EVOMG_OK) return rc;
        }
        alpha = sc_div(rho_new, ra);
        axpy(L, hh, Sol, alpha, u);
        axpy(L, s, Res, sc_neg(alpha), ap);
        if ((rc = precondition(s)) != EVOMG_OK) return rc;
        u = fieldptr(L, EVOMG_F_SOL);
        T ts, tt;
        if (fuse) {
          hipLaunchKernelGGL((evomg_apply_dots<T, 2>), dim3(dot_blocks), dim3(256), 0, stream_, g, outer_st_, (const T*)u, t, (const T*)s, partials_);
          if ((rc = fetch_dots(2, &ts, &tt)) != EVOMG_OK) return rc;
        } else {
          apply_operator(L, outer_st_, u, t);
          if ((rc = dot_to_host(L, t, s, &ts)) != EVOMG_OK) return rc;
          if ((rc = dot_to_host(L, t, t, &tt)) != EVOMG_OK) return rc;
        }
        omega = sc_div(ts, tt);
        axpy(L, Sol, hh, omega, u);
        apply_bc(L, Sol);
        if (fuse) {
          hipLaunchKernelGGL((evomg_axpy_dots<T>), dim3(dot_blocks), dim3(256), 0, stream_, g, Res, (const T*)s, sc_neg(omega), (const T*)t, (const T*)rhat, partials_);
          if ((rc = fetch_dots(2, &d, &rho_next)) != EVOMG_OK) return rc;
          have_rho = true;
        } else {
          axpy(L, Res, s, sc_neg(omega), t);
          if ((rc = dot_to_host(L, Res, Res, &d)) != EVOMG_OK) return rc;
        }
        res = sc_norm_from_dot(d);
        if (hist && nh < cap) hist[nh] = res;
        ++nh;
        if (!(res == res) || res > 1.7e308) { status = EVOMG_SOLVE_DIVERGED; break; }
        if (res < prob_.target_res_reduction * res0) { status = EVOMG_SOLVE_CONVERGED; break; }
        if (wall_limit_ms > 0 && wall_ms() - t0 > wall_limit_ms) { status = EVOMG_SOLVE_WALL_LIMIT; break; }
      }
    if (it > prob_.max_iterations) it = prob_.max_iterations;
    HIP_TRY(hipStreamSynchronize(stream_));
    out->time_ms = wall_ms() - t0;
    HIP_TRY(hipMemcpyAsync(lv.f[EVOMG_F_CORR], Sol, bytes, hipMemcpyDeviceToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(lv.f[EVOMG_F_RES], Res, bytes, hipMemcpyDeviceToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    out->n_iterations = (double)it;
    out->init_res_norm = res0;
    out->final_res_norm = res;
    out->status = status;
    out->n_history = (nh < cap || !hist) ? nh : cap;
    out->conv_factor = res0 > 0 ? res / res0 : 0.0;
    return EVOMG_OK;
  }

  // algorithmic bytes of one op launch: SURVEY.md 8(d) table x node count of the op's level
  double algorithmic_bytes(const evomg_op_t& op) const {
    const Geo& g = lv_[op.level].g;
    const double N = pow((double)g.n, dim_);
    const double w = (double)sizeof(T);
    const double frac = 1.0 / (double)(1 << dim_);
    switch (op.opcode) {
      case EVOMG_OP_SMOOTH: case EVOMG_OP_RESIDUAL: case EVOMG_OP_RICHARDSON: return 3 * w * N;
      case EVOMG_OP_RESTRICT: { const double Nf = pow((double)lv_[op.level + 1].g.n, dim_); return (w + w * frac) * Nf; }
      case EVOMG_OP_PROLONG_ADD: return (2 * w + w * frac) * N;
      case EVOMG_OP_PROLONG_SET: return (w + w * frac) * N;
      case EVOMG_OP_SET_ZERO: return w * N;
      case EVOMG_OP_COPY: return 2 * w * N;
      default: return 0.0;
    }
  }

  evomg_problem_t prob_;
  int device_;
  int dim_;
  Level<T> lv_[EVOMG_MAX_LEVELS];
  Stencil<T> outer_st_{};
  T* outer_[8] = {};
  hipStream_t stream_ = nullptr;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  T* partials_ = nullptr;
  T* dscalar_ = nullptr;
  T* hscalar_ = nullptr;
  int* dinfo_ = nullptr;
  bool mirror_enabled_ = getenv("EVOMG_NO_BOUNDARY_MIRROR") == nullptr;
  bool x_zero_next_ = false;
  bool zero_elision_ = getenv("EVOMG_NO_ZERO_ELISION") == nullptr;
  bool bnd_zero_[24][EVOMG_F_COUNT];   // boundary of the field's current array is known to be all zero
  static inline const bool pow2_reciprocal_ = getenv("EVOMG_NO_POW2_RECIPROCAL") == nullptr;
  bool tail_norm_enabled_ = getenv("EVOMG_NO_TAIL_NORM") == nullptr;
  bool tail_restrict_enabled_ = getenv("EVOMG_NO_TAIL_RESTRICT") == nullptr;
  hipEvent_t norm_event_ = nullptr;
  hipEvent_t outer_event_ = nullptr;
  bool speculate_ = getenv("EVOMG_NO_SPECULATION") == nullptr;
  // experiments only: the oracle restates the default (krylov_bicgstab_per_launch)
  int kb_min_n_ = getenv("EVOMG_KB_MIN_N") ? atoi(getenv("EVOMG_KB_MIN_N")) : 129;
  bool kcg_enabled_ = getenv("EVOMG_NO_LAUNCH_CG") == nullptr;       // large coarsest grids: CG as a sequence of launches
  KcgState* kcg_state_ = nullptr;
  KcgState* kcg_host_ = nullptr;
  KbState<T>* kb_state_ = nullptr;
  KbState<T>* kb_host_ = nullptr;
  T* kb_scal_ = nullptr;
  bool outer_fuse_ = getenv("EVOMG_NO_OUTER_FUSE") == nullptr;      // fused passes of the preconditioned BiCGStab loop
  bool outer_dev_steps_ = getenv("EVOMG_NO_OUTER_DEV_STEPS") == nullptr;   // ... with alpha and omega formed on the device
  bool strip_bc_ = getenv("EVOMG_NO_STRIP_BC") == nullptr;
  T* norm_partials_ = nullptr;
  size_t norm_partials_cap_ = 0;
  T* cg_xpart_ = nullptr;
  T* cg_xface_ = nullptr;
  unsigned* cg_flags_ = nullptr;
  bool cg_multi_attr_set_ = false;
  bool cg_multi_used_ = false;
  bool cg_lds_attr_set_[24] = {};
  int prof_opcode_ = -1, prof_level_ = -1;
  size_t prof_used_ = 0;
  double prof_bytes_ = 0.0;
  std::vector<hipEvent_t> prof_events_;
  bool use_reg_krylov_ = (getenv("EVOMG_NO_REG_KRYLOV") == nullptr);
  bool block_fast_ = (getenv("EVOMG_NO_BLOCK_FAST") == nullptr);
  bool fuse_enabled_ = (getenv("EVOMG_NO_FUSE") == nullptr);
  bool tile3d_rr_ = (getenv("EVOMG_NO_TILE3D_RR") == nullptr);
  bool balance_rows_ = (getenv("EVOMG_NO_BALANCED_ROWS") == nullptr);
  bool strip_block_ = (getenv("EVOMG_NO_STRIP_BLOCK") == nullptr);
  // off by default: measured slower than the op-fused multi-CU launches it replaces (one compute unit
  // moves 64 B/clk, so a 129^2 op costs one workgroup what a launch costs; t9 2.90 vs 2.31 ms). EVOMG_BATCH=1: A/B, tests
  bool small3d_enabled_ = getenv("EVOMG_NO_SMALL3D") == nullptr;
  bool small3d_attr_set_ = false;
  bool small2d_enabled_ = getenv("EVOMG_NO_SMALL2D") == nullptr;
  // largest grid that runs as LDS tiles: Helmholtz levels 3..7 149 (33) -> 137-146 (65) -> 131 ms (129), nothing beyond; real
  // sweeps at 129^2 are no faster than the strips (4.2 vs 3.7 us back to back), at 65^2 they are (3.7 vs 5.8 us)
  int small2d_max_n_ = getenv("EVOMG_SMALL2D_MAX_N") ? atoi(getenv("EVOMG_SMALL2D_MAX_N")) : (scalar_traits<T>::is_complex ? 129 : 65);
  std::set<const void*> small2d_attr_set_;
  bool batch_enabled_ = (getenv("EVOMG_BATCH") != nullptr) && (getenv("EVOMG_NO_STRIP") == nullptr);
  int batch_max_n_ = getenv("EVOMG_BATCH_MAX_N") ? atoi(getenv("EVOMG_BATCH_MAX_N")) : 129;
  MicroBatch batch_ = {};
  int n_cu_ = 256;
  int twice_planes_ = getenv("EVOMG_TWICE_PLANES") ? atoi(getenv("EVOMG_TWICE_PLANES")) : 16;
  int rr_cplanes_ = getenv("EVOMG_RR_CPLANES") ? atoi(getenv("EVOMG_RR_CPLANES")) : 8;
  bool strip_enabled_ = (getenv("EVOMG_NO_STRIP") == nullptr);
  // complex (Helmholtz) problems: from 17^2 — the strip kernels write the Robin columns themselves, where the generic ones
  // need a boundary launch behind every op (reference-range solve 212-224 -> 197-215 ms)
  // real problems too since the levels up to 33^2 run in LDS (evomg_small2d: the fused forms instead of one generic launch per op)
  int strip_min_n_ = getenv("EVOMG_STRIP_MIN_N") ? atoi(getenv("EVOMG_STRIP_MIN_N")) : ((scalar_traits<T>::is_complex || getenv("EVOMG_NO_SMALL2D") == nullptr) ? 17 : 33);
  int strip_rows_ = getenv("EVOMG_STRIP_ROWS") ? atoi(getenv("EVOMG_STRIP_ROWS")) : 0;
  int strip3d_min_n_ = getenv("EVOMG_STRIP3D_MIN_N") ? atoi(getenv("EVOMG_STRIP3D_MIN_N")) : 0;
  int strip_planes_ = getenv("EVOMG_STRIP_PLANES") ? atoi(getenv("EVOMG_STRIP_PLANES")) : 0;
  bool tile3d_enabled_ = getenv("EVOMG_NO_TILE3D") == nullptr;
  bool tile3d_rbgs_ = getenv("EVOMG_NO_TILE3D_RBGS") == nullptr;
  bool tile3d_twice_ = getenv("EVOMG_NO_TILE3D_TWICE") == nullptr;
  bool tile3d_pro_ = getenv("EVOMG_NO_TILE3D_PRO") == nullptr;
  int tile3d_nt_ = getenv("EVOMG_TILE3D_NT") ? atoi(getenv("EVOMG_TILE3D_NT")) : 0;   // bit 0: non-temporal f loads (measured slower at 513^3: profiles/r02_pmc_3d.md)
  int tile3d_order_ = getenv("EVOMG_TILE3D_ORDER") ? atoi(getenv("EVOMG_TILE3D_ORDER")) : 1;
  int tile3d_waves_ = getenv("EVOMG_TILE3D_WAVES") ? atoi(getenv("EVOMG_TILE3D_WAVES")) : 0;
  bool strip_nt_ = getenv("EVOMG_STRIP_NT") ? atoi(getenv("EVOMG_STRIP_NT")) != 0 : true;
  double nt_lo_mib_ = getenv("EVOMG_NT_LO_MIB") ? atof(getenv("EVOMG_NT_LO_MIB")) : 256.0;   // 3 fields above this: stream x, y
  double nt_hi_mib_ = getenv("EVOMG_NT_HI_MIB") ? atof(getenv("EVOMG_NT_HI_MIB")) : 192.0;   // ... unless one field exceeds this
};

// ---- systems of PDEs (evomg_system.h): own storage and a plain interpreter, nothing shared with the tuned scalar path ----
class SystemImpl : public ImplBase {
 public:
  SystemImpl(const evomg_problem_t& p, int device) : prob_(p), device_(device), nf_(p.n_unknowns) {}
  ~SystemImpl() override {
    (void)hipSetDevice(device_);
    for (int L = prob_.min_level; L <= prob_.max_level; ++L) {
      SLevel& lv = lv_[L];
      for (auto* q : lv.f) if (q) (void)hipFree(q);
      for (auto* q : lv.init) if (q) (void)hipFree(q);
      for (auto* q : lv.alt) if (q) (void)hipFree(q);
      for (auto* q : lv.tmp) if (q) (void)hipFree(q);
    }
    if (partials_) (void)hipFree(partials_);
    if (dscalar_) (void)hipFree(dscalar_);
    if (dinfo_) (void)hipFree(dinfo_);
    if (hscalar_) (void)hipHostFree(hscalar_);
    if (stream_) (void)hipStreamDestroy(stream_);
  }
  int init() override {
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device_ < 0 || device_ >= count) { snprintf(err, sizeof err, "device %d not available (%d devices)", device_, count); return EVOMG_ERR_DEVICE; }
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&partials_, sizeof(double) * kDotBlocks));
    HIP_TRY(hipMalloc(&dscalar_, sizeof(double) * 8));
    HIP_TRY(hipMalloc(&dinfo_, sizeof(int) * 4));
    HIP_TRY(hipHostMalloc(&hscalar_, sizeof(double) * 8));
    for (int L = prob_.min_level; L <= prob_.max_level; ++L) {
      SLevel& lv = lv_[L];
      const int n = (1 << L) + 1;
      lv.g.n = n;
      lv.g.pitch = ((n + 15) / 16) * 16;
      lv.g.plane = 0;
      lv.elems = (size_t)lv.g.pitch * n;
      memcpy(lv.st.a, prob_.sys_stencil[L], sizeof lv.st.a);
      for (int u = 0; u < nf_; ++u) {
        for (int slot : {EVOMG_F_SOL, EVOMG_F_RHS, EVOMG_F_RES}) HIP_TRY(alloc_zero(&lv.f[EVOMG_FIELD(slot, u)], lv.elems));
        if (L == prob_.max_level) HIP_TRY(alloc_zero(&lv.f[EVOMG_FIELD(EVOMG_F_CORR, u)], lv.elems));
        if (L == prob_.min_level) {
          HIP_TRY(alloc_zero(&lv.f[EVOMG_FIELD(EVOMG_F_CGS_RHS, u)], lv.elems));
          HIP_TRY(alloc_zero(&lv.tmp[2 * u], lv.elems));
          HIP_TRY(alloc_zero(&lv.tmp[2 * u + 1], lv.elems));
        }
        HIP_TRY(alloc_zero(&lv.alt[u], lv.elems));
      }
    }
    HIP_TRY(hipStreamSynchronize(stream_));
    return EVOMG_OK;
  }
  int set_initial_field(int level, int field, const double* host) override {
    if (!level_ok(level) || !host) return invalid("set_initial_field: bad level/pointer");
    HIP_TRY(hipSetDevice(device_));
    const int k = canon(level, field);
    SLevel& lv = lv_[level];
    if (k < 0 || !lv.f[k]) return invalid("set_initial_field: no such field on this level");
    if (!lv.init[k]) HIP_TRY(alloc_zero(&lv.init[k], lv.elems));
    HIP_TRY(upload(lv, lv.init[k], host));
    HIP_TRY(hipMemcpyAsync(lv.f[k], lv.init[k], lv.elems * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return EVOMG_OK;
  }
  int set_field(int level, int field, const double* host) override {
    if (!level_ok(level) || !host) return invalid("set_field: bad level/pointer");
    HIP_TRY(hipSetDevice(device_));
    double* d = fieldptr(level, field);
    if (!d) return invalid("set_field: no such field on this level");
    HIP_TRY(upload(lv_[level], d, host));
    return EVOMG_OK;
  }
  int get_field(int level, int field, double* host) override {
    if (!level_ok(level) || !host) return invalid("get_field: bad level/pointer");
    HIP_TRY(hipSetDevice(device_));
    const double* d = fieldptr(level, field);
    if (!d) return invalid("get_field: no such field on this level");
    const SLevel& lv = lv_[level];
    HIP_TRY(hipStreamSynchronize(stream_));
    HIP_TRY(hipMemcpy2D(host, lv.g.n * sizeof(double), d, lv.g.pitch * sizeof(double), lv.g.n * sizeof(double), lv.g.n, hipMemcpyDeviceToHost));
    return EVOMG_OK;
  }
  int set_operators(const evomg_problem_t*) override { return unsupported("set_operators: not available on systems of PDEs"); }
  int run_ops(const evomg_op_t* ops, int n_ops) override {
    HIP_TRY(hipSetDevice(device_));
    int rc = validate(ops, n_ops);
    if (rc != EVOMG_OK) return rc;
    if ((rc = exec_all(ops, n_ops)) != EVOMG_OK) return rc;
    HIP_TRY(hipStreamSynchronize(stream_));
    return EVOMG_OK;
  }
  int norm(int level, int field, double* out) override {
    if (!level_ok(level) || !out) return invalid("norm: bad level/pointer");
    HIP_TRY(hipSetDevice(device_));
    const double* d = fieldptr(level, field);
    if (!d) return invalid("norm: no such field on this level");
    double v;
    int rc = dot_to_host(level, d, &v);
    if (rc != EVOMG_OK) return rc;
    *out = sqrt(v);
    return EVOMG_OK;
  }
  int evaluate(const evomg_op_t* ops, int n_ops, int samples, double wall_limit_ms, evomg_result_t* out, double* hist, int cap) override {
    HIP_TRY(hipSetDevice(device_));
    if (!out || samples < 1 || n_ops < 0 || (!ops && n_ops > 0)) return invalid("evaluate: bad arguments");
    memset(out, 0, sizeof *out);
    int rc = validate(ops, n_ops);
    if (rc != EVOMG_OK) return rc;
    double tsum = 0.0;
    int done = 0;
    evomg_result_t r;
    for (int smp = 0; smp < samples; ++smp) {
      memset(&r, 0, sizeof r);
      for (int L = prob_.min_level; L <= prob_.max_level; ++L)
        for (int k = 0; k < kSlots; ++k) {
          SLevel& lv = lv_[L];
          if (!lv.f[k]) continue;
          if (lv.init[k]) HIP_TRY(hipMemcpyAsync(lv.f[k], lv.init[k], lv.elems * sizeof(double), hipMemcpyDeviceToDevice, stream_));
          else HIP_TRY(hipMemsetAsync(lv.f[k], 0, lv.elems * sizeof(double), stream_));
        }
      HIP_TRY(hipStreamSynchronize(stream_));
      int nh = 0;
      const double t0 = wall_ms();
      double res0 = 0.0;
      if ((rc = residual_norm(&res0)) != EVOMG_OK) return rc;
      if (hist && nh < cap) hist[nh] = res0;
      ++nh;
      double res = res0;
      int it = 0, status = EVOMG_SOLVE_MAXITER;
      if (res0 == 0.0) status = EVOMG_SOLVE_CONVERGED;
      else
        for (it = 1; it <= prob_.max_iterations; ++it) {
          if ((rc = exec_all(ops, n_ops)) != EVOMG_OK) return rc;
          if ((rc = residual_norm(&res)) != EVOMG_OK) return rc;
          if (hist && nh < cap) hist[nh] = res;
          ++nh;
          if (!(res == res) || res > 1.7e308) { status = EVOMG_SOLVE_DIVERGED; break; }
          if (res <= prob_.target_res_reduction * res0) { status = EVOMG_SOLVE_CONVERGED; break; }
          if (wall_limit_ms > 0 && wall_ms() - t0 > wall_limit_ms) { status = EVOMG_SOLVE_WALL_LIMIT; break; }
        }
      if (it > prob_.max_iterations) it = prob_.max_iterations;
      r.time_ms = wall_ms() - t0;
      r.n_iterations = (double)it;
      r.init_res_norm = res0;
      r.final_res_norm = res;
      r.status = status;
      r.n_history = (nh < cap || !hist) ? nh : cap;
      r.conv_factor = (it > 0 && res0 > 0) ? pow(res / res0, 1.0 / (double)it) : 0.0;
      tsum += r.time_ms;
      ++done;
      if (status != EVOMG_SOLVE_CONVERGED) break;
    }
    *out = r;
    out->time_ms = tsum / (double)done;
    return EVOMG_OK;
  }
  int bench_op(const evomg_op_t*, int, int, evomg_bench_t*) override { return unsupported("bench_op: not available on systems of PDEs"); }
  int kernel_name(const evomg_op_t* op, char* buf, int len) override {
    if (!op || !buf || len < 8) return invalid("kernel_name: bad arguments");
    snprintf(buf, len, "%s", op->opcode == EVOMG_OP_SMOOTH ? "evomg_sys_smooth" : op->opcode == EVOMG_OP_RESIDUAL ? "evomg_sys_residual" :
                             op->opcode == EVOMG_OP_COARSE_SOLVE ? "evomg_sys_cg" : "generic");
    return EVOMG_OK;
  }
  int profile_begin(int, int) override { return EVOMG_OK; }
  int profile_end(evomg_bench_t* out) override { if (out) memset(out, 0, sizeof *out); return EVOMG_OK; }

 private:
  static constexpr int kSlots = EVOMG_F_COUNT * EVOMG_MAX_UNKNOWNS;
  struct SLevel {
    Geo g{};
    size_t elems = 0;
    double* f[kSlots] = {};
    double* init[kSlots] = {};
    double* alt[EVOMG_MAX_UNKNOWNS] = {};
    double* tmp[2 * EVOMG_MAX_UNKNOWNS] = {};
    SysStencil st{};
  };
  hipError_t alloc_zero(double** p, size_t elems) {
    hipError_t e = hipMalloc(p, elems * sizeof(double));
    if (e != hipSuccess) return e;
    return hipMemsetAsync(*p, 0, elems * sizeof(double), stream_);
  }
  hipError_t upload(const SLevel& lv, double* dst, const double* host) {
    hipError_t e = hipStreamSynchronize(stream_);
    if (e != hipSuccess) return e;
    return hipMemcpy2D(dst, lv.g.pitch * sizeof(double), host, lv.g.n * sizeof(double), lv.g.n * sizeof(double), lv.g.n, hipMemcpyHostToDevice);
  }
  bool level_ok(int L) const { return L >= prob_.min_level && L <= prob_.max_level; }
  int canon(int level, int field) const {
    if (field < 0 || field >= EVOMG_F_COUNT * nf_) return -1;
    const int slot = field % EVOMG_F_COUNT, u = field / EVOMG_F_COUNT;
    if (slot == EVOMG_F_CORR && level < prob_.max_level) return EVOMG_FIELD(EVOMG_F_SOL, u);
    return field;
  }
  double* fieldptr(int level, int field) { const int k = canon(level, field); return k < 0 ? nullptr : lv_[level].f[k]; }
  int invalid(const char* msg) { snprintf(err, sizeof err, "%s", msg); return EVOMG_ERR_INVALID; }
  int unsupported(const char* msg) { snprintf(err, sizeof err, "%s", msg); return EVOMG_ERR_UNSUPPORTED; }
  static dim3 blk() { return dim3(64, 4, 1); }
  static dim3 grd(int nx, int ny) { return dim3((nx + 63) / 64, (ny + 3) / 4, 1); }

  // mirrors the oracle's sys_check_op
  int validate(const evomg_op_t* ops, int n_ops) {
    for (int i = 0; i < n_ops; ++i) {
      const evomg_op_t& op = ops[i];
      const int lo = prob_.min_level, hi = prob_.max_level;
      int rc = EVOMG_ERR_INVALID;
      bool ok = op.level >= lo && op.level <= hi && fieldptr(op.level, op.dst) != nullptr;
      if (ok) switch (op.opcode) {
        case EVOMG_OP_SMOOTH: {
          const int mask = op.unknowns;
          ok = fieldptr(op.level, op.src1) != nullptr && mask > 0 && mask < (1 << nf_) && op.block[0] >= 1 && op.block[1] >= 1;
          if (!ok) break;
          int k = 0, first = 0;
          for (int u = 0; u < nf_; ++u) k += (mask >> u) & 1;
          while (!((mask >> first) & 1)) ++first;
          const bool other = op.block_other[0] != 0 || op.block_other[1] != 0;       // second unknown: its own block shape
          if (other && (k != 2 || op.block_other[0] < 1 || op.block_other[1] < 1)) { ok = false; break; }
          const int cells = op.block[0] * op.block[1] + (k == 2 ? (other ? op.block_other[0] * op.block_other[1] : op.block[0] * op.block[1]) : 0);
          ok = cells <= EVOMG_MAX_BLOCK_CELLS && op.dst / EVOMG_F_COUNT == first && op.src1 / EVOMG_F_COUNT == first &&
               (op.partitioning == EVOMG_PART_SINGLE || op.partitioning == EVOMG_PART_RED_BLACK);
          if (ok && op.partitioning == EVOMG_PART_RED_BLACK && cells != k) { ok = false; rc = EVOMG_ERR_UNSUPPORTED; }
          break;
        }
        case EVOMG_OP_RESIDUAL: ok = fieldptr(op.level, op.src0) && fieldptr(op.level, op.src1) && op.src0 / EVOMG_F_COUNT == 0; break;
        case EVOMG_OP_RESTRICT: ok = op.level < hi && fieldptr(op.level + 1, op.src0); break;
        case EVOMG_OP_PROLONG_ADD: case EVOMG_OP_PROLONG_SET: ok = op.level > lo && fieldptr(op.level - 1, op.src0); break;
        case EVOMG_OP_SET_ZERO: break;
        case EVOMG_OP_COPY: ok = fieldptr(op.level, op.src0) != nullptr; break;
        case EVOMG_OP_COARSE_SOLVE: ok = op.level == lo; break;
        case EVOMG_OP_RICHARDSON: ok = false; rc = EVOMG_ERR_UNSUPPORTED; break;
        default: ok = false;
      }
      if (!ok) { snprintf(err, sizeof err, "op %d (opcode %d, level %d) rejected", i, op.opcode, op.level); return rc; }
    }
    return EVOMG_OK;
  }

  SysConstFields slot_fields(int level, int slot) {
    SysConstFields x{};
    for (int u = 0; u < nf_; ++u) x.p[u] = fieldptr(level, EVOMG_FIELD(slot, u));
    return x;
  }
  int exec(const evomg_op_t& op) {
    const int L = op.level;
    SLevel& lv = lv_[L];
    const Geo& g = lv.g;
    switch (op.opcode) {
      case EVOMG_OP_SMOOTH: {
        const int mask = op.unknowns, inner = g.n - 2;
        // per-unknown block shapes: the first unknown of the mask has `block`, the second `block_other` where that is set
        SysShapes sh{};
        int bx = 1, by = 1, seen = 0;
        for (int u = 0; u < nf_; ++u) {
          if (!((mask >> u) & 1)) { sh.bx[u] = sh.by[u] = 0; continue; }
          const bool second = seen++ > 0 && op.block_other[0] > 0;
          sh.bx[u] = second ? op.block_other[0] : op.block[0];
          sh.by[u] = second ? op.block_other[1] : op.block[1];
          bx = sh.bx[u] > bx ? sh.bx[u] : bx;
          by = sh.by[u] > by ? sh.by[u] : by;
        }
        const int sol_slot = canon(L, op.dst) % EVOMG_F_COUNT, rhs_slot = op.src1 % EVOMG_F_COUNT;
        const int tx = (inner + bx - 1) / bx, ty = (inner + by - 1) / by;
        const int passes = op.partitioning == EVOMG_PART_RED_BLACK ? 2 : 1;
        const SysConstFields f = slot_fields(L, rhs_slot);
        // point blocks: the straight-line kernel that writes every node of y (no copy of x first)
        const bool point = bx == 1 && by == 1 && sys_point_fast_ && nf_ == 2;
        for (int pass = 0; pass < passes; ++pass) {
          const SysConstFields x = slot_fields(L, sol_slot);
          SysFields y{};
          for (int u = 0; u < nf_; ++u) {
            y.p[u] = lv.alt[u];
            if (!point && ((mask >> u) & 1)) HIP_TRY(hipMemcpyAsync(lv.alt[u], x.p[u], lv.elems * sizeof(double), hipMemcpyDeviceToDevice, stream_));
          }
          const int col = passes == 2 ? pass : -1;
          if (point && mask == 3) hipLaunchKernelGGL(evomg_sys_point_smooth<3>, grd(g.n, g.n), blk(), 0, stream_, g, lv.st, nf_, col, x, f, y, op.omega);
          else if (point && mask == 1) hipLaunchKernelGGL(evomg_sys_point_smooth<1>, grd(g.n, g.n), blk(), 0, stream_, g, lv.st, nf_, col, x, f, y, op.omega);
          else if (point && mask == 2) hipLaunchKernelGGL(evomg_sys_point_smooth<2>, grd(g.n, g.n), blk(), 0, stream_, g, lv.st, nf_, col, x, f, y, op.omega);
          else
          hipLaunchKernelGGL(evomg_sys_smooth, grd(tx, ty), blk(), 0, stream_, g, lv.st, nf_, mask, sh, bx, by, tx, ty,
                             passes == 2 ? pass : -1, x, f, y, op.omega);
          // a colour pass always takes effect; a `with jacobi` sweep only where the next slot is advanced (evomg.h)
          const bool lost = passes == 1 && prob_.jacobi_semantics == EVOMG_JACOBI_EXASTENCILS_V11;
          if (!lost)
            for (int u = 0; u < nf_; ++u)
              if ((mask >> u) & 1) std::swap(lv.f[canon(L, EVOMG_FIELD(sol_slot, u))], lv.alt[u]);
        }
        break;
      }
      case EVOMG_OP_RESIDUAL: {
        const SysConstFields x = slot_fields(L, canon(L, op.src0) % EVOMG_F_COUNT);
        hipLaunchKernelGGL(evomg_sys_residual, grd(g.n, g.n), blk(), 0, stream_, g, lv.st, nf_, op.dst / EVOMG_F_COUNT, x,
                           (const double*)fieldptr(L, op.src1), fieldptr(L, op.dst));
        break;
      }
      case EVOMG_OP_RESTRICT:
        hipLaunchKernelGGL((evomg_restrict<double, 2>), grd(g.n, g.n), blk(), 0, stream_, g, lv_[L + 1].g, (const double*)fieldptr(L + 1, op.src0), fieldptr(L, op.dst));
        break;
      case EVOMG_OP_PROLONG_ADD:
        hipLaunchKernelGGL((evomg_prolong<double, 2, true>), grd(g.n, g.n), blk(), 0, stream_, g, lv_[L - 1].g, (const double*)fieldptr(L - 1, op.src0), fieldptr(L, op.dst), op.omega);
        break;
      case EVOMG_OP_PROLONG_SET:
        hipLaunchKernelGGL((evomg_prolong<double, 2, false>), grd(g.n, g.n), blk(), 0, stream_, g, lv_[L - 1].g, (const double*)fieldptr(L - 1, op.src0), fieldptr(L, op.dst), 1.0);
        break;
      case EVOMG_OP_SET_ZERO:
        hipLaunchKernelGGL((evomg_zero_inner<double, 2>), grd(g.n, g.n), blk(), 0, stream_, g, fieldptr(L, op.dst), 0);
        break;
      case EVOMG_OP_COPY: {
        double* d = fieldptr(L, op.dst);
        const double* s2 = fieldptr(L, op.src0);
        if (d != s2) HIP_TRY(hipMemcpyAsync(d, s2, lv.elems * sizeof(double), hipMemcpyDeviceToDevice, stream_));
        break;
      }
      case EVOMG_OP_COARSE_SOLVE: {
        SysFields x{}, r{}, p{}, ap{};
        SysConstFields b{};
        for (int u = 0; u < nf_; ++u) {
          x.p[u] = fieldptr(L, EVOMG_FIELD(EVOMG_F_CORR, u));
          b.p[u] = fieldptr(L, EVOMG_FIELD(EVOMG_F_CGS_RHS, u));
          r.p[u] = fieldptr(L, EVOMG_FIELD(EVOMG_F_RES, u));
          p.p[u] = lv.tmp[2 * u];
          ap.p[u] = lv.tmp[2 * u + 1];
        }
        if ((long)(g.n - 2) * (g.n - 2) * nf_ <= 512 && sys_cg_small_)
          hipLaunchKernelGGL(evomg_sys_cg_small, dim3(1), dim3(512), (size_t)nf_ * g.n * g.n * sizeof(double), stream_, g, lv.st, nf_, x, b, r,
                             prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_);
        else
        hipLaunchKernelGGL(evomg_sys_cg, dim3(1), dim3(1024), 0, stream_, g, lv.st, nf_, x, b, r, p, ap, prob_.cgs_max_iterations,
                           prob_.cgs_target_res_reduction, dinfo_);
        break;
      }
      default: return invalid("exec: unknown opcode");
    }
    HIP_TRY(hipGetLastError());
    return EVOMG_OK;
  }
  // ops[i], ops[i + 1]: the same statement for unknown 0 and unknown 1 (what a cycle emits back to back), independent of
  // each other -> one launch with blockIdx.z = unknown (evomg_system.h)
  bool same_statement_for_both_unknowns(const evomg_op_t& a, const evomg_op_t& b) const {
    if (!sys_pair_fuse_ || nf_ != 2 || a.opcode != b.opcode || a.level != b.level) return false;
    if (a.dst / EVOMG_F_COUNT != 0 || b.dst != a.dst + EVOMG_F_COUNT) return false;
    switch (a.opcode) {
      case EVOMG_OP_RESIDUAL:        // row i reads every unknown's solution slot (src0 names unknown 0's) and its own rhs
        return b.src0 == a.src0 && a.src1 / EVOMG_F_COUNT == 0 && b.src1 == a.src1 + EVOMG_F_COUNT &&
               canon(a.level, a.dst) % EVOMG_F_COUNT != canon(a.level, a.src0) % EVOMG_F_COUNT &&
               canon(a.level, a.dst) % EVOMG_F_COUNT != canon(a.level, a.src1) % EVOMG_F_COUNT;
      case EVOMG_OP_RESTRICT: return a.src0 / EVOMG_F_COUNT == 0 && b.src0 == a.src0 + EVOMG_F_COUNT;
      case EVOMG_OP_PROLONG_ADD: case EVOMG_OP_PROLONG_SET:
        return a.src0 / EVOMG_F_COUNT == 0 && b.src0 == a.src0 + EVOMG_F_COUNT && a.omega == b.omega;
      case EVOMG_OP_SET_ZERO: return true;
      default: return false;
    }
  }
  int exec_pair(const evomg_op_t& op) {
    const int L = op.level;
    const Geo& g = lv_[L].g;
    const dim3 grid2((g.n + 63) / 64, (g.n + 3) / 4, 2);
    SysFields d{};
    for (int u = 0; u < 2; ++u) d.p[u] = fieldptr(L, op.dst + EVOMG_F_COUNT * u);
    switch (op.opcode) {
      case EVOMG_OP_RESIDUAL: {
        const SysConstFields x = slot_fields(L, canon(L, op.src0) % EVOMG_F_COUNT);
        const SysConstFields f = slot_fields(L, op.src1 % EVOMG_F_COUNT);
        hipLaunchKernelGGL(evomg_sys_residual2, grid2, blk(), 0, stream_, g, lv_[L].st, nf_, x, f, d);
        break;
      }
      case EVOMG_OP_RESTRICT: {
        SysConstFields s{};
        for (int u = 0; u < 2; ++u) s.p[u] = fieldptr(L + 1, op.src0 + EVOMG_F_COUNT * u);
        hipLaunchKernelGGL(evomg_sys_restrict2, grid2, blk(), 0, stream_, g, lv_[L + 1].g, s, d);
        break;
      }
      case EVOMG_OP_PROLONG_ADD: case EVOMG_OP_PROLONG_SET: {
        SysConstFields s{};
        for (int u = 0; u < 2; ++u) s.p[u] = fieldptr(L - 1, op.src0 + EVOMG_F_COUNT * u);
        if (op.opcode == EVOMG_OP_PROLONG_ADD) hipLaunchKernelGGL(evomg_sys_prolong2<true>, grid2, blk(), 0, stream_, g, lv_[L - 1].g, s, d, op.omega);
        else hipLaunchKernelGGL(evomg_sys_prolong2<false>, grid2, blk(), 0, stream_, g, lv_[L - 1].g, s, d, 1.0);
        break;
      }
      case EVOMG_OP_SET_ZERO: hipLaunchKernelGGL(evomg_sys_zero2, grid2, blk(), 0, stream_, g, d); break;
      default: return invalid("exec_pair: unsupported opcode");
    }
    HIP_TRY(hipGetLastError());
    return EVOMG_OK;
  }
  int exec_all(const evomg_op_t* ops, int n_ops) {
    for (int i = 0; i < n_ops; ++i) {
      int rc;
      if (i + 1 < n_ops && same_statement_for_both_unknowns(ops[i], ops[i + 1])) {
        if ((rc = exec_pair(ops[i])) != EVOMG_OK) return rc;
        ++i;
      } else if ((rc = exec(ops[i])) != EVOMG_OK) return rc;
    }
    return EVOMG_OK;
  }
  int dot_to_host(int L, const double* a, double* out) {
    const Geo& g = lv_[L].g;
    const int m = g.n - 2;
    const int blocks = m < kDotBlocks ? m : kDotBlocks;
    hipLaunchKernelGGL((evomg_dot_partial<double, 2>), dim3(blocks), dim3(256), 0, stream_, g, a, a, partials_);
    hipLaunchKernelGGL((evomg_dot_final<double>), dim3(1), dim3(256), 0, stream_, partials_, blocks, dscalar_);
    HIP_TRY(hipMemcpyAsync(hscalar_, dscalar_, sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    *out = hscalar_[0];
    return EVOMG_OK;
  }
  // residual of every equation on the finest level + the combined norm sqrt(sum over all unknowns)
  int residual_norm(double* out) {
    const int L = prob_.max_level;
    SLevel& lv = lv_[L];
    const SysConstFields x = slot_fields(L, EVOMG_F_SOL);
    double total = 0.0;
    for (int u = 0; u < nf_; ++u) {
      double* r = fieldptr(L, EVOMG_FIELD(EVOMG_F_RES, u));
      hipLaunchKernelGGL(evomg_sys_residual, grd(lv.g.n, lv.g.n), blk(), 0, stream_, lv.g, lv.st, nf_, u, x,
                         (const double*)fieldptr(L, EVOMG_FIELD(EVOMG_F_RHS, u)), r);
      double d;
      const int rc = dot_to_host(L, r, &d);
      if (rc != EVOMG_OK) return rc;
      total += d;
    }
    *out = sqrt(total);
    return EVOMG_OK;
  }

  evomg_problem_t prob_;
  int device_;
  int nf_;
  bool sys_cg_small_ = getenv("EVOMG_NO_SYS_CG_SMALL") == nullptr;       // register / LDS form of the stacked CG (<= 512 unknowns)
  bool sys_pair_fuse_ = getenv("EVOMG_NO_SYS_PAIR_FUSE") == nullptr;     // one launch for a statement emitted once per unknown
  bool sys_point_fast_ = getenv("EVOMG_NO_SYS_POINT_FAST") == nullptr;   // straight-line kernel for point blocks (evomg_system.h)
  SLevel lv_[EVOMG_MAX_LEVELS];
  hipStream_t stream_ = nullptr;
  double* partials_ = nullptr;
  double* dscalar_ = nullptr;
  double* hscalar_ = nullptr;
  int* dinfo_ = nullptr;
};

// ---- non-linear FAS problems (evomg_fas.h): own storage and a plain interpreter, like the systems of PDEs -----------------
class FASImpl : public ImplBase {
 public:
  FASImpl(const evomg_problem_t& p, int device) : prob_(p), device_(device) {}
  ~FASImpl() override {
    (void)hipSetDevice(device_);
    for (int L = prob_.min_level; L <= prob_.max_level; ++L) {
      FLevel& lv = lv_[L];
      for (auto* q : lv.f) if (q) (void)hipFree(q);
      for (auto* q : lv.init) if (q) (void)hipFree(q);
      if (lv.next) (void)hipFree(lv.next);
      if (lv.xchg) (void)hipFree(lv.xchg);
    }
    if (fas_flags_) (void)hipFree(fas_flags_);
    if (fas_info_) (void)hipFree(fas_info_);
    if (fas_info_host_) (void)hipHostFree(fas_info_host_);
    if (partials_) (void)hipFree(partials_);
    if (dscalar_) (void)hipFree(dscalar_);
    if (hscalar_) (void)hipHostFree(hscalar_);
    if (stream_) (void)hipStreamDestroy(stream_);
  }
  int init() override {
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device_ < 0 || device_ >= count) { snprintf(err, sizeof err, "device %d not available (%d devices)", device_, count); return EVOMG_ERR_DEVICE; }
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&partials_, sizeof(double) * kDotBlocks));
    HIP_TRY(hipMalloc(&dscalar_, sizeof(double) * 8));
    HIP_TRY(hipHostMalloc(&hscalar_, sizeof(double) * 8));
    for (int L = prob_.min_level; L <= prob_.max_level; ++L) {
      FLevel& lv = lv_[L];
      const int n = (1 << L) + 1;
      lv.g.n = n;
      lv.g.pitch = ((n + 15) / 16) * 16;
      lv.g.plane = 0;
      lv.elems = (size_t)lv.g.pitch * n;
      for (int e = 0; e < EVOMG_STENCIL_ENTRIES; ++e) lv.st.c[e] = prob_.mg_stencil[L][e].re;
      for (int k = 0; k < kFields; ++k) HIP_TRY(alloc_zero(&lv.f[k], lv.elems));     // Solution, RHS, Residual, Approximation
      HIP_TRY(alloc_zero(&lv.next, lv.elems));                                          // Solution<next>
    }
    HIP_TRY(hipStreamSynchronize(stream_));
    return EVOMG_OK;
  }
  int set_initial_field(int level, int field, const double* host) override {
    if (!level_ok(level) || !host || !fieldptr(level, field)) return invalid("set_initial_field: bad level/field/pointer");
    HIP_TRY(hipSetDevice(device_));
    FLevel& lv = lv_[level];
    if (!lv.init[field]) HIP_TRY(alloc_zero(&lv.init[field], lv.elems));
    HIP_TRY(upload(lv, lv.init[field], host));
    HIP_TRY(hipMemcpyAsync(lv.f[field], lv.init[field], lv.elems * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return EVOMG_OK;
  }
  int set_field(int level, int field, const double* host) override {
    if (!level_ok(level) || !host || !fieldptr(level, field)) return invalid("set_field: bad level/field/pointer");
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(upload(lv_[level], fieldptr(level, field), host));
    return EVOMG_OK;
  }
  int get_field(int level, int field, double* host) override {
    if (!level_ok(level) || !host || !fieldptr(level, field)) return invalid("get_field: bad level/field/pointer");
    HIP_TRY(hipSetDevice(device_));
    const FLevel& lv = lv_[level];
    HIP_TRY(hipStreamSynchronize(stream_));
    HIP_TRY(hipMemcpy2D(host, lv.g.n * sizeof(double), fieldptr(level, field), lv.g.pitch * sizeof(double), lv.g.n * sizeof(double), lv.g.n, hipMemcpyDeviceToHost));
    return EVOMG_OK;
  }
  int set_operators(const evomg_problem_t*) override { return unsupported("set_operators: not available on FAS problems"); }
  int run_ops(const evomg_op_t* ops, int n_ops) override {
    HIP_TRY(hipSetDevice(device_));
    int rc = validate(ops, n_ops);
    if (rc != EVOMG_OK) return rc;
    for (int i = 0; i < n_ops; ++i) if ((rc = exec(ops[i])) != EVOMG_OK) return rc;
    HIP_TRY(hipStreamSynchronize(stream_));
    return fas_health();
  }
  int norm(int level, int field, double* out) override {
    if (!level_ok(level) || !out || !fieldptr(level, field)) return invalid("norm: bad level/field/pointer");
    HIP_TRY(hipSetDevice(device_));
    double v;
    int rc = dot_to_host(level, fieldptr(level, field), &v);
    if (rc != EVOMG_OK) return rc;
    *out = sqrt(v);
    return EVOMG_OK;
  }
  // the template's Solve (FAS_2D_Basic_template.exa4:130-160): UpResidual, norm; repeat the cycle until the norm has dropped
  // below target * initial or max_iterations cycles have run
  int evaluate(const evomg_op_t* ops, int n_ops, int samples, double wall_limit_ms, evomg_result_t* out, double* hist, int cap) override {
    HIP_TRY(hipSetDevice(device_));
    if (!out || samples < 1 || n_ops < 0 || (!ops && n_ops > 0)) return invalid("evaluate: bad arguments");
    // The tiled coarse sweeps need their workgroups co-resident; with several candidates in flight on one GPU they can miss
    // a rendezvous. As for the scalar problems' eight-workgroup CG: a property of the moment, not of the candidate — quiesce
    // the device and evaluate once more, alone, before reporting a device error.
    int rc = EVOMG_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {
      fas_missed_ = false;
      rc = evaluate_attempt(ops, n_ops, samples, wall_limit_ms, out, hist, cap);
      if (!fas_missed_) return rc;
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipMemsetAsync(fas_info_, 0, sizeof(int), stream_));
    }
    return rc;
  }
  int evaluate_attempt(const evomg_op_t* ops, int n_ops, int samples, double wall_limit_ms, evomg_result_t* out, double* hist, int cap) {
    memset(out, 0, sizeof *out);
    int rc = validate(ops, n_ops);
    if (rc != EVOMG_OK) return rc;
    const int L = prob_.max_level;
    double tsum = 0.0;
    int done = 0;
    evomg_result_t r;
    for (int smp = 0; smp < samples; ++smp) {
      memset(&r, 0, sizeof r);
      for (int l = prob_.min_level; l <= prob_.max_level; ++l) {
        FLevel& lv = lv_[l];
        for (int k = 0; k < kFields; ++k) {
          if (lv.init[k]) HIP_TRY(hipMemcpyAsync(lv.f[k], lv.init[k], lv.elems * sizeof(double), hipMemcpyDeviceToDevice, stream_));
          else HIP_TRY(hipMemsetAsync(lv.f[k], 0, lv.elems * sizeof(double), stream_));
        }
        HIP_TRY(hipMemsetAsync(lv.next, 0, lv.elems * sizeof(double), stream_));
      }
      HIP_TRY(hipStreamSynchronize(stream_));
      int nh = 0;
      const double t0 = wall_ms();
      double res0 = 0.0;
      if ((rc = residual_norm(L, &res0)) != EVOMG_OK) return rc;
      if (hist && nh < cap) hist[nh] = res0;
      ++nh;
      double res = res0;
      int it = 0, status = EVOMG_SOLVE_MAXITER;
      if (res0 == 0.0) status = EVOMG_SOLVE_CONVERGED;
      else
        for (it = 1; it <= prob_.max_iterations; ++it) {
          for (int i = 0; i < n_ops; ++i) if ((rc = exec(ops[i])) != EVOMG_OK) return rc;
          if ((rc = residual_norm(L, &res)) != EVOMG_OK) return rc;
          if (hist && nh < cap) hist[nh] = res;
          ++nh;
          if (!(res == res) || res > 1.7e308) { status = EVOMG_SOLVE_DIVERGED; break; }
          if (res < prob_.target_res_reduction * res0) { status = EVOMG_SOLVE_CONVERGED; break; }
          if (wall_limit_ms > 0 && wall_ms() - t0 > wall_limit_ms) { status = EVOMG_SOLVE_WALL_LIMIT; break; }
        }
      if (it > prob_.max_iterations) it = prob_.max_iterations;
      r.time_ms = wall_ms() - t0;
      r.n_iterations = (double)it;
      r.init_res_norm = res0;
      r.final_res_norm = res;
      r.status = status;
      r.n_history = (nh < cap || !hist) ? nh : cap;
      r.conv_factor = (it > 0 && res0 > 0) ? pow(res / res0, 1.0 / (double)it) : 0.0;
      tsum += r.time_ms;
      ++done;
      if ((rc = fas_health()) != EVOMG_OK) return rc;
      if (status != EVOMG_SOLVE_CONVERGED) break;
    }
    *out = r;
    out->time_ms = tsum / (double)done;
    return EVOMG_OK;
  }
  int bench_op(const evomg_op_t*, int, int, evomg_bench_t*) override { return unsupported("bench_op: not available on FAS problems"); }
  int kernel_name(const evomg_op_t* op, char* buf, int len) override {
    if (!op || !buf || len < 8) return invalid("kernel_name: bad arguments");
    snprintf(buf, len, "%s", op->opcode == EVOMG_OP_FAS_SMOOTH || op->opcode == EVOMG_OP_FAS_CGS ? "evomg_fas_smooth" :
                             op->opcode == EVOMG_OP_FAS_RESIDUAL ? "evomg_fas_residual" : "generic");
    return EVOMG_OK;
  }
  int profile_begin(int, int) override { return EVOMG_OK; }
  int profile_end(evomg_bench_t* out) override { if (out) memset(out, 0, sizeof *out); return EVOMG_OK; }

 private:
  static constexpr int kFields = 4;      // EVOMG_F_SOL, _RHS, _RES, _CORR (= Approximation, a separate array on every level)
  struct FLevel {
    Geo g{};
    size_t elems = 0;
    double* f[kFields] = {};
    double* init[kFields] = {};
    double* next = nullptr;
    double* xchg = nullptr;      // exchange slots of evomg_fas_cgs_tiles
    Stencil<double> st{};
  };
  hipError_t alloc_zero(double** p, size_t elems) {
    hipError_t e = hipMalloc(p, elems * sizeof(double));
    if (e != hipSuccess) return e;
    return hipMemsetAsync(*p, 0, elems * sizeof(double), stream_);
  }
  hipError_t upload(const FLevel& lv, double* dst, const double* host) {
    hipError_t e = hipStreamSynchronize(stream_);
    if (e != hipSuccess) return e;
    return hipMemcpy2D(dst, lv.g.pitch * sizeof(double), host, lv.g.n * sizeof(double), lv.g.n * sizeof(double), lv.g.n, hipMemcpyHostToDevice);
  }
  bool level_ok(int L) const { return L >= prob_.min_level && L <= prob_.max_level; }
  double* fieldptr(int level, int field) { return (field < 0 || field >= kFields) ? nullptr : lv_[level].f[field]; }
  int invalid(const char* msg) { snprintf(err, sizeof err, "%s", msg); return EVOMG_ERR_INVALID; }
  int unsupported(const char* msg) { snprintf(err, sizeof err, "%s", msg); return EVOMG_ERR_UNSUPPORTED; }
  static dim3 blk() { return dim3(64, 4, 1); }
  static dim3 grd(int n) { return dim3((n + 63) / 64, (n + 3) / 4, 1); }

  // mirrors the oracle's fas_check_op
  int validate(const evomg_op_t* ops, int n_ops) {
    for (int i = 0; i < n_ops; ++i) {
      const evomg_op_t& op = ops[i];
      const int lo = prob_.min_level, hi = prob_.max_level, L = op.level;
      int rc = EVOMG_ERR_INVALID;
      bool ok = L >= lo && L <= hi;
      if (ok) switch (op.opcode) {
        case EVOMG_OP_FAS_SMOOTH:
          ok = op.block[0] >= 0 && op.block[0] <= 16 && (op.partitioning == EVOMG_PART_SINGLE || op.partitioning == EVOMG_PART_RED_BLACK);
          break;
        case EVOMG_OP_FAS_RESIDUAL: case EVOMG_OP_FAS_SUBTRACT: case EVOMG_OP_FAS_CGS: break;
        case EVOMG_OP_FAS_RESTRICT_SOLUTION: case EVOMG_OP_FAS_COARSE_RHS: ok = L < hi; break;
        case EVOMG_OP_PROLONG_ADD: ok = L > lo && op.dst == EVOMG_F_SOL && op.src0 == EVOMG_F_SOL; break;
        default: ok = false; rc = EVOMG_ERR_UNSUPPORTED;
      }
      if (!ok) { snprintf(err, sizeof err, "op %d (opcode %d, level %d) rejected", i, op.opcode, op.level); return rc; }
    }
    return EVOMG_OK;
  }
  void smooth(int L, double om, bool red_black, int n_newton) {
    FLevel& lv = lv_[L];
    const double gam = prob_.nonlinear_gamma;
    double* x = lv.f[EVOMG_F_SOL];
    const double* f = lv.f[EVOMG_F_RHS];
    if (red_black) {
      for (int colour = 0; colour < 2; ++colour)
        hipLaunchKernelGGL(evomg_fas_smooth, grd(lv.g.n), blk(), 0, stream_, lv.g, lv.st, gam, (const double*)x, f, x, om, n_newton > 0 ? 1 : 0, n_newton, colour);
      return;
    }
    hipLaunchKernelGGL(evomg_fas_smooth, grd(lv.g.n), blk(), 0, stream_, lv.g, lv.st, gam, (const double*)x, f, lv.next, om, n_newton > 0 ? 1 : 0, n_newton, -1);
    std::swap(lv.f[EVOMG_F_SOL], lv.next);        // advance Solution
  }
  int exec(const evomg_op_t& op) {
    const int L = op.level;
    FLevel& lv = lv_[L];
    const double gam = prob_.nonlinear_gamma;
    switch (op.opcode) {
      case EVOMG_OP_FAS_SMOOTH: smooth(L, op.omega, op.partitioning == EVOMG_PART_RED_BLACK, op.block[0]); break;
      case EVOMG_OP_FAS_RESIDUAL:
        hipLaunchKernelGGL(evomg_fas_residual, grd(lv.g.n), blk(), 0, stream_, lv.g, lv.st, gam, (const double*)lv.f[EVOMG_F_SOL], (const double*)lv.f[EVOMG_F_RHS], lv.f[EVOMG_F_RES]);
        break;
      case EVOMG_OP_FAS_RESTRICT_SOLUTION:
        hipLaunchKernelGGL(evomg_fas_restrict_solution, grd(lv.g.n), blk(), 0, stream_, lv.g, lv_[L + 1].g, (const double*)lv_[L + 1].f[EVOMG_F_SOL], lv.f[EVOMG_F_CORR], lv.f[EVOMG_F_SOL]);
        break;
      case EVOMG_OP_FAS_COARSE_RHS:
        hipLaunchKernelGGL(evomg_fas_coarse_rhs, grd(lv.g.n), blk(), 0, stream_, lv.g, lv_[L + 1].g, lv.st, gam, (const double*)lv_[L + 1].f[EVOMG_F_RES],
                           (const double*)lv.f[EVOMG_F_CORR], (const double*)lv.f[EVOMG_F_SOL], lv.f[EVOMG_F_RHS]);
        break;
      case EVOMG_OP_FAS_SUBTRACT:
        hipLaunchKernelGGL(evomg_fas_subtract, grd(lv.g.n), blk(), 0, stream_, lv.g, lv.f[EVOMG_F_SOL], (const double*)lv.f[EVOMG_F_CORR]);
        break;
      case EVOMG_OP_FAS_CGS:
        if (lv.g.n >= 65 && lv.g.n <= 129 && fas_cgs_tiles_ && prob_.cgs_max_iterations > 0) {
          // 65^2 (the reference's coarsest grid) and 129^2: overlapping tiles on 16 / 64 compute units (evomg_fas_cgs_tiles)
          const int groups = fas_tiles_per_side(lv.g.n) * fas_tiles_per_side(lv.g.n);
          if (!lv.xchg) HIP_TRY(hipMalloc(&lv.xchg, 2 * lv.elems * sizeof(double)));
          if (!fas_flags_) {
            HIP_TRY(hipMalloc(&fas_flags_, sizeof(unsigned) * 64));
            HIP_TRY(hipMalloc(&fas_info_, sizeof(int)));
            HIP_TRY(hipMemsetAsync(fas_info_, 0, sizeof(int), stream_));
            HIP_TRY(hipHostMalloc(&fas_info_host_, sizeof(int)));
            *fas_info_host_ = 0;
          }
          HIP_TRY(hipMemsetAsync(fas_flags_, 0, sizeof(unsigned) * 64, stream_));
          hipLaunchKernelGGL(evomg_fas_cgs_tiles, dim3(groups), dim3(1024), 0, stream_, lv.g, lv.st, gam, lv.f[EVOMG_F_SOL], lv.next,
                             (const double*)lv.f[EVOMG_F_RHS], prob_.fas_cgs_omega, prob_.cgs_max_iterations, lv.xchg, fas_flags_, fas_info_);
          fas_tiles_used_ = true;
          if (prob_.cgs_max_iterations & 1) std::swap(lv.f[EVOMG_F_SOL], lv.next);
        } else if (lv.g.n <= 65 && fas_cgs_one_launch_) {
          // all sweeps in one single-workgroup launch (evomg_fas.h); an odd count leaves the result in the other slot
          if (fas_cgs_lds_) {
            // both slots in LDS for the whole call (evomg_fas_cgs_lds)
            const size_t lds_bytes = 2 * (size_t)lv.g.n * lv.g.n * sizeof(double);
            if (!fas_lds_attr_set_) {
              HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(evomg_fas_cgs_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
              fas_lds_attr_set_ = true;
            }
            hipLaunchKernelGGL(evomg_fas_cgs_lds, dim3(1), dim3(1024), lds_bytes, stream_, lv.g, lv.st, gam, lv.f[EVOMG_F_SOL], lv.next,
                               (const double*)lv.f[EVOMG_F_RHS], prob_.fas_cgs_omega, prob_.cgs_max_iterations);
          } else
          hipLaunchKernelGGL(evomg_fas_cgs, dim3(1), dim3(1024), 0, stream_, lv.g, lv.st, gam, lv.f[EVOMG_F_SOL], lv.next,
                             (const double*)lv.f[EVOMG_F_RHS], prob_.fas_cgs_omega, prob_.cgs_max_iterations);
          if (prob_.cgs_max_iterations & 1) std::swap(lv.f[EVOMG_F_SOL], lv.next);
        } else {
          for (int s = 0; s < prob_.cgs_max_iterations; ++s) smooth(L, prob_.fas_cgs_omega, false, 1);
        }
        break;
      case EVOMG_OP_PROLONG_ADD:
        hipLaunchKernelGGL((evomg_prolong<double, 2, true>), grd(lv.g.n), blk(), 0, stream_, lv.g, lv_[L - 1].g, (const double*)lv_[L - 1].f[EVOMG_F_SOL], lv.f[EVOMG_F_SOL], op.omega);
        break;
      default: return invalid("exec: unknown opcode");
    }
    HIP_TRY(hipGetLastError());
    return EVOMG_OK;
  }
  int dot_to_host(int L, const double* a, double* out) {
    const Geo& g = lv_[L].g;
    const int m = g.n - 2;
    const int blocks = m < kDotBlocks ? m : kDotBlocks;
    hipLaunchKernelGGL((evomg_dot_partial<double, 2>), dim3(blocks), dim3(256), 0, stream_, g, a, a, partials_);
    hipLaunchKernelGGL((evomg_dot_final<double>), dim3(1), dim3(256), 0, stream_, partials_, blocks, dscalar_);
    HIP_TRY(hipMemcpyAsync(hscalar_, dscalar_, sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    *out = hscalar_[0];
    return EVOMG_OK;
  }
  int residual_norm(int L, double* out) {      // UpResidual@finest ; NormResidual_0@finest
    FLevel& lv = lv_[L];
    hipLaunchKernelGGL(evomg_fas_residual, grd(lv.g.n), blk(), 0, stream_, lv.g, lv.st, prob_.nonlinear_gamma, (const double*)lv.f[EVOMG_F_SOL],
                       (const double*)lv.f[EVOMG_F_RHS], lv.f[EVOMG_F_RES]);
    double d;
    const int rc = dot_to_host(L, lv.f[EVOMG_F_RES], &d);
    if (rc != EVOMG_OK) return rc;
    *out = sqrt(d);
    return EVOMG_OK;
  }

  evomg_problem_t prob_;
  int device_;
  bool fas_cgs_one_launch_ = getenv("EVOMG_NO_FAS_CGS_KERNEL") == nullptr;
  bool fas_cgs_lds_ = getenv("EVOMG_NO_FAS_CGS_LDS") == nullptr;      // ... with both slots in LDS
  bool fas_cgs_tiles_ = getenv("EVOMG_NO_FAS_CGS_TILES") == nullptr;  // 65^2 / 129^2: overlapping tiles on several compute units
  bool fas_tiles_used_ = false;
  bool fas_missed_ = false;           // the last health check saw a missed rendezvous
  unsigned* fas_flags_ = nullptr;
  int* fas_info_ = nullptr;
  int* fas_info_host_ = nullptr;
  // a missed rendezvous of the tiled coarse sweeps is an error, not a hang (call after a stream synchronisation point)
  int fas_health() {
    if (!fas_tiles_used_) return EVOMG_OK;
    HIP_TRY(hipMemcpyAsync(fas_info_host_, fas_info_, sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    if (*fas_info_host_ != 0) {
      fas_missed_ = true;
      snprintf(err, sizeof err, "FAS coarse sweeps: the workgroups did not rendezvous (GPU oversubscribed?)");
      return EVOMG_ERR_DEVICE;
    }
    return EVOMG_OK;
  }
  bool fas_lds_attr_set_ = false;
  FLevel lv_[EVOMG_MAX_LEVELS];
  hipStream_t stream_ = nullptr;
  double* partials_ = nullptr;
  double* dscalar_ = nullptr;
  double* hscalar_ = nullptr;
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
  if (p->nonlinear_kind != EVOMG_NONLINEAR_NONE) {
    if (p->nonlinear_kind != EVOMG_NONLINEAR_GAM_U_EXP_U || p->dimension != 2 || p->scalar_kind != EVOMG_REAL_F64 ||
        p->bc_kind != EVOMG_BC_DIRICHLET || p->outer_kind != EVOMG_OUTER_MG_ITERATION || p->cgs_kind != EVOMG_CGS_FAS_SMOOTHER ||
        p->n_unknowns > 1) {
      snprintf(err, len, "non-linear FAS: 2D real Dirichlet scalar problems, N(u) = gam u exp(u), smoother as coarse solver");
      return EVOMG_ERR_UNSUPPORTED;
    }
  } else if (p->cgs_kind == EVOMG_CGS_FAS_SMOOTHER) { snprintf(err, len, "EVOMG_CGS_FAS_SMOOTHER needs a non-linear problem"); return EVOMG_ERR_INVALID; }
  if (p->n_unknowns > 1) {
    if (p->n_unknowns > EVOMG_MAX_UNKNOWNS) { snprintf(err, len, "at most %d unknowns per node", EVOMG_MAX_UNKNOWNS); return EVOMG_ERR_UNSUPPORTED; }
    if (p->dimension != 2 || p->scalar_kind != EVOMG_REAL_F64 || p->bc_kind != EVOMG_BC_DIRICHLET ||
        p->outer_kind != EVOMG_OUTER_MG_ITERATION || p->cgs_kind != EVOMG_CGS_CG) {
      snprintf(err, len, "systems of PDEs: 2D real Dirichlet problems with the generated solver loop and CG only");
      return EVOMG_ERR_UNSUPPORTED;
    }
  }
  return EVOMG_OK;
}

}  // namespace

struct evomg_handle {
  ImplBase* impl;
};

extern "C" {

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
  if (problem->nonlinear_kind != EVOMG_NONLINEAR_NONE) impl = new FASImpl(*problem, device_id);
  else if (problem->n_unknowns > 1) impl = new SystemImpl(*problem, device_id);
  else if (problem->scalar_kind == EVOMG_COMPLEX_F64) impl = new Impl<cplx>(*problem, device_id);
  else impl = new Impl<double>(*problem, device_id);
  rc = impl->init();
  if (rc != EVOMG_OK) {
    snprintf(g_create_error, sizeof g_create_error, "%s", impl->err);
    delete impl;
    return rc;
  }
  *out = new evomg_handle{impl};
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
  return h ? h->impl->evaluate(ops, n_ops, evaluation_samples, wall_limit_ms, out, res_history, history_capacity)
           : EVOMG_ERR_INVALID;
}
int evomg_run_ops(evomg_handle_t* h, const evomg_op_t* ops, int n_ops) {
  if (!h || n_ops < 0 || (!ops && n_ops > 0)) return EVOMG_ERR_INVALID;
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

}  // extern "C"
