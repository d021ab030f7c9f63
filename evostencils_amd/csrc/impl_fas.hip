// impl_fas.hip — non-linear FAS problems (evomg_fas.h): own storage and a plain interpreter, like the systems of PDEs.
#include "evomg_runtime.h"
#include "evomg_kernels.h"
#include "evomg_fas.h"

#include <atomic>
#include <mutex>

namespace evomg_detail {

// evomg_fas_cgs_tiles needs its 16 / 64 workgroups co-resident. One launch always fits the chip; several at once — candidates
// in flight on one GPU, each with its own handle and stream — may not (12 x 64 workgroups of 1024 threads exceed what can be
// resident, and launches waiting for each other's peers would all time out). While more than one FAS handle lives in the
// process, a tile launch therefore runs under this mutex from launch to completion: tile kernels never overlap each other
// (they still overlap every other kernel, which always finishes). A miss that happens anyway (another process on the GPU)
// is answered by ONE re-evaluation on the single-workgroup / launch-per-sweep path, which cannot miss.
static std::mutex g_fas_tiles_mutex;
static std::atomic<int> g_fas_handles{0};

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
    if (fas_info_host_) (void)hipHostFree(fas_info_host_);
    if (counted_) --g_fas_handles;
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
    {
      int lds = 0;
      if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device_) == hipSuccess && lds > 0) max_lds_ = (size_t)lds;
    }
    for (int L = prob_.min_level; L <= prob_.max_level; ++L) {
      FLevel& lv = lv_[L];
      const int n = (1 << L) + 1;
      lv.g.n = n;
      lv.g.prio = 0;
      lv.g.pitch = ((n + 15) / 16) * 16;
      lv.g.plane = 0;
      lv.elems = (size_t)lv.g.pitch * n;
      for (int e = 0; e < EVOMG_STENCIL_ENTRIES; ++e) lv.st.c[e] = prob_.mg_stencil[L][e].re;
      for (int k = 0; k < kFields; ++k) HIP_TRY(alloc_zero(&lv.f[k], lv.elems));     // Solution, RHS, Residual, Approximation
      HIP_TRY(alloc_zero(&lv.next, lv.elems));                                          // Solution<next>
    }
    // the tiled coarse sweeps' exchange slots, sequence flags and miss flag — allocated here, not inside the first timed solve.
    // The miss flag lives in pinned host memory the kernel writes directly (only on a miss): reading it after any stream
    // synchronisation costs nothing, so the solver loop can check it every iteration.
    {
      FLevel& lc = lv_[prob_.min_level];
      if (lc.g.n >= 65 && lc.g.n <= 129) HIP_TRY(hipMalloc(&lc.xchg, 2 * lc.elems * sizeof(double)));
      HIP_TRY(hipMalloc(&fas_flags_, sizeof(unsigned) * 64));
      HIP_TRY(hipHostMalloc(&fas_info_host_, sizeof(int), hipHostMallocMapped));
      *fas_info_host_ = 0;
      HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&fas_info_), fas_info_host_, 0));
    }
    ++g_fas_handles;
    counted_ = true;
    HIP_TRY(wait_stream(stream_));
    return EVOMG_OK;
  }
  int set_initial_field(int level, int field, const double* host) override {
    if (!level_ok(level) || !host || !fieldptr(level, field)) return invalid("set_initial_field: bad level/field/pointer");
    HIP_TRY(hipSetDevice(device_));
    FLevel& lv = lv_[level];
    if (!lv.init[field]) HIP_TRY(alloc_zero(&lv.init[field], lv.elems));
    HIP_TRY(upload(lv, lv.init[field], host));
    HIP_TRY(hipMemcpyAsync(lv.f[field], lv.init[field], lv.elems * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    HIP_TRY(wait_stream(stream_));
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
    HIP_TRY(wait_stream(stream_));
    HIP_TRY(hipMemcpy2D(host, lv.g.n * sizeof(double), fieldptr(level, field), lv.g.pitch * sizeof(double), lv.g.n * sizeof(double), lv.g.n, hipMemcpyDeviceToHost));
    return EVOMG_OK;
  }
  int set_operators(const evomg_problem_t*) override { return unsupported("set_operators: not available on FAS problems"); }
  int run_ops(const evomg_op_t* ops, int n_ops) override {
    HIP_TRY(hipSetDevice(device_));
    int rc = validate(ops, n_ops);
    if (rc != EVOMG_OK) return rc;
    HIP_TRY(wait_stream(stream_));
    *fas_info_host_ = 0;                  // a miss of an earlier call must not fail this one
    fas_missed_ = false;
    for (int i = 0; i < n_ops; ++i) if ((rc = exec(ops[i])) != EVOMG_OK) return rc;
    HIP_TRY(wait_stream(stream_));
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
      HIP_TRY(wait_stream(stream_));
      *fas_info_host_ = 0;
      fas_missed_ = false;
      fas_tiles_off_ = attempt > 0;        // the second attempt: single-workgroup / launch-per-sweep coarse sweeps (same bits)
      rc = evaluate_attempt(ops, n_ops, samples, wall_limit_ms, out, hist, cap);
      fas_tiles_off_ = false;
      if (!fas_missed_) return rc;
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
      HIP_TRY(wait_stream(stream_));
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
          if ((rc = fas_health()) != EVOMG_OK) return rc;      // after the norm's stream synchronisation: a plain host read
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
  evomg_problem_t& problem() override { return prob_; }
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
  FasNl nl() const {
    FasNl v;
    v.kind = prob_.nonlinear_kind;
    v.gam = prob_.nonlinear_gamma;
    v.prog = prob_.nonlinear_program;
    return v;
  }
  void smooth(int L, double om, bool red_black, int n_newton) {
    FLevel& lv = lv_[L];
    const FasNl gam = nl();      // the non-linear stencil entry (evomg_fas.h)
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
    const FasNl gam = nl();      // the non-linear stencil entry (evomg_fas.h)
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
        if (lv.g.n >= 65 && lv.g.n <= 129 && fas_cgs_tiles_ && !fas_tiles_off_ && lv.xchg && prob_.cgs_max_iterations > 0) {
          // 65^2 (the reference's coarsest grid) and 129^2: overlapping tiles on 16 / 64 compute units (evomg_fas_cgs_tiles)
          const int groups = fas_tiles_per_side(lv.g.n) * fas_tiles_per_side(lv.g.n);
          const bool exclusive = g_fas_handles.load() > 1;      // other FAS handles in this process: one tile launch at a time
          if (exclusive) g_fas_tiles_mutex.lock();
          hipError_t e = hipMemsetAsync(fas_flags_, 0, sizeof(unsigned) * 64, stream_);
          if (e == hipSuccess) {
            hipLaunchKernelGGL(evomg_fas_cgs_tiles, dim3(groups), dim3(1024), 0, stream_, lv.g, lv.st, gam, lv.f[EVOMG_F_SOL], lv.next,
                               (const double*)lv.f[EVOMG_F_RHS], prob_.fas_cgs_omega, prob_.cgs_max_iterations, lv.xchg, fas_flags_, fas_info_);
            e = hipGetLastError();
          }
          if (exclusive) {
            if (e == hipSuccess) e = hipStreamSynchronize(stream_);
            g_fas_tiles_mutex.unlock();
          }
          HIP_TRY(e);
          fas_tiles_used_ = true;
          if (prob_.cgs_max_iterations & 1) std::swap(lv.f[EVOMG_F_SOL], lv.next);
        } else if (lv.g.n <= 65 && fas_cgs_one_launch_) {
          // all sweeps in one single-workgroup launch (evomg_fas.h); an odd count leaves the result in the other slot
          const size_t lds_bytes = 2 * (size_t)lv.g.n * lv.g.n * sizeof(double);
          if (fas_cgs_lds_ && lds_bytes <= max_lds_) {
            // both slots in LDS for the whole call (evomg_fas_cgs_lds)
            if (!fas_lds_attr_set_) {
              HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(evomg_fas_cgs_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_));
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
    HIP_TRY(wait_stream(stream_));
    *out = hscalar_[0];
    return EVOMG_OK;
  }
  int residual_norm(int L, double* out) {      // UpResidual@finest ; NormResidual_0@finest
    FLevel& lv = lv_[L];
    hipLaunchKernelGGL(evomg_fas_residual, grd(lv.g.n), blk(), 0, stream_, lv.g, lv.st, nl(), (const double*)lv.f[EVOMG_F_SOL],
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
  bool fas_tiles_off_ = false;        // set for the re-evaluation after a miss: the paths that cannot miss
  bool counted_ = false;              // this handle is in g_fas_handles
  unsigned* fas_flags_ = nullptr;
  int* fas_info_ = nullptr;           // device address of fas_info_host_ (pinned, mapped)
  int* fas_info_host_ = nullptr;
  // a missed rendezvous of the tiled coarse sweeps is an error, not a hang (call after a stream synchronisation point: the
  // kernel writes the flag straight into pinned host memory)
  int fas_health() {
    if (!fas_tiles_used_) return EVOMG_OK;
    if (*static_cast<volatile int*>(fas_info_host_) != 0) {
      fas_missed_ = true;
      snprintf(err, sizeof err, "FAS coarse sweeps: the workgroups did not rendezvous (GPU oversubscribed?)");
      return EVOMG_ERR_DEVICE;
    }
    return EVOMG_OK;
  }
  bool fas_lds_attr_set_ = false;
  size_t max_lds_ = 64 * 1024;        // hipDeviceAttributeMaxSharedMemoryPerBlock, read at init
  FLevel lv_[EVOMG_MAX_LEVELS];
  hipStream_t stream_ = nullptr;
  double* partials_ = nullptr;
  double* dscalar_ = nullptr;
  double* hscalar_ = nullptr;
};

ImplBase* make_impl_fas(const evomg_problem_t& p, int device) { return new FASImpl(p, device); }

}  // namespace evomg_detail
