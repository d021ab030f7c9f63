// impl_system.hip — systems of PDEs (evomg_system.h): own storage and a plain interpreter, nothing shared with the tuned scalar path.
#include "evomg_runtime.h"
#include "evomg_kernels.h"
#include "evomg_system.h"

namespace evomg_detail {

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
      lv.g.prio = 0;
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
    HIP_TRY(wait_stream(stream_));
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
    HIP_TRY(wait_stream(stream_));
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
    HIP_TRY(wait_stream(stream_));
    HIP_TRY(hipMemcpy2D(host, lv.g.n * sizeof(double), d, lv.g.pitch * sizeof(double), lv.g.n * sizeof(double), lv.g.n, hipMemcpyDeviceToHost));
    return EVOMG_OK;
  }
  int set_operators(const evomg_problem_t*) override { return unsupported("set_operators: not available on systems of PDEs"); }
  int run_ops(const evomg_op_t* ops, int n_ops) override {
    HIP_TRY(hipSetDevice(device_));
    int rc = validate(ops, n_ops);
    if (rc != EVOMG_OK) return rc;
    if ((rc = exec_all(ops, n_ops)) != EVOMG_OK) return rc;
    HIP_TRY(wait_stream(stream_));
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
      HIP_TRY(wait_stream(stream_));
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
  evomg_problem_t& problem() override { return prob_; }
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

  // the unknowns a SMOOTH lists; a scalar op program (mask 0) on a system of ONE unknown — a scalar problem with a compact
  // nine-point stencil, EVOMG_STENCIL_COMPACT — means that unknown
  int mask_of(const evomg_op_t& op) const { return op.unknowns ? op.unknowns : (nf_ == 1 ? 1 : 0); }

  // mirrors the oracle's sys_check_op
  int validate(const evomg_op_t* ops, int n_ops) {
    for (int i = 0; i < n_ops; ++i) {
      const evomg_op_t& op = ops[i];
      const int lo = prob_.min_level, hi = prob_.max_level;
      int rc = EVOMG_ERR_INVALID;
      bool ok = op.level >= lo && op.level <= hi && fieldptr(op.level, op.dst) != nullptr;
      if (ok) switch (op.opcode) {
        case EVOMG_OP_SMOOTH: {
          const int mask = mask_of(op);
          ok = fieldptr(op.level, op.src1) != nullptr && mask > 0 && mask < (1 << nf_) && op.block[0] >= 1 && op.block[1] >= 1;
          if (!ok) break;
          int k = 0, first = 0;
          for (int u = 0; u < nf_; ++u) k += (mask >> u) & 1;
          while (!((mask >> first) & 1)) ++first;
          const bool other = op.block_other[0] != 0 || op.block_other[1] != 0;       // the listed unknowns after the first: their own block shapes
          if (other && (k < 2 || op.block_other[0] < 1 || op.block_other[1] < 1)) { ok = false; break; }
          if (op.block_third != 0 && (k < 3 || !other || (op.block_third & 0xff) < 1 || ((op.block_third >> 8) & 0xff) < 1)) { ok = false; break; }
          int cells = 0;
          for (int q = 0; q < k; ++q) {
            int sx, sy;
            listed_shape(op, q, &sx, &sy);
            cells += sx * sy;
          }
          ok = cells <= EVOMG_MAX_BLOCK_CELLS && op.dst / EVOMG_F_COUNT == first && op.src1 / EVOMG_F_COUNT == first &&
               (op.partitioning == EVOMG_PART_SINGLE || op.partitioning == EVOMG_PART_RED_BLACK);
          if (ok && op.partitioning == EVOMG_PART_RED_BLACK && cells != k) { ok = false; rc = EVOMG_ERR_UNSUPPORTED; }
          // Differing per-unknown shapes: how ExaStencils tiles the grid around such a `solve locally` is not visible from
          // the reference ([EXTERNAL]; the bounding-box tiling used here leaves nodes of the smaller shape unrelaxed, the IR's
          // block_diagonal(A_ij, shape_i) repeats every unknown with its own period). Where the sweep's RESULT is used
          // (textbook semantics) the candidate is therefore rejected; under the reference pipeline's semantics the sweep is
          // launched for its time only and its output never becomes the field, so rho and the iteration count are exact.
          if (ok && other && prob_.jacobi_semantics == EVOMG_JACOBI_TEXTBOOK) {
            bool differ = false;
            for (int q = 1; q < k; ++q) {
              int sx, sy;
              listed_shape(op, q, &sx, &sy);
              differ = differ || sx != op.block[0] || sy != op.block[1];
            }
            if (differ) { ok = false; rc = EVOMG_ERR_UNSUPPORTED; }
          }
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

  // block shape of the q-th LISTED unknown of a SMOOTH (include/evomg.h: block, block_other, block_third)
  static void listed_shape(const evomg_op_t& op, int q, int* sx, int* sy) {
    *sx = op.block[0];
    *sy = op.block[1];
    if (q >= 1 && op.block_other[0] > 0) { *sx = op.block_other[0]; *sy = op.block_other[1]; }
    if (q >= 2 && op.block_third != 0) { *sx = op.block_third & 0xff; *sy = (op.block_third >> 8) & 0xff; }
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
        const int mask = mask_of(op), inner = g.n - 2;
        // per-unknown block shapes: the first unknown of the mask has `block`, the second `block_other`, the third
        // `block_third` where those are set (listed_shape)
        SysShapes sh{};
        int bx = 1, by = 1, seen = 0;
        for (int u = 0; u < nf_; ++u) {
          if (!((mask >> u) & 1)) { sh.bx[u] = sh.by[u] = 0; continue; }
          listed_shape(op, seen++, &sh.bx[u], &sh.by[u]);
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
  // ops[i .. i + nf): the same statement for unknown 0, 1, ... (what a cycle emits back to back), independent of each other
  // -> one launch with blockIdx.z = unknown (evomg_system.h)
  bool same_statement_for_both_unknowns(const evomg_op_t& a, const evomg_op_t& b, int u) const {
    if (a.opcode != b.opcode || a.level != b.level) return false;
    if (a.dst / EVOMG_F_COUNT != 0 || b.dst != a.dst + EVOMG_F_COUNT * u) return false;
    switch (a.opcode) {
      case EVOMG_OP_RESIDUAL:        // row i reads every unknown's solution slot (src0 names unknown 0's) and its own rhs
        return b.src0 == a.src0 && a.src1 / EVOMG_F_COUNT == 0 && b.src1 == a.src1 + EVOMG_F_COUNT * u &&
               canon(a.level, a.dst) % EVOMG_F_COUNT != canon(a.level, a.src0) % EVOMG_F_COUNT &&
               canon(a.level, a.dst) % EVOMG_F_COUNT != canon(a.level, a.src1) % EVOMG_F_COUNT;
      case EVOMG_OP_RESTRICT: return a.src0 / EVOMG_F_COUNT == 0 && b.src0 == a.src0 + EVOMG_F_COUNT * u;
      case EVOMG_OP_PROLONG_ADD: case EVOMG_OP_PROLONG_SET:
        return a.src0 / EVOMG_F_COUNT == 0 && b.src0 == a.src0 + EVOMG_F_COUNT * u && a.omega == b.omega;
      case EVOMG_OP_SET_ZERO: return true;
      default: return false;
    }
  }
  bool same_statement_for_all_unknowns(const evomg_op_t* ops, int left) const {
    if (!sys_pair_fuse_ || nf_ < 2 || left < nf_) return false;
    for (int u = 1; u < nf_; ++u)
      if (!same_statement_for_both_unknowns(ops[0], ops[u], u)) return false;
    return true;
  }
  int exec_pair(const evomg_op_t& op) {
    const int L = op.level;
    const Geo& g = lv_[L].g;
    const dim3 grid2((g.n + 63) / 64, (g.n + 3) / 4, nf_);
    SysFields d{};
    for (int u = 0; u < nf_; ++u) d.p[u] = fieldptr(L, op.dst + EVOMG_F_COUNT * u);
    switch (op.opcode) {
      case EVOMG_OP_RESIDUAL: {
        const SysConstFields x = slot_fields(L, canon(L, op.src0) % EVOMG_F_COUNT);
        const SysConstFields f = slot_fields(L, op.src1 % EVOMG_F_COUNT);
        hipLaunchKernelGGL(evomg_sys_residual2, grid2, blk(), 0, stream_, g, lv_[L].st, nf_, x, f, d);
        break;
      }
      case EVOMG_OP_RESTRICT: {
        SysConstFields s{};
        for (int u = 0; u < nf_; ++u) s.p[u] = fieldptr(L + 1, op.src0 + EVOMG_F_COUNT * u);
        hipLaunchKernelGGL(evomg_sys_restrict2, grid2, blk(), 0, stream_, g, lv_[L + 1].g, s, d);
        break;
      }
      case EVOMG_OP_PROLONG_ADD: case EVOMG_OP_PROLONG_SET: {
        SysConstFields s{};
        for (int u = 0; u < nf_; ++u) s.p[u] = fieldptr(L - 1, op.src0 + EVOMG_F_COUNT * u);
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
      if (same_statement_for_all_unknowns(ops + i, n_ops - i)) {
        if ((rc = exec_pair(ops[i])) != EVOMG_OK) return rc;
        i += nf_ - 1;
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
    HIP_TRY(wait_stream(stream_));
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

ImplBase* make_impl_system(const evomg_problem_t& p, int device) { return new SystemImpl(p, device); }

}  // namespace evomg_detail
