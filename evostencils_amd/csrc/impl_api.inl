// impl_api.inl — the public operations behind the C-ABI: lifetime, field transfer, evaluate, bench, profiling
// Part of the body of `template <typename T> class Impl` (evomg_impl.h includes this file INSIDE the class).

  Impl(const evomg_problem_t& p, int device) : prob_(p), device_(device), dim_(p.dimension) {}

  ~Impl() override {
    (void)hipSetDevice(device_);
    drop_shared_rhs();
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
    if (xf_rhs_spare_) (void)hipFree(xf_rhs_spare_);
    if (xf_lane_sums_) (void)hipFree(xf_lane_sums_);
    if (norm_ticket_) (void)hipFree(norm_ticket_);
    if (cg_ll_) (void)hipFree(cg_ll_);
    if (kcg_state_) (void)hipFree(kcg_state_);
    if (kb_state_) (void)hipFree(kb_state_);
    if (kb_scal_) (void)hipFree(kb_scal_);
    if (kb_host_) (void)hipHostFree(kb_host_);
    if (kcg_host_) (void)hipHostFree(kcg_host_);
    if (hscalar_) (void)hipHostFree(hscalar_);
    if (hnorm_) (void)hipHostFree(hnorm_);
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
    // LDS a workgroup may ask for (gfx950: 160 KiB). The LDS-resident paths (small 3D level visits, 2D tiles) fall back to the
    // per-op launches on a device that offers less instead of failing the launch
    {
      int lds = 0;
      if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device_) == hipSuccess && lds > 0) max_lds_ = (size_t)lds;
    }
    HIP_TRY(hipEventCreate(&ev0_));
    HIP_TRY(hipEventCreate(&ev1_));
    HIP_TRY(hipMalloc(&partials_, sizeof(T) * 5 * kDotBlocks));   // two dot products per fused pass; five sets in kb_*
    HIP_TRY(hipMalloc(&dscalar_, sizeof(T) * 8));
    HIP_TRY(hipMalloc(&dinfo_, sizeof(int) * 4));
    HIP_TRY(hipMemset(dinfo_, 0, sizeof(int) * 4));
    // multi-workgroup CG exchange area: [2][G] partial sums, [G][2][1024] face planes, [2][G] sequence flags
    HIP_TRY(hipMalloc(&cg_ll_, sizeof(unsigned long long) * EVOMG_CG_LL_WORDS));
    HIP_TRY(hipMemset(cg_ll_, 0, sizeof(unsigned long long) * EVOMG_CG_LL_WORDS));
    HIP_TRY(hipHostMalloc(&hscalar_, sizeof(T) * 8));
    HIP_TRY(hipHostMalloc(&hnorm_, sizeof(T) * 8, hipHostMallocMapped));
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&hnorm_dev_), hnorm_, 0) != hipSuccess || !hnorm_dev_) host_norm_ = false;
    const int align = 128 / (int)sizeof(T);
    for (int L = prob_.min_level; L <= prob_.max_level; ++L) {
      Level<T>& lv = lv_[L];
      const int n = (1 << L) + 1;
      lv.g.n = n;
      lv.g.pitch = ((n + align - 1) / align) * align;
      lv.g.plane = (dim_ == 3) ? (long)lv.g.pitch * n : 0;
      lv.g.prio = (L < prob_.max_level) ? 1 : 0;       // evomg_kernels.h: prio_level
      lv.elems = (size_t)lv.g.pitch * n * (dim_ == 3 ? n : 1);
      for (int f = 0; f < EVOMG_F_COUNT; ++f) { bnd_zero_[L][f] = true; dirty_[L][f] = true; clean_ptr_[L][f] = nullptr; }
      for (int f : {EVOMG_F_SOL, EVOMG_F_RHS, EVOMG_F_RES}) HIP_TRY(alloc_zero(&lv.f[f], lv.elems));
      if (L == prob_.max_level) HIP_TRY(alloc_zero(&lv.f[EVOMG_F_CORR], lv.elems));
      if (L == prob_.min_level) HIP_TRY(alloc_zero(&lv.f[EVOMG_F_CGS_RHS], lv.elems));
      HIP_TRY(alloc_zero(&lv.alt, lv.elems));
    }
    load_stencils(prob_);
    HIP_TRY(wait_stream(stream_));
    return EVOMG_OK;
  }

  int set_initial_field(int level, int field, const double* host) override {
    if (!level_ok(level) || !host) return invalid("set_initial_field: bad level/pointer");
    HIP_TRY(hipSetDevice(device_));
    field = canon(level, field);
    Level<T>& lv = lv_[level];
    if (field < 0 || !lv.f[field]) return invalid("set_initial_field: no such field on this level");
    const bool rhs_max = level == prob_.max_level && field == EVOMG_F_RHS;
    if (rhs_max) drop_shared_rhs();           // (a second upload: the handle owns plain arrays again before anything else)
    if (!lv.init[field]) HIP_TRY(alloc_zero(&lv.init[field], lv.elems));
    bnd_zero_[level][field] = false;
    dirty_[level][field] = true;
    lv.alt_mirror = nullptr;
    HIP_TRY(upload(lv, lv.init[field], host));
    // The finest level's right-hand side is read-only to a multigrid solve and the same for every candidate of a problem: the
    // handles of a process read ONE device copy of it (evomg_runtime.h). Not under the preconditioned outer solver, whose
    // cycles use RHS@max as a work array.
    if (rhs_max && share_rhs_ && prob_.outer_kind == EVOMG_OUTER_MG_ITERATION) {
      HIP_TRY(wait_stream(stream_));
      const size_t host_bytes = (size_t)lv.g.n * lv.g.n * (dim_ == 3 ? lv.g.n : 1) * sizeof(T);
      T* mine = lv.init[field];
      T* shared = static_cast<T*>(shared_array_acquire(device_, lv.elems * sizeof(T), hash_words(host, host_bytes), mine));
      if (shared != mine) (void)hipFree(mine);
      lv.init[field] = shared;
      rhs_shared_ = true;
      rhs_private_ = lv.f[field];
      lv.f[field] = shared;
      rhs_aliased_ = true;
      dirty_[level][field] = false;
      clean_ptr_[level][field] = shared;
      return EVOMG_OK;
    }
    HIP_TRY(hipMemcpyAsync(lv.f[field], lv.init[field], lv.elems * sizeof(T), hipMemcpyDeviceToDevice, stream_));
    HIP_TRY(wait_stream(stream_));
    return EVOMG_OK;
  }
  // RHS@max is about to be written (set_field, an op program that names it as a destination): the handle's own array takes
  // over, with the shared array's content
  int privatize_rhs() {
    if (!rhs_aliased_) return EVOMG_OK;
    Level<T>& lv = lv_[prob_.max_level];
    HIP_TRY(hipMemcpyAsync(rhs_private_, lv.init[EVOMG_F_RHS], lv.elems * sizeof(T), hipMemcpyDeviceToDevice, stream_));
    lv.f[EVOMG_F_RHS] = rhs_private_;
    rhs_private_ = nullptr;
    rhs_aliased_ = false;
    clean_ptr_[prob_.max_level][EVOMG_F_RHS] = lv.f[EVOMG_F_RHS];
    lv.alt_mirror = nullptr;
    return EVOMG_OK;
  }
  // plain ownership again (before a new upload, before the destructor's frees)
  void drop_shared_rhs() {
    Level<T>& lv = lv_[prob_.max_level];
    if (rhs_aliased_) {
      lv.f[EVOMG_F_RHS] = rhs_private_;
      rhs_private_ = nullptr;
      rhs_aliased_ = false;
      dirty_[prob_.max_level][EVOMG_F_RHS] = true;
      clean_ptr_[prob_.max_level][EVOMG_F_RHS] = nullptr;
    }
    if (rhs_shared_) {
      shared_array_release(lv.init[EVOMG_F_RHS]);
      lv.init[EVOMG_F_RHS] = nullptr;
      rhs_shared_ = false;
    }
  }

  int set_field(int level, int field, const double* host) override {
    if (!level_ok(level) || !host) return invalid("set_field: bad level/pointer");
    HIP_TRY(hipSetDevice(device_));
    { const int rc = flush_pending_residual(); if (rc != EVOMG_OK) return rc; }
    if (level == prob_.max_level && canon(level, field) == EVOMG_F_RHS) { const int rc = privatize_rhs(); if (rc != EVOMG_OK) return rc; }
    T* d = fieldptr(level, field);
    if (!d) return invalid("set_field: no such field on this level");
    bnd_zero_[level][canon(level, field)] = false;
    dirty_[level][canon(level, field)] = true;
    lv_[level].alt_mirror = nullptr;
    HIP_TRY(upload(lv_[level], d, host));
    HIP_TRY(wait_stream(stream_));
    return EVOMG_OK;
  }

  int get_field(int level, int field, double* host) override {
    if (!level_ok(level) || !host) return invalid("get_field: bad level/pointer");
    HIP_TRY(hipSetDevice(device_));
    const T* d = fieldptr(level, field);
    if (!d) return invalid("get_field: no such field on this level");
    const Level<T>& lv = lv_[level];
    const size_t rows = (size_t)lv.g.n * (dim_ == 3 ? lv.g.n : 1);
    { const int rc = flush_pending_residual(); if (rc != EVOMG_OK) return rc; }
    d = fieldptr(level, field);
    HIP_TRY(wait_stream(stream_));
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
    if ((rc = flush_pending_residual()) != EVOMG_OK) return rc;
    info_fresh_ = false;
    if (cg_multi_used_) HIP_TRY(hipMemsetAsync(dinfo_, 0, sizeof(int) * 4, stream_));
    mark_written(ops, n_ops, /*loop=*/false);
    if (dirty_[prob_.max_level][EVOMG_F_RHS] && (rc = privatize_rhs()) != EVOMG_OK) return rc;
    rc = exec_program(ops, n_ops, /*cyclic=*/false);
    if (rc != EVOMG_OK) return rc;
    HIP_TRY(wait_stream(stream_));
    return coarse_solver_health();
  }

  // the multi-workgroup coarse solver reports a missed rendezvous in info[1] instead of hanging
  int coarse_solver_health() {
    if (!cg_multi_used_) return EVOMG_OK;
    int info[4] = {0, 0, 0, 0};
    if (info_fresh_) memcpy(info, last_info_, sizeof info);      // read behind the evaluation a moment ago: no second copy
    else HIP_TRY(hipMemcpy(info, dinfo_, sizeof info, hipMemcpyDeviceToHost));
    info_fresh_ = false;
    if (info[1] != 0) {
      snprintf(err, sizeof err, "coarse-grid CG: its workgroups did not rendezvous (GPU oversubscribed?)");
      cg_local_exchange_ = false;       // whatever the cause: the retry (and everything after it) exchanges over the fabric
      return EVOMG_ERR_DEVICE;
    }
    return EVOMG_OK;
  }

  int norm(int level, int field, double* out) override {
    if (!level_ok(level) || !out) return invalid("norm: bad level/pointer");
    HIP_TRY(hipSetDevice(device_));
    { const int rc = flush_pending_residual(); if (rc != EVOMG_OK) return rc; }
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
      if (attempt == 0) {
        if (getenv("EVOMG_LOG_RETRIES")) fprintf(stderr, "evomg: coarse-grid rendezvous missed with %d evaluations in flight; evaluating once more, alone\n", evaluations_in_flight());
        HIP_TRY(hipDeviceSynchronize());
      }
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
      HIP_TRY(hipMemsetAsync(dinfo_ + 2, 0, sizeof(int), stream_));      // coarse-solver iterations of this sample
      mark_written(ops, n_ops, /*loop=*/true);
      if (dirty_[prob_.max_level][EVOMG_F_RHS] && (rc = privatize_rhs()) != EVOMG_OK) return rc;
      if (prob_.outer_kind == EVOMG_OUTER_PBICGSTAB) rc = solve_pbicgstab(ops, n_ops, wall_limit_ms, &r, hist, cap);
      else rc = solve_mg(ops, n_ops, wall_limit_ms, &r, hist, cap);
      if (rc != EVOMG_OK) return rc;
      tsum += r.time_ms;
      ++done;
      if (r.status != EVOMG_SOLVE_CONVERGED) break;
    }
    *out = r;
    out->time_ms = tsum / (double)done;
    // behind the solve's own synchronisation, outside its timer: the coarse solver's words in ONE copy (iterations of all
    // coarse solves of the last sample; the rendezvous flag coarse_solver_health looks at)
    HIP_TRY(hipMemcpy(last_info_, dinfo_, sizeof last_info_, hipMemcpyDeviceToHost));
    coarse_iterations_ = last_info_[2];
    info_fresh_ = true;
    return EVOMG_OK;
  }
  long long coarse_iterations() override { return coarse_iterations_; }
  bool counts_coarse_iterations() const override { return true; }

  int bench_op(const evomg_op_t* op, int warmup, int reps, evomg_bench_t* out) override {
    HIP_TRY(hipSetDevice(device_));
    if (!op || !out || reps < 1 || warmup < 0) return invalid("bench_op: bad arguments");
    int rc = validate(op, 1);
    if (rc != EVOMG_OK) return rc;
    if ((rc = flush_pending_residual()) != EVOMG_OK) return rc;
    if (op->level == prob_.max_level && canon(op->level, op->dst) == EVOMG_F_RHS && (rc = privatize_rhs()) != EVOMG_OK) return rc;
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
        if (cells == 1 && op->partitioning == EVOMG_PART_RED_BLACK && rbgs3d_block_ok(lv_[op->level].g)) {
          snprintf(buf, len, "evomg_rbgs3d_block");
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
                 : (dim_ == 2 && kcg_enabled_ && n >= kb_min_n_) ? (kb_fused_ ? "kb_init / kb2_ps / kb2_hst / kb_xr / kb2_finish" : "kb_init / kb_p / kb_spmv / kb_hs / kb_xr / kb_check")
                                                                  : "evomg_bicgstab_single_wg";
        } else if (cg1_serves(lv_[prob_.min_level].g)) base = "evomg_cg1_band2d";
        else if (real_cg && dim_ == 3 && krylov_cg_groups<3>(n) > 1) base = prob_.cgs_kind == EVOMG_CGS_CG ? "evomg_cg1_slab3d" : "evomg_cg_multi_wg3d";
        else if (real_cg && kcg_enabled_ && (dim_ == 2 ? krylov_cg_per_launch<2>(n) : krylov_cg_per_launch<3>(n))) base = "kcg_init / kcg_apply / kcg_update / kcg_direct";
        else if (real_cg && use_reg_krylov_ && dim_ == 2 && krylov_own<2>(n, true).wave) base = "evomg_cg_wave2d";
        else if (real_cg && use_reg_krylov_ && dim_ == 2 && krylov_own<2>(n, true).tile && krylov_own<2>(n, true).lanes == 512) base = "evomg_cg_reg2d";
        else if (real_cg && use_reg_krylov_ && dim_ == 3 && krylov_own<3>(n, true).column && krylov_own<3>(n, true).m <= EVOMG_CG_WAVE3D_MAX_M) base = "evomg_cg_wave3d";
        else base = "evomg_cg_single_wg";
        snprintf(buf, len, "%s", base);          // the solvers' template arguments are tile shapes, not <type, dimension>
        return EVOMG_OK;
      }
    }
    snprintf(buf, len, "%s<%s, %d", base, type_name<T>::get(), dim_);
    return EVOMG_OK;
  }

  evomg_problem_t& problem() override { return prob_; }
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
    HIP_TRY(wait_stream(stream_));
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
