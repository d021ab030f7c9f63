// impl_solvers.inl — reductions, field reset and the two outer solver loops (generated MG iteration, preconditioned BiCGStab)
// Part of the body of `template <typename T> class Impl` (evomg_impl.h includes this file INSIDE the class).

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
    HIP_TRY(wait_stream(stream_));
    *out = hscalar_[0];
    return EVOMG_OK;
  }

  // Restores every field to its initial content before a solve — except the fields nothing has written since the last
  // restore (dirty_): the right-hand side of the finest level and gen_error@max are never written by an MG-iteration solve, and
  // restoring them anyway was 400 MB of the ~10 GB a C2 solve moves (with candidates in flight the GPU is bandwidth bound).
  // the residual field of the finest level, owed by the last solve (solve_mg)
  int flush_pending_residual() {
    if (!res_pending_) return EVOMG_OK;
    res_pending_ = false;
    const int L = prob_.max_level;
    residual(L, lv_[L].st, fieldptr(L, EVOMG_F_SOL), fieldptr(L, EVOMG_F_RHS), fieldptr(L, EVOMG_F_RES));
    HIP_TRY(hipGetLastError());
    HIP_TRY(wait_stream(stream_));
    return EVOMG_OK;
  }
  int reset_fields() {
    res_pending_ = false;                   // the fields it was owed for are about to be restored
    // fields of up to 16 MiB go through ONE launch (evomg_reset_batch), larger ones through the copy / fill engines' kernels
    ResetBatch batch;
    batch.n = 0;
    unsigned long long longest = 0;
    auto flush = [&]() {
      if (batch.n == 0) return;
      unsigned long long bx = (longest + 255) / 256;
      if (bx > 256) bx = 256;
      hipLaunchKernelGGL(evomg_reset_batch, dim3((unsigned)bx, (unsigned)batch.n), dim3(256), 0, stream_, batch);
      batch.n = 0;
      longest = 0;
    };
    for (int L = prob_.min_level; L <= prob_.max_level; ++L) {
      Level<T>& lv = lv_[L];
      lv.alt_mirror = nullptr;
      for (int f = 0; f < EVOMG_F_COUNT; ++f) {
        if (!lv.f[f]) continue;
        if (lazy_reset_ && !dirty_[L][f] && clean_ptr_[L][f] == lv.f[f]) continue;
        const size_t bytes = lv.elems * sizeof(T);
        if (reset_batch_ && bytes <= (16u << 20) && bytes % 16 == 0) {
          if (batch.n == EVOMG_RESET_MAX) flush();
          batch.seg[batch.n++] = ResetSeg{lv.f[f], lv.init[f], (unsigned long long)(bytes / 16)};
          if (bytes / 16 > longest) longest = bytes / 16;
          if (!lv.init[f]) bnd_zero_[L][f] = true;
        } else if (lv.init[f]) HIP_TRY(hipMemcpyAsync(lv.f[f], lv.init[f], bytes, hipMemcpyDeviceToDevice, stream_));
        else {
          HIP_TRY(hipMemsetAsync(lv.f[f], 0, bytes, stream_));
          bnd_zero_[L][f] = true;
        }
        dirty_[L][f] = false;
        clean_ptr_[L][f] = lv.f[f];
      }
    }
    flush();
    HIP_TRY(hipGetLastError());
    HIP_TRY(wait_stream(stream_));
    return EVOMG_OK;
  }

  // every field an op program may write (conservative: the destination of every op, the coarse solver's fields); `loop`: the
  // fields the solver loop itself writes on the finest level
  void mark_written(const evomg_op_t* ops, int n_ops, bool loop) {
    for (int i = 0; i < n_ops; ++i) {
      const evomg_op_t& op = ops[i];
      if (op.level < prob_.min_level || op.level > prob_.max_level) continue;
      const int d = canon(op.level, op.dst);
      if (d >= 0) dirty_[op.level][d] = true;
      if (op.opcode == EVOMG_OP_COARSE_SOLVE) {
        dirty_[op.level][canon(op.level, EVOMG_F_CORR)] = true;
        dirty_[op.level][EVOMG_F_RES] = true;
        dirty_[op.level][EVOMG_F_CGS_RHS] = true;
      }
    }
    if (loop) {
      const int L = prob_.max_level;
      dirty_[L][EVOMG_F_RES] = true;
      if (prob_.outer_kind == EVOMG_OUTER_PBICGSTAB)
        for (int f : {EVOMG_F_SOL, EVOMG_F_RHS, EVOMG_F_RES, EVOMG_F_CORR}) dirty_[L][f] = true;
    }
  }

  int solve_mg(const evomg_op_t* ops, int n_ops, double wall_limit_ms, evomg_result_t* out, double* hist, int cap) {
    const int L = prob_.max_level;
    Level<T>& lv = lv_[L];
    int nh = 0;
    const double t0 = wall_ms();
    double t_end = -1.0;
    T d;
    if (!norm_event_) HIP_TRY(hipEventCreateWithFlags(&norm_event_, hipEventDisableTiming));
    int ahead = 0;                // ops of the coming iteration that have already been launched (speculatively)
    bool res_stale = false;
    // residual + norm of the current iterate: launched, the scalar copied back asynchronously, an event recorded
    auto residual_norm_launch_all = [&]() -> int {
      int rc = EVOMG_OK;
      const bool timed = prof_opcode_ == EVOMG_PROF_LOOP_NORM && prof_level_ == L;
      bool wrote_r = true;
      if (timed && (rc = prof_open()) != EVOMG_OK) return rc;
      if (norm_fused(lv.g)) rc = residual_norm_launch(L, lv.st, fieldptr(L, EVOMG_F_SOL), fieldptr(L, EVOMG_F_RHS), fieldptr(L, EVOMG_F_RES));
      else if (norm_fused3d(lv.g)) {
        // the residual field itself is written only if the cycle reads it before overwriting it
        const bool dead = fuse_enabled_ && loop_residual_is_dead(ops, n_ops);
        if (dead) res_stale = true;
        wrote_r = !dead;
        rc = residual_norm_launch3d(L, lv.st, fieldptr(L, EVOMG_F_SOL), fieldptr(L, EVOMG_F_RHS), fieldptr(L, EVOMG_F_RES), !dead);
      } else if (residual_dot_fused_ && prob_.bc_kind == EVOMG_BC_DIRICHLET && !use_strip(lv.g) && !use_tile3d(lv.g) && !use_strip3d(lv.g, false)) {
        // small finest levels (the reference's 3D example: 65^3): residual and r . r in one pass (evomg_residual_dot_partial)
        const int m = lv.g.n - 2;
        const long rows = dim_ == 2 ? m : (long)m * m;
        const int blocks = (int)(rows < kDotBlocks ? rows : kDotBlocks);
        if (dim_ == 2) hipLaunchKernelGGL((evomg_residual_dot_partial<T, 2>), dim3(blocks), dim3(256), 0, stream_, lv.g, lv.st, (const T*)fieldptr(L, EVOMG_F_SOL), (const T*)fieldptr(L, EVOMG_F_RHS), fieldptr(L, EVOMG_F_RES), partials_);
        else hipLaunchKernelGGL((evomg_residual_dot_partial<T, 3>), dim3(blocks), dim3(256), 0, stream_, lv.g, lv.st, (const T*)fieldptr(L, EVOMG_F_SOL), (const T*)fieldptr(L, EVOMG_F_RHS), fieldptr(L, EVOMG_F_RES), partials_);
        hipLaunchKernelGGL((evomg_dot_final<T>), dim3(1), dim3(256), 0, stream_, partials_, blocks, norm_out());
      } else {
        residual(L, lv.st, fieldptr(L, EVOMG_F_SOL), fieldptr(L, EVOMG_F_RHS), fieldptr(L, EVOMG_F_RES));
        dot_async(L, fieldptr(L, EVOMG_F_RES), fieldptr(L, EVOMG_F_RES), norm_out());
      }
      if (rc != EVOMG_OK) return rc;
      // x, f read (+ the residual field written where the cycle reads it); the reduction launches behind it are included
      if (timed && (rc = prof_close((wrote_r ? 3.0 : 2.0) * sizeof(T) * pow((double)lv.g.n, dim_))) != EVOMG_OK) return rc;
      if (!host_norm_) HIP_TRY(hipMemcpyAsync(hscalar_, dscalar_, sizeof(T), hipMemcpyDeviceToHost, stream_));
      HIP_TRY(hipEventRecord(norm_event_, stream_));
      return EVOMG_OK;
    };
    // While the host waits for that scalar the GPU would idle. If the cycle starts with an out-of-place smoother, its
    // launch is queued behind the norm right away (speculatively): should the norm say "converged", the pointer swap
    // is simply undone and the sweep's output discarded; otherwise the cycle continues from its second op.
    T *saved_alt = nullptr, *saved_x = nullptr;
    const T* saved_mirror = nullptr;
    // Cross-iteration fusion (evomg_strip_cross.h): the cycle's last sweep (+ prolongation, + the norm) and the next cycle's
    // first sweep (+ residual + restriction) on the finest level as ONE speculative launch per iteration
    const CrossPlan xplan = cross_plan(ops, n_ops);
    bool head_done = false;       // ops[0 .. head_len) of the coming iteration ran inside the previous iteration's cross launch
    auto speculate = [&](int* launched) -> int {
      *launched = 0;
      if (xplan.ok) return EVOMG_OK;
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
      *launched = rc == EVOMG_OK ? 1 : 0;
      return rc;
    };
    // 3D, a cycle that opens with a double sweep on the iterate (C4: V(2,2)): the norm of the CURRENT iterate rides on that
    // double sweep of the COMING cycle (evomg_tile3d_twice<.., NORM0>), launched speculatively like the sweep above — x and f
    // are read once for both instead of once by the norm pass (439 us at 513^3) and once by the sweeps. Should the norm
    // say "stop", the pointer swap is undone; the iterate was only read.
    auto norm_on_head = [&](int* launched) -> int {
      *launched = 0;
      if (!head_norm_ || xplan.ok || n_ops < 2 || !norm_fused3d(lv.g) || !fuse_enabled_ || !fusable_twice(ops, n_ops, 0)) return EVOMG_OK;
      const evomg_op_t& sm = ops[0];
      if (sm.level != L || canon(L, sm.dst) != EVOMG_F_SOL || canon(L, sm.src1) != EVOMG_F_RHS) return EVOMG_OK;
      if (!loop_residual_is_dead(ops, n_ops) || !twice_strips_canonical(lv.g, sm.block[0] == 2)) return EVOMG_OK;
      const Geo& g = lv.g;
      const int strips = (g.n + EVOMG_STRIP_COLS - 1) / EVOMG_STRIP_COLS;
      const long count = (long)(g.n - 2) * (g.n - 2) * strips;
      int rc = ensure_norm_partials((size_t)count + kDotBlocks);
      if (rc != EVOMG_OK) return rc;
      // profiling: the launch counts as the loop's norm pass under EVOMG_PROF_LOOP_NORM (with its two reduction launches) and
      // as a smoothing launch under the sweeps' opcode; it reads x and f and writes the swept field either way
      const bool timed_norm = prof_opcode_ == EVOMG_PROF_LOOP_NORM && prof_level_ == L;
      const bool timed_sm = !timed_norm && prof_matches(sm);
      saved_alt = lv.alt;
      saved_mirror = lv.alt_mirror;
      saved_x = fieldptr(L, sm.dst);
      if ((timed_norm || timed_sm) && (rc = prof_open()) != EVOMG_OK) return rc;
      if ((rc = exec_twice(sm, nullptr, norm_partials_)) != EVOMG_OK) return rc;
      if (timed_sm && (rc = prof_close(algorithmic_bytes(sm))) != EVOMG_OK) return rc;
      hipLaunchKernelGGL((evomg_sum_stage1<T>), dim3(kDotBlocks), dim3(256), 0, stream_, norm_partials_, count, norm_partials_ + count);
      hipLaunchKernelGGL((evomg_dot_final<T>), dim3(1), dim3(256), 0, stream_, norm_partials_ + count, kDotBlocks, norm_out());
      if (timed_norm && (rc = prof_close(3.0 * sizeof(T) * pow((double)g.n, dim_))) != EVOMG_OK) return rc;
      if (!host_norm_) HIP_TRY(hipMemcpyAsync(hscalar_, dscalar_, sizeof(T), hipMemcpyDeviceToHost, stream_));
      HIP_TRY(hipEventRecord(norm_event_, stream_));
      res_stale = true;
      *launched = 2;
      return EVOMG_OK;
    };
    auto undo_speculation = [&]() {
      const int l0 = ops[0].level;
      lv_[l0].alt = saved_alt;
      lv_[l0].alt_mirror = saved_mirror;
      set_fieldptr(l0, ops[0].dst, saved_x);
    };
    auto wait_norm = [&]() -> int {
      HIP_TRY(wait_event(norm_event_));
      d = host_norm_ ? hnorm_[0] : hscalar_[0];
      return EVOMG_OK;
    };
    int rc = prob_.max_iterations >= 1 ? norm_on_head(&ahead) : EVOMG_OK;
    if (rc != EVOMG_OK) return rc;
    if (!ahead) {
      if ((rc = residual_norm_launch_all()) != EVOMG_OK) return rc;
      if ((rc = speculate(&ahead)) != EVOMG_OK) return rc;
    }
    if ((rc = wait_norm()) != EVOMG_OK) return rc;
    const double res0 = sc_norm_from_dot(d);
    if (hist && nh < cap) hist[nh] = res0;
    ++nh;
    double res = res0;
    int it = 0, status = EVOMG_SOLVE_MAXITER;
    if (res0 == 0.0) status = EVOMG_SOLVE_CONVERGED;
    else
      for (it = 1; it <= prob_.max_iterations; ++it) {
        bool norm_rode_along = false, norm_summed = false;
        // not in the last iteration the loop may run: there is no next cycle to start
        const bool cross = xplan.ok && it < prob_.max_iterations;
        const int first = head_done ? xplan.head_len : ahead;
        head_done = false;
        if (cross) {
          if ((rc = exec_program(ops, n_ops, /*cyclic=*/true, first, nullptr, n_ops - xplan.tail_len)) != EVOMG_OK) return rc;
          if ((rc = launch_cross(ops, n_ops, xplan, &norm_summed)) != EVOMG_OK) return rc;
          norm_rode_along = true;
        } else rc = exec_program(ops, n_ops, /*cyclic=*/true, first, &norm_rode_along);
        ahead = 0;
        if (rc != EVOMG_OK) return rc;
        if (norm_rode_along) {
          // the last sweep left the canonical partial sums behind: reduce them, no residual sweep
          res_stale = true;
          const Geo& g = lv.g;
          const int parts = ((g.n + EVOMG_NORM_STRIP_COLS - 1) / EVOMG_NORM_STRIP_COLS) * ((g.n + EVOMG_NORM_ROWS - 1) / EVOMG_NORM_ROWS);
          if (!norm_summed) hipLaunchKernelGGL((evomg_dot_final<T>), dim3(1), dim3(256), 0, stream_, norm_partials_, parts, norm_out());
          if (!host_norm_) HIP_TRY(hipMemcpyAsync(hscalar_, dscalar_, sizeof(T), hipMemcpyDeviceToHost, stream_));
          HIP_TRY(hipEventRecord(norm_event_, stream_));
        } else {
          // not after the last iteration the loop may run: there is no coming cycle for the norm to ride on
          if (it < prob_.max_iterations && (rc = norm_on_head(&ahead)) != EVOMG_OK) return rc;
          if (!ahead && (rc = residual_norm_launch_all()) != EVOMG_OK) return rc;
        }
        if (!ahead && it < prob_.max_iterations && (rc = speculate(&ahead)) != EVOMG_OK) return rc;
        if ((rc = wait_norm()) != EVOMG_OK) return rc;
        res = sc_norm_from_dot(d);
        if (hist && nh < cap) hist[nh] = res;
        ++nh;
        bool stop = true;
        if (!(res == res) || res > 1.7e308) status = EVOMG_SOLVE_DIVERGED;
        else if (res <= prob_.target_res_reduction * res0) status = EVOMG_SOLVE_CONVERGED;
        else if (wall_limit_ms > 0 && wall_ms() - t0 > wall_limit_ms) status = EVOMG_SOLVE_WALL_LIMIT;
        else stop = false;
        if (cross) {
          if (stop) {
            if ((rc = cross_finalize(ops, n_ops, xplan)) != EVOMG_OK) return rc;      // the iterate of THIS iteration, from x
          } else {
            cross_commit(ops, xplan);                                                  // the next cycle has already begun
            head_done = true;
          }
        }
        if (stop) {
          // the clock stops where the solve is over: the norm that ended the loop has been waited for, so every launch of the
          // last cycle has finished; what is still queued is the speculative sweep of a cycle that will not happen (drained
          // below, outside the timer — as the BiCGStab loop does). With the cross launch the final iterate is still to
          // be formed (cross_finalize, queued above): then the clock runs until the stream is empty.
          if (!cross) t_end = wall_ms();
          break;
        }
      }
    if (ahead) undo_speculation();                 // the sweep queued for an iteration that is not going to happen
    HIP_TRY(wait_stream(stream_));
    if (t_end < 0.0) t_end = wall_ms();
    // RES@max is left as the reference's loop would leave it (the residual of the returned iterate) — when somebody looks:
    // the pass is owed (flush_pending_residual: get_field, norm, run_ops, bench_op), not run behind every solve (round 6: one
    // launch and one host wait per evaluation, 20-25 us of the 0.12 ms an evaluation cost besides its solve at 513^2)
    res_pending_ = res_stale;
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
    HIP_TRY(wait_stream(stream_));
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
      HIP_TRY(wait_stream(stream_));
      *out0 = hscalar_[0];
      if (ndot > 1) *out1 = hscalar_[1];
      return EVOMG_OK;
    };
    bool have_rho = false;        // rho_new of the next iteration came out of the residual update
    double t_end = -1.0;          // set when the device-steps loop stops on its norm test
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
      HIP_TRY(wait_stream(stream_));         // `init` is a stack array
      if (!outer_event_) HIP_TRY(hipEventCreateWithFlags(&outer_event_, hipEventDisableTiming));
      auto front = [&](bool beta_done) -> int {       // beta (unless the reduction before formed it) ; p = Res + beta (p - omega ap) ; u = M^-1 p
        if (!beta_done) hipLaunchKernelGGL((evomg_bicg_beta<T>), dim3(1), dim3(64), 0, stream_, dscalar_);
        hipLaunchKernelGGL((evomg_axpy2_dev<T>), node_grid(g), node_block(), 0, stream_, g, p, (const T*)p, (const T*)ap, (const T*)Res, (const T*)dscalar_);
        return precondition(p);
      };
      if ((rc = front(false)) != EVOMG_OK) return rc;
      const bool robin_cols = outer_folds_ && prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN && scalar_traits<T>::is_complex;
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
        // Sol = hh + omega u, its Robin columns by the same launch (apply_bc's division)
        if (robin_cols) hipLaunchKernelGGL((evomg_axpy_dev<T, true>), node_grid(g), node_block(), 0, stream_, g, Sol, (const T*)hh, (const T*)(dscalar_ + 3), (const T*)u, kh(L));
        else {
          hipLaunchKernelGGL((evomg_axpy_dev<T>), node_grid(g), node_block(), 0, stream_, g, Sol, (const T*)hh, (const T*)(dscalar_ + 3), (const T*)u, 0.0);
          apply_bc(L, Sol);
        }
        hipLaunchKernelGGL((evomg_axpy_dots_dev<T>), dim3(dot_blocks), dim3(256), 0, stream_, g, Res, (const T*)s, (const T*)(dscalar_ + 3), (const T*)t, (const T*)rhat, partials_);
        hipLaunchKernelGGL((evomg_dot_final2<T>), dim3(1), dim3(256), 0, stream_, partials_, dot_blocks, 2, dscalar_, host_norm_ ? hnorm_dev_ : (T*)nullptr, outer_folds_ ? 1 : 0);
        if (!host_norm_) HIP_TRY(hipMemcpyAsync(hscalar_, dscalar_, sizeof(T), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipEventRecord(outer_event_, stream_));
        if (it < prob_.max_iterations && (rc = front(outer_folds_)) != EVOMG_OK) return rc;  // the next iteration's first half (beta: formed by the reduction above)
        HIP_TRY(wait_event(outer_event_));
        d = host_norm_ ? hnorm_[0] : hscalar_[0];
        res = sc_norm_from_dot(d);
        if (hist && nh < cap) hist[nh] = res;
        ++nh;
        // the solve ends HERE when one of the tests below fires: what is still queued is the next iteration's speculative
        // first half (a whole preconditioner application), which is drained after the clock has stopped
        t_end = wall_ms();
        if (!(res == res) || res > 1.7e308) { status = EVOMG_SOLVE_DIVERGED; break; }
        if (res < prob_.target_res_reduction * res0) { status = EVOMG_SOLVE_CONVERGED; break; }
        if (wall_limit_ms > 0 && t_end - t0 > wall_limit_ms) { status = EVOMG_SOLVE_WALL_LIMIT; break; }
        t_end = -1.0;
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
    HIP_TRY(wait_stream(stream_));
    out->time_ms = (t_end >= 0.0 ? t_end : wall_ms()) - t0;
    HIP_TRY(hipMemcpyAsync(lv.f[EVOMG_F_CORR], Sol, bytes, hipMemcpyDeviceToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(lv.f[EVOMG_F_RES], Res, bytes, hipMemcpyDeviceToDevice, stream_));
    HIP_TRY(wait_stream(stream_));
    out->n_iterations = (double)it;
    out->init_res_norm = res0;
    out->final_res_norm = res;
    out->status = status;
    out->n_history = (nh < cap || !hist) ? nh : cap;
    out->conv_factor = res0 > 0 ? res / res0 : 0.0;
    return EVOMG_OK;
  }
