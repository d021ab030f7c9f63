// impl_interp.inl — one op -> one launch: the plain interpreter, `with jacobi` slot handling, coarse-level batches, live profiling
// Part of the body of `template <typename T> class Impl` (evomg_impl.h includes this file INSIDE the class).

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
    // a sweep that consumes an elided SET_ZERO (zero-input variant) does not read x
    const double x_bytes = (timed && x_zero_next_ && op.opcode == EVOMG_OP_SMOOTH) ? (double)sizeof(T) * pow((double)lv_[op.level].g.n, dim_) : 0.0;
    if (timed) {
      const int rc = prof_open();
      if (rc != EVOMG_OK) return rc;
    }
    const int rc = exec_impl(op);
    if (timed) {
      const int rc2 = prof_close(algorithmic_bytes(op) - x_bytes);
      if (rc2 != EVOMG_OK) return rc2;
    }
    return rc;
  }
  // the 3D double sweep under live profiling: ONE launch doing two SMOOTH ops (x, f read once, y written once; with the
  // prolongation folded in, the coarse correction read as well)
  int exec_twice_timed(const evomg_op_t& sm, const evomg_op_t* pr = nullptr) {
    const bool timed = prof_matches(sm);
    if (timed) {
      const int rc = prof_open();
      if (rc != EVOMG_OK) return rc;
    }
    const int rc = exec_twice(sm, pr);
    if (timed) {
      const double coarse = pr ? (double)sizeof(T) * pow((double)lv_[sm.level - 1].g.n, dim_) : 0.0;
      const int rc2 = prof_close(algorithmic_bytes(sm) + coarse);
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
        } else if (op.partitioning == EVOMG_PART_RED_BLACK && rbgs3d_block_ok(g)) {
          // 3D below the tile threshold: both colours in one launch (evomg_small3d.h: evomg_rbgs3d_block), out of place
          launch_rbgs3d_block(lv, x, f, om, omm, take_x_zero(), nullptr, g, 0.0);
          advance(lv, L, op, x, lv.alt);
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

  int exec_twice(const evomg_op_t& sm, const evomg_op_t* pr = nullptr, T* norm_aux = nullptr) {
    const int L = sm.level;
    Level<T>& lv = lv_[L];
    T* x = fieldptr(L, sm.dst);
    const T* f = fieldptr(L, sm.src1);
    T* y = lv.alt;
    const double om = sm.omega, omm = 1.0 - sm.omega;
    const bool xzero = take_x_zero() != 0 && !pr && !norm_aux;      // the SET_ZERO in front of the sweeps, elided (zero_input_sweep)
    ensure_faces(lv, x, y);
    if constexpr (!scalar_traits<T>::is_complex) {
      if (sm.block[0] == 2) {
        const BlockPlan<T> pl = make_block_plan<T, 3>(lv.st, 2, 1, 1);
        const T* cz = pr ? fieldptr(L - 1, pr->src0) : nullptr;
        launch_tile3d_twice<TILE_BLOCK211>(lv.g, lv.st, x, f, y, om, omm, TwoCellPlan<T>{pl.fac[1][0], pl.upper[0][1], pl.pivot[0], pl.pivot[1]},
                                           cz, &lv_[L - 1 >= prob_.min_level ? L - 1 : L].g, pr ? pr->omega : 0.0, norm_aux, xzero);
      } else {
        const T* cz = pr ? fieldptr(L - 1, pr->src0) : nullptr;
        launch_tile3d_twice<STRIP_JACOBI>(lv.g, lv.st, x, f, y, om, omm, TwoCellPlan<T>{}, cz, &lv_[L - 1 >= prob_.min_level ? L - 1 : L].g,
                                          pr ? pr->omega : 0.0, norm_aux, xzero);
      }
    }
    advance(lv, L, sm, x, y);      // two sweeps, one pass: the result lives in the spare array -> ONE swap
    HIP_TRY(hipGetLastError());
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
