// impl_coarse.inl — the coarse-grid solver: kernel choice by grid size, Krylov solvers as launch sequences
// Part of the body of `template <typename T> class Impl` (evomg_impl.h includes this file INSIDE the class).

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
        if (cg1_serves(lv.g)) {
          const int rc = launch_cg1(lv, x, src, r, cgs);
          if (rc != EVOMG_OK) return rc;
        } else if (krylov_own<2>(lv.g.n, true).wave) launch_cg_wave2d<true>(lv, x, src, r, cgs);
        else {
          bool launched = false;
          const int rc = launch_cg_reg2d_folded(lv, x, src, r, cgs, &launched);
          if (rc != EVOMG_OK) return rc;
          if (!launched) return invalid("coarse_solve: no folded form for this grid");
        }
        bnd_zero_[L][canon(L, copy_op->dst)] = bnd_zero_[L][canon(L, copy_op->src0)];     // the COPY's bookkeeping
        lv.alt_mirror = nullptr;
        apply_bc(L, x);
        HIP_TRY(hipGetLastError());
        return EVOMG_OK;
      }
      return invalid("coarse_solve: folded copy on a complex problem");
    }
    T* w[kTmpVectors];
    const bool kb_route = prob_.cgs_kind == EVOMG_CGS_BICGSTAB && dim_ == 2 && kcg_enabled_ && lv.g.n >= kb_min_n_;
    const int need = prob_.cgs_kind == EVOMG_CGS_BICGSTAB ? (kb_route && kb_fused_ ? 8 : 6) : 2;
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
    } else if (cg1_serves(lv.g)) {
      // real 2D grids up to 129^2: the single-reduction recurrence (evomg_cg1.h)
      const int rc = launch_cg1(lv, x, b, r, nullptr);
      if (rc != EVOMG_OK) return rc;
      apply_bc(L, x);
    } else if (!scalar_traits<T>::is_complex && dim_ == 3 && krylov_cg_groups<3>(lv.g.n) > 1) {
      // 33^3: eight co-resident workgroups, one slab of planes each — the single-reduction form (evomg_cg1.h: evomg_cg1_slab3d)
      // or the classical one (evomg_cg_multi_wg3d); the canonical reduction order of this grid is the grouped one either way, so
      // there is no single-workgroup alternative
      const KrylovOwn own = krylov_own<3>(lv.g.n);
      const bool single_reduction = prob_.cgs_kind == EVOMG_CGS_CG;       // evomg_cg1_slab3d; else the classical evomg_cg_multi_wg3d
      const size_t lds_bytes = single_reduction ? sizeof(double) * (258 + (size_t)EVOMG_CG_SLAB * (own.m + 2) * (own.m + 2))
                                                : sizeof(T) * (34 + 2 * EVOMG_CG_GROUPS + (size_t)EVOMG_CG_SLAB * (own.m + 2) * (own.m + 2));
      if (!cg_multi_attr_set_) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(evomg_cg_multi_wg3d<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(evomg_cg1_slab3d), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_));
        cg_multi_attr_set_ = true;
      }
      cg_multi_used_ = true;
      unsigned seq_base = 0;
      const int rc_seq = take_ll_sequence((unsigned)(prob_.cgs_max_iterations > 0 ? prob_.cgs_max_iterations : 0) + 3u, &seq_base);
      if (rc_seq != EVOMG_OK) return rc_seq;
      const int blocks = EVOMG_CG_GROUPS * cg_stride_;
      if constexpr (!scalar_traits<T>::is_complex) {       // complex problems are 2D only
        if (single_reduction)
          hipLaunchKernelGGL(evomg_cg1_slab3d, dim3(blocks), dim3(1024), lds_bytes, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations,
                             prob_.cgs_target_res_reduction, dinfo_, cg_ll_, seq_base, cg_stride_, cg_local_exchange_ ? 1 : 0,
                             (cg_stride_ > 1 && evaluations_in_flight() > 1) ? next_cg_phase() % cg_stride_ : 0);
        else
          hipLaunchKernelGGL(evomg_cg_multi_wg3d<double>, dim3(blocks), dim3(1024), lds_bytes, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations,
                             prob_.cgs_target_res_reduction, dinfo_, cg_ll_, seq_base, cg_stride_);
      }
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

  // ---- the single-reduction CG (evomg_cg1.h): which grids it serves, and the launch ------------------------------------
  bool cgs_is_cg() const { return prob_.cgs_kind == EVOMG_CGS_CG || prob_.cgs_kind == EVOMG_CGS_CG_CLASSICAL; }
  bool cg1_serves(const Geo& g) const {
    // (EVOMG_NO_REG_KRYLOV does not reach here: it chooses among kernels of ONE arithmetic form, this form has one kernel)
    if (scalar_traits<T>::is_complex || dim_ != 2 || prob_.cgs_kind != EVOMG_CGS_CG) return false;
    const int m = g.n - 2;
    return m == 1 || m == 3 || m == 7 || m == 15 || m == 31 || m == 63 || m == 127;     // the level grids 3^2 .. 129^2
  }
  // sequence numbers of the exchange words: one launch uses at most `span` per exchange kind, above a base that only
  // grows; before it could wrap, the words are cleared (stream-ordered) and counting starts over
  int take_ll_sequence(unsigned span, unsigned* base) {
    if (cg_seq_base_ > 0x7fffffffu - span) {
      HIP_TRY(hipMemsetAsync(cg_ll_, 0, sizeof(unsigned long long) * EVOMG_CG_LL_WORDS, stream_));
      cg_seq_base_ = 0;
    }
    *base = cg_seq_base_;
    cg_seq_base_ += span;
    return EVOMG_OK;
  }
  // b_copy != nullptr: the folded form — b is the COPY's source, the solve starts from x = 0 (fusable_copy_zero_solve)
  template <int R, int E, int H, int WAVES, bool MULTI, int MASK>
  int launch_cg1_as(const Level<T>& lv, T* x, const T* b, T* r, T* b_copy, int groups) {
    static_assert(EVOMG_CG1_LL_WORDS <= EVOMG_CG_LL_WORDS, "the exchange area is shared with evomg_cg_multi_wg3d");
    if constexpr (scalar_traits<T>::is_complex) return invalid("launch_cg1: real problems only");
    else {
      unsigned base = 0;
      int stride = 1, phase = 0;
      bool wave_autonomous = false;
      if (MULTI) {
        const int rc = take_ll_sequence((unsigned)(prob_.cgs_max_iterations > 0 ? prob_.cgs_max_iterations : 0) + 3u, &base);
        if (rc != EVOMG_OK) return rc;
        // All workgroups of a solve on ONE XCD (every eighth block of the grid takes part): its exchanges then stay in that
        // XCD's L2. Alone on the device that is XCD 0; while other evaluations of this process are under way — candidates in
        // flight — the solves are dealt to the XCDs round-robin, so that they neither pile onto one XCD's 32 units (290
        // solves/s with twelve in flight) nor pay the fabric's latency with one workgroup per XCD (EVOMG_CG_SPREAD_XCDS=1: the
        // population of BASELINE config C5 52 against 55 solves/s, sixteen identical candidates 387 against 381:
        // profiles/r05_cg1_placement_raw.txt)
        stride = cg_stride_;
        if (stride > 1 && evaluations_in_flight() > 1) phase = next_cg_phase() % stride;
        // on one XCD the words of an exchange stay in its L2: every wave then exchanges for itself (no collector, no barrier:
        // 1.58 -> 1.49 us per iteration); over the fabric the collector form moves fewer words (2.06 against 2.15 us)
        wave_autonomous = stride > 1 && cg_local_exchange_ && cg1_wave_autonomous_;
        cg_multi_used_ = true;
      }
      const dim3 grid(MULTI ? groups * stride : 1), block(64 * WAVES);
      const int maxit = prob_.cgs_max_iterations, local = cg_local_exchange_ ? 1 : 0;
      const double tol = prob_.cgs_target_res_reduction;
#define EVOMG_CG1_LAUNCH(XZ, AUTO, COPY) hipLaunchKernelGGL((evomg_cg1_band2d<R, E, H, WAVES, MULTI, XZ, MASK, AUTO>), grid, block, 0, stream_, lv.g, lv.st, x, b, r, \
                                                            COPY, maxit, tol, dinfo_, cg_ll_, base, groups, stride, local, phase)
      if constexpr (MULTI) {
        if (wave_autonomous) {
          if (b_copy) EVOMG_CG1_LAUNCH(true, true, b_copy);
          else EVOMG_CG1_LAUNCH(false, true, (T*)nullptr);
          return EVOMG_OK;
        }
      }
      if (b_copy) EVOMG_CG1_LAUNCH(true, false, b_copy);
      else EVOMG_CG1_LAUNCH(false, false, (T*)nullptr);
#undef EVOMG_CG1_LAUNCH
      return EVOMG_OK;
    }
  }
  // one instantiation per level grid: the band layout is cg1_plan()'s (the oracle restates it), the grouping of the waves
  // into workgroups is free
  int launch_cg1(const Level<T>& lv, T* x, const T* b, T* r, T* b_copy) {
    switch (lv.g.n - 2) {
      case 127:                                    // 32 waves as sixteen workgroups of two (round 6: 1.03 against 1.10 us per iteration as eight of four — the grouping is not part of the arithmetic; four of eight: slower)
        return launch_cg1_as<4, 2, 1, 2, true, 3>(lv, x, b, r, b_copy, 16);
      case 63: return launch_cg1_as<16, 1, 1, 4, false, 15>(lv, x, b, r, b_copy, 1);
      case 31: return launch_cg1_as<4, 1, 2, 4, false, 3>(lv, x, b, r, b_copy, 1);
      case 15: return launch_cg1_as<2, 1, 4, 2, false, 1>(lv, x, b, r, b_copy, 1);
      case 7: return launch_cg1_as<1, 1, 8, 1, false, 0>(lv, x, b, r, b_copy, 1);
      case 3:
      case 1: return launch_cg1_as<1, 1, 16, 1, false, 0>(lv, x, b, r, b_copy, 1);
    }
    return invalid("launch_cg1: no banded form for this grid");
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
        HIP_TRY(wait_stream(stream_));
        if (kcg_host_->done) break;
      }
      hipLaunchKernelGGL(evomg_accumulate_iterations, dim3(1), dim3(64), 0, stream_, dinfo_, (const int*)&kcg_state_[k & 1].it);
      HIP_TRY(hipGetLastError());
      return EVOMG_OK;
    }
    return invalid("coarse_cg_per_launch: real problems only");
  }

  // BiCGStab as a sequence of launches (2D, 129^2 and up). The batches between two reads of the done flag start small
  // (C3 converges in two iterations) and double up to 16. Three launches per iteration (kb2_ps, kb2_hst, kb_xr; w[6], w[7]: the
  // second buffers of p and nu) or, with EVOMG_NO_KB_FUSE, the six of round 4 — the same bits either way.
  template <bool ROBIN> int coarse_bicgstab_per_launch(int L, const Level<T>& lv, T* x, const T* b, T* r, T* const* w) {
    if constexpr (ROBIN && !scalar_traits<T>::is_complex) return invalid("Robin columns on a real problem");
    else {
      if (!kb_state_) {
        HIP_TRY(hipMalloc(&kb_state_, 3 * sizeof(KbState<T>)));          // [2]: where kb2_finish's second copy goes when the host memory is not mapped
        HIP_TRY(hipMalloc(&kb_scal_, 4 * sizeof(T)));
        HIP_TRY(hipHostMalloc(&kb_host_, sizeof(KbState<T>), hipHostMallocMapped));
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&kb_host_dev_), kb_host_, 0) != hipSuccess) kb_host_dev_ = nullptr;
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
      const int maxit = prob_.cgs_max_iterations;           // >= 1 (evomg_create)
      int k = 0, batch = 2, slot = 0;            // slot: where the state the host reads (and the iteration count) ends up
      kb_host_->done = 0;
      if (kb_fused_) {
        T *p2 = w[6], *nu2 = w[7];
        while (k < maxit) {
          const int todo = maxit - k < batch ? maxit - k : batch;
          int cur = 0;
          for (int j = 0; j < todo; ++j, ++k) {
            cur = k & 1;
            const T *p_in = cur ? p2 : p, *nu_in = cur ? nu2 : nu;
            T *p_out = cur ? p : p2, *nu_out = cur ? nu : nu2;
            if (k == 0)
              hipLaunchKernelGGL((kb2_ps<T, ROBIN, true>), grid, block, 0, stream_, g, lv.st, (const T*)r, (const T*)rhat, nu_in, nu_out, p_in, p_out,
                                 (const T*)part_r, blocks, part_nu, kb_state_, cur, (const T*)kb_scal_, khv, tol);
            else
              hipLaunchKernelGGL((kb2_ps<T, ROBIN, false>), grid, block, 0, stream_, g, lv.st, (const T*)r, (const T*)rhat, nu_in, nu_out, p_in, p_out,
                                 (const T*)part_r, blocks, part_nu, kb_state_, cur, (const T*)kb_scal_, khv, tol);
            hipLaunchKernelGGL((kb2_hst<T, ROBIN>), grid, block, 0, stream_, g, lv.st, (const T*)x, (const T*)r, (const T*)p_out, (const T*)nu_out, hh, s, t,
                               (const T*)part_nu, blocks, part_t, (const KbState<T>*)kb_state_, cur, kb_scal_, khv);
            hipLaunchKernelGGL((kb_xr<T, ROBIN>), grid, block, 0, stream_, g, x, r, (const T*)s, (const T*)hh, (const T*)t, (const T*)rhat,
                               (const T*)part_t, blocks, part_r, (const KbState<T>*)kb_state_, cur, kb_scal_, khv);
          }
          // the state the host reads: written by the kernel itself into pinned host memory (no copy launch), else copied
          hipLaunchKernelGGL((kb2_finish<T>), dim3(1), block, 0, stream_, (const T*)part_r, blocks, kb_state_, cur, tol,
                             kb_host_dev_ ? kb_host_dev_ : kb_state_ + 2);
          slot = cur ^ 1;
          if (!kb_host_dev_) HIP_TRY(hipMemcpyAsync(kb_host_, kb_state_ + slot, sizeof(KbState<T>), hipMemcpyDeviceToHost, stream_));
          HIP_TRY(wait_stream(stream_));
          if (kb_host_->done) break;
          if (k >= 4 && batch < 16) batch *= 2;            // 2, 2, 4, 8, 16: C3's solves end after two to four iterations
        }
      } else {
        hipLaunchKernelGGL((kb_start<T>), dim3(1), block, 0, stream_, (const T*)part_r, blocks, kb_state_, kb_scal_);
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
          slot = k & 1;
          HIP_TRY(hipMemcpyAsync(kb_host_, kb_state_ + slot, sizeof(KbState<T>), hipMemcpyDeviceToHost, stream_));
          HIP_TRY(wait_stream(stream_));
          if (kb_host_->done) break;
          if (batch < 16) batch *= 2;
        }
      }
      hipLaunchKernelGGL(evomg_accumulate_iterations, dim3(1), dim3(64), 0, stream_, dinfo_, (const int*)&kb_state_[slot].it);
      HIP_TRY(hipGetLastError());
      return EVOMG_OK;
    }
  }

  // register-resident CG when a thread's share of the coarsest grid fits its registers (2D, up to 129^2); otherwise the
  // caller falls back to the global-memory kernel
  template <typename K> int set_lds_limit(K kern, int slot, size_t lds_bytes) {
    if (!cg_lds_attr_set_[slot]) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_));
      cg_lds_attr_set_[slot] = true;
    }
    (void)lds_bytes;
    return EVOMG_OK;
  }
  // evomg_cg_reg2d<.., POW2>: equal off-centre coefficients cc = +-2^k and a centre c0 = +-2^j cc
  bool cg_reg2d_pow2(const Stencil<T>& st) const {
    if constexpr (scalar_traits<T>::is_complex) return false;
    else {
      if (!pow2_fma_ || dim_ != 2) return false;
      int ex = 0;
      const double m1 = frexp(st.c[1], &ex);
      if (!(m1 == 0.5 || m1 == -0.5) || ex < -400 || ex > 400) return false;
      if (st.c[2] != st.c[1] || st.c[3] != st.c[1] || st.c[4] != st.c[1]) return false;
      const double m0 = frexp(st.c[0], &ex);
      return (m0 == 0.5 || m0 == -0.5) && ex > -400 && ex < 400;
    }
  }
  // b_copy != nullptr: the folded form (fusable_copy_zero_solve) — b is the COPY's source, the solve starts from x = 0; only
  // the variants for nearly full last bands exist in that form (every 2^k + 1 grid), `launched` stays false otherwise
  template <int RT, int E> int launch_cg_reg2d(const Level<T>& lv, T* x, const T* b, T* r, const KrylovOwn& own, int slot, bool* launched,
                                               T* b_copy = nullptr) {
    const size_t lds_bytes = (size_t)(cg_reg2d_lds_elems(RT, E) + 32) * sizeof(T);
    // the kernel assumes one band per wave of the 512-thread workgroup and exactly RT rows per band
    if (lds_bytes > max_lds_ || own.lanes != 512 || own.bands != 8 || own.R != RT || E != (own.pair ? 2 : 1)) return EVOMG_OK;
    // rows every wave is guaranteed to own: all of own.R except in the last band(s)
    const int last_rows = own.m - (own.bands - 1) * own.R;
    constexpr int kMaskLast = RT > 1 ? RT - 1 : 0;
    const bool nearly_full = last_rows >= kMaskLast;
    const dim3 one(1), thr(own.lanes);
    if constexpr (RT == 16 && E == 2 && !scalar_traits<T>::is_complex) {
      if (nearly_full && cg_reg2d_pow2(lv.st)) {           // 129^2 Poisson: the headline's coarse solve
        if (b_copy) {
          auto kern = evomg_cg_reg2d<T, RT, E, kMaskLast, true, true>;
          int rc = set_lds_limit(kern, 12 + 2 * slot + 1, lds_bytes);
          if (rc != EVOMG_OK) return rc;
          hipLaunchKernelGGL(kern, one, thr, lds_bytes, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_, b_copy);
        } else {
          auto kern = evomg_cg_reg2d<T, RT, E, kMaskLast, true>;
          int rc = set_lds_limit(kern, 2 * slot + 2, lds_bytes);
          if (rc != EVOMG_OK) return rc;
          hipLaunchKernelGGL(kern, one, thr, lds_bytes, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_, (T*)nullptr);
        }
        *launched = true;
        return EVOMG_OK;
      }
    }
    if (b_copy) {
      if (!nearly_full) return EVOMG_OK;
      auto kern = evomg_cg_reg2d<T, RT, E, kMaskLast, false, true>;
      int rc = set_lds_limit(kern, 12 + 2 * slot, lds_bytes);
      if (rc != EVOMG_OK) return rc;
      hipLaunchKernelGGL(kern, one, thr, lds_bytes, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_, b_copy);
      *launched = true;
      return EVOMG_OK;
    }
    if (nearly_full) {
      auto kern = evomg_cg_reg2d<T, RT, E, kMaskLast>;
      int rc = set_lds_limit(kern, 2 * slot, lds_bytes);
      if (rc != EVOMG_OK) return rc;
      hipLaunchKernelGGL(kern, one, thr, lds_bytes, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_, (T*)nullptr);
    } else {
      auto kern = evomg_cg_reg2d<T, RT, E, 0>;
      int rc = set_lds_limit(kern, 2 * slot + 1, lds_bytes);
      if (rc != EVOMG_OK) return rc;
      hipLaunchKernelGGL(kern, one, thr, lds_bytes, stream_, lv.g, lv.st, x, b, r, prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_, (T*)nullptr);
    }
    *launched = true;
    return EVOMG_OK;
  }
  template <bool XZERO> void launch_cg_wave2d(const Level<T>& lv, T* x, const T* b, T* r, T* b_copy) {
    if constexpr (!scalar_traits<T>::is_complex) {
      // rows every lane is guaranteed to own: the upper half's share (33^2: 15 of 16)
      const int m = lv.g.n - 2, upper = m - EVOMG_CG_WAVE_ROWS;
      if (upper >= EVOMG_CG_WAVE_ROWS - 1 && cg_reg2d_pow2(lv.st))       // 33^2 Poisson: the tutorial's coarse solve
        hipLaunchKernelGGL((evomg_cg_wave2d<T, XZERO, EVOMG_CG_WAVE_ROWS - 1, true>), dim3(1), dim3(64), 0, stream_, lv.g, lv.st, x, b, r, b_copy,
                           prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_);
      else if (upper >= EVOMG_CG_WAVE_ROWS - 1)
        hipLaunchKernelGGL((evomg_cg_wave2d<T, XZERO, EVOMG_CG_WAVE_ROWS - 1>), dim3(1), dim3(64), 0, stream_, lv.g, lv.st, x, b, r, b_copy,
                           prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_);
      else
        hipLaunchKernelGGL((evomg_cg_wave2d<T, XZERO, 0>), dim3(1), dim3(64), 0, stream_, lv.g, lv.st, x, b, r, b_copy,
                           prob_.cgs_max_iterations, prob_.cgs_target_res_reduction, dinfo_);
    }
  }
  // the register CG in its folded form (COPY + SET_ZERO inside the launch): which grids have it, and the launch
  bool cg_reg2d_folds(const Geo& g) const {
    if (scalar_traits<T>::is_complex || dim_ != 2 || !use_reg_krylov_) return false;
    const KrylovOwn own = krylov_own<2>(g.n, true);
    if (own.wave || !own.tile || own.lanes != 512 || own.bands != 8) return false;
    const int e = own.pair ? 2 : 1;
    if (own.pair ? own.R != 16 : (own.R != 1 && own.R != 2 && own.R != 4 && own.R != 8)) return false;
    if ((size_t)(cg_reg2d_lds_elems(own.R, e) + 32) * sizeof(T) > max_lds_) return false;
    return own.m - (own.bands - 1) * own.R >= (own.R > 1 ? own.R - 1 : 0);
  }
  int launch_cg_reg2d_folded(const Level<T>& lv, T* x, const T* src, T* r, T* cgs, bool* launched) {
    *launched = false;
    if constexpr (!scalar_traits<T>::is_complex) {
      const KrylovOwn own = krylov_own<2>(lv.g.n, true);
      if (!own.pair) {
        if (own.R == 1) return launch_cg_reg2d<1, 1>(lv, x, src, r, own, 0, launched, cgs);
        if (own.R == 2) return launch_cg_reg2d<2, 1>(lv, x, src, r, own, 1, launched, cgs);
        if (own.R == 4) return launch_cg_reg2d<4, 1>(lv, x, src, r, own, 2, launched, cgs);
        if (own.R == 8) return launch_cg_reg2d<8, 1>(lv, x, src, r, own, 3, launched, cgs);
        return EVOMG_OK;
      }
      if (own.R == 16) return launch_cg_reg2d<16, 2>(lv, x, src, r, own, 4, launched, cgs);
    }
    return EVOMG_OK;
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
