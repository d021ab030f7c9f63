// impl_peephole.inl — the op-program walker and every peephole predicate: which neighbouring ops run as ONE launch
// Part of the body of `template <typename T> class Impl` (evomg_impl.h includes this file INSIDE the class).

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

  bool fusable_copy_zero_solve(const evomg_op_t* ops, int n_ops, int i) const {
    if (!fuse_enabled_ || !use_reg_krylov_ || i + 2 >= n_ops || dim_ != 2) return false;
    const int L = prob_.min_level;
    // the one-wave or the register CG (real, Dirichlet) or the one-wave BiCGStab (up to 9^2, real or complex with its Robin columns)
    const bool cg_case = !scalar_traits<T>::is_complex && cgs_is_cg() && prob_.bc_kind == EVOMG_BC_DIRICHLET;
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
    return bicg_case || cg1_serves(lv_[L].g) || krylov_own<2>(lv_[L].g.n, true).wave != 0 || (cg_fold_reg_ && cg_reg2d_folds(lv_[L].g));
  }
  bool fusable_residual_restrict(const evomg_op_t* ops, int n_ops, int i) const {
    if (!fuse_enabled_ || i + 1 >= n_ops) return false;
    const evomg_op_t &a = ops[i], &b = ops[i + 1];
    if (a.opcode != EVOMG_OP_RESIDUAL || b.opcode != EVOMG_OP_RESTRICT) return false;
    if (b.level != a.level - 1 || canon(a.level, b.src0) != canon(a.level, a.dst)) return false;
    if (canon(a.level, a.dst) == canon(a.level, a.src0) || canon(a.level, a.dst) == canon(a.level, a.src1)) return false;
    if (!use_strip(lv_[a.level].g)) return false;
    if (prof_matches(b)) return false;       // profiling the RESTRICT op alone: keep it a launch of its own (RESIDUAL: timed fused)
    return true;
  }

  // SET_ZERO elision: set by the interpreter right before the sweep that consumes it
  int take_x_zero() { const int z = x_zero_next_ ? 1 : 0; x_zero_next_ = false; return z; }
  // SET_ZERO(L, X) immediately followed by a strip point smoother on X whose boundary is known to be zero: the zero
  // launch is skipped and the sweep reads no x at all (same arithmetic on literal zeros => same bits)
  bool elidable_zero(const evomg_op_t* ops, int n_ops, int i) const {
    if (!fuse_enabled_ || !zero_elision_ || i + 1 >= n_ops) return false;
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
    if (!fuse_enabled_ || !zero_elision_ || i >= n_ops) return false;
    const evomg_op_t& sm = ops[i];
    if (dim_ == 3) {
      // a double sweep of the tiles on a zero field loads nothing of it (evomg_tile3d_twice<.., XZ>)
      if (twice_xzero_ && fusable_twice(ops, n_ops, i) && !prof_matches(sm) && !prof_matches(ops[i + 1])) return true;
      // small 3D levels: the one-launch red-black sweep takes the zero on its load (evomg_rbgs3d_block: zero_first)
      // (levels up to 17^3 belong to evomg_small3d_chain, which takes the SET_ZERO itself)
      return sm.opcode == EVOMG_OP_SMOOTH && sm.partitioning == EVOMG_PART_RED_BLACK && sm.block[0] * sm.block[1] * sm.block[2] == 1 &&
             canon(sm.level, sm.src1) != canon(sm.level, sm.dst) && !prof_matches(sm) && lv_[sm.level].g.n > 17 &&
             rbgs3d_block_ok(lv_[sm.level].g);
    }
    if (sm.opcode != EVOMG_OP_SMOOTH || sm.block[0] * sm.block[1] != 1) return false;
    if (jacobi_lost(sm)) return false;        // the sweep's output is discarded: the field has to be zeroed for real
    if (canon(sm.level, sm.src1) == canon(sm.level, sm.dst)) return false;
    // the zero-input variants exist for the plain sweep and the sweep + residual + restriction; a sweep that is the
    // tail of the iteration (norm) would go another way
    if (tail_norm_candidate(ops, n_ops, i)) return false;
    return use_strip(lv_[sm.level].g);
  }

  bool fusable_two_sweeps_residual_restrict(const evomg_op_t* ops, int n_ops, int i, bool cyclic) const {
    if (i + 3 >= n_ops || !fusable_smooth_residual_restrict(ops, n_ops, i + 1, cyclic)) return false;
    const evomg_op_t &a = ops[i], &b = ops[i + 1];
    if (a.opcode != EVOMG_OP_SMOOTH || a.partitioning != EVOMG_PART_RED_BLACK || b.partitioning != EVOMG_PART_RED_BLACK) return false;
    if (a.level != b.level || a.omega != b.omega || a.block[0] * a.block[1] != 1) return false;
    if (canon(a.level, a.dst) != canon(b.level, b.dst) || canon(a.level, a.src1) != canon(b.level, b.src1)) return false;
    if (prof_matches(a) || prof_matches(b) || prof_matches(ops[i + 2]) || prof_matches(ops[i + 3])) return false;
    const Geo& g = lv_[a.level].g;
    if (small2d_ok(g) && small2d_tiles(g.n) == 1 && use_strip(g)) return true;
    // a grid of several LDS tiles: both sweeps in each tile's launch, with a halo of six nodes (evomg_small2d.h; tiles read
    // their halo from x while others store y: y is the level's spare array)
    if (xgen_enabled_ && small2d_ok(g) && use_strip(g) && small2d_tile_size(g.n) + 2 * EVOMG_SMALL2D_HALO2 <= 45 &&
        2 * small2d_image_elems(g.n, EVOMG_SMALL2D_HALO2) * sizeof(T) <= max_lds_)
      return true;
    // strip levels: when the sweeps start from the zero field (the SET_ZERO in front of them elided, or the cycle's known-zero
    // input), the first sweep's result is formed from f inside the second's launch (evomg_strip2d<.., XGEN>); with Helmholtz's
    // Robin columns only when the kernels write them themselves
    if (!xgen_enabled_ || !x_zero_next_ || !use_strip(g) || small2d_ok(g)) return false;
    if (prob_.bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN && !strip_bc_) return false;
    return !jacobi_lost(a);
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

  // PROLONG_ADD@L, red-black SMOOTH on a small 3D level: the correction folded into the one-launch sweep's load
  bool fusable_prolong_rbgs3d(const evomg_op_t* ops, int n_ops, int i) const {
    if (!fuse_enabled_ || dim_ != 3 || i + 1 >= n_ops) return false;
    const evomg_op_t &a = ops[i], &b = ops[i + 1];
    if (a.opcode != EVOMG_OP_PROLONG_ADD || b.opcode != EVOMG_OP_SMOOTH || a.level != b.level || a.level <= prob_.min_level) return false;
    if (b.partitioning != EVOMG_PART_RED_BLACK || b.block[0] * b.block[1] * b.block[2] != 1) return false;
    if (canon(a.level, a.dst) != canon(b.level, b.dst) || canon(b.level, b.src1) == canon(b.level, b.dst)) return false;
    if (prof_matches(a) || prof_matches(b)) return false;
    const int cs = canon(a.level - 1, a.src0);
    return cs >= 0 && lv_[a.level - 1].f[cs] != nullptr && rbgs3d_block_ok(lv_[a.level].g);
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
    const Geo& g = lv_[a.level].g;
    return use_tile3d_blocks(g);
  }

  // RESIDUAL@L, RESTRICT@(L-1) of that residual on a 3D level the tiles serve, the residual field dead afterwards
  // (evomg_tile3d_residual_restrict never writes it)
  bool fusable_residual_restrict3d(const evomg_op_t* ops, int n_ops, int i, bool cyclic) const {
    if (!fuse_enabled_ || !tile3d_rr_ || dim_ != 3 || scalar_traits<T>::is_complex || !cyclic || i + 1 >= n_ops) return false;
    const evomg_op_t &a = ops[i], &b = ops[i + 1];
    if (a.opcode != EVOMG_OP_RESIDUAL || b.opcode != EVOMG_OP_RESTRICT) return false;
    if (b.level != a.level - 1 || canon(a.level, b.src0) != canon(a.level, a.dst)) return false;
    if (canon(a.level, a.dst) == canon(a.level, a.src0) || canon(a.level, a.dst) == canon(a.level, a.src1)) return false;
    if (prob_.bc_kind != EVOMG_BC_DIRICHLET || !(use_tile3d(lv_[a.level].g) || rr3d_block_ok(lv_[a.level].g))) return false;
    if (a.level <= prob_.min_level || prof_matches(b)) return false;
    return residual_dead_after(ops, n_ops, i);
  }

  // ---- outer loop A: generated MG iteration (oracle FN(solve_mg)) ----------------------------------------------------------
  // an out-of-place smoother reads its field and writes the level's spare array; only the host-side pointer swap makes
  // the result current, so launching it speculatively is free of side effects
  bool smoother_is_out_of_place(const evomg_op_t& op) const {
    if (op.opcode != EVOMG_OP_SMOOTH) return false;
    const Geo& g = lv_[op.level].g;
    const int cells = op.block[0] * op.block[1] * (dim_ == 3 ? op.block[2] : 1);
    const bool rb = op.partitioning == EVOMG_PART_RED_BLACK;
    return cells > 1 || use_strip(g) || use_strip3d(g, rb) || (rb && use_tile3d(g) && tile3d_rbgs_) || (rb && rbgs3d_block_ok(g)) || !rb;
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
      if (lds_bytes > max_lds_) return 0;               // not this device: the per-op launches take over
      if (lds_bytes > 48 * 1024 && !small3d_attr_set_) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(evomg_small3d_chain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_) != hipSuccess) {
          (void)hipGetLastError();
          small3d_enabled_ = false;
          return 0;
        }
        small3d_attr_set_ = true;
      }
      const Geo gc = L > prob_.min_level ? lv_[L - 1].g : g;
      hipLaunchKernelGGL(evomg_small3d_chain, dim3(1), dim3(1024), lds_bytes, stream_, g, lv.st, fieldptr(L, sm.dst), (const T*)fieldptr(L, sm.src1),
                         sm.omega, 1.0 - sm.omega, nsweeps, zero_first, coarse_in, gc, om_p, res_out, coarse_out);
      if (hipGetLastError() != hipSuccess) { snprintf(err, sizeof err, "evomg_small3d_chain: launch failed"); return -1; }
      return p - i;
    }
  }

  // ---- cross-iteration fusion (evomg_strip_cross.h) ------------------------------------------------------------------
  // The cycle ends with [PROLONG_ADD +] a red-black point sweep on SOL@max and starts with one, (normally) followed by
  // RESIDUAL + RESTRICT of a residual nobody else reads: the solver loop then runs tail and head of consecutive iterations
  // as ONE launch. head_len / tail_len: ops the launch stands for at the start / end of the program.
  struct CrossPlan {
    bool ok = false, pro = false, tailr = false;
    int head_len = 0, tail_len = 0;
  };
  CrossPlan cross_plan(const evomg_op_t* ops, int n_ops) const {
    CrossPlan p;
    if constexpr (scalar_traits<T>::is_complex) return p;
    if (!xfuse_enabled_ || !fuse_enabled_ || dim_ != 2 || prob_.bc_kind != EVOMG_BC_DIRICHLET || n_ops < 2 || batch_enabled_) return p;
    const int L = prob_.max_level;
    const Geo& g = lv_[L].g;
    if (L <= prob_.min_level || g.n < xfuse_min_n_ || !use_strip(g) || small2d_ok(g)) return p;
    const evomg_op_t &first = ops[0], &last = ops[n_ops - 1];
    if (!tail_norm_candidate(ops, n_ops, n_ops - 1) || last.partitioning != EVOMG_PART_RED_BLACK) return p;
    if (first.opcode != EVOMG_OP_SMOOTH || first.level != L || first.partitioning != EVOMG_PART_RED_BLACK ||
        first.block[0] * first.block[1] != 1 || canon(L, first.dst) != EVOMG_F_SOL || canon(L, first.src1) != EVOMG_F_RHS) return p;
    p.pro = fusable_prolong_smooth(ops, n_ops, n_ops - 2);
    p.tail_len = p.pro ? 2 : 1;
    p.tailr = fusable_smooth_residual_restrict(ops, n_ops, 0, true);
    p.head_len = p.tailr ? 3 : 1;
    if (p.head_len + p.tail_len > n_ops) return p;
    if (p.tailr) {
      // the coarse right-hand side goes to a spare array that a pointer swap makes current: its boundary (never written by a
      // restriction) has to be what the field's is — zero
      const int slot = canon(L - 1, ops[2].dst);
      if (slot < 0 || !bnd_zero_[L - 1][slot] || !lv_[L - 1].f[slot]) return p;
      const int csrc = p.pro ? canon(L - 1, ops[n_ops - 2].src0) : 0;
      if (csrc < 0 || !lv_[L - 1].f[csrc]) return p;
    }
    p.ok = true;
    return p;
  }

  // tail_norm (solver loop only): set to true when the last sweep of the program also produced the canonical partial
  // sums of the squared residual norm (norm_partials_)
  // stop >= 0: execute ops[first .. stop) only (the solver loop runs the cycle's last ops itself: cross_plan); the liveness
  // scans still see the whole program
  int exec_program(const evomg_op_t* ops, int n_ops, bool cyclic, int first = 0, bool* tail_norm = nullptr, int stop = -1) {
    if (tail_norm) *tail_norm = false;
    const int end = stop >= 0 ? stop : n_ops;
    for (int i = first; i < end; ++i) {
      if (batchable(ops[i])) {              // small level: the op joins the list one workgroup runs in one launch
        const int rc = exec_batched(ops[i]);
        if (rc != EVOMG_OK) return rc;
        continue;
      }
      flush_batch();
      if (i + 1 < end && elidable_zero(ops, n_ops, i)) {
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
        const int rc = exec_twice_timed(ops[i + 1], &ops[i]);
        if (rc != EVOMG_OK) return rc;
        i += 2;
        continue;
      }
      if (fusable_twice(ops, n_ops, i)) {
        const int rc = exec_twice_timed(ops[i]);
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
        const double x_bytes = x_zero_next_ ? (double)sizeof(T) * lv.g.n * lv.g.n : 0.0;     // zero-input variant: x is not read
        if (timed) { const int rc = prof_open(); if (rc != EVOMG_OK) return rc; }
        if (sm.partitioning == EVOMG_PART_RED_BLACK)
          launch_strip_restrict_tail<STRIP_RBGS>(lv.g, lv_[L - 1].g, lv.st, x, fieldptr(L, sm.src1), y, om, omm, fieldptr(L - 1, rs.dst));
        else
          launch_strip_restrict_tail<STRIP_JACOBI>(lv.g, lv_[L - 1].g, lv.st, x, fieldptr(L, sm.src1), y, om, omm, fieldptr(L - 1, rs.dst));
        // x, f read + y written + the coarse right-hand side written
        if (timed) { const int rc = prof_close(algorithmic_bytes(sm) - x_bytes + (double)sizeof(T) * lv_[L - 1].g.n * lv_[L - 1].g.n); if (rc != EVOMG_OK) return rc; }
        lv.alt = x;
        set_fieldptr(L, sm.dst, y);
        lv.alt_mirror = y;             // the old array (now the spare one) has the same boundary values
        HIP_TRY(hipGetLastError());
        i += 2;
        continue;
      }
      if (fusable_residual_restrict3d(ops, n_ops, i, cyclic)) {
        const int L = ops[i].level;
        const bool timed = prof_matches(ops[i]);
        if (timed) { const int rc = prof_open(); if (rc != EVOMG_OK) return rc; }
        if (use_tile3d(lv_[L].g)) launch_tile3d_residual_restrict(L, fieldptr(L, ops[i].src0), fieldptr(L, ops[i].src1), fieldptr(L - 1, ops[i + 1].dst));
        else launch_rr3d_block(L, fieldptr(L, ops[i].src0), fieldptr(L, ops[i].src1), fieldptr(L, ops[i].dst), fieldptr(L - 1, ops[i + 1].dst));
        // x, f read + the coarse right-hand side written; the residual field is never written
        if (timed) { const int rc = prof_close(2.0 * sizeof(T) * pow((double)lv_[L].g.n, dim_) + (double)sizeof(T) * pow((double)lv_[L - 1].g.n, dim_)); if (rc != EVOMG_OK) return rc; }
        apply_bc(L - 1, fieldptr(L - 1, ops[i + 1].dst));
        HIP_TRY(hipGetLastError());
        ++i;
        continue;
      }
      if (fusable_residual_restrict(ops, n_ops, i)) {
        const bool write_r = !(cyclic && residual_dead_after(ops, n_ops, i));
        const int L = ops[i].level;
        const bool timed = prof_matches(ops[i]);
        if (timed) { const int rc = prof_open(); if (rc != EVOMG_OK) return rc; }
        launch_residual_restrict(L, fieldptr(L, ops[i].src0), fieldptr(L, ops[i].src1), fieldptr(L, ops[i].dst),
                                 fieldptr(L - 1, ops[i + 1].dst), write_r);
        if (timed) { const int rc = prof_close((write_r ? 3.0 : 2.0) * sizeof(T) * pow((double)lv_[L].g.n, dim_) + (double)sizeof(T) * pow((double)lv_[L - 1].g.n, dim_)); if (rc != EVOMG_OK) return rc; }
        if (!strip_bc_) {                                   // otherwise the kernel wrote the Robin columns itself
          if (write_r) apply_bc(L, fieldptr(L, ops[i].dst));
          apply_bc(L - 1, fieldptr(L - 1, ops[i + 1].dst));
        }
        HIP_TRY(hipGetLastError());
        ++i;
        continue;
      }
      if (fusable_prolong_rbgs3d(ops, n_ops, i)) {
        const evomg_op_t &pr = ops[i], &sm = ops[i + 1];
        const int L = pr.level;
        Level<T>& lv = lv_[L];
        T* x = fieldptr(L, sm.dst);
        launch_rbgs3d_block(lv, x, fieldptr(L, sm.src1), sm.omega, 1.0 - sm.omega, 0, fieldptr(L - 1, pr.src0), lv_[L - 1].g, pr.omega);
        advance(lv, L, sm, x, lv.alt);
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
