// impl_storage.inl — device storage helpers, launch geometry, op-program validation
// Part of the body of `template <typename T> class Impl` (evomg_impl.h includes this file INSIDE the class).

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
