// impl_launch2d.inl — launch geometry of the 2D strip / LDS-tile kernels and of the block smoothers
// Part of the body of `template <typename T> class Impl` (evomg_impl.h includes this file INSIDE the class).

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
  // complex: a flag — non-zero asks the kernels to divide by |c0|^2 with its reciprocal formed once per launch (evomg_strip.h: div_c0)
  static double inv_c0(const Stencil<cplx>&) { return cplx_const_div_ ? 1.0 : 0.0; }

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
  // (two LDS images at most: a device that cannot hold them keeps the strip kernels)
  bool small2d_ok(const Geo& g) const {
    return small2d_enabled_ && dim_ == 2 && g.n <= small2d_max_n_ && 2 * small2d_image_elems(g.n) * sizeof(T) <= max_lds_;
  }
  template <int MODE, bool PRO, bool TAILR, bool XZ>
  void launch_small2d(const Geo& g, const Stencil<T>& st, const T* x, const T* f, T* y, double om, double omm, double rkh, int rout,
                      const ProlongIn<T>& pro, int nsweeps = 1) {
    auto kern = evomg_small2d<T, MODE, PRO, TAILR, XZ>;
    const int tiles = small2d_tiles(g.n);
    const int halo = (nsweeps > 1 && tiles > 1) ? EVOMG_SMALL2D_HALO2 : EVOMG_SMALL2D_HALO;      // (as the kernel picks it)
    const size_t lds_bytes = (size_t)((MODE == STRIP_JACOBI || TAILR) ? 2 : 1) * small2d_image_elems(g.n, halo) * sizeof(T);
    const void* key = reinterpret_cast<const void*>(kern);
    if (lds_bytes > 48 * 1024 && !small2d_attr_set_.count(key)) {
      if (hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_) != hipSuccess) (void)hipGetLastError();   // the launch reports it
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
      if constexpr (MODE == STRIP_RBGS) {
        if (nsweeps == 2) {
          // two identical sweeps from zero + residual + restriction: the first sweep's result is formed from f inside the
          // launch (evomg_strip2d<.., XGEN>; three halo lanes per side: its own strip count)
          const int gstrips = (g.n + EVOMG_XGEN_STRIP_COLS - 1) / EVOMG_XGEN_STRIP_COLS;
          int grows = strip_rows_ > 0 ? ((strip_rows_ + 1) & ~1) : balanced_rows(g.n, gstrips, 2, 8, 96);
          const int gwaves = gstrips * ((g.n + grows - 1) / grows);
          const dim3 ggrid((((gwaves + 3) / 4 + 7) / 8) * 8);
          hipLaunchKernelGGL((evomg_strip2d<T, MODE, EVOMG_FUSED_PF, false, false, false, true, true, true>), ggrid, block, 0, stream_, g, st, x, f, y, om, omm, grows, gstrips, gwaves, rkh, (T*)nullptr, rout, pro);
          return;
        }
      }
      hipLaunchKernelGGL((evomg_strip2d<T, MODE, EVOMG_FUSED_PF, false, false, false, true, true>), grid, block, 0, stream_, g, st, x, f, y, om, omm, rows, strips, waves, rkh, (T*)nullptr, rout, pro);
      return;
    }
    if (strip_streams(g)) hipLaunchKernelGGL((evomg_strip2d<T, MODE, EVOMG_FUSED_PF, true, false, false, true>), grid, block, 0, stream_, g, st, x, f, y, om, omm, rows, strips, waves, rkh, (T*)nullptr, rout, pro);
    else hipLaunchKernelGGL((evomg_strip2d<T, MODE, EVOMG_FUSED_PF, false, false, false, true>), grid, block, 0, stream_, g, st, x, f, y, om, omm, rows, strips, waves, rkh, (T*)nullptr, rout, pro);
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

  // Fused residual + squared norm (the solver loop's residual check): the residual field is still written, the norm
  // costs no second pass. The summation order — per (strip, 16-row chunk) wave, lane = column pair, rows ascending —
  // is canonical for 2D levels of at least 33 nodes (the oracle's FN(norm_strip2d)) and does NOT depend on
  // EVOMG_NO_STRIP or the tuning knobs.
  // where the solver loop's final reduction writes the norm: straight into pinned host memory (no copy launch behind it), or the device scalar
  T* norm_out() { return host_norm_ ? hnorm_dev_ : dscalar_; }
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
    hipLaunchKernelGGL((evomg_dot_final<T>), dim3(1), dim3(256), 0, stream_, norm_partials_ + count, kDotBlocks, norm_out());
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
    hipLaunchKernelGGL((evomg_dot_final<T>), dim3(1), dim3(256), 0, stream_, norm_partials_, waves, norm_out());
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

  // Chunk height of the cross kernel (multiples of 16 rows). It holds 144 VGPRs — three waves per SIMD, i.e. at most three
  // 4-wave blocks per compute unit — and every block is meant to be resident at once, so a launch lasts as long as the unit
  // with the most blocks: ceil(blocks / units) x (rows + 9 pipeline steps); heights that need more than three blocks on a unit
  // pay a whole extra round. At 4097^2 (36 strips): 64 rows -> 585 blocks, three on some units, two on most (measured 103 us);
  // 80 rows -> 468 blocks, two on nearly every unit.
  int cross_rows(const Geo& g, int strips) const {
    const int n = g.n;
    // With the device full of other evaluations (candidates in flight) a launch on a level that streams from HBM neither needs
    // every unit for itself nor ends when its slowest unit does: what counts is the bytes it moves, so the chunks are tall —
    // 256 rows: eleven halo rows are 4 % of x and f instead of 14 %. Same-box A/B with sixteen C2 solves (4097^2) in flight:
    // 402-405 -> 412-417 and 368 -> 375 solves/s on two boxes (384 and 512 rows: less, 1024: a loss; the same rule for the
    // other strips of the big levels: less; with eight in flight +1 %). NOT on levels the Infinity Cache holds: 2049^2 with
    // sixteen in flight 849 -> 780 solves/s — there the march of a wave is what a launch costs. Placement only: the same bits.
    if (strip_streams(g) && evaluations_in_flight() >= 8) return 16 * EVOMG_NORM_ROWS;
    long best_cost = -1;
    int best = 4 * EVOMG_NORM_ROWS;
    for (int r = 2 * EVOMG_NORM_ROWS; r <= 8 * EVOMG_NORM_ROWS; r += EVOMG_NORM_ROWS) {
      const long blocks = ((long)strips * ((n + r - 1) / r) + 3) / 4;
      const long per_unit = (blocks + n_cu_ - 1) / n_cu_;
      const long rounds = (per_unit + 2) / 3;                           // three resident blocks per unit
      const long cost = (rounds > 1 ? rounds * 3 : per_unit) * (r + 9);
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; }
    }
    return best;
  }
  // ---- cross-iteration launch (evomg_strip_cross.h): tail of this iteration + head of the next on the finest level -------
  // Reads SOL@max (x), RHS@max, the coarse correction; writes w into the level's spare array, the next iteration's coarse
  // right-hand side into xf_rhs_spare_, the per-pair norm sums; then reduces those to the canonical partial sums in
  // norm_partials_. Nothing it writes is current until cross_commit().
  // *norm_done: the launch has also summed the norm's partial sums (the solver loop queues no reduction of its own)
  int launch_cross(const evomg_op_t* ops, int n_ops, const CrossPlan& plan, bool* norm_done = nullptr) {
    if constexpr (scalar_traits<T>::is_complex) return invalid("cross-iteration fusion: real problems only");
    else {
      const int L = prob_.max_level;
      Level<T>& lv = lv_[L];
      const Geo& g = lv.g;
      const Geo& gc = lv_[L - 1].g;
      const evomg_op_t &sa = ops[n_ops - 1], &sb = ops[0];
      const int strips = (g.n + EVOMG_CROSS_STRIP_COLS - 1) / EVOMG_CROSS_STRIP_COLS;
      const int forced_rows = xfuse_rows_ > 0 ? xfuse_rows_ : strip_rows_;
      int rows = forced_rows > 0 ? ((forced_rows + EVOMG_NORM_ROWS - 1) / EVOMG_NORM_ROWS) * EVOMG_NORM_ROWS : cross_rows(g, strips);
      const int chunks = (g.n + rows - 1) / rows;
      const int waves = strips * chunks;
      const int chunks16 = (g.n + EVOMG_NORM_ROWS - 1) / EVOMG_NORM_ROWS;
      const int sums_pitch = (((g.n + 1) / 2 + 15) / 16) * 16;
      const size_t need = (size_t)chunks16 * sums_pitch;
      if (need > xf_lane_sums_cap_) {
        if (xf_lane_sums_) (void)hipFree(xf_lane_sums_);
        xf_lane_sums_ = nullptr;
        HIP_TRY(hipMalloc(&xf_lane_sums_, sizeof(T) * need));
        xf_lane_sums_cap_ = need;
      }
      if (plan.tailr && !xf_rhs_spare_) HIP_TRY(alloc_zero(&xf_rhs_spare_, lv_[L - 1].elems));
      const int nstrips_norm = (g.n + EVOMG_NORM_STRIP_COLS - 1) / EVOMG_NORM_STRIP_COLS;
      {
        const int rc = ensure_norm_partials((size_t)nstrips_norm * chunks16);
        if (rc != EVOMG_OK) return rc;
      }
      CrossArgs a;
      a.coarse_in = plan.pro ? fieldptr(L - 1, ops[n_ops - 2].src0) : nullptr;
      a.gc = gc;
      a.om_p = plan.pro ? ops[n_ops - 2].omega : 0.0;
      a.coarse_out = xf_rhs_spare_;
      a.om_a = sa.omega; a.omm_a = 1.0 - sa.omega;
      a.om_b = sb.omega; a.omm_b = 1.0 - sb.omega;
      a.inv_c0 = inv_c0(lv.st);
      a.lane_sums = xf_lane_sums_;
      a.sums_pitch = sums_pitch;
      const T* x = fieldptr(L, EVOMG_F_SOL);
      const T* f = fieldptr(L, EVOMG_F_RHS);
      T* y = lv.alt;
      const dim3 grid((((waves + 3) / 4 + 7) / 8) * 8), block(256);
      const bool nt = strip_streams(g);
      const bool timed = prof_matches(sa) || prof_matches(sb);
      if (timed) { const int rc = prof_open(); if (rc != EVOMG_OK) return rc; }
      // the kernel's fused form needs EVERY coefficient to be a power of two (2D Poisson: -1/h^2, 4/h^2); else the general form
      bool recip = a.inv_c0 != 0.0;
      for (int e = 1; e <= 4; ++e) {
        int ex = 0;
        const double m = frexp(lv.st.c[e], &ex);
        recip = recip && (m == 0.5 || m == -0.5) && ex > -900 && ex < 900;
      }
      // unit weights on both sweeps and on the folded prolongation (the C2 cycle): the kernel's shortest form
      const bool unit = recip && cross_unit_ && a.om_a == 1.0 && a.om_b == 1.0 && (!plan.pro || a.om_p == 1.0);
#define EVOMG_CROSS_LAUNCH(NT, PRO, TR, RC, UN) hipLaunchKernelGGL((evomg_strip2d_cross<NT, PRO, TR, RC, UN>), grid, block, 0, stream_, g, lv.st, x, f, y, rows, strips, waves, a)
#define EVOMG_CROSS_NT(PRO, TR) do { if (nt && unit) EVOMG_CROSS_LAUNCH(true, PRO, TR, true, true); else if (unit) EVOMG_CROSS_LAUNCH(false, PRO, TR, true, true); \
                                     else if (nt && recip) EVOMG_CROSS_LAUNCH(true, PRO, TR, true, false); else if (nt) EVOMG_CROSS_LAUNCH(true, PRO, TR, false, false); \
                                     else if (recip) EVOMG_CROSS_LAUNCH(false, PRO, TR, true, false); else EVOMG_CROSS_LAUNCH(false, PRO, TR, false, false); } while (0)
      if (plan.pro && plan.tailr) EVOMG_CROSS_NT(true, true);
      else if (plan.pro) EVOMG_CROSS_NT(true, false);
      else if (plan.tailr) EVOMG_CROSS_NT(false, true);
      else EVOMG_CROSS_NT(false, false);
#undef EVOMG_CROSS_NT
#undef EVOMG_CROSS_LAUNCH
      // x, f read, w written; the coarse correction read and the coarse right-hand side written (8 B per coarse node each)
      if (timed) {
        const double nc2 = (double)gc.n * gc.n * sizeof(T);
        const int rc = prof_close(3.0 * sizeof(T) * (double)g.n * g.n + (plan.pro ? nc2 : 0.0) + (plan.tailr ? nc2 : 0.0));
        if (rc != EVOMG_OK) return rc;
      }
      const int nwaves_norm = nstrips_norm * chunks16;
      if (norm_fold_) {
        // canonical partial sums AND their sum — the loop's squared norm — in one launch (the last workgroup to arrive adds up)
        if (!norm_ticket_) {
          HIP_TRY(hipMalloc(&norm_ticket_, 128));
          HIP_TRY(hipMemsetAsync(norm_ticket_, 0, 128, stream_));
        }
        hipLaunchKernelGGL(evomg_norm_from_lane_sums_final, dim3((nwaves_norm + 4 * EVOMG_NORM_FOLD_PER - 1) / (4 * EVOMG_NORM_FOLD_PER)), dim3(256), 0, stream_, (const double*)xf_lane_sums_,
                           sums_pitch, g.n, nstrips_norm, nwaves_norm, (double*)norm_partials_, norm_ticket_, (double*)norm_out());
        if (norm_done) *norm_done = true;
      } else {
        hipLaunchKernelGGL(evomg_norm_from_lane_sums, dim3((nwaves_norm + 3) / 4), dim3(256), 0, stream_, (const double*)xf_lane_sums_, sums_pitch,
                           g.n, nstrips_norm, nwaves_norm, norm_partials_);
      }
      HIP_TRY(hipGetLastError());
      return EVOMG_OK;
    }
  }
  // the loop goes on: what the cross launch wrote becomes current (w = the iterate after the next cycle's first sweep; the
  // coarse right-hand side of the next cycle)
  void cross_commit(const evomg_op_t* ops, const CrossPlan& plan) {
    const int L = prob_.max_level;
    Level<T>& lv = lv_[L];
    T* x = fieldptr(L, EVOMG_F_SOL);
    T* y = lv.alt;
    lv.alt = x;
    set_fieldptr(L, EVOMG_F_SOL, y);
    lv.alt_mirror = y;
    if (plan.tailr) {
      const int slot = canon(L - 1, ops[2].dst);
      std::swap(lv_[L - 1].f[slot], xf_rhs_spare_);
    }
  }
  // the loop stops: the iterate it returns is z = the cycle's last sweep applied to x — formed now, by the kernel the
  // unfused path runs for that op (x is untouched; w and the spare coarse field are simply dropped)
  int cross_finalize(const evomg_op_t* ops, int n_ops, const CrossPlan& plan) {
    const int L = prob_.max_level;
    Level<T>& lv = lv_[L];
    const evomg_op_t& sa = ops[n_ops - 1];
    T* x = fieldptr(L, EVOMG_F_SOL);
    T* y = lv.alt;
    const T* f = fieldptr(L, EVOMG_F_RHS);
    const double om = sa.omega, omm = 1.0 - sa.omega;
    int rc;
    if (plan.pro) rc = launch_strip_fused<STRIP_RBGS, true, false>(lv.g, lv_[L - 1].g, lv.st, x, f, y, om, omm, fieldptr(L - 1, ops[n_ops - 2].src0), ops[n_ops - 2].omega);
    else rc = launch_strip_fused<STRIP_RBGS, false, false>(lv.g, lv.g, lv.st, x, f, y, om, omm, nullptr, 0.0);
    if (rc != EVOMG_OK) return rc;
    lv.alt = x;
    set_fieldptr(L, EVOMG_F_SOL, y);
    lv.alt_mirror = y;
    HIP_TRY(hipGetLastError());
    return EVOMG_OK;
  }
