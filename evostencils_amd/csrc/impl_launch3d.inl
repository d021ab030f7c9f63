// impl_launch3d.inl — launch geometry of the 3D marching tiles
// Part of the body of `template <typename T> class Impl` (evomg_impl.h includes this file INSIDE the class).

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
  // every off-centre coefficient of the 3D stencil is a (signed) power of two: the tile kernels' fused forms (evomg_strip.h:
  // msub / madd) give the same bits with one instruction per neighbour term instead of two
  bool offcentre_pow2(const Stencil<T>& st) const {
    if constexpr (scalar_traits<T>::is_complex) return false;
    else {
      if (!pow2_fma_) return false;
      for (int e = 1; e <= 6; ++e) {
        int ex = 0;
        const double m = frexp(st.c[e], &ex);
        if (!(m == 0.5 || m == -0.5) || ex < -900 || ex > 900) return false;
      }
      return true;
    }
  }
  // columns a strip of a 3D tile launch stores: the n - 1 columns left of the east boundary spread evenly over the strips
  // (even, <= cols), instead of `cols` each with a nearly empty last strip (513^3: 5 x 104 instead of 4 x 124 + 17) — with
  // one workgroup per compute unit and a round or two per launch the heaviest workgroup IS the launch. The 3D loop norm's
  // canonical order is defined on these strips (oracle: FN(norm_tile3d)).
  int tile3d_xpitch(int n, int cols) const {
    const int strips = (n + cols - 1) / cols;
    if (!xbalance3d_) return cols;
    const int p = 2 * ((n - 1 + 2 * strips - 1) / (2 * strips));
    return p < cols ? p : cols;
  }
  // z-chunk depth by workgroup balance: one 512-thread tile workgroup is resident per compute unit (245 VGPRs), so a launch
  // runs in ceil(workgroups / units) rounds of (depth + halo planes + prologue) plane steps each; deeper chunks re-read
  // fewer halo planes but leave the last round partly empty. Picks the depth (a multiple of `align`, at least `lo`) with
  // the fewest plane steps on the critical path; ties go to the shallower chunk (more workgroups in flight).
  int balanced_planes(long tiles, int inner, int halo, int lo, int align) const {
    // (a handful of distinct questions per handle, asked once per launch: remembered)
    for (const auto& e : balance_cache_)
      if (e.tiles == tiles && e.inner == inner && e.halo == halo && e.lo == lo && e.align == align) return e.planes;
    long best_cost = -1;
    int best = lo;
    for (int c = 1; c <= inner; ++c) {
      int p = (inner + c - 1) / c;
      p = ((p + align - 1) / align) * align;
      if (p < lo) break;
      if (balance3d_cap_ > 1 && p > balance3d_cap_) continue;
      const long chunks = (inner + p - 1) / p;
      const long rounds = (tiles * chunks + n_cu_ - 1) / n_cu_;
      const long cost = rounds * (p + halo + 2);
      if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best = p; }
    }
    balance_cache_[balance_cache_next_++ % balance_cache_.size()] = BalanceEntry{tiles, inner, halo, lo, align, best};
    return best;
  }
  // 3D red-black sweeps on the levels no marching kernel serves (below 129^3): one launch for both colours (evomg_rbgs3d_block)
  bool rbgs3d_block_ok(const Geo& g) const {
    return rbgs3d_block_ && dim_ == 3 && std::is_same<T, double>::value && prob_.bc_kind == EVOMG_BC_DIRICHLET && g.n >= 5 &&
           !use_strip3d(g, true) && !(use_tile3d(g) && tile3d_rbgs_);
  }
  // residual + restriction on those levels in one launch (evomg_residual_restrict3d_block), the residual field dead
  bool rr3d_block_ok(const Geo& g) const {
    return rr3d_block_ && dim_ == 3 && std::is_same<T, double>::value && prob_.bc_kind == EVOMG_BC_DIRICHLET && g.n >= 5 && !use_tile3d(g);
  }
  void launch_rr3d_block(int L, const T* x, const T* f, T* res, T* coarse) {
    if constexpr (std::is_same<T, double>::value) {
      const Geo &g = lv_[L].g, &gc = lv_[L - 1].g;
      const int inner = gc.n - 2;
      const int nbx = (inner + EVOMG_RR3D_CX - 1) / EVOMG_RR3D_CX, nby = (inner + EVOMG_RR3D_CY - 1) / EVOMG_RR3D_CY,
                nbz = (inner + EVOMG_RR3D_CZ - 1) / EVOMG_RR3D_CZ;
      hipLaunchKernelGGL(evomg_residual_restrict3d_block, dim3(nbx * nby * nbz), dim3(1024), 0, stream_, g, gc, lv_[L].st, x, f, res, coarse, nbx, nby, nbz);
    }
  }
  // x -> lv.alt (the caller advances); zero_first / coarse: the SET_ZERO or PROLONG_ADD in front of the sweep, folded in
  void launch_rbgs3d_block(Level<T>& lv, const T* x, const T* f, double om, double omm, int zero_first, const T* coarse, const Geo& gc, double om_p) {
    T* y = lv.alt;
    ensure_faces(lv, x, y);
    if constexpr (std::is_same<T, double>::value) {
      const Geo& g = lv.g;
      const int inner = g.n - 2;
      const int nbx = (inner + EVOMG_RB3D_BX - 1) / EVOMG_RB3D_BX, nby = (inner + EVOMG_RB3D_BY - 1) / EVOMG_RB3D_BY,
                nbz = (inner + EVOMG_RB3D_BZ - 1) / EVOMG_RB3D_BZ;
      hipLaunchKernelGGL(evomg_rbgs3d_block, dim3(nbx * nby * nbz), dim3(EVOMG_RB3D_THREADS), 0, stream_, g, lv.st, x, f, y, om, omm, nbx, nby,
                         zero_first, coarse, gc, om_p);
    }
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
    if (strip_planes_ <= 0 && balance3d_ && nw == 8) planes = balanced_planes((long)strips * ygroups, inner, 2, 8, 4);
    planes = (planes + 3) & ~3;
    const int chunks = (inner + planes - 1) / planes;
    const int blocks = strips * ygroups * chunks;
    const dim3 grid(((blocks + 7) / 8) * 8);
    const int nt = strip_streams3d(g) ? tile3d_nt_ : 0;
    if constexpr (!scalar_traits<T>::is_complex) {
      const bool p2 = offcentre_pow2(st);
#define EVOMG_TILE3D_LAUNCH(NW, NT) do { if (p2) hipLaunchKernelGGL((evomg_tile3d<T, MODE, NW, NT, true>), grid, dim3(64 * NW), 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks, two, tile3d_order_, aux, tile3d_xpitch(g.n, EVOMG_STRIP_COLS)); \
                                         else hipLaunchKernelGGL((evomg_tile3d<T, MODE, NW, NT, false>), grid, dim3(64 * NW), 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks, two, tile3d_order_, aux, tile3d_xpitch(g.n, EVOMG_STRIP_COLS)); } while (0)
#define EVOMG_TILE3D_NT(NW) if (nt & 1) EVOMG_TILE3D_LAUNCH(NW, 1); else EVOMG_TILE3D_LAUNCH(NW, 0);
      if (nw == 8) { EVOMG_TILE3D_NT(8) } else { EVOMG_TILE3D_NT(4) }
#undef EVOMG_TILE3D_NT
#undef EVOMG_TILE3D_LAUNCH
    }
  }
  // the double sweep's strips are those of the 3D loop norm's canonical order (always at 2^k + 1 nodes up to 1025^3)
  bool twice_strips_canonical(const Geo& g, bool b211) const {
    const int cols = b211 ? 120 : EVOMG_STRIP_COLS;
    return (g.n + cols - 1) / cols == (g.n + EVOMG_STRIP_COLS - 1) / EVOMG_STRIP_COLS &&
           tile3d_xpitch(g.n, cols) == tile3d_xpitch(g.n, EVOMG_STRIP_COLS);
  }
  // two identical Jacobi-type sweeps in a row on one field (evomg_tile3d_twice): same geometry as the red-black tiles
  template <int MODE> void launch_tile3d_twice(const Geo& g, const Stencil<T>& st, const T* x, const T* f, T* y, double om, double omm,
                                               TwoCellPlan<T> two, const T* coarse = nullptr, const Geo* gc = nullptr, double om_p = 0.0,
                                               T* norm_aux = nullptr, bool xzero = false) {
    constexpr int TY = EVOMG_TILE3D_TY;
    constexpr int cols = (MODE == TILE_BLOCK211) ? 120 : EVOMG_STRIP_COLS;      // two halo lanes per side for the blocks
    const int strips = (g.n + cols - 1) / cols;
    const int nw = 8;
    const int stored = nw * TY - 2;
    const int ygroups = (g.n - 3 + stored - 1) / stored > 0 ? (g.n - 3 + stored - 1) / stored : 1;
    int planes = strip_planes_ > 0 ? strip_planes_ : twice_planes_;
    if (strip_planes_ <= 0 && balance3d_ && !twice_planes_forced_) planes = balanced_planes((long)strips * ygroups, g.n - 2, 4, 8, 4);
    planes = (planes + 3) & ~3;
    const int chunks = (g.n - 2 + planes - 1) / planes;
    const int blocks = strips * ygroups * chunks;
    const dim3 grid(((blocks + 7) / 8) * 8);
    if constexpr (!scalar_traits<T>::is_complex) {
      const bool p2 = offcentre_pow2(st);
#define EVOMG_TWICE_LAUNCH(PRO, P2, PROLONG) hipLaunchKernelGGL((evomg_tile3d_twice<T, MODE, 8, 0, PRO, P2>), grid, dim3(512), 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks, two, PROLONG, tile3d_xpitch(g.n, cols), (T*)nullptr)
#define EVOMG_TWICE_NORM_LAUNCH(P2) hipLaunchKernelGGL((evomg_tile3d_twice<T, MODE, 8, 0, false, P2, true>), grid, dim3(512), 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks, two, (TwiceProlong<T>{nullptr, g, 0.0}), tile3d_xpitch(g.n, cols), norm_aux)
#define EVOMG_TWICE_XZ_LAUNCH(P2) hipLaunchKernelGGL((evomg_tile3d_twice<T, MODE, 8, 0, false, P2, false, true>), grid, dim3(512), 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks, two, (TwiceProlong<T>{nullptr, g, 0.0}), tile3d_xpitch(g.n, cols), (T*)nullptr)
      if (xzero && !coarse && !norm_aux) { if (p2) EVOMG_TWICE_XZ_LAUNCH(true); else EVOMG_TWICE_XZ_LAUNCH(false); }
      else if (norm_aux && !coarse) { if (p2) EVOMG_TWICE_NORM_LAUNCH(true); else EVOMG_TWICE_NORM_LAUNCH(false); }
      else if (coarse) { if (p2) EVOMG_TWICE_LAUNCH(true, true, (TwiceProlong<T>{coarse, *gc, om_p})); else EVOMG_TWICE_LAUNCH(true, false, (TwiceProlong<T>{coarse, *gc, om_p})); }
      else { if (p2) EVOMG_TWICE_LAUNCH(false, true, (TwiceProlong<T>{nullptr, g, 0.0})); else EVOMG_TWICE_LAUNCH(false, false, (TwiceProlong<T>{nullptr, g, 0.0})); }
#undef EVOMG_TWICE_LAUNCH
#undef EVOMG_TWICE_NORM_LAUNCH
#undef EVOMG_TWICE_XZ_LAUNCH
    }
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
    if (strip_planes_ <= 0 && balance3d_ && nw == 8) planes = balanced_planes((long)strips * ygroups, g.n - 2, 3, 8, 4);
    planes = (planes + 3) & ~3;
    const int chunks = (g.n - 2 + planes - 1) / planes;
    const int blocks = strips * ygroups * chunks;
    const dim3 grid(((blocks + 7) / 8) * 8);
    if constexpr (!scalar_traits<T>::is_complex) {
      const bool p2 = offcentre_pow2(st);
#define EVOMG_RBGS3D_LAUNCH(NW, P2) hipLaunchKernelGGL((evomg_tile3d_rbgs<T, NW, 0, P2>), grid, dim3(64 * NW), 0, stream_, g, st, x, f, y, om, omm, planes, strips, ygroups, blocks, tile3d_xpitch(g.n, EVOMG_STRIP_COLS))
      if (nw == 8) { if (p2) EVOMG_RBGS3D_LAUNCH(8, true); else EVOMG_RBGS3D_LAUNCH(8, false); }
      else { if (p2) EVOMG_RBGS3D_LAUNCH(4, true); else EVOMG_RBGS3D_LAUNCH(4, false); }
#undef EVOMG_RBGS3D_LAUNCH
    }
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
    if (strip_planes_ <= 0 && balance3d_ && nw == 8 && !rr_cplanes_forced_) cplanes = balanced_planes((long)strips * ((rows_c + 14) / 15), rows_c, 1, 4, 1);
    const int chunks = (rows_c + cplanes - 1) / cplanes;
    const int blocks = strips * ygroups * chunks;
    const dim3 grid(((blocks + 7) / 8) * 8);
    if constexpr (!scalar_traits<T>::is_complex) {
      const bool p2 = offcentre_pow2(lv_[L].st);
#define EVOMG_RR3D_LAUNCH(NW, P2) hipLaunchKernelGGL((evomg_tile3d_residual_restrict<T, NW, 0, P2>), grid, dim3(64 * NW), 0, stream_, g, gc, lv_[L].st, x, f, coarse, cplanes, strips, ygroups, blocks, tile3d_xpitch(g.n, EVOMG_STRIP_COLS))
      if (nw == 8) { if (p2) EVOMG_RR3D_LAUNCH(8, true); else EVOMG_RR3D_LAUNCH(8, false); }
      else { if (p2) EVOMG_RR3D_LAUNCH(4, true); else EVOMG_RR3D_LAUNCH(4, false); }
#undef EVOMG_RR3D_LAUNCH
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
