// impl_members.inl — data members and environment knobs
// Part of the body of `template <typename T> class Impl` (evomg_impl.h includes this file INSIDE the class).

  evomg_problem_t prob_;
  int device_;
  int dim_;
  Level<T> lv_[EVOMG_MAX_LEVELS];
  Stencil<T> outer_st_{};
  T* outer_[8] = {};
  hipStream_t stream_ = nullptr;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  T* partials_ = nullptr;
  T* dscalar_ = nullptr;
  T* hscalar_ = nullptr;
  T* hnorm_ = nullptr;             // pinned host scalar the solver loop's final reduction writes directly (host_norm_)
  T* hnorm_dev_ = nullptr;         // ... its device address
  bool host_norm_ = true;        // the final reduction writes the norm into pinned host memory (round 6: environment switch retired — measured, adopted); false where the mapping fails
  int* dinfo_ = nullptr;            // [0] iterations of the last coarse solve, [1] rendezvous miss, [2] iterations of all coarse solves
  long long coarse_iterations_ = 0;
  bool res_pending_ = false;         // RES@max still has to be formed from the iterate the last solve returned (flush_pending_residual)
  int last_info_[4] = {0, 0, 0, 0};  // dinfo_ as read behind the last evaluation; info_fresh_: coarse_solver_health may use it
  bool info_fresh_ = false;
  bool mirror_enabled_ = getenv("EVOMG_NO_BOUNDARY_MIRROR") == nullptr;
  bool x_zero_next_ = false;
  bool zero_elision_ = getenv("EVOMG_NO_ZERO_ELISION") == nullptr;
  bool bnd_zero_[24][EVOMG_F_COUNT];   // boundary of the field's current array is known to be all zero
  // lazy restore of the initial fields (reset_fields): written since the last restore / which array holds the pristine content
  static constexpr bool lazy_reset_ = true;      // (round 6: environment switch retired — measured, adopted)
  bool dirty_[24][EVOMG_F_COUNT];
  const T* clean_ptr_[24][EVOMG_F_COUNT];
  static inline const bool pow2_reciprocal_ = getenv("EVOMG_NO_POW2_RECIPROCAL") == nullptr;
  static inline const bool cplx_const_div_ = getenv("EVOMG_NO_CPLX_CONST_DIV") == nullptr;   // complex relaxations: the division by the centre with its divisor-only part hoisted
  static constexpr bool pow2_fma_ = true;       // 3D tiles: fused neighbour terms where the coefficients are powers of two
  // cross-iteration fusion on the finest 2D level (evomg_strip_cross.h): EVOMG_NO_XFUSE switches it off (same bits either way)
  bool xfuse_enabled_ = getenv("EVOMG_NO_XFUSE") == nullptr;
  // from 2049^2: one long marching pass per wave is a latency chain on smaller levels (same-box A/B of whole V(1,1) solves,
  // fused vs EVOMG_NO_XFUSE: 513^2 1.808 vs 1.747 ms, 1025^2 3.85 vs 3.79, 2049^2 11.89 vs 11.99, 4097^2 13.68 vs 14.12)
  int xfuse_min_n_ = getenv("EVOMG_XFUSE_MIN_N") ? atoi(getenv("EVOMG_XFUSE_MIN_N")) : 2049;
  bool norm_fold_ = getenv("EVOMG_NO_NORM_FOLD") == nullptr;        // the cross launch's norm: partial sums and their sum in ONE launch (same bits)
  // RHS@max shared between the handles of a process (evomg_runtime.h: shared_array_acquire): init[RHS]@max is the registry's
  // array; as long as nothing writes RHS@max the working field f[RHS]@max IS that array (rhs_aliased_) and the handle's own
  // array waits in rhs_private_ for the first writer (privatize_rhs)
  bool share_rhs_ = getenv("EVOMG_NO_SHARED_RHS") == nullptr;
  bool rhs_shared_ = false, rhs_aliased_ = false;
  T* rhs_private_ = nullptr;
  unsigned* norm_ticket_ = nullptr;
  bool xgen_enabled_ = getenv("EVOMG_NO_XGEN") == nullptr;            // two sweeps from zero + residual + restriction on strips: the first sweep formed from f (same bits; tests switch it off)
  bool cross_unit_ = getenv("EVOMG_NO_CROSS_UNIT") == nullptr;     // the cross kernel's unit-weight form (same bits; tests switch it off)
  static constexpr int xfuse_rows_ = 0;      // chunk height of the cross kernel: cross_rows() decides (EVOMG_STRIP_ROWS still forces one, tests)
  T* xf_rhs_spare_ = nullptr;        // spare coarse right-hand side of level max - 1 (speculative output, swapped in on commit)
  T* xf_lane_sums_ = nullptr;        // per (16-row chunk, column pair) sums of the squared residual
  size_t xf_lane_sums_cap_ = 0;
  bool tail_norm_enabled_ = getenv("EVOMG_NO_TAIL_NORM") == nullptr;
  bool tail_restrict_enabled_ = getenv("EVOMG_NO_TAIL_RESTRICT") == nullptr;
  hipEvent_t norm_event_ = nullptr;
  hipEvent_t outer_event_ = nullptr;
  bool speculate_ = getenv("EVOMG_NO_SPECULATION") == nullptr;
  // part of the arithmetic (the oracle restates it: krylov_bicgstab_per_launch): the launch-per-step BiCGStab from 129^2 up
  static constexpr int kb_min_n_ = 129;
  bool outer_folds_ = getenv("EVOMG_NO_OUTER_FOLDS") == nullptr;      // outer BiCGStab: beta formed by the reduction before it, Robin columns by the update before them
  bool kb_fused_ = getenv("EVOMG_NO_KB_FUSE") == nullptr;            // the launch-per-step BiCGStab in three launches per iteration (kb2_*), else six
  static constexpr bool kcg_enabled_ = true;       // large coarsest grids: CG as a sequence of launches (part of the arithmetic: the oracle restates its summation order)
  KcgState* kcg_state_ = nullptr;
  KcgState* kcg_host_ = nullptr;
  KbState<T>* kb_state_ = nullptr;
  KbState<T>* kb_host_ = nullptr;
  KbState<T>* kb_host_dev_ = nullptr;     // its device address (kb2_finish writes the state there itself)
  T* kb_scal_ = nullptr;
  bool outer_fuse_ = getenv("EVOMG_NO_OUTER_FUSE") == nullptr;      // fused passes of the preconditioned BiCGStab loop
  static constexpr bool outer_dev_steps_ = true;   // ... with alpha and omega formed on the device
  static constexpr bool strip_bc_ = true;          // the strip kernels write the Robin columns themselves
  T* norm_partials_ = nullptr;
  size_t norm_partials_cap_ = 0;
  unsigned long long* cg_ll_ = nullptr;   // exchange words of evomg_cg_multi_wg3d (EVOMG_CG_LL_WORDS, zeroed at creation)
  unsigned cg_seq_base_ = 0;              // sequence numbers handed out so far: they never repeat between launches
  static constexpr int cg_stride_ = 8;   // the eight workgroups of a multi-workgroup coarse solve on ONE XCD (every eighth block of the grid); one per XCD measured and dropped (profiles/r05_cg1_placement_raw.txt)
  bool cg_multi_attr_set_ = false;
  bool cg_multi_used_ = false;
  // exchanges of a multi-workgroup solve inside one XCD's L2 (evomg_krylov.h: ll_store16): asked for when all workgroups are
  // placed on one XCD; a solve that misses its rendezvous in that mode turns it off for the handle before the retry
  bool cg_local_exchange_ = true;
  static constexpr bool cg1_wave_autonomous_ = true;      // one-XCD placement: every wave exchanges for itself (the collector form stays the fallback when the workgroups do not share an XCD)
  bool cg_lds_attr_set_[24] = {};
  int prof_opcode_ = -1, prof_level_ = -1;
  size_t prof_used_ = 0;
  double prof_bytes_ = 0.0;
  std::vector<hipEvent_t> prof_events_;
  bool use_reg_krylov_ = (getenv("EVOMG_NO_REG_KRYLOV") == nullptr);
  bool block_fast_ = (getenv("EVOMG_NO_BLOCK_FAST") == nullptr);
  bool fuse_enabled_ = (getenv("EVOMG_NO_FUSE") == nullptr);
  bool tile3d_rr_ = (getenv("EVOMG_NO_TILE3D_RR") == nullptr);
  static constexpr bool balance_rows_ = true;
  bool strip_block_ = (getenv("EVOMG_NO_STRIP_BLOCK") == nullptr);
  // off by default: measured slower than the op-fused multi-CU launches it replaces (one compute unit
  // moves 64 B/clk, so a 129^2 op costs one workgroup what a launch costs; t9 2.90 vs 2.31 ms). EVOMG_BATCH=1: A/B, tests
  static constexpr bool reset_batch_ = true;     // restoring the initial fields: the small ones in one launch
  static constexpr bool residual_dot_fused_ = true;   // generic loop norm: residual and r . r in one launch
  bool rr3d_block_ = getenv("EVOMG_NO_RR3D_BLOCK") == nullptr;       // small 3D levels: residual + restriction in one launch
  bool rbgs3d_block_ = getenv("EVOMG_NO_RBGS3D_BLOCK") == nullptr;   // small 3D levels: both colours of a red-black sweep in one launch
  bool small3d_enabled_ = getenv("EVOMG_NO_SMALL3D") == nullptr;
  bool small3d_attr_set_ = false;
  bool small2d_enabled_ = getenv("EVOMG_NO_SMALL2D") == nullptr;
  // largest grid that runs as LDS tiles: Helmholtz levels 3..7 149 (33) -> 137-146 (65) -> 131 ms (129), nothing beyond; real
  // sweeps at 129^2 are no faster than the strips (4.2 vs 3.7 us back to back), at 65^2 they are (3.7 vs 5.8 us)
  int small2d_max_n_ = getenv("EVOMG_SMALL2D_MAX_N") ? atoi(getenv("EVOMG_SMALL2D_MAX_N")) : (scalar_traits<T>::is_complex ? 257 : 65);   // complex 257^2 as 8 x 8 tiles: C3 48.9 -> 47.8 ms (round 5); real levels: no difference between 65 and 257
  std::set<const void*> small2d_attr_set_;
  bool batch_enabled_ = (getenv("EVOMG_BATCH") != nullptr) && (getenv("EVOMG_NO_STRIP") == nullptr);
  int batch_max_n_ = getenv("EVOMG_BATCH_MAX_N") ? atoi(getenv("EVOMG_BATCH_MAX_N")) : 129;
  MicroBatch batch_ = {};
  int n_cu_ = 256;
  size_t max_lds_ = 64 * 1024;   // hipDeviceAttributeMaxSharedMemoryPerBlock, read at init
  int twice_planes_ = 16;
  bool balance3d_ = getenv("EVOMG_NO_BALANCE3D") == nullptr;   // z-chunk depth of the 3D tiles by workgroup balance (launch3d: balanced_planes)
  bool cg_fold_reg_ = getenv("EVOMG_NO_CG_FOLD") == nullptr;     // COPY + SET_ZERO folded into the register CG too (65^2 .. 129^2)
  bool twice_xzero_ = getenv("EVOMG_NO_TWICE_XZERO") == nullptr;   // SET_ZERO + 3D double sweep: the zero field is not loaded
  bool head_norm_ = getenv("EVOMG_NO_HEAD_NORM") == nullptr;   // 3D: the loop norm rides on the coming cycle's first double sweep (solve_mg: norm_on_head)
  static constexpr bool twice_planes_forced_ = false, rr_cplanes_forced_ = false;
  struct BalanceEntry { long tiles; int inner, halo, lo, align, planes; };
  mutable std::array<BalanceEntry, 16> balance_cache_{};      // balanced_planes: answers already worked out
  mutable unsigned balance_cache_next_ = 0;
  static constexpr bool xbalance3d_ = true;   // 3D tile strips of equal width (launch3d: tile3d_xpitch) — part of the 3D loop norm's canonical order
  int balance3d_cap_ = getenv("EVOMG_BALANCE3D") ? atoi(getenv("EVOMG_BALANCE3D")) : 0;   // > 1: deepest chunk it may pick
  int rr_cplanes_ = 8;
  bool strip_enabled_ = (getenv("EVOMG_NO_STRIP") == nullptr);
  // complex (Helmholtz) problems: from 17^2 — the strip kernels write the Robin columns themselves, where the generic ones
  // need a boundary launch behind every op (reference-range solve 212-224 -> 197-215 ms)
  // real problems too since the levels up to 33^2 run in LDS (evomg_small2d: the fused forms instead of one generic launch per op)
  int strip_min_n_ = getenv("EVOMG_STRIP_MIN_N") ? atoi(getenv("EVOMG_STRIP_MIN_N")) : ((scalar_traits<T>::is_complex || getenv("EVOMG_NO_SMALL2D") == nullptr) ? 17 : 33);
  int strip_rows_ = getenv("EVOMG_STRIP_ROWS") ? atoi(getenv("EVOMG_STRIP_ROWS")) : 0;
  int strip3d_min_n_ = getenv("EVOMG_STRIP3D_MIN_N") ? atoi(getenv("EVOMG_STRIP3D_MIN_N")) : 0;
  int strip_planes_ = getenv("EVOMG_STRIP_PLANES") ? atoi(getenv("EVOMG_STRIP_PLANES")) : 0;
  bool tile3d_enabled_ = getenv("EVOMG_NO_TILE3D") == nullptr;
  static constexpr bool tile3d_rbgs_ = true;
  bool tile3d_twice_ = getenv("EVOMG_NO_TILE3D_TWICE") == nullptr;
  bool tile3d_pro_ = getenv("EVOMG_NO_TILE3D_PRO") == nullptr;
  static constexpr int tile3d_nt_ = 0;   // non-temporal f loads measured slower at 513^3 (profiles/r02_pmc_3d.md)
  static constexpr int tile3d_order_ = 1;
  int tile3d_waves_ = getenv("EVOMG_TILE3D_WAVES") ? atoi(getenv("EVOMG_TILE3D_WAVES")) : 0;
  bool strip_nt_ = getenv("EVOMG_STRIP_NT") ? atoi(getenv("EVOMG_STRIP_NT")) != 0 : true;
  static constexpr double nt_lo_mib_ = 256.0;   // 3 fields above this: stream x, y
  static constexpr double nt_hi_mib_ = 192.0;   // ... unless one field exceeds this
