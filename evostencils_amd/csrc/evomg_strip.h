// evomg_strip.h — the 2D "marching strip" kernels: Jacobi (K1), fused two-colour red-black Gauss-Seidel (K2) and
// residual (K4) for 5-point stencils at the HBM roofline.
//
// Work decomposition (no LDS, no barriers, no atomics):
//   * one wave64 owns a vertical strip of 128 consecutive columns: lane l holds the column PAIR (2l, 2l+1) of the
//     strip, so every global access is one aligned 16-byte load/store per lane (1 KiB per wave instruction);
//   * the wave marches down its chunk of rows with a three-row register window (rows j-1, j, j+1 of x), so each
//     x row is fetched once per strip; rows are prefetched PF steps ahead through a register ring to keep
//     (PF+1) x 2 KiB per wave in flight;
//   * horizontal neighbours live in the adjacent lanes: one DPP wave shift (v_mov_b32_dpp wave_shr:1 / wave_shl:1)
//     per 32-bit half, no LDS round trip. Lanes 0 and 63 are halo lanes: only the 62 interior pairs (124 columns)
//     are stored, which is what makes the strip self-contained (3% redundant column loads, served by L2);
//   * red-black Gauss-Seidel runs BOTH colours in the same pass, out of place: at step j the wave relaxes the red
//     nodes of row j from old values, then the black nodes of row j-1 from the new reds of rows j-2, j-1, j held in
//     registers. Algorithmic traffic 24 B/node (x, f read once; y written once) instead of 48 B/node for two
//     in-place colour passes; the result is bit-identical to the two-pass form (same formula, same operands).
//
// Vertical halo: a chunk of H rows re-reads 2 (Jacobi/residual) or 4 (RB-GS) rows of x: (H+4)/H.
#pragma once
#include "evomg_kernels.h"

#define EVOMG_STRIP_COLS 124   // stored columns per strip (62 interior lane pairs)

enum { STRIP_JACOBI = 0, STRIP_RBGS = 1, STRIP_RESIDUAL = 2, STRIP_RESNORM = 3 };
#define EVOMG_NORM_ROWS 16   // chunk height of the fused residual + norm pass: part of the canonical summation order
// Strips that carry the residual norm have TWO halo lanes per side (120 stored columns): a fused red-black sweep leaves
// garbage in the black nodes of its outermost lanes (their red neighbour has no west / east value), which is harmless
// for the sweep itself — nobody reads them — but the residual of the first stored column would. With two halo lanes
// every value the residual touches is exact. The width is part of the canonical summation order of the norm.
#define EVOMG_NORM_STRIP_COLS 120
#define EVOMG_XGEN_STRIP_COLS 116   // three halo lanes per side (XGEN below)

template <typename T> struct Pair {
  T a, b;
};

template <typename T> __device__ __forceinline__ Pair<T> load_pair(const T* __restrict__ p);
template <> __device__ __forceinline__ Pair<double> load_pair<double>(const double* __restrict__ p) {
  const double2 v = *reinterpret_cast<const double2*>(p);
  return Pair<double>{v.x, v.y};
}
template <> __device__ __forceinline__ Pair<cplx> load_pair<cplx>(const cplx* __restrict__ p) {
  const double2 v0 = reinterpret_cast<const double2*>(p)[0];
  const double2 v1 = reinterpret_cast<const double2*>(p)[1];
  return Pair<cplx>{cplx{v0.x, v0.y}, cplx{v1.x, v1.y}};
}
typedef double evomg_f64x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ Pair<T> load_pair_nt(const T* __restrict__ p);
template <> __device__ __forceinline__ Pair<double> load_pair_nt<double>(const double* __restrict__ p) {
  const evomg_f64x2 v = __builtin_nontemporal_load(reinterpret_cast<const evomg_f64x2*>(p));
  return Pair<double>{v.x, v.y};
}
template <> __device__ __forceinline__ Pair<cplx> load_pair_nt<cplx>(const cplx* __restrict__ p) {
  const evomg_f64x2 v0 = __builtin_nontemporal_load(reinterpret_cast<const evomg_f64x2*>(p));
  const evomg_f64x2 v1 = __builtin_nontemporal_load(reinterpret_cast<const evomg_f64x2*>(p) + 1);
  return Pair<cplx>{cplx{v0.x, v0.y}, cplx{v1.x, v1.y}};
}
template <bool NT = false> __device__ __forceinline__ void store_pair(double* __restrict__ p, const Pair<double>& v) {
  if constexpr (NT) {
    evomg_f64x2 t;
    t.x = v.a;
    t.y = v.b;
    __builtin_nontemporal_store(t, reinterpret_cast<evomg_f64x2*>(p));
  } else {
    *reinterpret_cast<double2*>(p) = make_double2(v.a, v.b);
  }
}
template <bool NT = false> __device__ __forceinline__ void store_pair(cplx* __restrict__ p, const Pair<cplx>& v) {
  if constexpr (NT) {
    evomg_f64x2 t0, t1;
    t0.x = v.a.re; t0.y = v.a.im;
    t1.x = v.b.re; t1.y = v.b.im;
    __builtin_nontemporal_store(t0, reinterpret_cast<evomg_f64x2*>(p));
    __builtin_nontemporal_store(t1, reinterpret_cast<evomg_f64x2*>(p) + 1);
  } else {
    reinterpret_cast<double2*>(p)[0] = make_double2(v.a.re, v.a.im);
    reinterpret_cast<double2*>(p)[1] = make_double2(v.b.re, v.b.im);
  }
}

// same operation order as relax_point / apply_stencil (evomg_kernels.h) and the oracle's FN(relax2)/FN(apply_st2)
// inv_c0 != 0: the centre coefficient is a power of two (4 / h^2 of the 2D Poisson stencil on a dyadic mesh), so the
// division is an exact scaling and r * (1 / c0) gives the same bits as r / c0 at a tenth of the instructions
// (host-checked, real scalars only)
__device__ __forceinline__ double div_c0(double r, double c0, double inv_c0) {
  return inv_c0 != 0.0 ? r * inv_c0 : r / c0;
}
// complex: inv_c0 != 0 is the refined reciprocal y of d = |c0|^2 (relax_div_const below, formed once per kernel): sc_div's two
// quotients by d — the divisor is the same for every node of the launch — as three operations each (evomg_scalar.h: const_div,
// the compiler's own division sequence with its divisor-only part hoisted: the same bits) instead of eleven, a rcp among them;
// a complex relaxation is 50 instead of 69 FP64 instructions
__device__ __forceinline__ cplx div_c0(cplx r, cplx c0, double inv_c0) {
  if (inv_c0 == 0.0) return sc_div(r, c0);
  const double d = c0.re * c0.re + c0.im * c0.im;            // sc_div's own numerators and denominator
  const ConstDiv<double> cd{d, inv_c0};
  return cplx{const_div(r.re * c0.re + r.im * c0.im, cd), const_div(r.im * c0.re - r.re * c0.im, cd)};
}
// what a kernel hands relax5 as inv_c0: real problems the host's exact reciprocal of a power-of-two centre (or 0); complex
// problems the device-formed reciprocal of |c0|^2 when the host asks for it (host_inv_c0 != 0: impl_launch2d.inl inv_c0)
__device__ __forceinline__ double relax_div_const(const Stencil<double>&, double host_inv_c0) { return host_inv_c0; }
__device__ __forceinline__ double relax_div_const(const Stencil<cplx>& st, double host_inv_c0) {
  if (host_inv_c0 == 0.0) return 0.0;
  return const_div_make(st.c[0].re * st.c[0].re + st.c[0].im * st.c[0].im).y;
}
template <typename T>
__device__ __forceinline__ T relax5(const Stencil<T>& st, T c, T w, T e, T s, T n, T f, double om, double omm, double inv_c0 = 0.0) {
  T r = f;
  r = sc_sub(r, sc_mul(st.c[1], w));
  r = sc_sub(r, sc_mul(st.c[2], e));
  r = sc_sub(r, sc_mul(st.c[3], s));
  r = sc_sub(r, sc_mul(st.c[4], n));
  const T xs = div_c0(r, st.c[0], inv_c0);
  return sc_add(sc_rmul(omm, c), sc_rmul(om, xs));
}
template <typename T>
__device__ __forceinline__ T residual5(const Stencil<T>& st, T c, T w, T e, T s, T n, T f) {
  T a = sc_mul(st.c[0], c);
  a = sc_add(a, sc_mul(st.c[1], w));
  a = sc_add(a, sc_mul(st.c[2], e));
  a = sc_add(a, sc_mul(st.c[3], s));
  a = sc_add(a, sc_mul(st.c[4], n));
  return sc_sub(f, a);
}

// PRO: the sweep's input is x + om * P(coarse) instead of x — the prolongation-and-correction that precedes a
// post-smoothing sweep, formed on the fly from the coarse rows (8 B per lane and fine row) with the formulas and masks
// of evomg_strip_prolong, so the result is bit-identical to running PROLONG_ADD and then the sweep, while the
// corrected field is neither written nor read back (2 x 8 B/node less traffic). With the Helmholtz Robin condition the
// boundary columns of the corrected rows are re-derived as the prolongation op's apply_bc would.
template <typename T> struct ProlongIn {
  const T* coarse;     // PRO: coarse correction to prolongate into the input
  Geo gc;              // geometry of the coarse level (PRO and TAIL_RESTRICT)
  double om;           // PRO: weight of the correction
  T* coarse_out;       // TAIL_RESTRICT: coarse right-hand side to write
  double inv_c0;       // 1 / centre coefficient when that is a power of two (exact), else 0
  double robin_kh_coarse;   // TAIL_RESTRICT on Helmholtz problems: k h of the coarse level
};
// TAIL_NORM (smoother modes): the sweep is the last op of a solver-loop iteration, so the squared norm of the residual
// of its OUTPUT is accumulated in the same pass: the rows the sweep has finalised are kept in a three-row register
// window and the residual of the middle one is formed two steps behind the relaxation front (one more halo row per
// chunk side). Partial sums follow the canonical order of STRIP_RESNORM (one per strip and 16-row chunk; a wave
// covering a taller chunk emits several), so the solver loop needs no separate residual sweep: 24 B/node less.
// grid: ceil(strips * chunks / 4) blocks of 256 threads (4 independent waves); rows_per_chunk = H
// TAIL_RESTRICT (smoother modes): the sweep is followed by RESIDUAL + RESTRICT of a residual nobody else reads. Same
// finalised-row window, one more halo row per side; the residual rows feed a three-row residual window from which
// every second step emits one row of the coarse right-hand side exactly as evomg_strip_residual_restrict does.
// XGEN (round 6; red-black, with TAIL_RESTRICT and XZERO): the sweep is the SECOND of two identical sweeps that start from the
// zero field (the pre-smoothing of a V(2,.) cycle on every level below the finest, and on the finest one inside a
// preconditioner). What the first sweep leaves in memory is a function of f alone — its red nodes see nothing but zeros, its
// black nodes the reds of three rows — so the rows of x are FORMED here from f rows r-1, r, r+1 (evaluating the first sweep's
// own expressions on literal zeros, Robin columns included) instead of being written by one launch and read back by the
// next: f read once, y and the coarse right-hand side written — 24 + 2 instead of 56 + 2 B/node for SMOOTH, SMOOTH,
// RESIDUAL, RESTRICT. The black node a generated row holds in the outermost lane has no red neighbour inside the wave:
// three halo lanes per side (116 stored columns), as in the cross-iteration kernel.
template <typename T, int MODE, int PF, bool NT_XY, bool PRO = false, bool TAIL_NORM = false, bool TAIL_RESTRICT = false,
          bool XZERO = false, bool XGEN = false>
__global__ void __launch_bounds__(256)
evomg_strip2d(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f, T* __restrict__ y,
              double om, double omm, int rows_per_chunk, int n_strips, int n_waves, double robin_kh,
              T* __restrict__ partials, int robin_out, ProlongIn<T> pro) {
  // XCD-aware block order: the dispatcher deals blocks round-robin over the 8 XCDs (block b -> XCD b % 8), each
  // with a private L2. Remap so that every XCD works on one contiguous band of rows: vertically adjacent chunks
  // then share their halo rows through that XCD's L2 instead of re-reading them from HBM. gridDim.x is a multiple
  // of 8 (host side), so the map is a bijection; placement only affects speed, never results.
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int wave = __builtin_amdgcn_readfirstlane((int)(logical_block * 4 + (threadIdx.x >> 6)));
  if (wave >= n_waves) return;
  prio_level(g);
  const int lane = threadIdx.x & 63;
  const int strip = wave % n_strips;
  const int chunk = wave / n_strips;
  const int n = g.n;
  const long pitch = g.pitch;
  // halo lanes per side: 1 (124 stored columns), 2 for the kernels that carry the residual norm (120 columns)
  static_assert(!XGEN || (MODE == STRIP_RBGS && TAIL_RESTRICT && XZERO && !PRO), "generated input: second of two red-black sweeps from zero");
  constexpr int HALO = XGEN ? 3 : (TAIL_NORM || TAIL_RESTRICT || MODE == STRIP_RESNORM) ? 2 : 1;
  constexpr int COLS = 128 - 4 * HALO;
  static_assert(HALO == 1 ? COLS == EVOMG_STRIP_COLS : HALO == 2 ? COLS == EVOMG_NORM_STRIP_COLS : COLS == EVOMG_XGEN_STRIP_COLS, "strip widths");
  const int c0 = strip * COLS - 2 * HALO + 2 * lane;        // first column of this lane's pair (even)
  const bool col_ok = c0 >= 0 && c0 < n;                    // c0+1 <= n < pitch: the pair load stays inside the row
  const int y0 = chunk * rows_per_chunk;
  const int y1 = min(y0 + rows_per_chunk, n);               // stored rows [y0, y1)
  const double inv_c0k = relax_div_const(st, pro.inv_c0);    // relax5's divisor-side constant
  const T zero = scalar_traits<T>::zero();
  const Pair<T> zpair{zero, zero};
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
  const bool store_lane = lane >= HALO && lane <= 63 - HALO && col_ok;

  // Branch-free loads: out-of-domain rows / columns are clamped to the nearest valid pair instead of being
  // predicated off. Whatever they deliver only ever feeds non-inner nodes, whose result is discarded in favour of
  // the old value — and straight-line code is what lets the compiler keep the PF-deep prefetch ring in flight
  // with counted s_waitcnt vmcnt(N) instead of draining to vmcnt(0) at every predicated block.
  const int c_clamped = min(max(c0, 0), n - 1);             // n-1 is even; (n-1, n) is still inside the row pitch
  // XZERO (SET_ZERO elided): the input field is known to be zero everywhere, nothing is loaded. A compile-time
  // switch on purpose: a run-time flag, even a uniform one, makes the compiler wrap the x loads in branches (and split
  // them), which costs every variant of this kernel its counted-vmcnt pipelining.
  auto load_row = [&](const T* __restrict__ base, int row) -> Pair<T> {
    if constexpr (XZERO) return zpair;
    const int r = min(max(row, 0), n - 1);                  // wave-uniform
    if constexpr (NT_XY) return load_pair_nt<T>(base + pitch * r + c_clamped);
    else return load_pair<T>(base + pitch * r + c_clamped);
  };
  // NT_XY (levels whose x, y and f together exceed the 256 MiB Infinity Cache while f alone fits): x is loaded and y
  // stored with the non-temporal hint, the right-hand side with plain loads — f is the one array every sweep of the level
  // reads again, x and y are streamed through once, so f stays resident across the coarse part of the cycle instead of
  // being evicted by them. Measured inside C2 solves (profiles/r03_nt_placement.md): fused sweeps 84-91 -> 76.5 us; the
  // round-1/2 placement (hint on f, kept x / y of back-to-back ping-pong sweeps resident) was the wrong one in a cycle
  auto load_row_f = [&](const T* __restrict__ base, int row) -> Pair<T> {
    const int r = min(max(row, 0), n - 1);
    return load_pair<T>(base + pitch * r + c_clamped);
  };

  // Helmholtz Robin boundary (K8) applied to what the sweep produced: x_b = x_in / (1 - i k h) in the first and the
  // last column of every inner row. The lane holding column 0 has its inner neighbour as second element; column n-1
  // is the first element of its lane and column n-2 the second element of the lane to the left (one DPP shift).
  // Must be called by the whole wave (DPP). robin_out = 0: the caller wants the boundary left alone (e.g. the
  // residual of the outer Krylov solver).
  const bool robin_a = scalar_traits<T>::is_complex && robin_out && robin_kh != 0.0 && (c0 == 0 || c0 == n - 1);
  auto robin_fix = [&](Pair<T>& out) {
    if constexpr (scalar_traits<T>::is_complex) {
      if (robin_out && robin_kh != 0.0) {                     // kernel arguments: uniform
        const cplx den{1.0, -robin_kh};
        const T lb = from_left(out.b);
        if (c0 == 0) out.a = sc_div(out.b, den);
        if (c0 == n - 1) out.a = sc_div(lb, den);
      }
    }
  };
  // ---- XGEN: the rows of x as the first sweep from zero leaves them, formed from f (rows in ascending order) ----
  // red of row q at this lane's red element of that row (a when q is even), zero where the first sweep relaxes nothing
  auto gen_red = [&](const Pair<T>& fq_, int q) -> T {
    const bool q_in = q >= 1 && q <= n - 2;
    if ((q & 1) == 0) return (q_in && inner_a) ? relax5(st, zero, zero, zero, zero, zero, fq_.a, om, omm, inv_c0k) : zero;
    return (q_in && inner_b) ? relax5(st, zero, zero, zero, zero, zero, fq_.b, om, omm, inv_c0k) : zero;
  };
  constexpr int GP = 2;                                      // f rows in flight ahead of the generator
  Pair<T> gen_f = zpair, gen_fq[GP];                         // f of the row formed next; f of the rows behind it
  T gen_rp = zero, gen_rc = zero;                            // reds of the row before the one formed next, and of that row
  int gen_row = 0;                                           // the row formed next
  auto gen_start = [&](int first) {
    gen_row = first;
    gen_rp = gen_red(load_row_f(f, first - 1), first - 1);
    gen_f = load_row_f(f, first);
    gen_rc = gen_red(gen_f, first);
#pragma unroll
    for (int k = 0; k < GP; ++k) gen_fq[k] = load_row_f(f, first + 1 + k);
  };
  // whole wave (DPP); k: which slot of the ring holds f of the next row (static at the call sites)
  auto gen_next = [&](int k) -> Pair<T> {
    const int r = gen_row;
    const Pair<T> fn = gen_fq[k];
    gen_fq[k] = load_row_f(f, r + 1 + GP);
    const T rn = gen_red(fn, r + 1);
    const bool r_in = r >= 1 && r <= n - 2;
    Pair<T> out;
    if ((r & 1) == 0) {                 // red = a, black = b
      T e = from_right(gen_rc);
      T w = gen_rc;
      if constexpr (scalar_traits<T>::is_complex) {
        if (robin_kh != 0.0) {          // (as below: the Robin condition between the colours, on the black node's old value — zero)
          const cplx den{1.0, -robin_kh};
          if (c0 == 0) w = sc_div(zero, den);
          if (c0 + 1 == n - 2) e = sc_div(zero, den);
        }
      }
      out.a = gen_rc;
      out.b = (r_in && inner_b) ? relax5(st, zero, w, e, gen_rp, rn, gen_f.b, om, omm, inv_c0k) : zero;
    } else {                            // red = b, black = a
      const T w = from_left(gen_rc);
      out.a = (r_in && inner_a) ? relax5(st, zero, w, gen_rc, gen_rp, rn, gen_f.a, om, omm, inv_c0k) : zero;
      out.b = gen_rc;
    }
    if (r_in) robin_fix(out);
    gen_rp = gen_rc;
    gen_rc = rn;
    gen_f = fn;
    gen_row = r + 1;
    return out;
  };
  constexpr int LEAD = (MODE == STRIP_RBGS) ? 1 : 0;        // RB-GS relaxes red one row ahead of black
  constexpr int EXT = TAIL_RESTRICT ? 2 : (TAIL_NORM ? 1 : 0);   // extra finalised rows on each side for the residual
  constexpr bool TAIL = TAIL_NORM || TAIL_RESTRICT;
  static_assert(!(TAIL_NORM && TAIL_RESTRICT), "one tail at a time");
  static_assert(!TAIL || MODE == STRIP_JACOBI || MODE == STRIP_RBGS, "tails are smoother options");
  static_assert(!TAIL || PF % 2 == 0, "tails rely on the static row parity");
  const int js = y0 - LEAD - EXT;                           // first row relaxed (red)
  const int je = y1 + LEAD + EXT;                           // one past the last row relaxed
  static_assert(!PRO || PF % 2 == 0, "PRO relies on the static row parity of an even prefetch depth");
  const int nc = PRO ? pro.gc.n : 1;
  const int Ic = min(max(c0 >> 1, 0), nc - 1);              // coarse column of the lane's first element
  // coarse row `row` at the lane's column and the next one (8-byte aligned 16-byte load; column nc lies inside the
  // row pitch). Every lane loads its own right neighbour — not a DPP shift as in evomg_strip_prolong — because here
  // the halo lane 63 must be right as well: its second element is the east neighbour of its first.
  auto load_coarse = [&](int row) -> Pair<T> {
    const int r = min(max(row, 0), nc - 1);
    return load_pair<T>(pro.coarse + (long)pro.gc.pitch * r + Ic);
  };
  const Pair<T> zcoarse{zero, zero};
  // fine row `row` of x + om P coarse; odd: the row lies between the coarse rows cJ and cJ1
  auto corrected = [&](const Pair<T>& raw, int row, bool odd, const Pair<T>& cJp, const Pair<T>& cJ1p) -> Pair<T> {
    const bool row_in = row >= 1 && row <= n - 2;
    const T cJ = cJp.a, cJr = cJp.b;
    T va, vb;
    if (!odd) {
      va = cJ;
      vb = sc_rmul(0.5, sc_add(cJ, cJr));
    } else {
      const T cJ1 = cJ1p.a, cJ1r = cJ1p.b;
      va = sc_rmul(0.5, sc_add(cJ, cJ1));
      vb = sc_rmul(0.25, sc_add(sc_add(cJ, cJr), sc_add(cJ1, cJ1r)));
    }
    Pair<T> out;
    out.a = (row_in && inner_a) ? sc_add(raw.a, sc_rmul(pro.om, va)) : raw.a;
    out.b = (row_in && inner_b) ? sc_add(raw.b, sc_rmul(pro.om, vb)) : raw.b;
    if constexpr (scalar_traits<T>::is_complex) {
      // Helmholtz: the prolongation op is followed by the Robin condition on the corrected field (whole-wave DPP)
      if (robin_kh != 0.0) {
        const cplx den{1.0, -robin_kh};
        const T lb = from_left(out.b);
        if (row_in && c0 == 0) out.a = sc_div(out.b, den);
        if (row_in && c0 == n - 1) out.a = sc_div(lb, den);
      }
    }
    return out;
  };
  Pair<T> xm, xc;
  if constexpr (XGEN) {
    gen_start(js - 1);
    xm = gen_next(0);
    xc = gen_next(1);
  } else {
    xm = load_row(x, js - 1);
    xc = load_row(x, js);
  }
  if constexpr (PRO) {
    // y0 is even: row js has the parity of LEAD + EXT, row js - 1 the other one
    constexpr bool odd_c = ((LEAD + EXT) & 1) != 0, odd_m = !odd_c;
    const int rm = max(js - 1, 0), rc = max(js, 0);
    xm = corrected(xm, js - 1, odd_m, load_coarse(rm >> 1), odd_m ? load_coarse((rm >> 1) + 1) : zcoarse);
    xc = corrected(xc, js, odd_c, load_coarse(rc >> 1), odd_c ? load_coarse((rc >> 1) + 1) : zcoarse);
  }
  Pair<T> xq[PF], fq[PF];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    if constexpr (XGEN) xq[d] = gen_next(d % GP);            // (two rows formed above: the ring slot of row js + 1 + d is d mod GP)
    else xq[d] = load_row(x, js + 1 + d);
    fq[d] = load_row_f(f, js + d);
  }
  // PRO: the coarse rows around the fine row that is corrected next (row j + 1 at step j) live in a two-row register
  // window: c_lo = coarse row floor(r / 2), c_hi the one above; an odd fine row uses both and then shifts the window,
  // whose new top row is loaded two steps before it is needed — one coarse row load per two fine rows
  Pair<T> c_lo = zcoarse, c_hi = zcoarse;
  if constexpr (PRO) {
    const int r0 = js + 1;                                   // may be negative: >> floors, load_coarse clamps
    c_lo = load_coarse(r0 >> 1);
    c_hi = load_coarse((r0 >> 1) + 1);
  }
  Pair<T> fm = zpair;          // f of row j-1 (black phase)
  T rm2 = zero, rm1 = zero;    // new red of rows j-2, j-1 (this lane's pair)
  T nrm = zero;                // STRIP_RESNORM: this lane's sum of squared residuals (rows ascending, a then b)
  Pair<T> fin1 = zpair, fin2 = zpair;   // TAIL_NORM: the two rows finalised before the current one
  T done0 = zero, done1 = zero, done2 = zero, done3 = zero;   // lane sums of the completed 16-row chunks, newest first
  int n_done = 0;                       // (host: a TAIL_NORM chunk has at most 4 x 16 rows)
  Pair<T> fprev = zpair;                // TAIL_NORM: f of the row whose residual is formed (two / one step behind)
  Pair<T> res1 = zpair, res2 = zpair;   // TAIL_RESTRICT: residual rows rr - 1, rr - 2
  const int ncr = TAIL_RESTRICT ? pro.gc.n : 1;
  const int Icr = c0 >> 1;                // coarse column of the lane's first element
  // residual of finalised row rr from the window (fin2, fin1, newest); rr_odd is static at the call sites
  auto tail = [&](const Pair<T>& newest, int rr, bool rr_odd, const Pair<T>& frow) {
    if constexpr (TAIL) {
      const T wa = from_left(fin1.b), eb = from_right(fin1.a);          // whole wave
      if constexpr (TAIL_NORM) {
        if (rr >= y0 && rr < y1) {                                       // wave-uniform
          if (rr >= 1 && rr <= n - 2 && store_lane) {
            const T ra = residual5(st, fin1.a, wa, fin1.b, fin2.a, newest.a, frow.a);
            const T rb = residual5(st, fin1.b, fin1.a, eb, fin2.b, newest.b, frow.b);
            if (inner_a) nrm = sc_add(nrm, sc_mul(ra, ra));
            if (inner_b) nrm = sc_add(nrm, sc_mul(rb, rb));
          }
          if ((rr & (EVOMG_NORM_ROWS - 1)) == EVOMG_NORM_ROWS - 1 || rr == n - 1) {
            // a canonical 16-row chunk is complete: park its lane sum (register shuffle only — reducing and storing
            // it here would drain the prefetch pipeline with a vmcnt(0) every 16 rows); flushed after the loop
            done3 = done2; done2 = done1; done1 = done0; done0 = nrm;
            ++n_done;
            nrm = zero;
          }
        }
      } else {
        // residual of row rr, zero outside the inner nodes (as evomg_strip_residual_restrict forms it)
        const bool rin = rr >= 1 && rr <= n - 2;
        Pair<T> rcur;
        rcur.a = (rin && inner_a) ? residual5(st, fin1.a, wa, fin1.b, fin2.a, newest.a, frow.a) : zero;
        rcur.b = (rin && inner_b) ? residual5(st, fin1.b, fin1.a, eb, fin2.b, newest.b, frow.b) : zero;
        if (rr_odd) {
          // rows rr-2, rr-1, rr = 2J-1, 2J, 2J+1: a complete coarse neighbourhood
          const int J = (rr - 1) >> 1;
          const T w = from_left(res1.b), sw = from_left(res2.b), nw = from_left(rcur.b);
          const T e4 = sc_add(sc_add(w, res1.b), sc_add(res2.a, rcur.a));
          const T q4 = sc_add(sc_add(sw, res2.b), sc_add(nw, rcur.b));
          T v = sc_rmul(0.25, res1.a);
          v = sc_add(v, sc_rmul(0.125, e4));
          v = sc_add(v, sc_rmul(0.0625, q4));
          T vs = v;
          bool coarse_robin = false;
          if constexpr (scalar_traits<T>::is_complex) {
            if (pro.robin_kh_coarse != 0.0) {                    // Robin columns of the coarse right-hand side
              const cplx den{1.0, -pro.robin_kh_coarse};
              const T vr = from_right(v), vl = from_left(v);
              if (Icr == 0) vs = sc_div(vr, den);
              if (Icr == ncr - 1) vs = sc_div(vl, den);
              coarse_robin = (Icr == 0 || Icr == ncr - 1) && c0 >= 0 && c0 < n;
            }
          }
          if (store_lane && ((Icr >= 1 && Icr <= ncr - 2) || coarse_robin) && J >= 1 && J <= ncr - 2 && 2 * J >= y0 && 2 * J < y1)
            pro.coarse_out[(long)pro.gc.pitch * J + Icr] = vs;
        }
        res2 = res1;
        res1 = rcur;
      }
      fin2 = fin1;
      fin1 = newest;
    }
  };

  // whole trips of the unrolled body (round 6): a wave whose rows do not fill the last trip marches on past `je` — loads are
  // clamped, nothing beyond its rows is stored or summed — instead of branching around every row step: the steps of a trip
  // are one stretch of code for the scheduler (C3 -1.2 %, the 513^2 solve -1.5 %, same bits)
  // (chunk heights are even and PF is 2: only a grid's LAST chunk can be short by a row, and the row it then marches on to
  // lies outside the grid — the residual modes, which store and sum whatever inner row they are given, never see a row of
  // another chunk)
  static_assert(PF == 2 || MODE == STRIP_JACOBI || MODE == STRIP_RBGS, "whole trips: the residual modes rely on PF == 2");
  const int je_loop = js + ((je - js + PF - 1) / PF) * PF;
  for (int base = js; base < je_loop; base += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      const int j = base + d;
      {
        Pair<T> xp = xq[d];
        const Pair<T> fc = fq[d];
        if constexpr (XGEN) xq[d] = gen_next((PF + d) % GP);   // rows js - 1 .. : the slot of the next row advances by one per row formed
        else xq[d] = load_row(x, j + 1 + PF);
        fq[d] = load_row_f(f, j + PF);
        const bool row_inner = j >= 1 && j <= n - 2;
        // chunk heights are even and js = y0 - LEAD, so with an even prefetch depth the row parity is static
        const int parity = (PF % 2 == 0) ? ((LEAD + EXT + d) & 1) : (j & 1);
        if constexpr (PRO) {
          const bool odd_next = ((parity + 1) & 1) != 0;         // row j + 1 (static)
          xp = corrected(xp, j + 1, odd_next, c_lo, c_hi);
          if (odd_next) {
            c_lo = c_hi;
            c_hi = load_coarse(((j + 1) >> 1) + 2);
          }
        }
        if constexpr (MODE == STRIP_JACOBI || MODE == STRIP_RESIDUAL || MODE == STRIP_RESNORM) {
          const T wa = from_left(xc.b), eb = from_right(xc.a);
          Pair<T> out;
          if constexpr (MODE == STRIP_JACOBI) {
            out.a = (row_inner && inner_a) ? relax5(st, xc.a, wa, xc.b, xm.a, xp.a, fc.a, om, omm, inv_c0k) : xc.a;
            out.b = (row_inner && inner_b) ? relax5(st, xc.b, xc.a, eb, xm.b, xp.b, fc.b, om, omm, inv_c0k) : xc.b;
            if (row_inner) robin_fix(out);
            if (store_lane && j >= y0 && j < y1) store_pair<NT_XY>(y + pitch * j + c0, out);
            if constexpr (TAIL) {
              tail(out, j - 1, ((parity + 1) & 1) != 0, fprev);  // row j is final now: residual of row j - 1
              fprev = fc;
            }
          } else {
            // residual: only inner nodes are written (plus the Robin columns of a Helmholtz residual field)
            if (row_inner) {                                     // wave-uniform
              out.a = residual5(st, xc.a, wa, xc.b, xm.a, xp.a, fc.a);
              out.b = residual5(st, xc.b, xc.a, eb, xm.b, xp.b, fc.b);
              robin_fix(out);
              if (store_lane) {
                T* dst = y + pitch * j + c0;
                const bool sa = inner_a || robin_a;
                if (sa && inner_b) store_pair(dst, out);
                else if (sa) dst[0] = out.a;
                else if (inner_b) dst[1] = out.b;
                if constexpr (MODE == STRIP_RESNORM) {
                  // unconjugated dot(r, r), unfused multiply + add like evomg_dot_partial
                  if (inner_a) nrm = sc_add(nrm, sc_mul(out.a, out.a));
                  if (inner_b) nrm = sc_add(nrm, sc_mul(out.b, out.b));
                }
              }
            }
          }
        } else {
          // ---- red nodes of row j (colour 0: even index sum; c0 is even => element a when j is even) ----
          T rc;
          if (parity == 0) {
            const T w = from_left(xc.b);
            rc = (row_inner && inner_a) ? relax5(st, xc.a, w, xc.b, xm.a, xp.a, fc.a, om, omm, inv_c0k) : xc.a;
          } else {
            const T e = from_right(xc.a);
            rc = (row_inner && inner_b) ? relax5(st, xc.b, xc.a, e, xm.b, xp.b, fc.b, om, omm, inv_c0k) : xc.b;
          }
          // ---- black nodes of row j-1 from the new reds of rows j-2 (rm2), j-1 (rm1), j (rc) ----
          const int jb = j - 1;
          const bool brow_inner = jb >= 1 && jb <= n - 2;
          Pair<T> out;
          if (parity == 1) {            // jb = j-1 even: red = a, black = b
            T e = from_right(rm1);
            T w = rm1;
            if constexpr (scalar_traits<T>::is_complex) {
              // Helmholtz Robin boundary (K8): the two-pass form re-applies x_b = x_in / (1 - i k h) between the
              // colours, so a black node next to the left/right boundary sees the boundary value derived from
              // its own (still old) value. Identical to the stored value whenever the field is BC-consistent.
              if (robin_kh != 0.0) {
                const cplx den{1.0, -robin_kh};
                if (c0 == 0) w = sc_div(xm.b, den);
                if (c0 + 1 == n - 2) e = sc_div(xm.b, den);
              }
            }
            const T bl = (brow_inner && inner_b) ? relax5(st, xm.b, w, e, rm2, rc, fm.b, om, omm, inv_c0k) : xm.b;
            out.a = rm1;
            out.b = bl;
          } else {                      // red = b, black = a
            const T w = from_left(rm1);
            const T bl = (brow_inner && inner_a) ? relax5(st, xm.a, w, rm1, rm2, rc, fm.a, om, omm, inv_c0k) : xm.a;
            out.a = bl;
            out.b = rm1;
          }
          if (brow_inner) robin_fix(out);
          if (store_lane && jb >= y0 && jb < y1) store_pair<NT_XY>(y + pitch * jb + c0, out);
          if constexpr (TAIL) {
            tail(out, jb - 1, (parity & 1) != 0, fprev);         // row j - 1 is final now: residual of row j - 2
            fprev = fm;
          }
          rm2 = rm1;
          rm1 = rc;
          fm = fc;
        }
        xm = xc;
        xc = xp;
      }
    }
  }
  if constexpr (TAIL_NORM) {
    // completed 16-row chunks: y0 / 16 + (n_done - 1 - k) for done_k; wave64 tree as in STRIP_RESNORM
    const int c16 = y0 / EVOMG_NORM_ROWS;
    const T d[4] = {done0, done1, done2, done3};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < n_done) {                                      // wave-uniform
        const T tot = wave_sum(d[k]);
        if (lane == 0) partials[(c16 + n_done - 1 - k) * n_strips + strip] = tot;
      }
    }
  }
  if constexpr (MODE == STRIP_RESNORM) {
    // one partial sum per wave = (strip, chunk): wave64 tree (offsets 32 .. 1), restated by the oracle's norm
    const T tot = wave_sum(nrm);
    if (lane == 0) partials[wave] = tot;
  }
}

// ---- fused residual + full-weighting restriction (K4 + K5) ------------------------------------------------------
// The residual of fine row j is computed exactly as in STRIP_RESIDUAL and kept in a three-row register window; every
// second row the wave emits one coarse row: lane l's pair starts at the even fine column c0, i.e. at coarse column
// c0/2, whose 3x3 fine neighbourhood is the lane's own pair plus the left lane's second element (one DPP shift per
// row). Consecutive lanes write consecutive coarse nodes. The residual field itself is written only if WRITE_R
// (it is skipped when the op program never reads it again before overwriting it): 16 + 8/4 instead of 24 + 10 B/node.
// Same operation order as evomg_apply<.., true> followed by evomg_restrict => bit-identical coarse right-hand side.
template <typename T, bool WRITE_R, bool NT_XY>
__global__ void __launch_bounds__(256)
evomg_strip_residual_restrict(Geo g, Geo gc, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f,
                              T* __restrict__ res, T* __restrict__ coarse, int rows_per_chunk, int n_strips, int n_waves,
                              double robin_kh, double robin_kh_coarse) {
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int wave = __builtin_amdgcn_readfirstlane((int)(logical_block * 4 + (threadIdx.x >> 6)));
  if (wave >= n_waves) return;
  prio_level(g);
  const int lane = threadIdx.x & 63;
  const int strip = wave % n_strips;
  const int chunk = wave / n_strips;
  const int n = g.n, nc = gc.n;
  const long pitch = g.pitch;
  const int c0 = strip * EVOMG_STRIP_COLS - 2 + 2 * lane;
  const bool col_ok = c0 >= 0 && c0 < n;
  const int y0 = chunk * rows_per_chunk;                     // even
  const int y1 = min(y0 + rows_per_chunk, n);
  const T zero = scalar_traits<T>::zero();
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
  const bool store_lane = lane >= 1 && lane <= 62 && col_ok;
  const int I = c0 >> 1;                                     // coarse column of element a
  const bool coarse_col_inner = I >= 1 && I <= nc - 2;
  const int c_clamped = min(max(c0, 0), n - 1);
  // NT_XY: as in evomg_strip2d — x streamed with the non-temporal hint, the right-hand side kept cacheable
  auto load_row = [&](const T* __restrict__ base, int row) -> Pair<T> {
    const int r = min(max(row, 0), n - 1);
    if constexpr (NT_XY) return load_pair_nt<T>(base + pitch * r + c_clamped);
    else return load_pair<T>(base + pitch * r + c_clamped);
  };
  auto load_row_f = [&](const T* __restrict__ base, int row) -> Pair<T> {
    const int r = min(max(row, 0), n - 1);
    return load_pair<T>(base + pitch * r + c_clamped);
  };
  constexpr int PF = 2;
  const int js = y0 - 1;            // first residual row needed (south neighbour row of the chunk's first coarse row)
  const int je = y1;                // one past the last
  Pair<T> xm = load_row(x, js - 1);
  Pair<T> xc = load_row(x, js);
  Pair<T> xq[PF], fq[PF];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    xq[d] = load_row(x, js + 1 + d);
    fq[d] = load_row_f(f, js + d);
  }
  Pair<T> r2{zero, zero}, r1{zero, zero};    // residual rows j-2, j-1
  for (int base = js; base < je; base += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      const int j = base + d;
      if (j < je) {
        const Pair<T> xp = xq[d];
        const Pair<T> fc = fq[d];
        xq[d] = load_row(x, j + 1 + PF);
        fq[d] = load_row_f(f, j + PF);
        const bool row_inner = j >= 1 && j <= n - 2;
        const T wa = from_left(xc.b), eb = from_right(xc.a);
        Pair<T> rc;
        rc.a = (row_inner && inner_a) ? residual5(st, xc.a, wa, xc.b, xm.a, xp.a, fc.a) : zero;
        rc.b = (row_inner && inner_b) ? residual5(st, xc.b, xc.a, eb, xm.b, xp.b, fc.b) : zero;
        if constexpr (WRITE_R) {
          // Helmholtz: the residual field carries the Robin condition too (stored copy only; the restriction below
          // never reads the boundary columns)
          Pair<T> rs = rc;
          bool robin_a = false;
          if constexpr (scalar_traits<T>::is_complex) {
            if (robin_kh != 0.0 && row_inner) {               // uniform
              const cplx den{1.0, -robin_kh};
              const T lb = from_left(rc.b);
              if (c0 == 0) rs.a = sc_div(rc.b, den);
              if (c0 == n - 1) rs.a = sc_div(lb, den);
              robin_a = c0 == 0 || c0 == n - 1;
            }
          }
          if (row_inner && store_lane && j >= y0) {
            T* dst = res + pitch * j + c0;
            const bool sa = inner_a || robin_a;
            if (sa && inner_b) store_pair(dst, rs);
            else if (sa) dst[0] = rs.a;
            else if (inner_b) dst[1] = rs.b;
          }
        }
        // js = y0 - 1 is odd, PF is even: d == 0 <=> j odd <=> rows j-2, j-1, j are a complete coarse neighbourhood
        if (d == 0) {
          const int J = (j - 1) >> 1;                        // coarse row; fine centre row j-1 = 2J
          const T w = from_left(r1.b), sw = from_left(r2.b), nw = from_left(rc.b);
          const T e4 = sc_add(sc_add(w, r1.b), sc_add(r2.a, rc.a));
          const T q4 = sc_add(sc_add(sw, r2.b), sc_add(nw, rc.b));
          T v = sc_rmul(0.25, r1.a);
          v = sc_add(v, sc_rmul(0.125, e4));
          v = sc_add(v, sc_rmul(0.0625, q4));
          T vs = v;
          bool coarse_robin = false;
          if constexpr (scalar_traits<T>::is_complex) {
            if (robin_kh_coarse != 0.0) {                      // Robin columns of the coarse right-hand side
              const cplx den{1.0, -robin_kh_coarse};
              const T vr = from_right(v), vl = from_left(v);
              if (I == 0) vs = sc_div(vr, den);
              if (I == nc - 1) vs = sc_div(vl, den);
              coarse_robin = (I == 0 || I == nc - 1) && c0 >= 0 && c0 < n;
            }
          }
          if (store_lane && (coarse_col_inner || coarse_robin) && J >= 1 && J <= nc - 2 && 2 * J >= y0) coarse[(long)gc.pitch * J + I] = vs;
        }
        r2 = r1;
        r1 = rc;
        xm = xc;
        xc = xp;
      }
    }
  }
}

// ---- bilinear prolongation (+ scaled correction), fine-row marching (K6 / K6') -------------------------------------
// Lane pair (c0, c0+1) with c0 even: element a sits on coarse column I = c0/2, element b between I and I+1 (the right
// lane's coarse value, one DPP shift). Each coarse row is loaded once per two fine rows. In place: a lane reads and
// writes only its own pair. Operation order of interpolate<T,2> + evomg_prolong => bit-identical.
template <typename T, bool ACC>
__global__ void __launch_bounds__(256)
evomg_strip_prolong(Geo g, Geo gc, const T* __restrict__ coarse, T* __restrict__ fine, double om, int rows_per_chunk,
                    int n_strips, int n_waves, double robin_kh) {
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int wave = __builtin_amdgcn_readfirstlane((int)(logical_block * 4 + (threadIdx.x >> 6)));
  if (wave >= n_waves) return;
  prio_level(g);
  const int lane = threadIdx.x & 63;
  const int strip = wave % n_strips;
  const int chunk = wave / n_strips;
  const int n = g.n, nc = gc.n;
  const long pitch = g.pitch;
  const int c0 = strip * EVOMG_STRIP_COLS - 2 + 2 * lane;
  const bool col_ok = c0 >= 0 && c0 < n;
  const int y0 = chunk * rows_per_chunk;                     // even
  const int y1 = min(y0 + rows_per_chunk, n);
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
  const bool store_lane = lane >= 1 && lane <= 62 && col_ok;
  const int c_clamped = min(max(c0, 0), n - 1);
  const int I = min(max(c0 >> 1, 0), nc - 1);
  auto load_fine = [&](int row) -> Pair<T> {
    const int r = min(max(row, 0), n - 1);
    return load_pair<T>(fine + pitch * r + c_clamped);
  };
  auto load_coarse = [&](int J) -> T {
    const int r = min(max(J, 0), nc - 1);
    return coarse[(long)gc.pitch * r + I];
  };
  // Helmholtz Robin columns of the updated fine field (see evomg_strip2d): whole-wave call
  auto robin_fix = [&](Pair<T>& out) {
    if constexpr (scalar_traits<T>::is_complex) {
      if (robin_kh != 0.0) {
        const cplx den{1.0, -robin_kh};
        const T lb = from_left(out.b);
        if (c0 == 0) out.a = sc_div(out.b, den);
        if (c0 == n - 1) out.a = sc_div(lb, den);
      }
    }
  };
  const int J0 = y0 >> 1;
  T cJ = load_coarse(J0), cJ1 = load_coarse(J0 + 1);
  Pair<T> x0 = load_fine(y0), x1 = load_fine(y0 + 1);
  for (int j = y0; j < y1; j += 2) {
    const int J = j >> 1;
    // prefetch the next coarse row and the next two fine rows
    const T cJ2 = load_coarse(J + 2);
    const Pair<T> x2 = load_fine(j + 2), x3 = load_fine(j + 3);
    const T cJr = from_right(cJ), cJ1r = from_right(cJ1);
    // even fine row j = 2J
    {
      const bool row_inner = j >= 1 && j <= n - 2;
      const T va = cJ;
      const T vb = sc_rmul(0.5, sc_add(cJ, cJr));
      Pair<T> out;
      out.a = (row_inner && inner_a) ? (ACC ? sc_add(x0.a, sc_rmul(om, va)) : va) : x0.a;
      out.b = (row_inner && inner_b) ? (ACC ? sc_add(x0.b, sc_rmul(om, vb)) : vb) : x0.b;
      if (row_inner) robin_fix(out);
      if (store_lane) store_pair(fine + pitch * j + c0, out);
    }
    // odd fine row j + 1 = 2J + 1
    if (j + 1 < y1) {
      const bool row_inner = j + 1 >= 1 && j + 1 <= n - 2;
      const T va = sc_rmul(0.5, sc_add(cJ, cJ1));
      const T vb = sc_rmul(0.25, sc_add(sc_add(cJ, cJr), sc_add(cJ1, cJ1r)));
      Pair<T> out;
      out.a = (row_inner && inner_a) ? (ACC ? sc_add(x1.a, sc_rmul(om, va)) : va) : x1.a;
      out.b = (row_inner && inner_b) ? (ACC ? sc_add(x1.b, sc_rmul(om, vb)) : vb) : x1.b;
      if (row_inner) robin_fix(out);
      if (store_lane) store_pair(fine + pitch * (j + 1) + c0, out);
    }
    cJ = cJ1;
    cJ1 = cJ2;
    x0 = x2;
    x1 = x3;
  }
}

// ---- 3D marching tiles: Jacobi (K1) and residual (K4) for 7-point stencils -----------------------------------------
// The 2D strip idea with one more dimension: a wave owns 128 columns (lane = column pair, 16-byte accesses, x
// neighbours by DPP) x TY rows and marches along z. It keeps a ring of four planes of its tile in registers — k-1, k,
// k+1 and the prefetched k+2 — each with one halo row above and below (rows y0-1 .. y0+TY), so every plane of x is
// loaded once per tile, (TY+2)/TY times overall; the halo rows are the neighbouring tiles' own rows, fetched at
// about the same time by the adjacent waves of the same workgroup (tiles of one workgroup are y-neighbours) and
// served by L1/L2. The ring is addressed with compile-time indices (the z loop is unrolled four times), so
// advancing a plane costs no register moves. No LDS, no barriers.
// Operation order = relax_point / apply_stencil (evomg_kernels.h) = the oracle's FN(relax3) / FN(apply_st3).
template <typename T>
__device__ __forceinline__ T relax7(const Stencil<T>& st, const ConstDiv<T>& cd, T c, T w, T e, T s, T n, T d, T u, T f, double om, double omm) {
  T r = f;
  r = sc_sub(r, sc_mul(st.c[1], w));
  r = sc_sub(r, sc_mul(st.c[2], e));
  r = sc_sub(r, sc_mul(st.c[3], s));
  r = sc_sub(r, sc_mul(st.c[4], n));
  r = sc_sub(r, sc_mul(st.c[5], d));
  r = sc_sub(r, sc_mul(st.c[6], u));
  const T xs = const_div(r, cd);         // = r / st.c[0], bit for bit (evomg_scalar.h)
  return sc_add(sc_rmul(omm, c), sc_rmul(om, xs));
}
template <typename T>
__device__ __forceinline__ T residual7(const Stencil<T>& st, T c, T w, T e, T s, T n, T d, T u, T f) {
  T a = sc_mul(st.c[0], c);
  a = sc_add(a, sc_mul(st.c[1], w));
  a = sc_add(a, sc_mul(st.c[2], e));
  a = sc_add(a, sc_mul(st.c[3], s));
  a = sc_add(a, sc_mul(st.c[4], n));
  a = sc_add(a, sc_mul(st.c[5], d));
  a = sc_add(a, sc_mul(st.c[6], u));
  return sc_sub(f, a);
}

// POW2 forms (round 4): where every OFF-CENTRE coefficient is a (signed) power of two — -1/h^2 on a dyadic mesh: the 3D Poisson
// stencil on every level (its centre 6/h^2 is not) — the product c * v is exact, so `r - c * v` and fma(-c, v, r) round the same
// real number once: the same bits, one instruction instead of two (the 3D double sweep is VALU co-bound). Host-checked
// (offcentre_pow2); the centre term and the quotient keep their own roundings. Out-of-range intermediates (beyond 1e300) are the
// only place the two forms could part.
template <bool POW2, typename T> __device__ __forceinline__ T msub(T r, T c, T v) {
  if constexpr (POW2 && !scalar_traits<T>::is_complex) return __builtin_fma(-c, v, r);
  else return sc_sub(r, sc_mul(c, v));
}
template <bool POW2, typename T> __device__ __forceinline__ T madd(T a, T c, T v) {
  if constexpr (POW2 && !scalar_traits<T>::is_complex) return __builtin_fma(c, v, a);
  else return sc_add(a, sc_mul(c, v));
}
template <bool POW2, typename T>
__device__ __forceinline__ T relax7p(const Stencil<T>& st, const ConstDiv<T>& cd, T c, T w, T e, T s, T n, T d, T u, T f, double om, double omm) {
  T r = f;
  r = msub<POW2>(r, st.c[1], w);
  r = msub<POW2>(r, st.c[2], e);
  r = msub<POW2>(r, st.c[3], s);
  r = msub<POW2>(r, st.c[4], n);
  r = msub<POW2>(r, st.c[5], d);
  r = msub<POW2>(r, st.c[6], u);
  const T xs = const_div(r, cd);
  return sc_add(sc_rmul(omm, c), sc_rmul(om, xs));
}
template <bool POW2, typename T>
__device__ __forceinline__ T residual7p(const Stencil<T>& st, T c, T w, T e, T s, T n, T d, T u, T f) {
  T a = sc_mul(st.c[0], c);
  a = madd<POW2>(a, st.c[1], w);
  a = madd<POW2>(a, st.c[2], e);
  a = madd<POW2>(a, st.c[3], s);
  a = madd<POW2>(a, st.c[4], n);
  a = madd<POW2>(a, st.c[5], d);
  a = madd<POW2>(a, st.c[6], u);
  return sc_sub(f, a);
}

// grid: n_blocks (rounded up to a multiple of 8) blocks of 256 threads; block -> (group of 4 y tiles fastest, strip,
// z chunk), wave w of the block = y tile 4*group + w. (Lock-stepping the four waves with a barrier per plane so that
// shared halo rows meet in L1/L2 was measured and does not help: profiles/r01_pmc_3d.md.)
template <typename T, int MODE, int TY, bool NT_F>
__global__ void __launch_bounds__(256)
evomg_strip3d(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f, T* __restrict__ y, double om,
              double omm, int planes_per_chunk, int n_strips, int n_ygroups, int n_blocks) {
  static_assert(MODE == STRIP_JACOBI || MODE == STRIP_RESIDUAL, "3D tiles: Jacobi and residual");
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (logical_block >= n_blocks) return;
  prio_level(g);
  const int lane = threadIdx.x & 63;
  const ConstDiv<T> cd0 = const_div_make(st.c[0]);     // quotients by the stencil centre (evomg_scalar.h)
  const int ytile = (logical_block % n_ygroups) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int strip = (logical_block / n_ygroups) % n_strips;
  const int chunk = logical_block / (n_ygroups * n_strips);
  const int n = g.n;
  const long pitch = g.pitch, plane = g.plane;
  const int c0 = strip * EVOMG_STRIP_COLS - 2 + 2 * lane;
  const bool col_ok = c0 >= 0 && c0 < n;
  const int y0 = ytile * TY;
  const int z0 = chunk * planes_per_chunk;
  const int z1 = min(z0 + planes_per_chunk, n);
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
  const bool store_lane = lane >= 1 && lane <= 62 && col_ok;
  const int c_clamped = min(max(c0, 0), n - 1);
  // branch-free clamped loads (see evomg_strip2d): whatever an out-of-domain slot delivers only feeds non-inner nodes
  auto load_x = [&](int row, int pl) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl, 0), n - 1);      // wave-uniform
    return load_pair<T>(x + plane * p + pitch * r + c_clamped);
  };
  auto load_f = [&](int row, int pl) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl, 0), n - 1);
    if constexpr (NT_F) return load_pair_nt<T>(f + plane * p + pitch * r + c_clamped);
    else return load_pair<T>(f + plane * p + pitch * r + c_clamped);
  };
  Pair<T> xr[4][TY + 2];      // ring slot (k + 1) & 3 holds plane k; rows y0-1 .. y0+TY
  Pair<T> fr[2][TY];          // ring slot k & 1 holds f of plane k; rows y0 .. y0+TY-1
  // z0 is a multiple of 4 (host side), so ring slots are compile-time constants inside the unrolled loop
#pragma unroll
  for (int t = 0; t < TY + 2; ++t) {
    xr[0][t] = load_x(y0 - 1 + t, z0 - 1);
    xr[1][t] = load_x(y0 - 1 + t, z0);
    xr[2][t] = load_x(y0 - 1 + t, z0 + 1);
  }
#pragma unroll
  for (int t = 0; t < TY; ++t) fr[0][t] = load_f(y0 + t, z0);

  for (int kb = z0; kb < z1; kb += 4) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int k = kb + d;
      if (k < z1) {                                            // uniform over the block (same chunk)
        // prefetch plane k+2 of x and plane k+1 of f
#pragma unroll
        for (int t = 0; t < TY + 2; ++t) xr[(d + 3) & 3][t] = load_x(y0 - 1 + t, k + 2);
#pragma unroll
        for (int t = 0; t < TY; ++t) fr[(d + 1) & 1][t] = load_f(y0 + t, k + 1);
        const Pair<T>* below = xr[d & 3];
        const Pair<T>* here = xr[(d + 1) & 3];
        const Pair<T>* above = xr[(d + 2) & 3];
        const Pair<T>* fk = fr[d & 1];
        const bool plane_inner = k >= 1 && k <= n - 2;
#pragma unroll
        for (int t = 0; t < TY; ++t) {
          const int row = y0 + t;
          if (row < n) {                                       // wave-uniform
            const Pair<T> c = here[t + 1], s = here[t], nn = here[t + 2], dn = below[t + 1], up = above[t + 1];
            const T wa = from_left(c.b), eb = from_right(c.a);
            const bool row_inner = plane_inner && row >= 1 && row <= n - 2;
            T* dst = y + plane * k + pitch * row + c0;
            Pair<T> out;
            if constexpr (MODE == STRIP_JACOBI) {
              out.a = (row_inner && inner_a) ? relax7(st, cd0, c.a, wa, c.b, s.a, nn.a, dn.a, up.a, fk[t].a, om, omm) : c.a;
              out.b = (row_inner && inner_b) ? relax7(st, cd0, c.b, c.a, eb, s.b, nn.b, dn.b, up.b, fk[t].b, om, omm) : c.b;
              if (store_lane) store_pair(dst, out);
            } else {
              if (row_inner && store_lane) {
                out.a = residual7(st, c.a, wa, c.b, s.a, nn.a, dn.a, up.a, fk[t].a);
                out.b = residual7(st, c.b, c.a, eb, s.b, nn.b, dn.b, up.b, fk[t].b);
                if (inner_a && inner_b) store_pair(dst, out);
                else if (inner_a) dst[0] = out.a;
                else if (inner_b) dst[1] = out.b;
              }
            }
          }
        }
      }
    }
  }
}

// ---- 3D fused two-colour red-black Gauss-Seidel (K2), out of place ---------------------------------------------------
// Same tiles as evomg_strip3d. Marching along z, step k relaxes the RED nodes of plane k from old values and then the
// BLACK nodes of plane k-1 from the new reds of planes k-2, k-1, k (the 2D kernel's lag, one dimension up). In the y
// direction there is no lag to exploit, so the wave also relaxes the reds of one halo row below and above its tile
// (rows y0-1 and y0+TY, needed by its first and last black row) and therefore loads x with a halo of two rows:
// (TY+4)/TY for x, (TY+2)/TY for f — 16.9 + 12.4 + 8 = 37 B/node at TY = 4 against 48 B/node for two in-place colour
// passes, and one launch instead of two. Colour 0 = even index sum; c0 is even, so element a of a pair is red iff
// row + plane is even: static inside the twice-unrolled z loop (y0 and z0 are even).
// Operation order = relax_point, so the result is bit-identical to the two-pass form (evomg_rbgs_colour x 2).
template <typename T, int TY, bool NT_F>
__global__ void __launch_bounds__(256)
evomg_strip3d_rbgs(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f, T* __restrict__ y, double om,
                   double omm, int planes_per_chunk, int n_strips, int n_ygroups, int n_blocks) {
  static_assert(TY % 2 == 0, "even tile heights keep the colour of a row static");
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (logical_block >= n_blocks) return;
  prio_level(g);
  const int lane = threadIdx.x & 63;
  const ConstDiv<T> cd0 = const_div_make(st.c[0]);     // quotients by the stencil centre (evomg_scalar.h)
  const int ytile = (logical_block % n_ygroups) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int strip = (logical_block / n_ygroups) % n_strips;
  const int chunk = logical_block / (n_ygroups * n_strips);
  const int n = g.n;
  const long pitch = g.pitch, plane = g.plane;
  const int c0 = strip * EVOMG_STRIP_COLS - 2 + 2 * lane;
  const bool col_ok = c0 >= 0 && c0 < n;
  const int y0 = ytile * TY;
  const int z0 = chunk * planes_per_chunk;                     // even (host side)
  const int z1 = min(z0 + planes_per_chunk, n);
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
  const bool store_lane = lane >= 1 && lane <= 62 && col_ok;
  const int c_clamped = min(max(c0, 0), n - 1);
  const T zero = scalar_traits<T>::zero();
  auto load_x = [&](int row, int pl) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl, 0), n - 1);
    return load_pair<T>(x + plane * p + pitch * r + c_clamped);
  };
  auto load_f = [&](int row, int pl) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl, 0), n - 1);
    if constexpr (NT_F) return load_pair_nt<T>(f + plane * p + pitch * r + c_clamped);
    else return load_pair<T>(f + plane * p + pitch * r + c_clamped);
  };
  const int ks = z0 - 1;                 // first plane whose reds are relaxed (odd)
  const int ke = z1 + 1;                 // one past the last
  Pair<T> xb[TY + 2];                    // old x, plane k-1, rows y0-1 .. y0+TY
  Pair<T> xc[TY + 4], xa[TY + 4], xq[TY + 4];   // old x, planes k, k+1, k+2 (prefetch), rows y0-2 .. y0+TY+1
  Pair<T> fb[TY];                        // f, plane k-1, rows y0 .. y0+TY-1
  Pair<T> fc[TY + 2], fq[TY + 2];        // f, planes k, k+1 (prefetch), rows y0-1 .. y0+TY
  T r2[TY], r1[TY + 2], rc[TY + 2];      // new reds: plane k-2 (rows y0 ..), planes k-1 and k (rows y0-1 .. y0+TY)
#pragma unroll
  for (int t = 0; t < TY + 2; ++t) xb[t] = load_x(y0 - 1 + t, ks - 1);
#pragma unroll
  for (int u = 0; u < TY + 4; ++u) {
    xc[u] = load_x(y0 - 2 + u, ks);
    xa[u] = load_x(y0 - 2 + u, ks + 1);
  }
#pragma unroll
  for (int t = 0; t < TY + 2; ++t) fc[t] = load_f(y0 - 1 + t, ks);
#pragma unroll
  for (int t = 0; t < TY; ++t) { fb[t] = Pair<T>{zero, zero}; r2[t] = zero; }
#pragma unroll
  for (int t = 0; t < TY + 2; ++t) r1[t] = zero;

  for (int kb = ks; kb < ke; kb += 2) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int k = kb + d;                                    // d == 0: k odd, d == 1: k even
      if (k < ke) {
#pragma unroll
        for (int u = 0; u < TY + 4; ++u) xq[u] = load_x(y0 - 2 + u, k + 2);
#pragma unroll
        for (int t = 0; t < TY + 2; ++t) fq[t] = load_f(y0 - 1 + t, k + 1);
        const bool plane_inner = k >= 1 && k <= n - 2;
        // ---- reds of plane k, rows y0-1 .. y0+TY ----
#pragma unroll
        for (int t = 0; t < TY + 2; ++t) {
          const int row = y0 - 1 + t;
          const bool row_inner = plane_inner && row >= 1 && row <= n - 2;
          const Pair<T> c = xc[t + 1], s = xc[t], nn = xc[t + 2], dn = xb[t], up = xa[t + 1];
          // row parity: y0 even => row = y0 - 1 + t has the parity of t + 1; k has the parity of d + 1
          if (((t + 1 + d + 1) & 1) == 0) {                    // a is red
            const T w = from_left(c.b);
            rc[t] = (row_inner && inner_a) ? relax7(st, cd0, c.a, w, c.b, s.a, nn.a, dn.a, up.a, fc[t].a, om, omm) : c.a;
          } else {                                             // b is red
            const T e = from_right(c.a);
            rc[t] = (row_inner && inner_b) ? relax7(st, cd0, c.b, c.a, e, s.b, nn.b, dn.b, up.b, fc[t].b, om, omm) : c.b;
          }
        }
        // ---- blacks of plane k-1, rows y0 .. y0+TY-1, from the new reds ----
        const int kbk = k - 1;
        const bool bplane_inner = kbk >= 1 && kbk <= n - 2;
        const bool bplane_stored = kbk >= z0 && kbk < z1;
#pragma unroll
        for (int t = 0; t < TY; ++t) {
          const int row = y0 + t;
          const bool row_inner = bplane_inner && row >= 1 && row <= n - 2;
          const Pair<T> old = xb[t + 1];
          const T red = r1[t + 1];
          Pair<T> out;
          // row = y0 + t has the parity of t; plane k-1 has the parity of d
          if (((t + d) & 1) == 0) {                            // a red, b black
            const T e = from_right(red);
            const T bl = (row_inner && inner_b) ? relax7(st, cd0, old.b, red, e, r1[t], r1[t + 2], r2[t], rc[t + 1], fb[t].b, om, omm) : old.b;
            out.a = red;
            out.b = bl;
          } else {                                             // b red, a black
            const T w = from_left(red);
            const T bl = (row_inner && inner_a) ? relax7(st, cd0, old.a, w, red, r1[t], r1[t + 2], r2[t], rc[t + 1], fb[t].a, om, omm) : old.a;
            out.a = bl;
            out.b = red;
          }
          if (store_lane && bplane_stored && row < n) store_pair(y + plane * kbk + pitch * row + c0, out);
        }
        // ---- advance one plane ----
#pragma unroll
        for (int t = 0; t < TY; ++t) { r2[t] = r1[t + 1]; fb[t] = fc[t + 1]; }
#pragma unroll
        for (int t = 0; t < TY + 2; ++t) { r1[t] = rc[t]; xb[t] = xc[t + 1]; fc[t] = fq[t]; }
#pragma unroll
        for (int u = 0; u < TY + 4; ++u) { xc[u] = xa[u]; xa[u] = xq[u]; }
      }
    }
  }
}
