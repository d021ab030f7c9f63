// evomg_strip_cross.h — the finest 2D level's work ACROSS the solver loop's iteration boundary in one marching-strip pass.
//
// A V-type cycle ends with [PROLONG_ADD +] SMOOTH on the finest level (followed by the loop's residual norm) and the next
// iteration starts with SMOOTH + RESIDUAL + RESTRICT on the same arrays: two passes of 26 B/node each over the one level that
// carries three quarters of a cycle's traffic (evomg_strip2d<.., PRO, TAIL_NORM> and <.., TAIL_RESTRICT>). This kernel runs
// both red-black sweeps, the norm and the restriction in ONE pass:
//
//   x (+ om_p * P coarse)  --sweep A-->  z  --sweep B-->  w  --residual, full weighting-->  coarse right-hand side
//                                        |
//                                        +--> squared residual norm of z (the iterate the loop tests)
//
// reading x, f and the coarse correction once and writing w and the coarse right-hand side: 28 B/node instead of 52. z — the
// iterate the loop would return if the norm says "converged" — is NOT written: the launch is speculative exactly like the
// solver loop's early launch of the next cycle's first sweep (w and the coarse right-hand side go to spare arrays that only a
// host-side pointer swap makes current); when the loop stops, the ordinary post-smoothing kernel forms z from x, which is
// still intact. Per node the formulas, operand order and masks are those of evomg_strip2d (relax5 / residual5, the
// prolongation's `corrected`, the restriction's e4 / q4 sums), so z, w, the coarse field and the norm carry the same bits.
//
// Geometry: wave = 128 columns (lane = column pair), THREE halo lanes per side — sweep A is exact on columns [2, 125] of the
// strip, sweep B on [4, 123], w's residual on [5, 122], so coarse columns with 2I in [6, 121] and the 116 stored columns
// [6, 121] tile the row. Rows: at step j the wave relaxes A's reds on row j and A's blacks on row j-1 (z row j-1 final), forms
// the norm's residual on z row j-2, relaxes B's reds on row j-2 and B's blacks on row j-3 (w row j-3 final), forms w's
// residual on row j-4 and, every second step, coarse row (j-5)/2: a chunk of H rows reads H + 11 rows of x (the step count is
// rounded up to the unroll factor: up to five more, whose stores are all masked by their row ranges).
// Register windows (x, z, w: 3 rows; f: 6 rows; reds and residuals: 2) are rings addressed with compile-time indices — the row
// loop is unrolled SIX times, the least common multiple of the ring lengths — so advancing a row costs no register moves.
//
// The norm keeps its canonical summation order although the strips are 116 instead of 120 columns wide: every lane writes
// the sum of its column pair per 16-row chunk (4 MB at 4097^2, 1 % of the pass), and evomg_norm_from_lane_sums replays
// TAIL_NORM's wave tree over the canonical 120-column strips on those sums.
#pragma once
#include <type_traits>

#include "evomg_strip.h"

#ifndef EVOMG_CROSS_TIMING
#define EVOMG_CROSS_TIMING 0     // developer builds only (tools/build_variant.py t2 -DEVOMG_CROSS_TIMING=2, tools/ab_lib_cross.sh): 1 halo rows from the chunk's own rows, 2 no loads of x / f, 4 no stores, 8 no loads of the coarse correction — WRONG results
#endif
#define EVOMG_CROSS_STRIP_COLS 116
#define EVOMG_CROSS_UNROLL 6
#ifndef EVOMG_CROSS_PF
#define EVOMG_CROSS_PF 2        // rows of x / f in flight per wave beyond the window (a divisor of the unroll factor); 3: 176 VGPRs, 2: 158 — same speed (profiles/r05_cross_kernel.md)
#endif

struct CrossArgs {
  const double* coarse_in;   // PRO: coarse correction
  Geo gc;                    // coarse level (PRO and TAILR)
  double om_p;               // PRO: weight of the correction
  double* coarse_out;        // TAILR: coarse right-hand side (spare array)
  double om_a, omm_a;        // sweep A (the cycle's last smoother)
  double om_b, omm_b;        // sweep B (the next cycle's first smoother)
  double inv_c0;             // exact reciprocal of the centre coefficient, or 0
  double* lane_sums;         // [chunk16][sums_pitch]: per column pair, sum over the chunk's rows of the squared residual of z
  int sums_pitch;
};

// relax5 / residual5 specialised at compile time. POW2: ALL five coefficients are (signed) powers of two — -1/h^2 and 4/h^2 on a
// dyadic mesh, i.e. every level of the 2D Poisson problem (host-checked, else the general form runs). A product with a power of
// two is exact, so `r - c * w` and fma(-c, w, r) round the same real number once: the SAME BITS with half the instructions
// (4 fused instead of 4 + 4 in a relaxation, 4 instead of 8 in a residual; out-of-range intermediates — |values| beyond 1e300
// or below 1e-300 — are the only place the two forms could part). Likewise om * (r * 2^-k) = (om * 2^-k) * r. The general form
// is relax5 / residual5 of evomg_strip.h with the division (or the exact reciprocal) chosen by `inv_c0` at run time.
// UNIT (with POW2 only): the relaxation factor is 1 — BASELINE's C2 cycle, and ExaStencils' own default smoothers are close to
// it. omm = 1 - 1 = 0, so `omm * c + om_inv * r` is (+-0) + om_inv * r = om_inv * r: the same value from ONE instruction instead
// of three (8 of a row step's 64 FP64 instructions, and FP64 issue is what two thirds of this launch are). The only bits that
// can differ are the SIGN of an exact zero ((+0) + (-0) = +0, the product alone gives -0): no comparison, norm or later value
// can tell them apart (a zero of either sign times a finite coefficient adds nothing to an fma).
template <bool POW2, bool UNIT = false>
__device__ __forceinline__ double cross_relax5(const Stencil<double>& st, double c, double w, double e, double s, double n, double f,
                                               double om, double omm, double inv_c0, double om_inv) {
  if constexpr (POW2) {
    double r = f;
    r = __builtin_fma(-st.c[1], w, r);
    r = __builtin_fma(-st.c[2], e, r);
    r = __builtin_fma(-st.c[3], s, r);
    r = __builtin_fma(-st.c[4], n, r);
    if constexpr (UNIT) return om_inv * r;
    else return omm * c + om_inv * r;
  } else {
    return relax5(st, c, w, e, s, n, f, om, omm, inv_c0);
  }
}
template <bool POW2>
__device__ __forceinline__ double cross_residual5(const Stencil<double>& st, double c, double w, double e, double s, double n, double f) {
  if constexpr (POW2) {
    double a = st.c[0] * c;
    a = __builtin_fma(st.c[1], w, a);
    a = __builtin_fma(st.c[2], e, a);
    a = __builtin_fma(st.c[3], s, a);
    a = __builtin_fma(st.c[4], n, a);
    return f - a;
  } else {
    return residual5(st, c, w, e, s, n, f);
  }
}

// one fused two-colour step: reds of row j from (xm, xc, xp) = rows j-1, j, j+1, then blacks of row j-1 from the new reds of
// rows j-2 (red_slot on entry), j-1 (rm1), j; returns the finished row j-1 and leaves row j's reds in red_slot
template <bool RECIP, bool UNIT = false>
__device__ __forceinline__ Pair<double> cross_rb_step(const Stencil<double>& st, int parity, const Pair<double>& xm, const Pair<double>& xc,
                                                       const Pair<double>& xp, const Pair<double>& fc, const Pair<double>& fm, double& red_slot,
                                                       double rm1, bool row_inner, bool brow_inner, bool inner_a, bool inner_b, double om,
                                                       double omm, double inv_c0, double om_inv) {
  // every relaxation is computed by all lanes and SELECTED where the node is an inner one (straight-line code: a conditional
  // evaluation becomes an exec-masked block per relaxation, which cuts the row step into a dozen basic blocks)
  const double rm2 = red_slot;
  double rc;
  if (parity == 0) {
    const double w = from_left(xc.b);
    const double t = cross_relax5<RECIP, UNIT>(st, xc.a, w, xc.b, xm.a, xp.a, fc.a, om, omm, inv_c0, om_inv);
    rc = (row_inner && inner_a) ? t : xc.a;
  } else {
    const double e = from_right(xc.a);
    const double t = cross_relax5<RECIP, UNIT>(st, xc.b, xc.a, e, xm.b, xp.b, fc.b, om, omm, inv_c0, om_inv);
    rc = (row_inner && inner_b) ? t : xc.b;
  }
  Pair<double> out;
  if (parity == 1) {            // row j-1 even: red = a, black = b
    const double e = from_right(rm1);
    const double t = cross_relax5<RECIP, UNIT>(st, xm.b, rm1, e, rm2, rc, fm.b, om, omm, inv_c0, om_inv);
    out.a = rm1;
    out.b = (brow_inner && inner_b) ? t : xm.b;
  } else {                      // red = b, black = a
    const double w = from_left(rm1);
    const double t = cross_relax5<RECIP, UNIT>(st, xm.a, w, rm1, rm2, rc, fm.a, om, omm, inv_c0, om_inv);
    out.a = (brow_inner && inner_a) ? t : xm.a;
    out.b = rm1;
  }
  red_slot = rc;
  return out;
}

// grid: ceil(strips * chunks / 4) blocks (multiple of 8) of 256 threads = 4 independent waves; rows_per_chunk: multiple of 16
// RECIP: every coefficient of the stencil is a power of two (see cross_relax5)
// UNIT: both relaxation factors and the weight of the folded prolongation are 1 (host-checked), RECIP only
template <bool NT_XY, bool PRO, bool TAILR, bool RECIP, bool UNIT = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))      // two waves per SIMD is what the chunks give it: scheduled for that, 1.5-2.5 % (same box, profiles/r05_cross_timing_raw.txt)
evomg_strip2d_cross(Geo g, Stencil<double> st, const double* __restrict__ x, const double* __restrict__ f, double* __restrict__ y,
                    int rows_per_chunk, int n_strips, int n_waves, CrossArgs a) {
  static_assert(!UNIT || RECIP, "the unit-weight form is a form of the power-of-two one");
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int wave = __builtin_amdgcn_readfirstlane((int)(logical_block * 4 + (threadIdx.x >> 6)));
  if (wave >= n_waves) return;
  const int lane = threadIdx.x & 63;
  const int strip = wave % n_strips;
  const int chunk = wave / n_strips;
  const int n = g.n;
  const long pitch = g.pitch;
  constexpr int HALO = 3, COLS = EVOMG_CROSS_STRIP_COLS, U = EVOMG_CROSS_UNROLL;
  static_assert(COLS == 128 - 4 * HALO, "strip width");
  const int c0 = strip * COLS - 2 * HALO + 2 * lane;
  const bool col_ok = c0 >= 0 && c0 < n;
  const int y0 = chunk * rows_per_chunk;                     // multiple of 16
  const int y1 = min(y0 + rows_per_chunk, n);
  const Pair<double> zpair{0.0, 0.0};
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
#if EVOMG_CROSS_TIMING & 4
  const bool store_lane = lane >= HALO && lane <= 63 - HALO && col_ok && n < 0;
#else
  const bool store_lane = lane >= HALO && lane <= 63 - HALO && col_ok;
#endif
  const int c_clamped = min(max(c0, 0), n - 1);
  auto load_row = [&](const double* __restrict__ base, int row) -> Pair<double> {
#if EVOMG_CROSS_TIMING & 1                                    // developer builds: WRONG results — what the halo rows cost (profiles/r05_cross_kernel.md)
    const int r = min(max(row, y0), y1 - 1);
#else
    const int r = min(max(row, 0), n - 1);
#endif
    if constexpr (NT_XY) return load_pair_nt<double>(base + pitch * r + c_clamped);
    else return load_pair<double>(base + pitch * r + c_clamped);
  };
  auto load_row_f = [&](const double* __restrict__ base, int row) -> Pair<double> {
#if EVOMG_CROSS_TIMING & 1
    const int r = min(max(row, y0), y1 - 1);
#else
    const int r = min(max(row, 0), n - 1);
#endif
    return load_pair<double>(base + pitch * r + c_clamped);
  };
  const int nc = a.gc.n;
  const int Ic = min(max(c0 >> 1, 0), nc - 1);
  auto load_coarse = [&](int row) -> Pair<double> {
    const int r = min(max(row, 0), nc - 1);
    return load_pair<double>(a.coarse_in + (long)a.gc.pitch * r + Ic);
  };
  // fine row `row` of x + om_p P coarse (evomg_strip2d's `corrected`, real Dirichlet case)
  auto corrected = [&](const Pair<double>& raw, int row, bool odd, const Pair<double>& cJp, const Pair<double>& cJ1p, bool edge = true) -> Pair<double> {
    const bool row_in = !edge || (row >= 1 && row <= n - 2);
    const double cJ = cJp.a, cJr = cJp.b;
    double ca, cb;
    if constexpr (UNIT) {
      // om_p = 1: 1 * v = v, and the products with 1/2 and 1/4 are exact, so fma(0.5, s, x) rounds x + 0.5 s once, as the
      // product followed by the sum does: the same bits from 3 / 6 instructions per row instead of 6 / 10
      if (!odd) {
        ca = raw.a + cJ;
        cb = __builtin_fma(0.5, cJ + cJr, raw.b);
      } else {
        const double cJ1 = cJ1p.a, cJ1r = cJ1p.b;
        ca = __builtin_fma(0.5, cJ + cJ1, raw.a);
        cb = __builtin_fma(0.25, (cJ + cJr) + (cJ1 + cJ1r), raw.b);
      }
    } else {
      double va, vb;
      if (!odd) {
        va = cJ;
        vb = 0.5 * (cJ + cJr);
      } else {
        const double cJ1 = cJ1p.a, cJ1r = cJ1p.b;
        va = 0.5 * (cJ + cJ1);
        vb = 0.25 * ((cJ + cJr) + (cJ1 + cJ1r));
      }
      ca = raw.a + a.om_p * va;
      cb = raw.b + a.om_p * vb;
    }
    Pair<double> out;
    out.a = (row_in && (!edge || inner_a)) ? ca : raw.a;
    out.b = (row_in && (!edge || inner_b)) ? cb : raw.b;
    return out;
  };
  const int js = y0 - 5;                                     // odd: first row whose A-reds are relaxed
  const int je = y1 + 4;                                     // one past the last
  // rings: slot of row js + k is k mod (ring length); the loop below is unrolled U = 6 times, so every index is static
  constexpr int PF = EVOMG_CROSS_PF;
  static_assert(EVOMG_CROSS_UNROLL % PF == 0, "the prefetch ring has to divide the unroll factor");
  Pair<double> xw[3], zw[3], ww[3], fw[6], xq[PF], fq[PF], rs[2];
  double ra[2] = {0.0, 0.0}, rb[2] = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 3; ++k) xw[k] = zw[k] = ww[k] = zpair;
#pragma unroll
  for (int k = 0; k < 6; ++k) fw[k] = zpair;
  rs[0] = rs[1] = zpair;
  xw[2] = load_row(x, js - 1);                               // k = -1
  xw[0] = load_row(x, js);                                   // k = 0
  if constexpr (PRO) {
    // js is odd, js - 1 even
    const int rm = max(js - 1, 0), rc0 = max(js, 0);
    xw[2] = corrected(xw[2], js - 1, false, load_coarse(rm >> 1), zpair);
    xw[0] = corrected(xw[0], js, true, load_coarse(rc0 >> 1), load_coarse((rc0 >> 1) + 1));
  }
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    xq[k] = load_row(x, js + 1 + k);
    fq[k] = load_row_f(f, js + k);
  }
  // coarse rows of the folded prolongation: a ring of three (static indices: three coarse rows per six fine ones) — the rows
  // (r >> 1), (r >> 1) + 1 the fine row r = j + 1 of this step lies between, and the next one, so that a coarse row is asked for
  // four steps before it is first read (the vector-memory counter is in order: the SHORTEST prefetch distance of the loop is
  // the one every wait observes)
  Pair<double> cr[3] = {zpair, zpair, zpair};
  if constexpr (PRO) {
    const int r0 = js + 1;                                   // may be negative: >> floors, load_coarse clamps
    cr[0] = load_coarse(r0 >> 1);
    cr[1] = load_coarse((r0 >> 1) + 1);
    cr[2] = load_coarse((r0 >> 1) + 2);
  }
  double nrm = 0.0;
  const double om_inv_a = a.om_a * a.inv_c0, om_inv_b = a.om_b * a.inv_c0;     // POW2: exact (om times a power of two)
  const int ncr = a.gc.n;
  const int Icr = c0 >> 1;

  // The row loop exists twice: EDGE — the wave touches the domain's boundary (first / last strip, first / last chunk), every
  // relaxation is masked to the inner nodes; interior waves (nine in ten at 4097^2) run the same steps without any of the
  // masks: no selects, no compares (a sixth fewer instructions — and instructions are what two thirds of this launch are: with
  // every load and store taken out it still takes 60 of its 92 us, a wave issuing one FP64 instruction per 8 cycles:
  // profiles/r05_cross_kernel.md, "What bounds it").
  auto march = [&](auto edge_tag) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    const bool in_a = !EDGE || inner_a, in_b = !EDGE || inner_b;
    for (int base = js; base < je; base += U) {
  #pragma unroll
      for (int d = 0; d < U; ++d) {
        const int j = base + d;
        {                                                      // steps at and beyond je (the last trip): clamped loads, every store guarded
          const int parity = (1 + d) & 1;                      // js is odd
          Pair<double> xp = xq[d % PF];
          const Pair<double> fc = fq[d % PF];
          if constexpr (UNIT) {
            // With unit weights the old value of a RED node is never read (x_new = (f - sum of the black neighbours) / c0), so
            // half of every 16-byte load of x is dead. The register allocator then hands that half out again right behind the
            // load's issue and the wave stalls at an `s_waitcnt vmcnt(0)` for the load it has just issued — a full memory round
            // trip every second row step (seen in the ISA: profiles/r06_cross_kernel.md). Naming both halves here, where the row
            // is consumed PF steps later, keeps the destination reserved until then; no instruction is emitted.
            asm volatile("" : : "v"(xp.a), "v"(xp.b));
          }
#if !(EVOMG_CROSS_TIMING & 2)                                  // 2: no loads of x / f in the loop (the prologue's rows go round)
          xq[d % PF] = load_row(x, j + 1 + PF);
          fq[d % PF] = load_row_f(f, j + PF);
#endif
          if constexpr (PRO) {
            const bool odd_next = ((parity + 1) & 1) != 0;     // row j + 1
            xp = corrected(xp, j + 1, odd_next, cr[(d >> 1) % 3], cr[((d >> 1) + 1) % 3], EDGE);
#if !(EVOMG_CROSS_TIMING & 8)                                  // 8: no loads of the coarse correction in the loop
            if (odd_next) cr[(d >> 1) % 3] = load_coarse(((j + 1) >> 1) + 3);      // the slot of the row just left behind
#endif
          }
          xw[(d + 1) % 3] = xp;
          fw[d % 6] = fc;
          // ---- sweep A: reds of row j, blacks of row j-1 -> z row j-1 ----
          {
            const bool row_inner = !EDGE || (j >= 1 && j <= n - 2), brow_inner = !EDGE || (j - 1 >= 1 && j - 1 <= n - 2);
            zw[(d + 2) % 3] = cross_rb_step<RECIP, UNIT>(st, parity, xw[(d + 2) % 3], xw[d % 3], xw[(d + 1) % 3], fc, fw[(d + 5) % 6], ra[d & 1], ra[(d + 1) & 1],
                                            row_inner, brow_inner, in_a, in_b, a.om_a, a.omm_a, a.inv_c0, om_inv_a);
          }
          const Pair<double>& z3 = zw[d % 3];                  // z rows j-3, j-2, j-1
          const Pair<double>& z2 = zw[(d + 1) % 3];
          const Pair<double>& z1 = zw[(d + 2) % 3];
          const Pair<double>& f2 = fw[(d + 4) % 6];            // f rows j-2, j-3, j-4
          const Pair<double>& f3 = fw[(d + 3) % 6];
          const Pair<double>& f4 = fw[(d + 2) % 6];
          // ---- the loop's norm: residual of z row j-2 (TAIL_NORM of evomg_strip2d) ----
          {
            const int rr = j - 2;
            const double wa = from_left(z2.b), eb = from_right(z2.a);
            // branch-free: where TAIL_NORM adds nothing, +0.0 is added (nrm >= +0.0: the same value, bit for bit)
            const bool counted = rr >= y0 && rr < y1 && (!EDGE || (rr >= 1 && rr <= n - 2)) && store_lane;
            const double qa = cross_residual5<RECIP>(st, z2.a, wa, z2.b, z3.a, z1.a, f2.a);
            const double qb = cross_residual5<RECIP>(st, z2.b, z2.a, eb, z3.b, z1.b, f2.b);
            nrm = nrm + ((counted && in_a) ? qa * qa : 0.0);
            nrm = nrm + ((counted && in_b) ? qb * qb : 0.0);
            if (rr >= y0 && rr < y1 && ((rr & (EVOMG_NORM_ROWS - 1)) == EVOMG_NORM_ROWS - 1 || rr == n - 1)) {      // wave-uniform
              if (store_lane) a.lane_sums[(long)(rr / EVOMG_NORM_ROWS) * a.sums_pitch + (c0 >> 1)] = nrm;
              nrm = 0.0;
            }
          }
          // ---- sweep B on z: reds of row j-2, blacks of row j-3 -> w row j-3 ----
          {
            const int jb = j - 2;
            const bool row_inner = !EDGE || (jb >= 1 && jb <= n - 2), brow_inner = !EDGE || (jb - 1 >= 1 && jb - 1 <= n - 2);
            const Pair<double> wrow = cross_rb_step<RECIP, UNIT>(st, parity, z3, z2, z1, f2, f3, rb[d & 1], rb[(d + 1) & 1], row_inner, brow_inner, in_a,
                                                    in_b, a.om_b, a.omm_b, a.inv_c0, om_inv_b);
            ww[d % 3] = wrow;
            const int jw = j - 3;
            if (store_lane && jw >= y0 && jw < y1) store_pair<NT_XY>(y + pitch * jw + c0, wrow);
          }
          if constexpr (TAILR) {
            // ---- residual of w row j-4 and, on odd rows, one coarse row (TAIL_RESTRICT of evomg_strip2d) ----
            const Pair<double>& w5 = ww[(d + 1) % 3];          // w rows j-5, j-4, j-3
            const Pair<double>& w4 = ww[(d + 2) % 3];
            const Pair<double>& w3 = ww[d % 3];
            const int rr = j - 4;
            const bool rin = !EDGE || (rr >= 1 && rr <= n - 2);
            const double wa = from_left(w4.b), eb = from_right(w4.a);
            Pair<double> rcur;
            const double ta = cross_residual5<RECIP>(st, w4.a, wa, w4.b, w5.a, w3.a, f4.a), tb = cross_residual5<RECIP>(st, w4.b, w4.a, eb, w5.b, w3.b, f4.b);
            rcur.a = (rin && in_a) ? ta : 0.0;
            rcur.b = (rin && in_b) ? tb : 0.0;
            const Pair<double>& res1 = rs[(d + 1) & 1];        // residual rows rr-1, rr-2
            const Pair<double>& res2 = rs[d & 1];
            if (parity == 1) {                                 // rr odd: rows rr-2, rr-1, rr = 2J-1, 2J, 2J+1
              const int J = (rr - 1) >> 1;
              const double w = from_left(res1.b), sw = from_left(res2.b), nw = from_left(rcur.b);
              const double e4 = (w + res1.b) + (res2.a + rcur.a);
              const double q4 = (sw + res2.b) + (nw + rcur.b);
              double v = 0.25 * res1.a;                        // products with 1/4, 1/8, 1/16 are exact: fma = multiply, then add
              v = __builtin_fma(0.125, e4, v);
              v = __builtin_fma(0.0625, q4, v);
              if (store_lane && (!EDGE || (Icr >= 1 && Icr <= ncr - 2 && J >= 1 && J <= ncr - 2)) && 2 * J >= y0 && 2 * J < y1)
                a.coarse_out[(long)a.gc.pitch * J + Icr] = v;
            }
            rs[d & 1] = rcur;
          }
        }
      }
    }
  };
  // every row any step touches is an inner row, every column of the strip an inner column
  const int j_last = js + ((je - js + U - 1) / U) * U - 1;
  const bool interior = js - 1 >= 1 && j_last + 1 <= n - 2 && strip * COLS - 2 * HALO >= 1 && strip * COLS - 2 * HALO + 127 <= n - 2;
  if (interior) march(std::false_type{});
  else march(std::true_type{});
}

// The canonical partial sums of the squared residual norm — one per (120-column strip, 16-row chunk), wave64 tree over the
// lanes' column-pair sums, exactly what evomg_strip2d<.., TAIL_NORM> and <STRIP_RESNORM> emit — from the per-pair sums the
// cross kernel left behind. One wave per partial; grid: ceil(n_waves / 4) blocks of 256 threads.
__global__ void __launch_bounds__(256)
static evomg_norm_from_lane_sums(const double* __restrict__ lane_sums, int sums_pitch, int n, int n_strips, int n_waves,
                                 double* __restrict__ partials) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (wave >= n_waves) return;
  prio_latency();
  const int lane = threadIdx.x & 63;
  const int strip = wave % n_strips, c16 = wave / n_strips;
  const int c0 = strip * EVOMG_NORM_STRIP_COLS - 4 + 2 * lane;
  const bool store_lane = lane >= 2 && lane <= 61 && c0 >= 0 && c0 < n;
  const double v = store_lane ? lane_sums[(long)c16 * sums_pitch + (c0 >> 1)] : 0.0;
  const double tot = wave_sum(v);
  if (lane == 0) partials[c16 * n_strips + strip] = tot;
}

// The same, and the loop's norm itself, in ONE launch (round 6): the workgroup whose arrival ticket is the last one adds up
// all partial sums in evomg_dot_final's order (thread t: partials t, t + 256, ... one after the other; then block_sum) and
// writes the squared norm where the solver loop waits for it — the reduction launch behind this one (3 us alone, 75-86 us with
// candidates in flight: a single workgroup launched into saturated units, profiles/r05_throughput.md) is gone.
// Hand-off between workgroups without fences (MI355X_MICROARCH.md, "Valid forms"): every partial is stored with an sc1
// (agent-scope) store, the storing waves drain their stores (s_waitcnt vmcnt(0)) before the workgroup's barrier, ONE lane
// then takes a ticket with an agent-scope atomic add; the workgroup that drew the last ticket reads the partials with sc1
// loads behind a barrier of its own. The ticket counter is left at zero for the next launch (same stream: ordered).
// EVOMG_NORM_FOLD_PER partial sums per wave: 2 249 workgroups taking tickets from ONE counter serialise on it (same-address
// atomics: ~10 ns each — the launch took 33 us alone, twice the two launches it replaces); with eight partials per wave 282 do.
#define EVOMG_NORM_FOLD_PER 8
__global__ void __launch_bounds__(256)
static evomg_norm_from_lane_sums_final(const double* __restrict__ lane_sums, int sums_pitch, int n, int n_strips, int n_waves,
                                       double* __restrict__ partials, unsigned* __restrict__ ticket, double* __restrict__ out) {
  prio_latency();
  __shared__ double red[17];
  __shared__ int is_last;
  constexpr int PER = EVOMG_NORM_FOLD_PER;
  const int first = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))) * PER;
  const int lane = threadIdx.x & 63;
  double v[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {                            // all loads first, then the trees
    const int wave = first + k;
    const int strip = wave % n_strips, c16 = wave / n_strips;
    const int c0 = strip * EVOMG_NORM_STRIP_COLS - 4 + 2 * lane;
    const bool store_lane = wave < n_waves && lane >= 2 && lane <= 61 && c0 >= 0 && c0 < n;
    v[k] = store_lane ? lane_sums[(long)c16 * sums_pitch + (c0 >> 1)] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int wave = first + k;
    const double tot = wave_sum(v[k]);
    if (lane == 0 && wave < n_waves)
      __hip_atomic_store(partials + wave, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  double s = 0.0;
  for (int i0 = threadIdx.x; i0 < n_waves; i0 += 8 * 256) {
    double q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + j * 256;
      q[j] = __hip_atomic_load(partials + (i < n_waves ? i : i0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (i0 + j * 256 < n_waves) s = s + q[j];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    out[0] = s;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
