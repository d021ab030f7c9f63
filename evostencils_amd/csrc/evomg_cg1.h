// evomg_cg1.h — the coarse-grid CG with ONE reduction per iteration (Chronopoulos & Gear 1989), 2D real grids up to 129^2.
//
// gen_mgCycle@minLevel with solver_cgs = "CG" (example_problems/Poisson/2D_FD_Poisson_fromL2.exa3:12-14: at most 1000
// iterations, reduction 1e-12). Classical CG (evomg_krylov.h: evomg_cg_reg2d and friends) has two dependent
// reduce -> divide chains per iteration (p . A p, then r . r), which is what bounded it: 2.05 us per iteration at 129^2 on
// ONE compute unit — more units did not pay, two grid rendezvous of 0.6-0.8 us each cost what they saved. The recurrence
// below needs both inner products at the SAME point of the iteration, so one exchange carries everything that crosses
// waves:
//     r0 = b - A x0, w0 = A r0, g = (r0, r0), d = (w0, r0), den = d, alpha = g / den, beta = 0, p = s = 0
//     repeat:  p = r + beta p;  s = w + beta s  (= A p);  x += alpha p;  r -= alpha s
//              w = A r;  g' = (r, r);  d = (w, r)                 <- the one reduction
//              stop when sqrt(g') <= tol * sqrt(g0) or at the iteration cap   (decided on g' itself: cg1_stop_threshold)
//              beta = g' / g;  den = d - beta^2 den;  alpha = g' / den;  g = g'
// (den_k = d_k - beta_k g_k / alpha_{k-1} with alpha_{k-1} = g_{k-1} / den_{k-1}: the same real number as in the paper,
// with two divisions on the scalar chain instead of three.) Same iterates as classical CG in exact arithmetic, same
// stopping rule on the recursively updated residual; iteration counts on the Poisson levels agree with the classical form
// (tests/test_cg_variants*.py). The oracle restates exactly this recurrence and summation order (oracle/evomg_oracle_impl.h:
// coarse_cg1); the classical form stays available on both sides as cgs_kind = EVOMG_CGS_CG_CLASSICAL.
//
// Ownership ("bands"): band k owns the R inner rows k R .. k R + R - 1. A wave holds H bands side by side (H = 1: 64 lanes
// per band; narrow grids: H = 2, 4, ... bands of C = 64 / H lanes, whose last lane never has a column — the zero it holds is
// the halo between two bands of a wave), lane (h, c) of global wave gw works on band gw H + h and owns the E columns
// E c .. E c + E - 1 of each of its rows; x, r, p, s and w of these nodes live in registers for the whole solve. West / east
// neighbours are the lane's other element or one DPP wave shift away. South / north neighbours: every lane ALSO carries r
// and s of the row below its first and above its last row (ghost rows) and updates them with the very operations their
// owner executes; the only thing it needs from the neighbouring band for that is that band's edge row of w = A r, which
// travels with the exchange of the sums — so an iteration has exactly one synchronisation.
// Waves are grouped into workgroups of WAVES waves; MULTI: `groups` co-resident workgroups share the grid (129^2: the five
// vectors are 645 KB, more than one unit's registers hold), otherwise one workgroup runs the solve out of one unit.
//   within a workgroup: wave sums and edge rows through LDS (double-buffered by the parity of the exchange), one barrier;
//   across workgroups:  every wave publishes its two sums, edge waves their edge row of w, as "LL" words (evomg_krylov.h:
//                       ll_store / ll_load); wave 0 of every workgroup collects all wave sums before the barrier.
// Reduction order (restated by the oracle): fused multiply-add accumulation over a lane's nodes (rows ascending, elements
// ascending), wave64 tree (offsets 32 .. 1), then the tree with offsets 16 .. 1 over the <= 32 wave sums — the same for
// every grouping of the waves into workgroups, so `groups` and WAVES are tuning parameters, not part of the arithmetic
// (R, E and H are). Both inner products share ONE tree: after the first step (one v_permlane32_swap of the two registers)
// lanes 0..31 carry (r, r), lanes 32..63 (w, r).
// Scalars: 1 / g is formed between the stores of the exchange and its first look (time the wave otherwise sleeps through: the
// words take ~400 cycles to land, and a look that comes back without them costs its instructions and a round trip), so the
// chain behind the sums is two products, one fused multiply-add and ONE division: beta = g' (1 / g), den = d - beta^2 den,
// alpha = g' / den — formed ahead of the stopping test, which compares g' with a threshold instead of taking its square root
// (cg1_stop_threshold: the same decisions bit for bit). Round 6, from the ISA of the loop: the compiler had sunk 1 / g to
// its use and put the square root of the test in front of everything — three dependent chains behind every exchange.
#pragma once
#include "evomg_krylov.h"

// developer builds (tools/bench_cg1.hip -DEVOMG_CG1_PROFILE): clock ticks of wave 0 of workgroup 0 per phase of an iteration
#ifdef EVOMG_CG1_PROFILE
#define PROF(k) do { if (gw == 0) { const long long t_ = clock64(); prof_acc[k] += t_ - prof_t; prof_t = t_; } } while (0)
#else
#define PROF(k) do { } while (0)
#endif

#ifndef EVOMG_CG1_SLEEP0
#define EVOMG_CG1_SLEEP0 8
#endif
#ifndef EVOMG_CG1_SLEEP_LOCAL
#define EVOMG_CG1_SLEEP_LOCAL 4   // x 64 cycles, before the first look inside one XCD
#endif
#define EVOMG_CG1_MAX_WAVES 32    // all wave sums of both inner products fit one wave64: (r, r) in lanes 0..31, (w, r) in 32..63
#define EVOMG_CG1_MAX_GROUPS 16
#define EVOMG_CG1_MAX_E 2
// exchange words across workgroups: [parity][quantity][wave][2 words] sums, then [parity][group][edge][element][lane][2 words]
#define EVOMG_CG1_LL_SUM_WORDS (2 * 2 * EVOMG_CG1_MAX_WAVES * 2)
#define EVOMG_CG1_LL_GROUP_SEAM_WORDS (2 * EVOMG_CG1_MAX_GROUPS * 2 * EVOMG_CG1_MAX_E * 64 * 2)
// AUTO (every wave exchanges for itself): [parity][wave][edge][element][lane][2 words] edge rows of every wave
#define EVOMG_CG1_LL_WAVE_SEAM_WORDS (2 * EVOMG_CG1_MAX_WAVES * 2 * EVOMG_CG1_MAX_E * 64 * 2)
#define EVOMG_CG1_LL_WORDS (EVOMG_CG1_LL_SUM_WORDS + EVOMG_CG1_LL_GROUP_SEAM_WORDS + EVOMG_CG1_LL_WAVE_SEAM_WORDS)

// the band layout of a grid: rows per band, columns per lane, bands per wave, waves in total; waves == 0: no banded form
struct Cg1Plan {
  int R, E, H, waves, mask_from;
};
__host__ __device__ inline Cg1Plan cg1_plan(int n) {
  const int m = n - 2;
  Cg1Plan p{0, 0, 0, 0, 0};
  if (m < 1 || m > 64 * EVOMG_CG1_MAX_E) return p;
  p.E = m > 64 ? 2 : 1;
  p.H = m >= 32 ? 1 : m >= 16 ? 2 : m >= 8 ? 4 : m >= 4 ? 8 : 16;     // a band of a narrow grid keeps one lane without a column
  // rows per band, by measurement (tools/bench_cg1.hip): 129^2 32 waves of 4 rows, 65^2 4 waves of 16, 33^2 4 waves of 2 x 4
  p.R = m >= 64 ? 4 : m >= 32 ? 16 : m >= 16 ? 4 : m >= 8 ? 2 : 1;
  const int bands = (m + p.R - 1) / p.R;
  p.waves = (bands + p.H - 1) / p.H;
  if (p.waves > EVOMG_CG1_MAX_WAVES) { p.waves = 0; return p; }
  // rows from mask_from on may lie outside the grid in the last band that has rows (2^k + 1 grids: it owns R - 1 rows); a
  // wave with bands that own no row at all masks every row
  const int last = m - (bands - 1) * p.R;
  p.mask_from = (bands == p.waves * p.H && last >= p.R - 1) ? p.R - 1 : 0;
  return p;
}

__device__ __forceinline__ double cg1_lane(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// wave sums of two quantities in one tree: the first step (offset 32) is one v_permlane32_swap per 32-bit half — afterwards
// lanes 0..31 hold a[l] + a[l + 32], lanes 32..63 hold b[l - 32] + b[l] — then offsets 16, 8, 4, 2, 1 inside each half.
// Lane 0 ends up with the sum of a, lane 32 with the sum of b; summation order = wave_sum_fast's.
// The stopping rule `sqrt(g) <= T` (T = tol * sqrt(g0)) decided on g ITSELF: sqrt is monotone and correctly rounded, so there is
// a largest double g* with sqrt(g*) <= T, and `sqrt(g) <= T` holds exactly when g <= g*. g* lies within a few units in the
// last place of T * T and is found once per solve by stepping; the square root (a dependent chain of ~200 cycles that every
// iteration ended with) leaves the loop, the decisions are the same ones bit for bit. Where the search does not settle
// within its few steps (T * T under- or overflows, T negative, infinite or not a number) the loop keeps the square root.
struct Cg1Stop { double gstar; double t; bool exact; };
__device__ __forceinline__ Cg1Stop cg1_stop_threshold(double t) {
  Cg1Stop s{0.0, t, false};
  if (t >= 0.0 && t < __builtin_inf()) {
    double g = sc_mul(t, t);
    if (g < __builtin_inf()) {
      auto up = [](double v) { return __longlong_as_double(__double_as_longlong(v) + 1); };      // v >= 0 and finite
      auto down = [](double v) { return __longlong_as_double(__double_as_longlong(v) - 1); };    // v > 0
      for (int k = 0; k < 4; ++k)
        if (g > 0.0 && sqrt(g) > t) g = down(g);
      for (int k = 0; k < 4; ++k) {
        const double u = up(g);
        if (u < __builtin_inf() && sqrt(u) <= t) g = u;
      }
      const double u = up(g);
      s.gstar = g;
      s.exact = sqrt(g) <= t && (!(u < __builtin_inf()) || sqrt(u) > t);
    }
  }
  return s;
}
// true: the iteration stops (also for a sum that is not a number, as `!(sqrt(g) > T)` did)
__device__ __forceinline__ bool cg1_stops(const Cg1Stop& s, double g) {
  if (s.exact) return !(g > s.gstar);
  asm volatile("" : "+v"(g));                        // a real branch: without this the compiler forms the root on both paths and selects
  return !(sc_norm_from_dot(g) > s.t);
}
// `v` is formed where this stands, not where it is first used (the compiler sinks a division to its use — onto the chain
// behind the exchange — however early the source forms it)
__device__ __forceinline__ double cg1_formed_here(double v) {
  asm volatile("" : "+v"(v));
  return v;
}
// ... and what is computed from the result is computed behind the stores that precede this in the source
__device__ __forceinline__ double cg1_formed_after_stores(double v) {
  asm volatile("" : "+v"(v) : : "memory");
  return v;
}

__device__ __forceinline__ double cg1_wave_sums(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  double t = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  t += lanes_plus16(t);
  return row_tree(t);
}
// the tree over the wave sums: lane q < 32 holds quantity 0 of wave q, lane 32 + q quantity 1 (zeros beyond the last wave:
// the steps that would only add them are left out, NW = number of waves rounded up to a power of two)
template <int NW> __device__ __forceinline__ double cg1_cross_tree(double t) {
  if constexpr (NW > 16) t += lanes_plus16(t);
  if constexpr (NW > 8) t += row_plus<8>(t);
  if constexpr (NW > 4) t += row_plus<4>(t);
  if constexpr (NW > 2) t += row_plus<2>(t);
  if constexpr (NW > 1) t += row_plus<1>(t);
  return t;
}
__host__ __device__ constexpr int cg1_pow2_ceil(int v) { return v <= 1 ? 1 : v <= 2 ? 2 : v <= 4 ? 4 : v <= 8 ? 8 : v <= 16 ? 16 : 32; }

// WAVES: waves per workgroup; waves beyond cg1_plan().waves (workgroup padding) own no rows and add zeros.
// MASK_FROM: first unrolled row that may lie outside the grid (cg1_plan().mask_from, or 0 when padding waves exist).
// XZERO / b_copy: the SET_ZERO of the solution and the COPY `gen_rhs = RHS` in front of the solve folded in, as in
// evomg_cg_reg2d: b is the copy's source, the copy (whole array) is written to b_copy, x is not read.
// AUTO (MULTI only): every wave publishes its sums AND both its edge rows as words and polls for what it needs itself — no
// collector, no LDS, no barrier in the loop; meant for the placement in which the words stay in one XCD's L2.
template <int R, int E, int H, int WAVES, bool MULTI, bool XZERO, int MASK_FROM, bool AUTO = false>
__global__ void __launch_bounds__(64 * WAVES)
evomg_cg1_band2d(Geo g, Stencil<double> st, double* __restrict__ x, const double* __restrict__ b, double* __restrict__ r_out,
                 double* __restrict__ b_copy, int maxit, double tol, int* __restrict__ info, evomg_u64* __restrict__ ll,
                 unsigned seq_base, int groups, int stride, int try_local, int phase = 0) {
  static_assert(E >= 1 && E <= EVOMG_CG1_MAX_E && (E == 1 || H == 1), "columns per lane");
  static_assert(!MULTI || H == 1, "several workgroups: full-width bands only");
  static_assert(!AUTO || MULTI, "wave-autonomous exchange: the multi-workgroup form");
  constexpr int C = 64 / H;                          // lanes per band
  prio_latency();                                    // with candidates in flight: ahead of the streaming waves on its SIMDs
  int group = 0;
  if constexpr (MULTI) {
    // only the blocks `phase` modulo `stride` take part: workgroups are dealt to the XCDs round-robin, so stride 8 puts the
    // solve on ONE XCD — XCD `phase`; the host deals the solves of candidates in flight round-robin (impl_coarse.inl) —,
    // stride 1 one workgroup on each
    if ((int)blockIdx.x % stride != phase) return;        // `phase`: which XCD (the host deals solves in flight round-robin)
    group = (int)blockIdx.x / stride;
  } else {
    groups = 1;
  }
  __shared__ double sums_lds[2][64];
  __shared__ double seam_lds[2][2][E][WAVES * 64];  // [parity][edge: first / last row][element][band slot = thread]
  __shared__ int lds_fail[2];                        // sticky; [parity of the exchange whose barrier publishes it]
  __shared__ int lds_same_xcd;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int gw = group * WAVES + w;
  const int m = g.n - 2;
  const int band = gw * H + l / C;
  const int iy0 = band * R;
  const int rows = max(0, min(R, m - iy0));
  const int ix0 = E * (l % C);
  bool ok_e[E];
#pragma unroll
  for (int e = 0; e < E; ++e) ok_e[e] = ix0 + e < m;
  // Only a lane's LAST element can be without a column (host-checked for E = 2: m = 127, element 0 is column 2 l <= 126):
  // it gets its own coefficients, zero where the column is missing — its stencil sum is then zero by construction, no mask
  // in the loop; the other elements use the stencil as it is (scalar registers).
  double sl[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) sl[k] = ok_e[E - 1] ? st.c[k] : 0.0;
  const bool has_south = band > 0 && rows > 0, has_north = iy0 + R < m;
  // where the neighbouring bands' edge rows come from: this workgroup's LDS, or (MULTI) the neighbouring workgroup
  const bool south_lds = tid >= C, north_lds = tid + C < WAVES * 64;
  if (tid < 64) { sums_lds[0][tid] = 0.0; sums_lds[1][tid] = 0.0; }
  if (tid == 0) { lds_fail[0] = 0; lds_fail[1] = 0; }
  if (b_copy) {
    const long elems = (long)g.pitch * g.n;
    for (long q = (long)group * blockDim.x + tid; q < elems; q += (long)groups * blockDim.x) b_copy[q] = b[q];
  }
  auto sum_slot = [&](int par, int quantity, int wave) { return ll + ((size_t)(par * 2 + quantity) * EVOMG_CG1_MAX_WAVES + wave) * 2; };
  auto seam_slot = [&](int par, int grp, int edge, int e) {
    return ll + EVOMG_CG1_LL_SUM_WORDS + (((((size_t)par * EVOMG_CG1_MAX_GROUPS + grp) * 2 + edge) * EVOMG_CG1_MAX_E + e) * 64 + l) * 2;
  };
  // the initial residual of one node (row iy may be a ghost row or lie outside the grid: then zero)
  auto r_init = [&](int iy, int e, double* xval) -> double {
    const bool ok = ok_e[e] && iy >= 0 && iy < m;
    const long c = gidx<2>(g, min(ix0 + e, m - 1) + 1, min(max(iy, 0), m - 1) + 1, 0);
    if constexpr (XZERO) {
      if (xval) *xval = 0.0;
      return ok ? b[c] : 0.0;
    } else {
      if (xval) *xval = x[c];
      return ok ? sc_sub(b[c], apply_stencil_fma<double, 2>(st, x, g, c)) : 0.0;
    }
  };
  double r[R][E], p[R][E], s[R][E], wv[R][E], xv[R][E];
  double r_gs[E], r_gn[E], s_gs[E], s_gn[E], w_gs[E], w_gn[E];
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      r[i][e] = r_init(iy0 + i, e, &xv[i][e]);
      p[i][e] = 0.0;
      s[i][e] = 0.0;
    }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    r_gs[e] = has_south ? r_init(iy0 - 1, e, nullptr) : 0.0;
    r_gn[e] = has_north ? r_init(iy0 + R, e, nullptr) : 0.0;
    s_gs[e] = 0.0;
    s_gn[e] = 0.0;
  }
  __syncthreads();                                   // the zeroed LDS slots

  double gam, del, gn = 0.0, dl = 0.0;
  bool same_xcd = false;                             // established below (MULTI): the exchanges may stay inside one XCD's L2
#ifdef EVOMG_CG1_PROFILE
  long long prof_acc[6] = {0, 0, 0, 0, 0, 0}, prof_t = clock64();
#endif
  // w = A r on the lane's rows, partial sums of (r, r) and (w, r); operation order of apply_stencil_fma
  auto apply = [&]() {
    gam = 0.0;
    del = 0.0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const double from_w = from_left(r[i][E - 1]);  // west of the lane's first element
      const double from_e = from_right(r[i][0]);     // east of the lane's last element
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const double rc = r[i][e];
        const double rw = (e == 0) ? from_w : r[i][e > 0 ? e - 1 : 0];
        const double re = (e == E - 1) ? from_e : r[i][e < E - 1 ? e + 1 : e];
        const double rs = (i == 0) ? r_gs[e] : r[i > 0 ? i - 1 : 0][e];
        const double rn = (i == R - 1) ? r_gn[e] : r[i < R - 1 ? i + 1 : i][e];
        double a;
        if (e == E - 1) {
          a = sc_mul(sl[0], rc);
          a = sc_fma(sl[1], rw, a);
          a = sc_fma(sl[2], re, a);
          a = sc_fma(sl[3], rs, a);
          a = sc_fma(sl[4], rn, a);
        } else {
          a = sc_mul(st.c[0], rc);
          a = sc_fma(st.c[1], rw, a);
          a = sc_fma(st.c[2], re, a);
          a = sc_fma(st.c[3], rs, a);
          a = sc_fma(st.c[4], rn, a);
        }
        if (i >= MASK_FROM) a = (i < rows) ? a : 0.0;      // a row outside the grid: r is zero there, the stencil sum is not
        wv[i][e] = a;
        gam = sc_fma(rc, rc, gam);
        del = sc_fma(a, rc, del);
      }
    }
  };
  // the one synchronisation of an iteration: sums of both inner products over all waves (gn, dl) and the neighbouring
  // bands' edge rows of w (w_gs, w_gn). false: a workgroup never arrived (not co-resident) — reported, not waited for again.
  auto wseam_slot = [&](int par, int wave, int edge, int e) {
    return ll + EVOMG_CG1_LL_SUM_WORDS + EVOMG_CG1_LL_GROUP_SEAM_WORDS +
           (((((size_t)par * EVOMG_CG1_MAX_WAVES + wave) * 2 + edge) * EVOMG_CG1_MAX_E + e) * 64 + l) * 2;
  };
  bool wave_alive = true;
  // AUTO: where an exchange publishes and polls depends on its parity only, not on the iteration: both address sets are
  // formed ONCE (round 6 — the ISA of the loop had 175 instructions between the end of the stencil pass and the first poll,
  // most of them 64-bit address arithmetic and branches around the five stores: a third of an iteration). The row loop below
  // is unrolled by two so that each exchange names its set statically. A wave publishes BOTH its edge rows always (the
  // first and the last wave's outer rows go to slots nobody polls), under ONE branch on the placement.
  constexpr int KA = 1 + 2 * E;
  struct AutoSlots {
    evomg_u64* sum;                 // lane 0: (r, r), lane 32: (w, r) of this wave
    evomg_u64* seam[2][E];          // this wave's first / last row of w
    const evomg_u64* poll[KA];      // all wave sums (lane = (quantity, wave)), the neighbours' edge rows
  };
  AutoSlots slots_a{}, slots_b{};
  bool auto_need[KA];
  if constexpr (AUTO) {
    const int total = groups * WAVES;
    auto fill = [&](AutoSlots& S, int par) {
      S.sum = sum_slot(par, l >> 5, gw);
      S.poll[0] = sum_slot(par, l >> 5, (l & 31) < total ? (l & 31) : gw);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        S.seam[0][e] = wseam_slot(par, gw, 0, e);
        S.seam[1][e] = wseam_slot(par, gw, 1, e);
        S.poll[1 + e] = gw > 0 ? wseam_slot(par, gw - 1, 1, e) : sum_slot(par, 0, gw);
        S.poll[1 + E + e] = gw + 1 < total ? wseam_slot(par, gw + 1, 0, e) : sum_slot(par, 0, gw);
      }
    };
    // exchange number n of this launch (n = 0: the one behind the first stencil pass) carries seq_base + 2 + n
    const int par_a = (int)((seq_base + 2u) & 1u);
    fill(slots_a, par_a);
    fill(slots_b, par_a ^ 1);
    auto_need[0] = (l & 31) < total;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      auto_need[1 + e] = gw > 0;
      auto_need[1 + E + e] = gw + 1 < total;
    }
  }
  // `recip`: in: the g of the iteration before; out: 1 / g, which the scalars behind the exchange start from. Its division is
  // formed HERE, between the stores of the exchange and the first look — time the wave would otherwise sleep through, waiting
  // for the words to land (left to the compiler the division sinks to its use, onto the chain behind the exchange)
  auto exchange_auto = [&](unsigned seq, const AutoSlots& S, double& recip) -> bool {
    PROF(0);
    const double ws = cg1_wave_sums(gam, del);       // lane 0: the wave's (r, r), lane 32: its (w, r)
    // (ONE branch on the placement around all five stores: the kind of store is then a constant of each arm)
    if (same_xcd) {
      if ((l & 31) == 0) ll_store16(S.sum, ws, seq, true);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        ll_store16(S.seam[0][e], wv[0][e], seq, true);
        ll_store16(S.seam[1][e], wv[R - 1][e], seq, true);
      }
      // the first look is not taken before the words can have landed: a look that comes back without them costs its ~25
      // instructions and a round trip (measured before this wait: 1.87 looks per poll)
      recip = cg1_formed_here(sc_div(1.0, cg1_formed_after_stores(recip)));
      __builtin_amdgcn_s_sleep(EVOMG_CG1_SLEEP_LOCAL);
    } else {
      recip = sc_div(1.0, recip);
      if ((l & 31) == 0) ll_store16(S.sum, ws, seq, false);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        ll_store16(S.seam[0][e], wv[0][e], seq, false);
        ll_store16(S.seam[1][e], wv[R - 1][e], seq, false);
      }
      __builtin_amdgcn_s_sleep(EVOMG_CG1_SLEEP0);   // across XCDs the words cannot have landed yet: no poll that must fail
    }
    PROF(1);
    double got[KA];
    const bool all = ll_poll16<KA>(S.poll, auto_need, seq, got);
    if (__builtin_amdgcn_ballot_w64(!all) != 0ull) wave_alive = false;
    PROF(2);
    PROF(3);
    double t = auto_need[0] ? got[0] : 0.0;
    t = cg1_cross_tree<32>(t);
    gn = cg1_lane(t, 0);
    dl = cg1_lane(t, 32);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      w_gs[e] = has_south ? got[1 + e] : 0.0;
      w_gn[e] = has_north ? got[1 + E + e] : 0.0;
    }
    PROF(4);
    return wave_alive;
  };
  auto exchange = [&](unsigned seq, double& recip) -> bool {
    const int par = (int)(seq & 1u);
    PROF(0);
    const double ws = cg1_wave_sums(gam, del);       // lane 0: the wave's (r, r), lane 32: its (w, r)
    recip = cg1_formed_here(sc_div(1.0, recip));
#pragma unroll
    for (int e = 0; e < E; ++e) {
      seam_lds[par][0][e][tid] = wv[0][e];
      seam_lds[par][1][e][tid] = wv[R - 1][e];
    }
    PROF(1);
    double ll_gs[E], ll_gn[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { ll_gs[e] = 0.0; ll_gn[e] = 0.0; }
    if constexpr (!MULTI) {
      if ((l & 31) == 0) sums_lds[par][(l & 32) + w] = ws;
    } else {
      if ((l & 31) == 0) ll_store16(sum_slot(par, l >> 5, gw), ws, seq, same_xcd);
      if (w == 0 && group > 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) ll_store16(seam_slot(par, group, 0, e), wv[0][e], seq, same_xcd);
      }
      if (w == WAVES - 1 && group < groups - 1) {
#pragma unroll
        for (int e = 0; e < E; ++e) ll_store16(seam_slot(par, group, 1, e), wv[R - 1][e], seq, same_xcd);
      }
      // wave 0 collects the sums of all waves (lane = (quantity, wave)); the edge waves fetch the neighbouring workgroups'
      // edge rows. Everything a lane waits for is asked for in one go and polled together: one round trip, not one per word.
      constexpr int K = 1 + 2 * E;
      const evomg_u64* slot[K];
      bool need[K];
      slot[0] = sum_slot(par, l >> 5, l & 31);
      need[0] = w == 0 && (l & 31) < groups * WAVES;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        slot[1 + e] = seam_slot(par, group > 0 ? group - 1 : 0, 1, e);
        need[1 + e] = w == 0 && group > 0;
        slot[1 + E + e] = seam_slot(par, group < groups - 1 ? group + 1 : group, 0, e);
        need[1 + E + e] = w == WAVES - 1 && group < groups - 1;
      }
      if (w == 0 || w == WAVES - 1) {
        // (a slot this lane does not need: any readable word will do — its own sum slot)
#pragma unroll
        for (int k = 0; k < K; ++k)
          if (!need[k]) slot[k] = sum_slot(par, 0, gw);
        double got[K];
        if (!same_xcd) __builtin_amdgcn_s_sleep(EVOMG_CG1_SLEEP0);      // across XCDs the words cannot have landed yet: no poll that must fail
        const bool all = ll_poll16<K>(slot, need, seq, got);
        // a missing sum is seen by everybody behind THIS barrier, a missing edge row behind the next one
        if (!all) lds_fail[need[0] ? par : par ^ 1] = 1;
        if (w == 0) sums_lds[par][l] = (need[0] && all) ? got[0] : 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if (need[1 + e] && all) ll_gs[e] = got[1 + e];
          if (need[1 + E + e] && all) ll_gn[e] = got[1 + E + e];
        }
      }
    }
    PROF(2);
    __syncthreads();
    PROF(3);
    double t = sums_lds[par][l];
    double vs[E], vn[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      vs[e] = south_lds ? seam_lds[par][1][e][south_lds ? tid - C : tid] : ll_gs[e];
      vn[e] = north_lds ? seam_lds[par][0][e][north_lds ? tid + C : tid] : ll_gn[e];
    }
    t = cg1_cross_tree<MULTI ? 32 : cg1_pow2_ceil(WAVES)>(t);
    gn = cg1_lane(t, 0);
    dl = cg1_lane(t, 32);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      w_gs[e] = has_south ? vs[e] : 0.0;
      w_gn[e] = has_north ? vn[e] : 0.0;
    }
    // failures seen before this barrier (a missing sum) or after the previous one (a missing edge row): the same answer in
    // every wave of the workgroup, so all of them leave the loop together
    bool ok = true;
    if constexpr (MULTI) ok = lds_fail[par] == 0;
    PROF(4);
    return ok;
  };

  unsigned seq = seq_base;
  if constexpr (MULTI) {
    // Do all workgroups sit on ONE XCD (the host asked for that with stride 8; workgroups are dealt round-robin)? Every wave
    // publishes its XCD's number the safe way, wave 0 of every workgroup compares all of them: if so, the exchanges of the
    // solve stay inside that XCD's L2 (ll_store16). Every workgroup reaches the same verdict from the same words.
    ++seq;
    const int par = (int)(seq & 1u);
    const unsigned xcc = ll_xcc_id();
    if (l == 0) ll_store16(sum_slot(par, 0, gw), (double)xcc, seq, false);
    if (w == 0) {
      const int total = groups * WAVES;
      const evomg_u64* const slot[1] = {sum_slot(par, 0, (l & 31) < total ? (l & 31) : gw)};
      const bool need[1] = {true};
      double got[1] = {0.0};
      const bool arrived = ll_poll16<1>(slot, need, seq, got);
      if (!arrived) lds_fail[par] = 1;
      const bool same = arrived && got[0] == (double)xcc;
      const bool all_same = __builtin_amdgcn_ballot_w64(same) == ~0ull;
      if (l == 0) lds_same_xcd = (try_local && stride > 1 && all_same) ? 1 : 0;
    }
    __syncthreads();
    same_xcd = lds_same_xcd != 0;
    // a rendezvous missed HERE (a workgroup not co-resident) ends the solve at once, in every form of the exchange: the
    // wave-autonomous one never looks at lds_fail again, and its verdict on the placement would differ between workgroups
    // 
    if (lds_fail[par] != 0) wave_alive = false;
  }
  apply();
  bool alive = wave_alive;
  if (alive) {
    ++seq;
    double none = 1.0;
    if constexpr (AUTO) alive = exchange_auto(seq, slots_a, none);
    else alive = exchange(seq, none);
  }
  const double res0 = sc_norm_from_dot(gn);
  const Cg1Stop stop = cg1_stop_threshold(tol * res0);
  int it = 0;
  if (alive && res0 != 0.0 && maxit > 0) {
    double den = dl, alpha = sc_div(gn, dl), beta = 0.0;
    // one iteration; `xchg`: its exchange. false: the loop is over (converged, capped, or a rendezvous was missed)
    auto iterate = [&](auto&& xchg) -> bool {
      double inv_g = gn;                             // for the next beta: the exchange turns it into 1 / g (see exchange_auto)
      const double nalpha = sc_neg(alpha);
#pragma unroll
      for (int i = 0; i < R; ++i)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          p[i][e] = sc_fma(beta, p[i][e], r[i][e]);
          s[i][e] = sc_fma(beta, s[i][e], wv[i][e]);
          xv[i][e] = sc_fma(alpha, p[i][e], xv[i][e]);
          r[i][e] = sc_fma(nalpha, s[i][e], r[i][e]);
        }
#pragma unroll
      for (int e = 0; e < E; ++e) {                  // the ghost rows: what their owners do to them
        s_gs[e] = sc_fma(beta, s_gs[e], w_gs[e]);
        r_gs[e] = sc_fma(nalpha, s_gs[e], r_gs[e]);
        s_gn[e] = sc_fma(beta, s_gn[e], w_gn[e]);
        r_gn[e] = sc_fma(nalpha, s_gn[e], r_gn[e]);
      }
      apply();
      alive = xchg(++seq, inv_g);
      ++it;
      if (!alive) return false;
      // the scalars of the NEXT iteration are formed before the stopping test is branched on (simply not used when the loop
      // ends here). What the ISA of the loop showed before round 6's second pass: square root of the test -> branch -> the
      // division 1 / g (sunk here from the top of the iteration) -> two products, one fused multiply-add -> the division
      // of alpha: three dependent chains one after the other, ~500 cycles of an iteration's ~2500. Now the one division.
      const double beta_next = sc_mul(gn, inv_g);
      const double den_next = sc_fma(sc_neg(sc_mul(beta_next, beta_next)), den, dl);
      const double alpha_next = cg1_formed_here(sc_div(gn, den_next));
      if (cg1_stops(stop, gn) || it >= maxit) return false;
      beta = beta_next;
      den = den_next;
      alpha = alpha_next;
      PROF(5);
      return true;
    };
    if constexpr (AUTO) {
      // unrolled by two: the exchanges alternate between the two address sets (the first one above took set a)
      auto xb = [&](unsigned q, double& recip) { return exchange_auto(q, slots_b, recip); };
      auto xa = [&](unsigned q, double& recip) { return exchange_auto(q, slots_a, recip); };
      for (;;) {
        if (!iterate(xb)) break;
        if (!iterate(xa)) break;
      }
    } else {
      auto xg = [&](unsigned q, double& recip) { return exchange(q, recip); };
      while (iterate(xg)) {}
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (ok_e[e] && i < rows) {
        const long c = gidx<2>(g, ix0 + e + 1, iy0 + i + 1, 0);
        x[c] = xv[i][e];
        r_out[c] = r[i][e];
      }
#ifdef EVOMG_CG1_PROFILE
  if (gw == 0 && l == 0) for (int k = 0; k < 6; ++k) reinterpret_cast<long long*>(info + 4)[k] = prof_acc[k];
#endif
  // info[1] is sticky (cleared by the host before an evaluation): any workgroup may be the one that saw a missed rendezvous
  if (tid == 0) {
    if (group == 0) { info[0] = it; info[2] += it; }      // [2]: iterations of all coarse solves since the host last cleared it
    if (!alive) info[1] = 1;
  }
}

// ---- the single-reduction CG on 33^3: eight co-resident workgroups, one slab of z planes each ----------------------------
// The 3D counterpart of evomg_cg1_band2d and the successor of evomg_cg_multi_wg3d (evomg_krylov.h) for cgs_kind =
// EVOMG_CGS_CG: same slabs, same column ownership (thread t of a workgroup owns the column (t % m, t / m) of its slab, up
// to four nodes, all vectors in registers, r additionally as a zero-haloed image in LDS for the in-plane neighbours), same
// summation order (fused accumulation along the column, wave64 tree, tree over the 16 wave sums, the eight slab sums added
// in the order 0 .. 7 — oracle: dot_canon_grouped) — but ONE exchange per iteration instead of two: the slab sums of
// (r, r) and (w, r) and the slab's first and last plane of w = A r. The planes of r and s next to the slab (ghost planes)
// are kept in registers and updated with the operations their owner executes, as the ghost rows of the 2D kernel.
// Per iteration: three workgroup barriers (image of r; wave sums; slab sums) and one grid rendezvous, where the classical
// form has five and two.
// ll: [parity][quantity][group][2 words] sums, then [parity][group][first | last plane][1024][2 words] faces of w — inside
// the exchange area of evomg_cg_multi_wg3d (EVOMG_CG_LL_WORDS).
// (static: the header is part of several translation units)
static __global__ void __launch_bounds__(1024)
evomg_cg1_slab3d(Geo g, Stencil<double> st, double* __restrict__ x, const double* __restrict__ b, double* __restrict__ r_out,
                 int maxit, double tol, int* __restrict__ info, evomg_u64* __restrict__ ll, unsigned seq_base, int stride, int try_local, int phase = 0) {
  constexpr int G = EVOMG_CG_GROUPS, S = EVOMG_CG_SLAB;
  if ((int)blockIdx.x % stride != phase) return;         // `phase`: which XCD (evomg_cg1_band2d)
  prio_latency();
  const int grp = (int)blockIdx.x / stride;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double* wave_sums = reinterpret_cast<double*>(lds_raw);       // [parity][64]: (r, r) of wave q at q, (w, r) at 32 + q
  double* slab_sums = wave_sums + 128;                           // [parity][64]: (r, r) of slab q at q, (w, r) at 32 + q
  int* lds_fail = reinterpret_cast<int*>(slab_sums + 128);       // [0], [1]: sticky failure flags; [2]: all workgroups on one XCD
  double* img = slab_sums + 128 + 2;                             // S planes x (m+2) x (m+2), zero halo
  const KrylovOwn own = krylov_own<3>(g.n);
  const int m = own.m, W = m + 2, PS = W * W;
  const int zs = (grp * m) / G, ze = ((grp + 1) * m) / G, cnt = ze - zs;      // 1 <= cnt <= S (m >= 2 G)
  const int t = threadIdx.x, wv_id = t >> 6, l = t & 63;
  for (int i = t; i < S * PS; i += blockDim.x) img[i] = 0.0;
  if (t < 128) { wave_sums[t] = 0.0; slab_sums[t] = 0.0; }
  if (t == 0) { lds_fail[0] = 0; lds_fail[1] = 0; }
  const bool col = t < m * m;
  const int ix = col ? t % m : 0, iy = col ? t / m : 0;
  const int o0 = col ? (iy + 1) * W + ix + 1 : 0;      // column-less threads sit on the halo corner (always zero)
  const long c0 = gidx<3>(g, ix + 1, iy + 1, zs + 1);
  auto sum_slot = [&](int par, int quantity, int group) { return ll + ((size_t)(par * 2 + quantity) * G + group) * 2; };
  auto face_slot = [&](int par, int group, int which) {
    return ll + EVOMG_CG_LL_PART_WORDS + ((((size_t)par * G + group) * 2 + which) * 1024 + t) * 2;
  };
  __syncthreads();

  double r[S], xr[S], p[S], s[S], wv[S];
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const bool ok = col && i < cnt;
    const long c = c0 + g.plane * (i < cnt ? i : 0);
    r[i] = ok ? sc_sub(b[c], apply_stencil_fma<double, 3>(st, x, g, c)) : 0.0;
    xr[i] = x[c];
    p[i] = 0.0;
    s[i] = 0.0;
    if (ok) img[i * PS + o0] = r[i];
  }
  // the planes of r below and above the slab (the neighbouring slabs' last / first plane): formed here as their owners
  // form them, updated below as their owners update them
  double r_lo = 0.0, r_hi = 0.0, s_lo = 0.0, s_hi = 0.0, w_lo = 0.0, w_hi = 0.0;
  if (col && grp > 0) { const long c = c0 - g.plane; r_lo = sc_sub(b[c], apply_stencil_fma<double, 3>(st, x, g, c)); }
  if (col && grp < G - 1) { const long c = c0 + g.plane * cnt; r_hi = sc_sub(b[c], apply_stencil_fma<double, 3>(st, x, g, c)); }
  auto last_of = [&](const double (&a)[S]) -> double {       // a[cnt - 1] without a dynamically indexed (scratch) array
    double v = a[0];
#pragma unroll
    for (int i = 1; i < S; ++i) {
      double cand = a[i];
      krylov_opaque(cand);
      v = (i < cnt) ? cand : v;
    }
    return v;
  };
  double gam, del, gn = 0.0, dl = 0.0;
  bool same_xcd = false;                             // established below: the exchanges may stay inside one XCD's L2
  auto apply = [&]() {
    gam = 0.0;
    del = 0.0;
#pragma unroll
    for (int i = 0; i < S; ++i) {
      const bool ok = col && i < cnt;
      const double* q = img + i * PS + o0;
      const double rc = r[i];
      const double rd = (i == 0) ? r_lo : r[i > 0 ? i - 1 : 0];
      const double ru = (i + 1 < cnt) ? r[(i + 1 < S) ? i + 1 : i] : r_hi;
      double a = sc_mul(st.c[0], rc);
      a = sc_fma(st.c[1], q[-1], a);
      a = sc_fma(st.c[2], q[1], a);
      a = sc_fma(st.c[3], q[-W], a);
      a = sc_fma(st.c[4], q[W], a);
      a = sc_fma(st.c[5], rd, a);
      a = sc_fma(st.c[6], ru, a);
      a = ok ? a : 0.0;
      wv[i] = a;
      gam = sc_fma(rc, rc, gam);
      del = sc_fma(a, rc, del);
    }
  };
  // sums over the whole grid (gn, dl); the neighbouring slabs' faces of w are fetched by faces() once the caller has done what
  // does not need them (they were published before the sums: by then they have landed)
  auto exchange = [&](unsigned seq) -> bool {
    const int par = (int)(seq & 1u);
    const double ws = cg1_wave_sums(gam, del);
    if ((l & 31) == 0) wave_sums[par * 64 + (l & 32) + wv_id] = ws;
    if (col && grp > 0) ll_store16(face_slot(par, grp, 0), wv[0], seq, same_xcd);
    if (col && grp < G - 1) ll_store16(face_slot(par, grp, 1), last_of(wv), seq, same_xcd);
    __syncthreads();
    if (wv_id == 0) {
      double v = wave_sums[par * 64 + l];            // lanes 0..15: (r, r) of the 16 waves, lanes 32..47: (w, r)
      v = cg1_cross_tree<16>(v);
      if ((l & 31) == 0) ll_store16(sum_slot(par, l >> 5, grp), v, seq, same_xcd);
      const int q = l & 31;
      const evomg_u64* const slot[1] = {sum_slot(par, l >> 5, q < G ? q : grp)};
      const bool need[1] = {q < G};
      double got[1];
      if (!ll_poll16<1>(slot, need, seq, got)) lds_fail[par] = 1;
      slab_sums[par * 64 + l] = q < G ? got[0] : 0.0;
    }
    __syncthreads();
    double a = slab_sums[par * 64], d = slab_sums[par * 64 + 32];
#pragma unroll
    for (int q = 1; q < G; ++q) {
      a = a + slab_sums[par * 64 + q];
      d = d + slab_sums[par * 64 + 32 + q];
    }
    gn = a;
    dl = d;
    return lds_fail[par] == 0;
  };
  bool faces_ok = true;
  auto faces = [&](unsigned seq) {
    const int par = (int)(seq & 1u);
    w_lo = 0.0;
    w_hi = 0.0;
    const evomg_u64* const slot[2] = {face_slot(par, grp > 0 ? grp - 1 : grp, 1), face_slot(par, grp < G - 1 ? grp + 1 : grp, 0)};
    const bool need[2] = {col && grp > 0, col && grp < G - 1};
    double got[2];
    if (need[0] || need[1]) {
      faces_ok &= ll_poll16<2>(slot, need, seq, got);
      if (need[0]) w_lo = got[0];
      if (need[1]) w_hi = got[1];
    }
  };

  unsigned seq = seq_base;
  {
    // all eight workgroups on ONE XCD (stride 8)? then the exchanges stay inside its L2 (evomg_cg1_band2d explains)
    ++seq;
    const int par = (int)(seq & 1u);
    const unsigned xcc = ll_xcc_id();
    if (t == 0) ll_store16(sum_slot(par, 0, grp), (double)xcc, seq, false);
    if (wv_id == 0) {
      const evomg_u64* const slot[1] = {sum_slot(par, 0, (l & 31) < G ? (l & 31) : grp)};
      const bool need[1] = {true};
      double got[1] = {0.0};
      const bool arrived = ll_poll16<1>(slot, need, seq, got);
      if (!arrived) lds_fail[par] = 1;
      const bool all_same = __builtin_amdgcn_ballot_w64(arrived && got[0] == (double)xcc) == ~0ull;
      if (l == 0) lds_fail[2] = (try_local && stride > 1 && all_same) ? 1 : 0;
    }
    __syncthreads();                                 // (also: the image of r)
    same_xcd = lds_fail[2] != 0;
  }
  apply();
  bool alive = exchange(++seq);
  const double res0 = sc_norm_from_dot(gn);
  const Cg1Stop stop = cg1_stop_threshold(tol * res0);
  int it = 0;
  if (alive && res0 != 0.0 && maxit > 0) {
    double den = dl, alpha = sc_div(gn, dl), beta = 0.0;
    for (;;) {
      const double inv_g = cg1_formed_here(sc_div(1.0, gn));
      const double nalpha = sc_neg(alpha);
#pragma unroll
      for (int i = 0; i < S; ++i) {
        p[i] = sc_fma(beta, p[i], r[i]);
        s[i] = sc_fma(beta, s[i], wv[i]);
        xr[i] = sc_fma(alpha, p[i], xr[i]);
        r[i] = sc_fma(nalpha, s[i], r[i]);
        if (col && i < cnt) img[i * PS + o0] = r[i];
      }
      faces(seq);                                    // their trip hid behind the update
      s_lo = sc_fma(beta, s_lo, w_lo);
      r_lo = sc_fma(nalpha, s_lo, r_lo);
      s_hi = sc_fma(beta, s_hi, w_hi);
      r_hi = sc_fma(nalpha, s_hi, r_hi);
      __syncthreads();                               // the image of r
      apply();
      alive = exchange(++seq);
      ++it;
      if (!alive) break;
      // (as in evomg_cg1_band2d: the next scalars ahead of the stopping branch)
      const double beta_next = sc_mul(gn, inv_g);
      const double den_next = sc_fma(sc_neg(sc_mul(beta_next, beta_next)), den, dl);
      const double alpha_next = cg1_formed_here(sc_div(gn, den_next));
      if (cg1_stops(stop, gn) || it >= maxit) break;
      beta = beta_next;
      den = den_next;
      alpha = alpha_next;
    }
  }
#pragma unroll
  for (int i = 0; i < S; ++i)
    if (col && i < cnt) {
      const long c = c0 + g.plane * i;
      x[c] = xr[i];
      r_out[c] = r[i];
    }
  if (t == 0) {
    if (grp == 0) { info[0] = it; info[2] += it; }      // [2]: iterations of all coarse solves since the host last cleared it
    if (!alive) info[1] = 1;
  }
  if (!faces_ok) info[1] = 1;
}
