// evomg_fas.h — non-linear FAS problems: -Lap u + gam u e^u = f, 2D, real, u = 0 on the boundary; the
// statement kinds the reference's second backend emits (evostencils/code_generation/exastencils_FAS.py:99-319) on its
// example problem (example_problems/FAS_2D_Basic/FAS_2D_Basic_template.exa4).
//
// First implementation, as for the systems of PDEs: correct and complete for the op vocabulary, plain kernels (thread per
// node, no fusion) — the reference runs this problem on levels 6..10. Operation order = the layer-4 statement texts, left
// to right = oracle/evomg_oracle_fas.h. exp() is the device library's: results agree with the CPU oracle to rounding of
// the exponential, not bit for bit (tests: 1e-12 relative per statement, 1e-10 on residual histories).
//   A(X; S) := ( ( Laplace * X ) + ( gamSten * X ) ),  gamSten = gam * exp( S )
#pragma once
#include "evomg_kernels.h"

// `centre`: the value of X at the node (the smoother's local iterate); s: Solution at the node (what gamSten reads)
__device__ __forceinline__ double fas_apply(const Stencil<double>& st, double gam, const double* x, const Geo& g, long c, double centre, double s) {
  double a = st.c[0] * centre;
  a = a + st.c[1] * x[c - 1];
  a = a + st.c[2] * x[c + 1];
  a = a + st.c[3] * x[c - g.pitch];
  a = a + st.c[4] * x[c + g.pitch];
  return a + (gam * exp(s)) * centre;
}
// Solution += w * ( ( RHS - A(Solution) ) / ( Laplace:[0,0] [+ gam*(Solution + 1)*exp(Solution)] ) ), n local steps
__device__ __forceinline__ double fas_relax(const Stencil<double>& st, double gam, const double* x, const double* f, const Geo& g, long c,
                                            double om, int newton, int n_steps) {
  double u = x[c];
  for (int s = 0; s < (n_steps > 1 ? n_steps : 1); ++s) {
    const double num = f[c] - fas_apply(st, gam, x, g, c, u, u);
    const double den = newton ? st.c[0] + (gam * (u + 1.0)) * exp(u) : st.c[0];
    u = u + om * (num / den);
  }
  return u;
}
// colour < 0: Jacobi sweep x -> y (inner nodes; y keeps its zero boundary); colour 0 | 1: one colour in place (y == x)
__global__ void __launch_bounds__(256) evomg_fas_smooth(Geo g, Stencil<double> st, double gam, const double* x, const double* f, double* y,
                                                        double om, int newton, int n_steps, int colour) {
  int i, j, k;
  if (!node_of_thread<2>(g, i, j, k) || !is_inner<2>(g, i, j, 0)) return;
  if (colour >= 0 && ((i + j) & 1) != colour) return;
  const long c = gidx<2>(g, i, j, 0);
  y[c] = fas_relax(st, gam, x, f, g, c, om, newton, n_steps);
}
// Residual = ( RHS - A(Solution) )
__global__ void __launch_bounds__(256) evomg_fas_residual(Geo g, Stencil<double> st, double gam, const double* __restrict__ x,
                                                          const double* __restrict__ f, double* __restrict__ r) {
  int i, j, k;
  if (!node_of_thread<2>(g, i, j, k) || !is_inner<2>(g, i, j, 0)) return;
  const long c = gidx<2>(g, i, j, 0);
  r[c] = f[c] - fas_apply(st, gam, x, g, c, x[c], x[c]);
}
__device__ __forceinline__ double fas_fw(const Geo& gf, const double* __restrict__ fine, int I, int J) {   // = evomg_restrict<double, 2>
  const long c = gidx<2>(gf, 2 * I, 2 * J, 0);
  const long p = gf.pitch;
  const double e = (fine[c - 1] + fine[c + 1]) + (fine[c - p] + fine[c + p]);
  const double q = (fine[c - p - 1] + fine[c - p + 1]) + (fine[c + p - 1] + fine[c + p + 1]);
  double v = 0.25 * fine[c];
  v = v + 0.125 * e;
  v = v + 0.0625 * q;
  return v;
}
// Approximation@c = R * Solution@f ; Solution@c = Approximation@c
__global__ void __launch_bounds__(256) evomg_fas_restrict_solution(Geo gc, Geo gf, const double* __restrict__ fine_sol,
                                                                   double* __restrict__ approx, double* __restrict__ sol) {
  int I, J, K;
  if (!node_of_thread<2>(gc, I, J, K) || !is_inner<2>(gc, I, J, 0)) return;
  const double v = fas_fw(gf, fine_sol, I, J);
  const long c = gidx<2>(gc, I, J, 0);
  approx[c] = v;
  sol[c] = v;
}
// RHS@c = ( ( R * Residual@f ) + A(Approximation@c; Solution@c) )
__global__ void __launch_bounds__(256) evomg_fas_coarse_rhs(Geo gc, Geo gf, Stencil<double> st, double gam, const double* __restrict__ fine_res,
                                                            const double* __restrict__ approx, const double* __restrict__ sol,
                                                            double* __restrict__ rhs) {
  int I, J, K;
  if (!node_of_thread<2>(gc, I, J, K) || !is_inner<2>(gc, I, J, 0)) return;
  const long c = gidx<2>(gc, I, J, 0);
  rhs[c] = fas_fw(gf, fine_res, I, J) + fas_apply(st, gam, approx, gc, c, approx[c], sol[c]);
}
// Solution -= Approximation
__global__ void __launch_bounds__(256) evomg_fas_subtract(Geo g, double* __restrict__ sol, const double* __restrict__ approx) {
  int i, j, k;
  if (!node_of_thread<2>(g, i, j, k) || !is_inner<2>(g, i, j, 0)) return;
  const long c = gidx<2>(g, i, j, 0);
  sol[c] = sol[c] - approx[c];
}

// CGS@coarsest: `sweeps` Newton-Jacobi sweeps (one local step, weight om) in ONE launch of ONE 1024-thread workgroup when
// the level has at most 65^2 nodes: a sweep there is a few thousand nodes, 200 launches of it are 200 x 5 us of launch
// floor for 200 x ~1 us of work. Thread t owns the inner nodes t, t + 1024, ...; a barrier between sweeps; the two slots
// alternate. Same per-node formula as evomg_fas_smooth (fas_relax): bit-identical to the launch-per-sweep form.
// After an even number of sweeps the result is in `a`, after an odd number in `b` (the host swaps accordingly).
__global__ void __launch_bounds__(1024) evomg_fas_cgs(Geo g, Stencil<double> st, double gam, double* a, double* b, const double* f,
                                                      double om, int sweeps) {
  const int m = g.n - 2;
  const int total = m * m;
  double* x = a;
  double* y = b;
  for (int s = 0; s < sweeps; ++s) {
    for (int q = threadIdx.x; q < total; q += 1024) {
      const int i = 1 + q % m, j = 1 + q / m;
      const long c = gidx<2>(g, i, j, 0);
      y[c] = fas_relax(st, gam, x, f, g, c, om, 1, 1);
    }
    __syncthreads();
    double* t = x; x = y; y = t;
  }
}

// The same sweeps with the iterate in LDS: both slots of the 65^2 field (2 x 33 KB) stay on the compute unit
// for the whole call (each with the boundary values its array holds), the right-hand side of a thread's (up to four) nodes in registers; a sweep reads its five values from
// one LDS image and writes the other — no global-memory round trip between the sweeps, what is left is the arithmetic
// (one exponential, one division per node). Per-node formula and operation order = fas_relax(.., newton = 1, n_steps = 1):
// bit-identical to evomg_fas_cgs and to the launch-per-sweep form. a / b: as evomg_fas_cgs (the result of the last sweep
// and of the one before it are written back to the slots where the global-memory form would have left them).
// dynamic LDS: 2 * n * n doubles.
__global__ void __launch_bounds__(1024) evomg_fas_cgs_lds(Geo g, Stencil<double> st, double gam, double* a, double* b, const double* f,
                                                          double om, int sweeps) {
  extern __shared__ __align__(16) double fas_lds[];
  const int n = g.n, m = n - 2, total = m * m, nn = n * n;
  double* cur = fas_lds;
  double* nxt = fas_lds + nn;
  for (int q = threadIdx.x; q < nn; q += 1024) {
    const int i = q % n, j = q / n;
    const bool inner = i >= 1 && i <= m && j >= 1 && j <= m;
    cur[q] = a[gidx<2>(g, i, j, 0)];
    nxt[q] = inner ? 0.0 : b[gidx<2>(g, i, j, 0)];        // each slot keeps its own boundary values, as in global memory
  }
  constexpr int MAXN = 4;                                  // 63^2 inner nodes / 1024 threads
  double fv[MAXN];
  int lq[MAXN];
#pragma unroll
  for (int e = 0; e < MAXN; ++e) {
    const int q = threadIdx.x + 1024 * e;
    const bool ok = q < total;
    const int i = 1 + (ok ? q % m : 0), j = 1 + (ok ? q / m : 0);
    lq[e] = ok ? j * n + i : -1;
    fv[e] = ok ? f[gidx<2>(g, i, j, 0)] : 0.0;
  }
  __syncthreads();
  for (int s = 0; s < sweeps; ++s) {
#pragma unroll
    for (int e = 0; e < MAXN; ++e) {
      if (lq[e] < 0) continue;
      const int c = lq[e];
      const double u = cur[c];
      double ap = st.c[0] * u;
      ap = ap + st.c[1] * cur[c - 1];
      ap = ap + st.c[2] * cur[c + 1];
      ap = ap + st.c[3] * cur[c - n];
      ap = ap + st.c[4] * cur[c + n];
      const double ex = exp(u);
      ap = ap + (gam * ex) * u;
      const double num = fv[e] - ap;
      const double den = st.c[0] + (gam * (u + 1.0)) * ex;
      nxt[c] = u + om * (num / den);
    }
    __syncthreads();
    double* t = cur; cur = nxt; nxt = t;
  }
  // cur = result of the last sweep, nxt = of the one before; the global-memory form leaves sweep k's output in b for odd k
  double* out_last = (sweeps & 1) ? b : a;
  double* out_prev = (sweeps & 1) ? a : b;
#pragma unroll
  for (int e = 0; e < MAXN; ++e) {
    if (lq[e] < 0) continue;
    const int q = threadIdx.x + 1024 * e;
    const long c = gidx<2>(g, 1 + q % m, 1 + q / m, 0);
    out_last[c] = cur[lq[e]];
    if (sweeps > 0) out_prev[c] = nxt[lq[e]];
  }
}
