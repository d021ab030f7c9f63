/* evomg_cpu_tuned.c — a TUNED CPU solver for the headline workload (TEST / BENCH INFRASTRUCTURE, like everything under oracle/).
 *
 * What bench.py's `cpu_baseline.tuned` times: the same algorithm as the headline candidate — 2D Poisson 5-point, V(pre, post)
 * red-black Gauss-Seidel cycle over levels lo..hi, CG on the coarsest grid (at most cgs_maxit iterations, reduction cgs_tol:
 * example_problems/Poisson/2D_FD_Poisson_fromL2.exa3:12-14), the generated solver loop (repeat the cycle until the residual
 * norm has dropped by tol or maxit: exa3:2-4) — written the way a CPU code generator writes it: OpenMP over rows, every loop
 * in its natural order, plain `reduction(+)` dot products, fused multiply-adds left to the compiler (-O3 -march=native).
 * That is what ExaStencils' generated C++ is for the reference (omp_numThreads in
 * example_problems/lib/parallelization_pureOmp.knowledge). It is NOT bit-comparable with the HIP backend or the parity oracle
 * (evomg_oracle.c sums in the device's tree order and keeps every operation unfused): same iteration count, residual history
 * equal to rounding (tests/test_cpu_tuned.py), and a wall time that scales with the host's cores — the honest CPU number to put
 * beside a GPU one.
 *
 * Restates (file:line of the reference): the statements generate_multigrid emits for a V-cycle (exastencils.py:684-925): smoother
 * `solve locally ... relax w` with red-black colouring (:778-822), residual (:837-853), full weighting (:868-873), bilinear
 * prolongation and correction (:742-743), zero initial coarse error (:895).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef struct {
  int n;
  double inv_h2;
  double *u, *f, *r;
} level_t;

static inline size_t at(int n, int i, int j) { return (size_t)i + (size_t)n * (size_t)j; }

/* one colour of a red-black sweep, in place: u <- (1 - w) u + w (f + (uw + ue + us + un) / h^2) / (4 / h^2) */
static void sweep_colour(level_t* L, int colour, double w) {
  const int n = L->n;
  const double o = L->inv_h2, c0 = 4.0 * o, wi = w / c0, wm = 1.0 - w;
  double* u = L->u;
  const double* f = L->f;
#pragma omp parallel for schedule(static) if (n >= 257)
  for (int j = 1; j < n - 1; ++j) {
    double* row = u + (size_t)n * j;
    const double *south = row - n, *north = row + n, *fr = f + (size_t)n * j;
    for (int i = 1 + ((j + 1 + colour) & 1); i < n - 1; i += 2)
      row[i] = wm * row[i] + wi * (fr[i] + o * (row[i - 1] + row[i + 1] + south[i] + north[i]));
  }
}
static void smooth(level_t* L, double w) {
  sweep_colour(L, 0, w);      /* colour 0 = even i0 + i1 first (the convention the recorded run pins, oracle/APPB_MATRIX.md) */
  sweep_colour(L, 1, w);
}
static void residual(level_t* L) {
  const int n = L->n;
  const double o = L->inv_h2, c0 = 4.0 * o;
#pragma omp parallel for schedule(static)
  for (int j = 1; j < n - 1; ++j)
    for (int i = 1; i < n - 1; ++i) {
      const size_t q = at(n, i, j);
      L->r[q] = L->f[q] - (c0 * L->u[q] - o * (L->u[q - 1] + L->u[q + 1] + L->u[q - n] + L->u[q + n]));
    }
}
static double norm_residual(level_t* L) {
  const int n = L->n;
  const double o = L->inv_h2, c0 = 4.0 * o;
  double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
  for (int j = 1; j < n - 1; ++j)
    for (int i = 1; i < n - 1; ++i) {
      const size_t q = at(n, i, j);
      const double v = L->f[q] - (c0 * L->u[q] - o * (L->u[q - 1] + L->u[q + 1] + L->u[q - n] + L->u[q + n]));
      s += v * v;
    }
  return sqrt(s);
}
/* coarse f = full weighting of the fine residual; coarse u = 0 */
static void restrict_fw(level_t* F, level_t* C) {
  const int nc = C->n, nf = F->n;
  const double* r = F->r;
#pragma omp parallel for schedule(static)
  for (int J = 1; J < nc - 1; ++J)
    for (int I = 1; I < nc - 1; ++I) {
      const size_t q = at(nf, 2 * I, 2 * J);
      C->f[at(nc, I, J)] = 0.25 * r[q] + 0.125 * (r[q - 1] + r[q + 1] + r[q - nf] + r[q + nf]) +
                           0.0625 * (r[q - nf - 1] + r[q - nf + 1] + r[q + nf - 1] + r[q + nf + 1]);
    }
  memset(C->u, 0, sizeof(double) * (size_t)nc * nc);
}
/* fine u += w * bilinear interpolation of the coarse u */
static void prolong_add(level_t* C, level_t* F, double w) {
  const int nc = C->n, nf = F->n;
  const double* c = C->u;
#pragma omp parallel for schedule(static)
  for (int j = 1; j < nf - 1; ++j) {
    const int J = j >> 1;
    for (int i = 1; i < nf - 1; ++i) {
      const int I = i >> 1;
      double v;
      if (!(j & 1)) v = (i & 1) ? 0.5 * (c[at(nc, I, J)] + c[at(nc, I + 1, J)]) : c[at(nc, I, J)];
      else if (!(i & 1)) v = 0.5 * (c[at(nc, I, J)] + c[at(nc, I, J + 1)]);
      else v = 0.25 * (c[at(nc, I, J)] + c[at(nc, I + 1, J)] + c[at(nc, I, J + 1)] + c[at(nc, I + 1, J + 1)]);
      F->u[at(nf, i, j)] += w * v;
    }
  }
}
/* classical CG on the coarsest grid, x0 = 0: vectors p, ap in the work arrays; returns the iterations */
static int coarse_cg(level_t* L, double* p, double* ap, int maxit, double tol) {
  const int n = L->n;
  const double o = L->inv_h2, c0 = 4.0 * o;
  const size_t N = (size_t)n * n;
  double* x = L->u;
  double* r = L->r;
  memset(x, 0, sizeof(double) * N);
  memset(p, 0, sizeof(double) * N);
  double rr = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : rr) if (n >= 257)
  for (int j = 1; j < n - 1; ++j)
    for (int i = 1; i < n - 1; ++i) {
      const size_t q = at(n, i, j);
      r[q] = L->f[q];
      p[q] = r[q];
      rr += r[q] * r[q];
    }
  const double res0 = sqrt(rr);
  if (res0 == 0.0) return 0;
  int it = 0;
  while (it < maxit) {
    double pap = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : pap) if (n >= 257)
    for (int j = 1; j < n - 1; ++j)
      for (int i = 1; i < n - 1; ++i) {
        const size_t q = at(n, i, j);
        const double v = c0 * p[q] - o * (p[q - 1] + p[q + 1] + p[q - n] + p[q + n]);
        ap[q] = v;
        pap += p[q] * v;
      }
    const double alpha = rr / pap;
    double rrn = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : rrn) if (n >= 257)
    for (int j = 1; j < n - 1; ++j)
      for (int i = 1; i < n - 1; ++i) {
        const size_t q = at(n, i, j);
        x[q] += alpha * p[q];
        r[q] -= alpha * ap[q];
        rrn += r[q] * r[q];
      }
    ++it;
    if (!(sqrt(rrn) > tol * res0)) break;
    const double beta = rrn / rr;
#pragma omp parallel for schedule(static) if (n >= 257)
    for (int j = 1; j < n - 1; ++j)
      for (int i = 1; i < n - 1; ++i) {
        const size_t q = at(n, i, j);
        p[q] = r[q] + beta * p[q];
      }
    rr = rrn;
  }
  return it;
}

/* u_fine: initial guess incl. the Dirichlet boundary values, overwritten with the solution; f_fine: right-hand side (both
 * (2^hi + 1)^2 doubles, i0 fastest). history: residual norms (initial one first), capacity `cap`. Returns the number of cycles
 * run, or -1 on a bad argument / out of memory. */
int evomg_cpu_tuned_poisson2d(int lo, int hi, double* u_fine, const double* f_fine, int pre, int post, double omega, double tol,
                              int maxit, int cgs_maxit, double cgs_tol, double* history, int cap, long long* coarse_iterations) {
  if (lo < 1 || hi <= lo || hi > 14 || !u_fine || !f_fine) return -1;
  level_t lv[16];
  memset(lv, 0, sizeof lv);
  for (int L = lo; L <= hi; ++L) {
    const int n = (1 << L) + 1;
    const size_t N = (size_t)n * n;
    lv[L].n = n;
    lv[L].inv_h2 = (double)(1 << L) * (double)(1 << L);
    lv[L].u = (L == hi) ? u_fine : (double*)calloc(N, sizeof(double));
    lv[L].f = (L == hi) ? (double*)f_fine : (double*)calloc(N, sizeof(double));
    lv[L].r = (double*)calloc(N, sizeof(double));
    if (!lv[L].u || !lv[L].f || !lv[L].r) return -1;
  }
  const size_t Nc = (size_t)lv[lo].n * lv[lo].n;
  double* p = (double*)calloc(Nc, sizeof(double));
  double* ap = (double*)calloc(Nc, sizeof(double));
  long long k_total = 0;
  double res0 = norm_residual(&lv[hi]);
  int n_hist = 0;
  if (history && cap > 0) history[n_hist++] = res0;
  int it = 0;
  double res = res0;
  while (it < maxit && res > tol * res0) {
    for (int L = hi; L > lo; --L) {
      for (int s = 0; s < pre; ++s) smooth(&lv[L], omega);
      residual(&lv[L]);
      restrict_fw(&lv[L], &lv[L - 1]);
    }
    k_total += coarse_cg(&lv[lo], p, ap, cgs_maxit, cgs_tol);
    for (int L = lo + 1; L <= hi; ++L) {
      prolong_add(&lv[L - 1], &lv[L], 1.0);
      for (int s = 0; s < post; ++s) smooth(&lv[L], omega);
    }
    res = norm_residual(&lv[hi]);
    ++it;
    if (history && n_hist < cap) history[n_hist++] = res;
    if (!(res == res)) break;
  }
  for (int L = lo; L <= hi; ++L) {
    if (L != hi) { free(lv[L].u); free(lv[L].f); }
    free(lv[L].r);
  }
  free(p);
  free(ap);
  if (coarse_iterations) *coarse_iterations = k_total;
  return it;
}
