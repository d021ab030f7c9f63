// evomg_system.h — systems of PDEs: several unknowns per node coupled by 9-point stencils; the linear
// elasticity problem of the reference (example_problems/LinearElasticity/2D_FD_LinearElasticity_fromL2.*; IR:
// evostencils/ir/system.py:32-135; lowering: code_generation/exastencils.py:769-873).
//
// First implementation: correct and complete for the op vocabulary the grammar produces on such problems, deliberately
// plain in its kernels (thread per node / thread per tile, no fusion) — these problems run on the reference's own small
// levels (4..8); the tuned strip / tile kernels of the scalar path are the model for a later round.
// 2D, real, Dirichlet. Semantics and operation order = oracle/evomg_oracle_system.h (bit-identical pointwise kernels;
// the coarse CG sums in the order that oracle restates).
#pragma once
#include "evomg_kernels.h"

struct SysStencil {
  double a[EVOMG_MAX_UNKNOWNS][EVOMG_MAX_UNKNOWNS][EVOMG_SYS_STENCIL_ENTRIES];
};
struct SysFields {
  double* p[EVOMG_MAX_UNKNOWNS];
};
struct SysConstFields {
  const double* p[EVOMG_MAX_UNKNOWNS];
};

__device__ __forceinline__ int sys_ox(int e) { return (e == 1 || e == 5 || e == 7) ? -1 : (e == 2 || e == 6 || e == 8) ? 1 : 0; }
__device__ __forceinline__ int sys_oy(int e) { return (e == 3 || e == 5 || e == 6) ? -1 : (e == 4 || e == 7 || e == 8) ? 1 : 0; }

// row i of the block operator at inner node (a, b): sum_j sum_e A[i][j][e] x_j(node + offset_e), j then e ascending
__device__ __forceinline__ double sys_apply_row(const Geo& g, const SysStencil& st, int nf, int i, const SysConstFields& x, int a, int b) {
  double s = 0.0;
  for (int j = 0; j < nf; ++j)
#pragma unroll
    for (int e = 0; e < 9; ++e) s = s + st.a[i][j][e] * x.p[j][(long)g.pitch * (b + sys_oy(e)) + a + sys_ox(e)];
  return s;
}

// r_i = f_i - (A x)_i on inner nodes
__global__ void __launch_bounds__(256) evomg_sys_residual(Geo g, SysStencil st, int nf, int i, SysConstFields x,
                                                          const double* __restrict__ f, double* __restrict__ r) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y * blockDim.y + threadIdx.y;
  if (a < 1 || b < 1 || a > g.n - 2 || b > g.n - 2) return;
  const long c = (long)g.pitch * b + a;
  r[c] = f[c] - sys_apply_row(g, st, nf, i, x, a, b);
}
// y_i = (A x)_i on inner nodes (CG)
__global__ void __launch_bounds__(256) evomg_sys_apply(Geo g, SysStencil st, int nf, int i, SysConstFields x, double* __restrict__ y) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y * blockDim.y + threadIdx.y;
  if (a < 1 || b < 1 || a > g.n - 2 || b > g.n - 2) return;
  y[(long)g.pitch * b + a] = sys_apply_row(g, st, nf, i, x, a, b);
}

// One `solve locally` block per thread: the tile anchored at inner node (1 + ti*bx, 1 + tj*by), cut at the far boundary;
// (bx, by) = the componentwise maximum of the listed unknowns' block shapes sh (they may differ per unknown, evomg.h:
// block / block_other): tiles repeat with the bounding box of the statement's unknowns, and a node of the box that no
// listed (unknown, cell) names keeps its value. Unknowns = (unknown of `mask`, cell of ITS shape), field order then
// cells with i0 outermost; every stencil entry that hits a listed unknown goes into the local matrix, the rest to the
// right-hand side at its current value; Gaussian elimination without pivoting, exactly-zero multipliers skipped;
// x <- (1-w) x + w x*. colour >= 0: only tiles whose anchor has that colour (point blocks). Reads x, writes y (the
// caller made y a copy of x).
struct SysShapes {
  int bx[EVOMG_MAX_UNKNOWNS], by[EVOMG_MAX_UNKNOWNS];
};
__global__ void __launch_bounds__(256) evomg_sys_smooth(Geo g, SysStencil st, int nf, int mask, SysShapes sh, int bx, int by, int tx, int ty,
                                                        int colour, SysConstFields x, SysConstFields f, SysFields y, double om) {
  const int ti = blockIdx.x * blockDim.x + threadIdx.x, tj = blockIdx.y * blockDim.y + threadIdx.y;
  if (ti >= tx || tj >= ty) return;
  const int n = g.n;
  const int a0 = 1 + ti * bx, b0 = 1 + tj * by;
  if (colour >= 0 && ((a0 + b0) & 1) != colour) return;
  int eu[8], ea[8], eb[8], m = 0;
  for (int u = 0; u < nf; ++u) {
    if (!((mask >> u) & 1)) continue;
    const int wx = (a0 + sh.bx[u] <= n - 1) ? sh.bx[u] : n - 1 - a0, wy = (b0 + sh.by[u] <= n - 1) ? sh.by[u] : n - 1 - b0;
    for (int ci = 0; ci < wx; ++ci)
      for (int cj = 0; cj < wy; ++cj) { eu[m] = u; ea[m] = a0 + ci; eb[m] = b0 + cj; ++m; }
  }
  double M[8][8], r[8];
  for (int p = 0; p < m; ++p) {
    for (int q = 0; q < m; ++q) M[p][q] = 0.0;
    const int i = eu[p];
    double s = f.p[i][(long)g.pitch * eb[p] + ea[p]];
    for (int j = 0; j < nf; ++j)
      for (int e = 0; e < 9; ++e) {
        const int ta = ea[p] + sys_ox(e), tb = eb[p] + sys_oy(e);
        int hit = -1;
        if ((mask >> j) & 1)
          for (int q = 0; q < m; ++q)
            if (eu[q] == j && ea[q] == ta && eb[q] == tb) hit = q;
        if (hit >= 0) M[p][hit] = st.a[i][j][e];
        else s = s - st.a[i][j][e] * x.p[j][(long)g.pitch * tb + ta];
      }
    r[p] = s;
  }
  for (int p = 0; p < m; ++p)
    for (int q = p + 1; q < m; ++q) {
      if (M[q][p] == 0.0) continue;
      const double fac = M[q][p] / M[p][p];
      for (int t = p + 1; t < m; ++t) M[q][t] = M[q][t] - fac * M[p][t];
      r[q] = r[q] - fac * r[p];
    }
  for (int p = m - 1; p >= 0; --p) {
    double s = r[p];
    for (int t = p + 1; t < m; ++t) s = s - M[p][t] * r[t];
    r[p] = s / M[p][p];
  }
  const double omm = 1.0 - om;
  for (int p = 0; p < m; ++p) {
    const long c = (long)g.pitch * eb[p] + ea[p];
    y.p[eu[p]][c] = omm * x.p[eu[p]][c] + om * r[p];
  }
}

// Point blocks (every listed unknown has the block shape (1, 1): collective / decoupled Jacobi and their red-black forms —
// the smoothers a cycle mostly consists of): one thread per NODE of the whole grid, straight-line code for the 1 x 1 or
// 2 x 2 local system instead of the generic kernel's run-time cell lists and 8 x 8 elimination in scratch memory, and
// every node of y written — the relaxed value, or x where the node is a boundary node / of the other colour — so the
// caller needs no copy of x into y first. Operation for operation the generic kernel (same order of the right-hand-side
// subtractions: unknown j ascending, entry e ascending, centre entries of listed unknowns skipped; same elimination with
// its skipped zero multiplier; same relaxation): bit-identical.
template <int MASK>
__global__ void __launch_bounds__(256) evomg_sys_point_smooth(Geo g, SysStencil st, int nf, int colour, SysConstFields x, SysConstFields f,
                                                              SysFields y, double om) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y * blockDim.y + threadIdx.y;
  const int n = g.n;
  if (a >= n || b >= n) return;
  const long c = (long)g.pitch * b + a;
  const bool inner = a >= 1 && b >= 1 && a <= n - 2 && b <= n - 2;
  if (!inner || (colour >= 0 && ((a + b) & 1) != colour)) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if ((MASK >> u) & 1) y.p[u][c] = x.p[u][c];
    return;
  }
  constexpr int M0 = (MASK & 1) ? 0 : 1;                  // first listed unknown
  constexpr bool TWO = MASK == 3;
  constexpr int NF = 2;                                   // this straight-line form is the two-unknown one (host-checked: nf == 2)
  double rhs[2], xv[NF][9];
  // the nine values of both unknowns around the node (loaded once; the generic kernel reads them entry by entry)
#pragma unroll
  for (int j = 0; j < NF; ++j)
#pragma unroll
    for (int e = 0; e < 9; ++e) xv[j][e] = x.p[j][(long)g.pitch * (b + sys_oy(e)) + a + sys_ox(e)];
#pragma unroll
  for (int p = 0; p < (TWO ? 2 : 1); ++p) {
    const int i = TWO ? p : M0;
    double s = f.p[i][c];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        if (e == 0 && ((MASK >> j) & 1)) continue;        // a listed unknown: goes into the local matrix
        s = s - st.a[i][j][e] * xv[j][e];
      }
    rhs[p] = s;
  }
  const double omm = 1.0 - om;
  if (TWO) {
    const double m00 = st.a[0][0][0], m01 = st.a[0][1][0], m10 = st.a[1][0][0];
    double m11 = st.a[1][1][0];
    double r0 = rhs[0], r1 = rhs[1];
    if (m10 != 0.0) {
      const double fac = m10 / m00;
      m11 = m11 - fac * m01;
      r1 = r1 - fac * r0;
    }
    r1 = r1 / m11;
    double s0 = r0;
    s0 = s0 - m01 * r1;
    r0 = s0 / m00;
    y.p[0][c] = omm * xv[0][0] + om * r0;
    y.p[1][c] = omm * xv[1][0] + om * r1;
  } else {
    const double r0 = rhs[0] / st.a[M0][M0][0];
    y.p[M0][c] = omm * xv[M0][0] + om * r0;
  }
}

// Coarse-grid CG on the stacked unknowns: ONE 1024-thread workgroup, vectors in global memory (L2-resident on the small
// coarsest grids these problems have), thread t owns the stacked inner nodes t, t + 1024, ... (unknown-major, rows,
// columns); dots: per-thread fma chain in that order, then the block_sum trees. Restated by the oracle (sys_dot).
__global__ void __launch_bounds__(1024) evomg_sys_cg(Geo g, SysStencil st, int nf, SysFields x, SysConstFields b, SysFields r,
                                                     SysFields p, SysFields ap, int maxit, double tol, int* __restrict__ info) {
  __shared__ double red[17];
  const int n = g.n, m = n - 2;
  const long per = (long)m * m, total = per * nf;
  const int t = threadIdx.x;
  auto node = [&](long q, int& u, long& c, int& a, int& bb) {
    u = (int)(q / per);
    const long k = q % per;
    a = 1 + (int)(k % m);
    bb = 1 + (int)(k / m);
    c = (long)g.pitch * bb + a;
  };
  SysConstFields xc, pc;
  for (int u = 0; u < EVOMG_MAX_UNKNOWNS; ++u) { xc.p[u] = x.p[u]; pc.p[u] = p.p[u]; }
  double loc = 0.0;
  for (long q = t; q < total; q += 1024) {
    int u, a, bb; long c;
    node(q, u, c, a, bb);
    const double v = b.p[u][c] - sys_apply_row(g, st, nf, u, xc, a, bb);
    r.p[u][c] = v;
    p.p[u][c] = v;
    loc = __builtin_fma(v, v, loc);
  }
  double rr = block_sum(loc, red);
  const double res0 = sqrt(rr);
  int it = 0;
  if (res0 != 0.0) {
    for (; it < maxit; ++it) {
      __syncthreads();                       // p complete (written by other threads' updates)
      loc = 0.0;
      for (long q = t; q < total; q += 1024) {
        int u, a, bb; long c;
        node(q, u, c, a, bb);
        const double v = sys_apply_row(g, st, nf, u, pc, a, bb);
        ap.p[u][c] = v;
        loc = __builtin_fma(p.p[u][c], v, loc);
      }
      const double alpha = rr / block_sum(loc, red);
      loc = 0.0;
      for (long q = t; q < total; q += 1024) {
        int u, a, bb; long c;
        node(q, u, c, a, bb);
        x.p[u][c] = __builtin_fma(alpha, p.p[u][c], x.p[u][c]);
        const double v = __builtin_fma(-alpha, ap.p[u][c], r.p[u][c]);
        r.p[u][c] = v;
        loc = __builtin_fma(v, v, loc);
      }
      const double rrn = block_sum(loc, red);
      if (!(sqrt(rrn) > tol * res0)) { ++it; break; }
      const double beta = rrn / rr;
      for (long q = t; q < total; q += 1024) {
        int u, a, bb; long c;
        node(q, u, c, a, bb);
        p.p[u][c] = __builtin_fma(beta, p.p[u][c], r.p[u][c]);
      }
      rr = rrn;
    }
  }
  if (t == 0) info[0] = it;
}

// The same solve for coarsest grids with at most 512 stacked unknowns (17^2 nodes x 2 unknowns: the reference's level range
// 4..8 ends there): every thread of ONE 512-thread workgroup owns the stacked node t (exactly evomg_sys_cg's ownership when
// there are fewer unknowns than threads) and keeps its x, r, p and A p in registers; only p lives in an LDS image (zero
// boundary) for the neighbours' 9-point reads; one barrier per reduction. Arithmetic, ownership and summation order are
// evomg_sys_cg's (the extra waves of the 1024-thread form contribute exact zeros): bit-identical to it and to the oracle.
// dynamic LDS: nf * n * n doubles.
__global__ void __launch_bounds__(512) evomg_sys_cg_small(Geo g, SysStencil st, int nf, SysFields x, SysConstFields b, SysFields r,
                                                          int maxit, double tol, int* __restrict__ info) {
  extern __shared__ __align__(16) double sys_img[];
  __shared__ double red[32];
  const int n = g.n, m = n - 2, nn = n * n;
  const int per = m * m, total = per * nf;
  const int t = threadIdx.x;
  const bool own = t < total;
  const int u = own ? t / per : 0, k = own ? t % per : 0;
  const int a = 1 + k % m, bb = 1 + k / m;
  const long c = (long)g.pitch * bb + a;
  const int li = u * nn + bb * n + a;
  // the thread's matrix row in registers: indexed by the thread's unknown, the by-value stencil would otherwise be re-read
  // from the kernel-argument segment with eighteen global loads in every iteration (2.0 -> 1.x us per iteration)
  double cf[EVOMG_MAX_UNKNOWNS][9];
#pragma unroll
  for (int j = 0; j < EVOMG_MAX_UNKNOWNS; ++j)
#pragma unroll
    for (int e = 0; e < 9; ++e) cf[j][e] = j < nf ? st.a[u][j][e] : 0.0;
  auto apply_row = [&](int) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < EVOMG_MAX_UNKNOWNS; ++j)
      if (j < nf) {
#pragma unroll
        for (int e = 0; e < 9; ++e) s = s + cf[j][e] * sys_img[j * nn + (bb + sys_oy(e)) * n + a + sys_ox(e)];
      }
    return s;
  };
  // image of x (all nodes, boundary included) for the initial residual
  for (int q = t; q < nf * nn; q += 512) {
    const int uu = q / nn, rem = q % nn;
    sys_img[q] = x.p[uu][(long)g.pitch * (rem / n) + rem % n];
  }
  __syncthreads();
  double xv = own ? sys_img[li] : 0.0;
  double rv = own ? b.p[u][c] - apply_row(u) : 0.0;
  double pv = rv, apv = 0.0;
  int phase = 0;
  double rr = block_sum_fast(__builtin_fma(rv, rv, 0.0), red, phase);     // its barrier: everybody has read the x image
  const double res0 = sqrt(rr);
  for (int q = t; q < nf * nn; q += 512) sys_img[q] = 0.0;               // from here on: the image of p (zero boundary)
  int it = 0;
  if (res0 != 0.0) {
    __syncthreads();
    for (; it < maxit; ++it) {
      if (own) sys_img[li] = pv;
      __syncthreads();                       // p complete
      apv = own ? apply_row(u) : 0.0;
      const double alpha = rr / block_sum_fast(__builtin_fma(pv, apv, 0.0), red, phase);   // its barrier: all reads of p done
      xv = __builtin_fma(alpha, pv, xv);
      rv = __builtin_fma(-alpha, apv, rv);
      const double rrn = block_sum_fast(__builtin_fma(rv, rv, 0.0), red, phase);
      if (!(sqrt(rrn) > tol * res0)) { ++it; break; }
      const double beta = rrn / rr;
      pv = __builtin_fma(beta, pv, rv);
      rr = rrn;
    }
  }
  if (own) {
    x.p[u][c] = xv;
    r.p[u][c] = rv;
  }
  if (t == 0) info[0] = it;
}

// ---- the same statement for both unknowns in ONE launch (blockIdx.z = unknown) ---------------------------------------------
// A cycle on a system emits its residual / restriction / prolongation / zero statements once per unknown, back to back and
// independent of each other; on the small levels these problems live on every launch is ~3.5 us of dispatch for well under
// a microsecond of work. Bodies = the per-unknown kernels' (evomg_sys_residual, evomg_restrict<double, 2>,
// evomg_prolong<double, 2, ACC>, evomg_zero_inner<double, 2>): bit-identical.
__global__ void __launch_bounds__(256) evomg_sys_residual2(Geo g, SysStencil st, int nf, SysConstFields x, SysConstFields f, SysFields r) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y * blockDim.y + threadIdx.y, i = blockIdx.z;
  if (a < 1 || b < 1 || a > g.n - 2 || b > g.n - 2) return;
  const long c = (long)g.pitch * b + a;
  r.p[i][c] = f.p[i][c] - sys_apply_row(g, st, nf, i, x, a, b);
}
__global__ void __launch_bounds__(256) evomg_sys_restrict2(Geo gc, Geo gf, SysConstFields fine_u, SysFields coarse_u) {
  const int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y * blockDim.y + threadIdx.y;
  if (I < 1 || J < 1 || I > gc.n - 2 || J > gc.n - 2) return;
  const double* __restrict__ fine = fine_u.p[blockIdx.z];
  const long c = gidx<2>(gf, 2 * I, 2 * J, 0);
  const long p = gf.pitch;
  const double e = sc_add(sc_add(fine[c - 1], fine[c + 1]), sc_add(fine[c - p], fine[c + p]));
  const double q = sc_add(sc_add(fine[c - p - 1], fine[c - p + 1]), sc_add(fine[c + p - 1], fine[c + p + 1]));
  double v = sc_rmul(0.25, fine[c]);
  v = sc_add(v, sc_rmul(0.125, e));
  v = sc_add(v, sc_rmul(0.0625, q));
  coarse_u.p[blockIdx.z][gidx<2>(gc, I, J, 0)] = v;
}
template <bool ACC>
__global__ void __launch_bounds__(256) evomg_sys_prolong2(Geo gf, Geo gc, SysConstFields coarse_u, SysFields fine_u, double om) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i < 1 || j < 1 || i > gf.n - 2 || j > gf.n - 2) return;
  const long c = gidx<2>(gf, i, j, 0);
  double* __restrict__ fine = fine_u.p[blockIdx.z];
  const double v = interpolate<double, 2>(gc, coarse_u.p[blockIdx.z], i, j, 0);
  fine[c] = ACC ? sc_add(fine[c], sc_rmul(om, v)) : v;
}
__global__ void __launch_bounds__(256) evomg_sys_zero2(Geo g, SysFields x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i < 1 || j < 1 || i > g.n - 2 || j > g.n - 2) return;
  x.p[blockIdx.z][gidx<2>(g, i, j, 0)] = 0.0;
}
