// evomg_kernels.h — HIP kernels for gfx950 (wave64), one family per ExaSlang statement kind
// (SURVEY.md 2.3, K1..K9). FP64 / complex-FP64 5- and 7-point stencils: bandwidth-bound, no MFMA.
//
// Data layout in HBM: one array per (field, level); node (i0,i1,i2) at i0 + pitch*i1 + plane*i2 with
// pitch = n rounded up to a 128-byte multiple (rows start cache-line aligned), plane = pitch*n.
// Boundary nodes are part of the array (index 0 and n-1 per dimension).
#pragma once
#include "evomg_scalar.h"

struct Geo {
  int n;          // nodes per dimension
  int pitch;      // elements per row
  long plane;     // elements per z-plane (pitch * n), 0 for 2D
  int prio;       // != 0: launches on this level sit on a candidate's critical path without filling the device (prio_level)
};

// ---- issue priority (round 6) --------------------------------------------------------------------------------------------
// With candidates in flight the waves of one candidate's latency-bound launches (coarse-grid solves on eight units, final
// reductions, one-workgroup levels, the levels below the finest) share SIMDs with the streaming waves of other candidates'
// finest-level sweeps: at equal priority the arbiter alternates and a 0.5 ms coarse solve takes 1.3 ms, a 3 us reduction 86
// (profiles/r05_throughput.md). s_setprio raises the issuing wave's priority for the rest of its life: the latency-bound
// waves win the issue slot whenever they are ready, the streaming waves take the gaps (there are plenty: a coarse-solve
// wave waits for its exchange most of an iteration). Alone on the device nothing changes. 0 = off (developer builds).
#ifndef EVOMG_PRIO_LATENCY
#define EVOMG_PRIO_LATENCY 3     // coarse solves, reductions, one-workgroup launches
#endif
#ifndef EVOMG_PRIO_LEVEL
#define EVOMG_PRIO_LEVEL 2       // streaming kernels on levels below the handle's finest (Geo::prio)
#endif
__device__ __forceinline__ void prio_latency() {
  if constexpr (EVOMG_PRIO_LATENCY > 0) __builtin_amdgcn_s_setprio(EVOMG_PRIO_LATENCY);
}
__device__ __forceinline__ void prio_level(const Geo& g) {
  if constexpr (EVOMG_PRIO_LEVEL > 0) {
    if (g.prio) __builtin_amdgcn_s_setprio(EVOMG_PRIO_LEVEL);       // kernel argument: a scalar branch
  }
}

template <int DIM> __device__ __forceinline__ long gidx(const Geo& g, int i, int j, int k) {
  if (DIM == 2) return (long)i + (long)g.pitch * j;
  return (long)i + (long)g.pitch * j + g.plane * k;
}

// thread -> node mapping of the generic kernels: block (64,4,1); grid (ceil(n/64), ceil(n/4), DIM==3 ? n : 1)
template <int DIM> __device__ __forceinline__ bool node_of_thread(const Geo& g, int& i, int& j, int& k) {
  i = blockIdx.x * blockDim.x + threadIdx.x;
  j = blockIdx.y * blockDim.y + threadIdx.y;
  k = (DIM == 3) ? (int)blockIdx.z : 0;
  return i < g.n && j < g.n && (DIM == 2 || k < g.n);
}
template <int DIM> __device__ __forceinline__ bool is_inner(const Geo& g, int i, int j, int k) {
  const int m = g.n - 1;
  return i > 0 && i < m && j > 0 && j < m && (DIM == 2 || (k > 0 && k < m));
}

// ---- pointwise formulas: identical operation order to oracle FN(apply_st2/3), FN(relax2/3) ---------------
template <typename T, int DIM>
__device__ __forceinline__ T apply_stencil(const Stencil<T>& st, const T* __restrict__ x, const Geo& g, long c) {
  T a = sc_mul(st.c[0], x[c]);
  a = sc_add(a, sc_mul(st.c[1], x[c - 1]));
  a = sc_add(a, sc_mul(st.c[2], x[c + 1]));
  a = sc_add(a, sc_mul(st.c[3], x[c - g.pitch]));
  a = sc_add(a, sc_mul(st.c[4], x[c + g.pitch]));
  if (DIM == 3) {
    a = sc_add(a, sc_mul(st.c[5], x[c - g.plane]));
    a = sc_add(a, sc_mul(st.c[6], x[c + g.plane]));
  }
  return a;
}
template <typename T, int DIM>
__device__ __forceinline__ T relax_point(const Stencil<T>& st, const T* __restrict__ x, const T* __restrict__ f,
                                         const Geo& g, long c, double om, double omm) {
  T s = f[c];
  s = sc_sub(s, sc_mul(st.c[1], x[c - 1]));
  s = sc_sub(s, sc_mul(st.c[2], x[c + 1]));
  s = sc_sub(s, sc_mul(st.c[3], x[c - g.pitch]));
  s = sc_sub(s, sc_mul(st.c[4], x[c + g.pitch]));
  if (DIM == 3) {
    s = sc_sub(s, sc_mul(st.c[5], x[c - g.plane]));
    s = sc_sub(s, sc_mul(st.c[6], x[c + g.plane]));
  }
  const T xs = sc_div(s, st.c[0]);
  return sc_add(sc_rmul(omm, x[c]), sc_rmul(om, xs));
}

// ---- K1: damped Jacobi, out of place (reads x, f; writes y incl. a copy of the boundary) -------------------
template <typename T, int DIM>
__global__ void __launch_bounds__(256) evomg_jacobi(Geo g, Stencil<T> st, const T* __restrict__ x,
                                                    const T* __restrict__ f, T* __restrict__ y, double om, double omm) {
  int i, j, k;
  if (!node_of_thread<DIM>(g, i, j, k)) return;
  const long c = gidx<DIM>(g, i, j, k);
  y[c] = is_inner<DIM>(g, i, j, k) ? relax_point<T, DIM>(st, x, f, g, c, om, omm) : x[c];
}

// ---- K2 (two-pass form): one colour of red-black Gauss-Seidel, in place ------------------------------------
// thread tx handles node i = 2*tx + offset so that all threads of the launch work on the requested colour.
template <typename T, int DIM>
__global__ void __launch_bounds__(256) evomg_rbgs_colour(Geo g, Stencil<T> st, T* __restrict__ x,
                                                         const T* __restrict__ f, double om, double omm, int colour) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = (DIM == 3) ? (int)blockIdx.z : 0;
  const int i = 2 * t + ((j + k + colour) & 1);
  if (!is_inner<DIM>(g, i, j, k)) return;
  const long c = gidx<DIM>(g, i, j, k);
  x[c] = relax_point<T, DIM>(st, x, f, g, c, om, omm);
}

// ---- K3: collective block-Jacobi, one thread per tile (tiles anchored at the first inner node) -----------
template <typename T, int DIM>
__global__ void __launch_bounds__(256) evomg_block_jacobi(Geo g, Stencil<T> st, const T* __restrict__ x,
                                                          const T* __restrict__ f, T* __restrict__ y, double om,
                                                          double omm, int bx, int by, int bz, int tx, int ty, int tz,
                                                          int only_truncated, int pinned_dim) {
  // pinned_dim = d >= 0: the launch covers only the last layer of tiles in dimension d (the tiles that can be cut
  // by the far boundary); the thread grid then enumerates the other two tile coordinates.
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  int tj = blockIdx.y * blockDim.y + threadIdx.y;
  int tk = (DIM == 3) ? (int)blockIdx.z : 0;
  if (pinned_dim == 0) { const int a = ti, b = tj; ti = tx - 1; tj = a; tk = (DIM == 3) ? b : 0; }
  else if (pinned_dim == 1) { const int a = ti, b = tj; ti = a; tj = ty - 1; tk = (DIM == 3) ? b : 0; }
  else if (pinned_dim == 2) { tk = tz - 1; }
  if (ti >= tx || tj >= ty || tk >= tz) return;
  const int n = g.n;
  const int i0 = 1 + ti * bx, j0 = 1 + tj * by, k0 = (DIM == 3) ? 1 + tk * bz : 0;
  const int mx = (i0 + bx <= n - 1) ? bx : (n - 1 - i0);
  const int my = (j0 + by <= n - 1) ? by : (n - 1 - j0);
  const int mz = (DIM == 3) ? ((k0 + bz <= n - 1) ? bz : (n - 1 - k0)) : 1;
  const int m = mx * my * mz;
  // full tiles are done by evomg_block_jacobi_fast when the runtime pairs the two kernels
  if (only_truncated && m == bx * by * bz) return;
  T A[8][8];
  T r[8];
  for (int p = 0; p < m; ++p)
    for (int q = 0; q < m; ++q) A[p][q] = scalar_traits<T>::zero();
  const int nnb = 1 + 2 * DIM;
  for (int ci = 0; ci < mx; ++ci)
    for (int cj = 0; cj < my; ++cj)
      for (int ck = 0; ck < mz; ++ck) {
        const int p = (ci * my + cj) * mz + ck;
        const long gc = gidx<DIM>(g, i0 + ci, j0 + cj, k0 + ck);
        T s = f[gc];
        A[p][p] = st.c[0];
        for (int e = 1; e < nnb; ++e) {
          const int ox = (e == 1) ? -1 : (e == 2) ? 1 : 0;
          const int oy = (e == 3) ? -1 : (e == 4) ? 1 : 0;
          const int oz = (e == 5) ? -1 : (e == 6) ? 1 : 0;
          const int di = ci + ox, dj = cj + oy, dk = ck + oz;
          if (di >= 0 && di < mx && dj >= 0 && dj < my && dk >= 0 && dk < mz) {
            A[p][(di * my + dj) * mz + dk] = st.c[e];
          } else {
            const long gn = gc + ox + (long)g.pitch * oy + g.plane * oz;
            s = sc_sub(s, sc_mul(st.c[e], x[gn]));
          }
        }
        r[p] = s;
      }
  for (int p = 0; p < m; ++p)
    for (int q = p + 1; q < m; ++q) {
      if (sc_iszero(A[q][p])) continue;
      const T fac = sc_div(A[q][p], A[p][p]);
      for (int t = p + 1; t < m; ++t) A[q][t] = sc_sub(A[q][t], sc_mul(fac, A[p][t]));
      r[q] = sc_sub(r[q], sc_mul(fac, r[p]));
    }
  for (int p = m - 1; p >= 0; --p) {
    T s = r[p];
    for (int t = p + 1; t < m; ++t) s = sc_sub(s, sc_mul(A[p][t], r[t]));
    r[p] = sc_div(s, A[p][p]);
  }
  for (int ci = 0; ci < mx; ++ci)
    for (int cj = 0; cj < my; ++cj)
      for (int ck = 0; ck < mz; ++ck) {
        const int p = (ci * my + cj) * mz + ck;
        const long gc = gidx<DIM>(g, i0 + ci, j0 + cj, k0 + ck);
        y[gc] = sc_add(sc_rmul(omm, x[gc]), sc_rmul(om, r[p]));
      }
}

// ---- K3, fast path: full tiles with a tile factorisation computed once on the host ---------------------------------
// Every full tile has the same local matrix (constant-coefficient stencil restricted to the block), so the forward
// elimination of A is identical for all tiles: the host runs the oracle's elimination once (same operation order,
// same skip-if-zero rule; host and device doubles are both IEEE, no contraction) and the kernel only replays it on
// the tile's right-hand side. One thread per tile, tiles x-fastest over lanes; everything indexed by compile-time
// cell numbers after unrolling, so plan entries are scalar kernel-argument loads and r[] stays in registers.
template <typename T> struct BlockPlan {
  int m, bx, by, bz;
  int ci[8], cj[8], ck[8];   // cell coordinates inside the tile (unknown order: i0 outermost)
  int inside[8][7];          // inside[p][e]: neighbour e of cell p is an unknown of the tile
  int skip[8][8];            // forward step (q, p) skipped because A[q][p] was exactly zero
  T fac[8][8];               // elimination factors A[q][p] / A[p][p]
  T upper[8][8];             // eliminated matrix, strictly upper part
  T pivot[8];
};
template <typename T, int DIM> inline BlockPlan<T> make_block_plan(const Stencil<T>& st, int bx, int by, int bz) {
  BlockPlan<T> pl{};
  pl.bx = bx; pl.by = by; pl.bz = (DIM == 3) ? bz : 1;
  const int mx = bx, my = by, mz = pl.bz;
  const int m = pl.m = mx * my * mz;
  T A[8][8];
  for (int p = 0; p < 8; ++p)
    for (int q = 0; q < 8; ++q) A[p][q] = scalar_traits<T>::zero();
  const int nnb = 1 + 2 * DIM;
  for (int ci = 0; ci < mx; ++ci)
    for (int cj = 0; cj < my; ++cj)
      for (int ck = 0; ck < mz; ++ck) {
        const int p = (ci * my + cj) * mz + ck;
        pl.ci[p] = ci; pl.cj[p] = cj; pl.ck[p] = ck;
        A[p][p] = st.c[0];
        for (int e = 1; e < nnb; ++e) {
          const int ox = (e == 1) ? -1 : (e == 2) ? 1 : 0;
          const int oy = (e == 3) ? -1 : (e == 4) ? 1 : 0;
          const int oz = (e == 5) ? -1 : (e == 6) ? 1 : 0;
          const int di = ci + ox, dj = cj + oy, dk = ck + oz;
          const bool in = di >= 0 && di < mx && dj >= 0 && dj < my && dk >= 0 && dk < mz;
          pl.inside[p][e] = in ? 1 : 0;
          if (in) A[p][(di * my + dj) * mz + dk] = st.c[e];
        }
      }
  for (int p = 0; p < m; ++p)
    for (int q = p + 1; q < m; ++q) {
      pl.skip[q][p] = sc_iszero(A[q][p]) ? 1 : 0;
      if (pl.skip[q][p]) continue;
      const T fac = sc_div(A[q][p], A[p][p]);
      pl.fac[q][p] = fac;
      for (int t = p + 1; t < m; ++t) A[q][t] = sc_sub(A[q][t], sc_mul(fac, A[p][t]));
    }
  for (int p = 0; p < m; ++p) {
    pl.pivot[p] = A[p][p];
    for (int t = p + 1; t < m; ++t) pl.upper[p][t] = A[p][t];
  }
  return pl;
}
// BX, BY, BZ > 0: the tile shape is a compile-time constant (the common shapes), so cell offsets and the
// inside-the-tile tests fold away and the compiler merges the x-adjacent cells' accesses into 16-byte loads/stores;
// BX == 0: shape taken from the plan at run time (any other shape with <= 8 cells).
template <typename T, int DIM, int M, int BX = 0, int BY = 0, int BZ = 0>
__global__ void __launch_bounds__(256) evomg_block_jacobi_fast(Geo g, Stencil<T> st, BlockPlan<T> pl, const T* __restrict__ x,
                                                               const T* __restrict__ f, T* __restrict__ y, double om,
                                                               double omm, int tx, int ty, int tz) {
  const int ti = blockIdx.x * blockDim.x + threadIdx.x;
  const int tj = blockIdx.y * blockDim.y + threadIdx.y;
  const int tk = (DIM == 3) ? (int)blockIdx.z : 0;
  if (ti >= tx || tj >= ty || tk >= tz) return;
  const int n = g.n;
  constexpr bool kStatic = BX > 0;
  const int bx = kStatic ? BX : pl.bx, by = kStatic ? BY : pl.by, bz = kStatic ? BZ : pl.bz;
  const int i0 = 1 + ti * bx, j0 = 1 + tj * by, k0 = (DIM == 3) ? 1 + tk * bz : 0;
  const long base = gidx<DIM>(g, i0, j0, k0);
  if (i0 + bx > n - 1 || j0 + by > n - 1 || (DIM == 3 && k0 + bz > n - 1)) {
    // Tile cut by the far boundary. With a static two-cell shape what is left is a single node and the local system
    // a point relaxation (same operation order as the generic kernel's one-unknown elimination = relax_point);
    // every other case is left to the generic kernel, launched over the last tile layer.
    if constexpr (kStatic && BX * BY * BZ == 2) y[base] = relax_point<T, DIM>(st, x, f, g, base, om, omm);
    return;
  }
  T r[M];
  long gc[M];
#pragma unroll
  for (int p = 0; p < M; ++p) {
    // unknown order: i0 outermost (make_block_plan)
    const int ci = kStatic ? p / (BY * BZ > 0 ? BY * BZ : 1) : pl.ci[p];
    const int cj = kStatic ? (p / (BZ > 0 ? BZ : 1)) % (BY > 0 ? BY : 1) : pl.cj[p];
    const int ck = kStatic ? p % (BZ > 0 ? BZ : 1) : pl.ck[p];
    const bool in1 = kStatic ? ci - 1 >= 0 : pl.inside[p][1];
    const bool in2 = kStatic ? ci + 1 < BX : pl.inside[p][2];
    const bool in3 = kStatic ? cj - 1 >= 0 : pl.inside[p][3];
    const bool in4 = kStatic ? cj + 1 < BY : pl.inside[p][4];
    const bool in5 = kStatic ? ck - 1 >= 0 : pl.inside[p][5];
    const bool in6 = kStatic ? ck + 1 < BZ : pl.inside[p][6];
    gc[p] = base + ci + (long)g.pitch * cj + g.plane * ck;
    T s = f[gc[p]];
    if (!in1) s = sc_sub(s, sc_mul(st.c[1], x[gc[p] - 1]));
    if (!in2) s = sc_sub(s, sc_mul(st.c[2], x[gc[p] + 1]));
    if (!in3) s = sc_sub(s, sc_mul(st.c[3], x[gc[p] - g.pitch]));
    if (!in4) s = sc_sub(s, sc_mul(st.c[4], x[gc[p] + g.pitch]));
    if (DIM == 3) {
      if (!in5) s = sc_sub(s, sc_mul(st.c[5], x[gc[p] - g.plane]));
      if (!in6) s = sc_sub(s, sc_mul(st.c[6], x[gc[p] + g.plane]));
    }
    r[p] = s;
  }
#pragma unroll
  for (int p = 0; p < M; ++p)
#pragma unroll
    for (int q = p + 1; q < M; ++q)
      if (!pl.skip[q][p]) r[q] = sc_sub(r[q], sc_mul(pl.fac[q][p], r[p]));
#pragma unroll
  for (int p = M - 1; p >= 0; --p) {
    T s = r[p];
#pragma unroll
    for (int t = p + 1; t < M; ++t) s = sc_sub(s, sc_mul(pl.upper[p][t], r[t]));
    r[p] = sc_div(s, pl.pivot[p]);
  }
#pragma unroll
  for (int p = 0; p < M; ++p) y[gc[p]] = sc_add(sc_rmul(omm, x[gc[p]]), sc_rmul(om, r[p]));
}

// copies the boundary shell x -> y (block smoother writes only inner tiles)
template <typename T, int DIM>
__global__ void __launch_bounds__(256) evomg_copy_boundary(Geo g, const T* __restrict__ x, T* __restrict__ y) {
  int i, j, k;
  if (!node_of_thread<DIM>(g, i, j, k)) return;
  if (is_inner<DIM>(g, i, j, k)) return;
  const long c = gidx<DIM>(g, i, j, k);
  y[c] = x[c];
}
// same, but launched over the faces only: blockIdx.z = 2 * face + side, face = 0 (i0 = 0, n-1), 1 (i1), 2 (i2);
// grid: (ceil(n/64), ceil(n/4) or 1, 2 * DIM)
template <typename T, int DIM>
__global__ void __launch_bounds__(256) evomg_copy_faces(Geo g, const T* __restrict__ x, T* __restrict__ y) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y * blockDim.y + threadIdx.y;
  const int face = blockIdx.z >> 1;                // all 2 * DIM faces in one launch
  const int side = (blockIdx.z & 1) ? g.n - 1 : 0;
  if (a >= g.n || b >= ((DIM == 3) ? g.n : 1)) return;
  int i, j, k;
  if (face == 0) { i = side; j = a; k = b; }
  else if (face == 1) { i = a; j = side; k = b; }
  else { i = a; j = b; k = side; }
  const long c = gidx<DIM>(g, i, j, (DIM == 3) ? k : 0);
  y[c] = x[c];
}

// ---- K4: residual r = f - A x on inner nodes; y = A x variant for the Krylov solvers ------------------------
template <typename T, int DIM, bool RESIDUAL>
__global__ void __launch_bounds__(256) evomg_apply(Geo g, Stencil<T> st, const T* __restrict__ x,
                                                   const T* __restrict__ f, T* __restrict__ r) {
  int i, j, k;
  if (!node_of_thread<DIM>(g, i, j, k)) return;
  if (!is_inner<DIM>(g, i, j, k)) return;
  const long c = gidx<DIM>(g, i, j, k);
  const T ax = apply_stencil<T, DIM>(st, x, g, c);
  r[c] = RESIDUAL ? sc_sub(f[c], ax) : ax;
}

// Richardson step y = x + w * (f - A x) on inner nodes (y pre-filled with x on the boundary by the caller)
template <typename T, int DIM>
__global__ void __launch_bounds__(256) evomg_richardson(Geo g, Stencil<T> st, const T* __restrict__ x,
                                                        const T* __restrict__ f, T* __restrict__ y, double om) {
  int i, j, k;
  if (!node_of_thread<DIM>(g, i, j, k)) return;
  const long c = gidx<DIM>(g, i, j, k);
  if (!is_inner<DIM>(g, i, j, k)) { y[c] = x[c]; return; }
  const T res = sc_sub(f[c], apply_stencil<T, DIM>(st, x, g, c));
  y[c] = sc_add(x[c], sc_mul(scalar_traits<T>::from_real(om), res));
}

// ---- K5: full-weighting restriction, coarse-node driven ----------------------------------------------------
template <typename T, int DIM>
__global__ void __launch_bounds__(256) evomg_restrict(Geo gc, Geo gf, const T* __restrict__ fine, T* __restrict__ coarse) {
  int I, J, K;
  if (!node_of_thread<DIM>(gc, I, J, K)) return;
  if (!is_inner<DIM>(gc, I, J, K)) return;
  const int i = 2 * I, j = 2 * J, k = 2 * K;
  if (DIM == 2) {
    const long c = gidx<2>(gf, i, j, 0);
    const long p = gf.pitch;
    const T e = sc_add(sc_add(fine[c - 1], fine[c + 1]), sc_add(fine[c - p], fine[c + p]));
    const T q = sc_add(sc_add(fine[c - p - 1], fine[c - p + 1]), sc_add(fine[c + p - 1], fine[c + p + 1]));
    T v = sc_rmul(0.25, fine[c]);
    v = sc_add(v, sc_rmul(0.125, e));
    v = sc_add(v, sc_rmul(0.0625, q));
    coarse[gidx<2>(gc, I, J, 0)] = v;
  } else {
    T face = scalar_traits<T>::zero(), edge = scalar_traits<T>::zero(), corner = scalar_traits<T>::zero();
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int nz = (dx != 0) + (dy != 0) + (dz != 0);
          if (nz == 0) continue;
          const T v = fine[gidx<3>(gf, i + dx, j + dy, k + dz)];
          if (nz == 1) face = sc_add(face, v);
          else if (nz == 2) edge = sc_add(edge, v);
          else corner = sc_add(corner, v);
        }
    T v = sc_rmul(0.125, fine[gidx<3>(gf, i, j, k)]);
    v = sc_add(v, sc_rmul(0.0625, face));
    v = sc_add(v, sc_rmul(0.03125, edge));
    v = sc_add(v, sc_rmul(0.015625, corner));
    coarse[gidx<3>(gc, I, J, K)] = v;
  }
}

// ---- K6 / K6': bi-/tri-linear prolongation, fine-node driven; ACC: fine += w * P coarse ---------------------
template <typename T, int DIM>
__device__ __forceinline__ T interpolate(const Geo& gc, const T* __restrict__ cz, int i, int j, int k) {
  const int I = i >> 1, J = j >> 1, K = k >> 1;
  const int ox = i & 1, oy = j & 1, oz = (DIM == 3) ? (k & 1) : 0;
  if (DIM == 2) {
    const long c = gidx<2>(gc, I, J, 0);
    const long p = gc.pitch;
    if (!ox && !oy) return cz[c];
    if (ox && !oy) return sc_rmul(0.5, sc_add(cz[c], cz[c + 1]));
    if (!ox && oy) return sc_rmul(0.5, sc_add(cz[c], cz[c + p]));
    return sc_rmul(0.25, sc_add(sc_add(cz[c], cz[c + 1]), sc_add(cz[c + p], cz[c + p + 1])));
  }
  T s = scalar_traits<T>::zero();
  for (int dz = 0; dz <= oz; ++dz)
    for (int dy = 0; dy <= oy; ++dy)
      for (int dx = 0; dx <= ox; ++dx) s = sc_add(s, cz[gidx<3>(gc, I + dx, J + dy, K + dz)]);
  const int nodd = ox + oy + oz;
  const double w = (nodd == 0) ? 1.0 : (nodd == 1) ? 0.5 : (nodd == 2) ? 0.25 : 0.125;
  return sc_rmul(w, s);
}
template <typename T, int DIM, bool ACC>
__global__ void __launch_bounds__(256) evomg_prolong(Geo gf, Geo gc, const T* __restrict__ coarse, T* __restrict__ fine, double om) {
  int i, j, k;
  if (!node_of_thread<DIM>(gf, i, j, k)) return;
  if (!is_inner<DIM>(gf, i, j, k)) return;
  const long c = gidx<DIM>(gf, i, j, k);
  const T v = interpolate<T, DIM>(gc, coarse, i, j, k);
  fine[c] = ACC ? sc_add(fine[c], sc_rmul(om, v)) : v;
}

// 3D, coarse-CELL driven: thread (I, J, K) owns the 2x2x2 fine nodes (2I+a, 2J+b, 2K+c) of one coarse cell. It loads
// the cell's 8 corner values once (instead of up to 8 loads for every fine node) and touches the fine array in
// x-adjacent pairs (2I, 2I+1): aligned 16-byte accesses. Sums are formed exactly as interpolate<T,3> forms them
// (start from zero, add corners in dz, dy, dx order, scale once), so the result is bit-identical to evomg_prolong.
// Only fine index 0 needs masking (a = 0 with I = 0 etc.): 2I+1 <= n-2 for every cell and n-1 is never produced.
// grid: (ceil(cells/64), ceil(cells/4), cells), cells = coarse n - 1
template <typename T, bool ACC>
__global__ void __launch_bounds__(256) evomg_prolong_cell3d(Geo gf, Geo gc, const T* __restrict__ coarse, T* __restrict__ fine, double om) {
  const int I = blockIdx.x * blockDim.x + threadIdx.x;
  const int J = blockIdx.y * blockDim.y + threadIdx.y;
  const int K = blockIdx.z;
  const int cells = gc.n - 1;
  if (I >= cells || J >= cells || K >= cells) return;
  T cv[2][2][2];
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) cv[dz][dy][dx] = coarse[gidx<3>(gc, I + dx, J + dy, K + dz)];
  const T zero = scalar_traits<T>::zero();
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int j = 2 * J + b, k = 2 * K + c;
      if (j < 1 || k < 1) continue;                     // boundary row / plane of the fine grid
      T s0 = zero, s1 = zero;
#pragma unroll
      for (int dz = 0; dz <= c; ++dz)
#pragma unroll
        for (int dy = 0; dy <= b; ++dy) {
          s0 = sc_add(s0, cv[dz][dy][0]);
          s1 = sc_add(s1, cv[dz][dy][0]);
          s1 = sc_add(s1, cv[dz][dy][1]);
        }
      const double w0 = (b + c == 0) ? 1.0 : (b + c == 1) ? 0.5 : 0.25;
      const T v0 = sc_rmul(w0, s0), v1 = sc_rmul(0.5 * w0, s1);
      T* dst = fine + gidx<3>(gf, 2 * I, j, k);
      if (ACC) {
        const T o0 = dst[0], o1 = dst[1];
        const T n1 = sc_add(o1, sc_rmul(om, v1));
        const T n0 = (I >= 1) ? sc_add(o0, sc_rmul(om, v0)) : o0;
        dst[0] = n0;
        dst[1] = n1;
      } else {
        if (I >= 1) dst[0] = v0;
        dst[1] = v1;
      }
    }
}

// ---- K7: zero the inner nodes ---------------------------------------------------------------------------------
template <typename T, int DIM>
// side_columns (2D Helmholtz): also the first and last column of the inner rows, which is what the Robin condition
// x_b = x_in / (1 - i k h) yields for a zeroed interior
__global__ void __launch_bounds__(256) evomg_zero_inner(Geo g, T* __restrict__ x, int side_columns) {
  int i, j, k;
  if (!node_of_thread<DIM>(g, i, j, k)) return;
  if (!is_inner<DIM>(g, i, j, k)) {
    if (!(DIM == 2 && side_columns && j >= 1 && j <= g.n - 2)) return;
  }
  x[gidx<DIM>(g, i, j, k)] = scalar_traits<T>::zero();
}

// ---- restoring the initial fields before a solve: every SMALL field in one launch (round 4) ---------------------------------
// A level range of five or six levels has two dozen fields to zero or to copy from their initial image; one hipMemsetAsync /
// hipMemcpyAsync each was two dozen launches per candidate (a tenth of a 513^2 evaluation's wall time, and launches are what
// candidates in flight compete for). Segments are 16-byte multiples (rows are padded to 128 bytes).
#define EVOMG_RESET_MAX 40
struct ResetSeg { void* dst; const void* src; unsigned long long n16; };
struct ResetBatch { int n; ResetSeg seg[EVOMG_RESET_MAX]; };
static __global__ void __launch_bounds__(256) evomg_reset_batch(ResetBatch b) {
  const ResetSeg s = b.seg[blockIdx.y];
  uint4* __restrict__ d = reinterpret_cast<uint4*>(s.dst);
  const uint4* __restrict__ src = reinterpret_cast<const uint4*>(s.src);
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < s.n16; i += (unsigned long long)gridDim.x * 256)
    d[i] = src ? src[i] : zero;
}

// ---- K8: Helmholtz Robin boundary at x = 0, 1:  x_b = x_in / (1 - i k h)  (2D complex only) ----------------
// (static: the header is part of several translation units)
static __global__ void __launch_bounds__(256) evomg_robin_bc(Geo g, cplx* __restrict__ x, double kh) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < 1 || j >= g.n - 1) return;
  const cplx den{1.0, -kh};
  const long row = (long)g.pitch * j;
  x[row] = sc_div(x[row + 1], den);
  x[row + g.n - 1] = sc_div(x[row + g.n - 2], den);
}

// ---- neighbouring lanes by DPP wave shifts (no LDS round trip) -----------------------------------------------------
__device__ __forceinline__ int dpp_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }   // bound_ctrl: lanes without a source read 0
__device__ __forceinline__ int dpp_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
// value held by lane-1 (lane 0 receives 0)
__device__ __forceinline__ double from_left(double v) {
  return __hiloint2double(dpp_shr1(__double2hiint(v)), dpp_shr1(__double2loint(v)));
}
// value held by lane+1 (lane 63 receives 0)
__device__ __forceinline__ double from_right(double v) {
  return __hiloint2double(dpp_shl1(__double2hiint(v)), dpp_shl1(__double2loint(v)));
}
__device__ __forceinline__ cplx from_left(cplx v) { return cplx{from_left(v.re), from_left(v.im)}; }
__device__ __forceinline__ cplx from_right(cplx v) { return cplx{from_right(v.re), from_right(v.im)}; }


// ---- reductions ---------------------------------------------------------------------------------------------------
// wave64 shuffle reduction followed by an LDS reduction across the block's waves; fixed order => deterministic.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ cplx wave_sum(cplx v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v.re += __shfl_down(v.re, off, 64);
    v.im += __shfl_down(v.im, off, 64);
  }
  return v;
}
// all threads of the block receive the sum; `red` = LDS scratch of >= 17 elements; blockDim multiple of 64
template <typename T> __device__ __forceinline__ T block_sum(T v, T* red) {
  const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
  const int nthreads = blockDim.x * blockDim.y * blockDim.z;
  const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
  v = wave_sum(v);
  __syncthreads();            // protects `red` against the previous use
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (wave == 0) {
    T t = (lane < nwaves) ? red[lane] : scalar_traits<T>::zero();
    t = wave_sum(t);
    if (lane == 0) red[16] = t;
  }
  __syncthreads();
  return red[16];
}

// ---- fast variants of the same reduction trees (used by the register-resident Krylov kernels) ---------------------------
// Same summation order as wave_sum / block_sum (v[l] += v[l+off], off = 32,16,8,4,2,1; then the same tree over the
// wave sums padded with zeros) but without the LDS crossbar: v_permlane32_swap / v_permlane16_swap (gfx950) bring
// lane l+32 / l+16, DPP row_shl:n brings lane l+n inside a 16-lane row. Only lane 0's value is meaningful.
__device__ __forceinline__ double lanes_plus32(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[1], (int)a[1]);
}
// value held by the lane 32 away (lane l <-> lane l ^ 32): one v_permlane32_swap per 32-bit half serves both directions
__device__ __forceinline__ double lanes_swap32(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const bool upper = (threadIdx.x & 32) != 0;
  return __hiloint2double((int)(upper ? b[0] : b[1]), (int)(upper ? a[0] : a[1]));
}
__device__ __forceinline__ cplx lanes_swap32(cplx v) { return cplx{lanes_swap32(v.re), lanes_swap32(v.im)}; }
__device__ __forceinline__ double lanes_plus16(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[1], (int)a[1]);
}
template <int N> __device__ __forceinline__ double row_plus(double v) {
  // bound_ctrl: lanes whose source lies outside the 16-lane row read 0 — no zero-initialised destination needed
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + N, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + N, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_tree(double v) {   // offsets 8,4,2,1 inside row 0
  v += row_plus<8>(v);
  v += row_plus<4>(v);
  v += row_plus<2>(v);
  v += row_plus<1>(v);
  return v;
}
__device__ __forceinline__ double wave_sum_fast(double v) {
  v += lanes_plus32(v);
  v += lanes_plus16(v);
  return row_tree(v);
}
__device__ __forceinline__ cplx wave_sum_fast(cplx v) { return cplx{wave_sum_fast(v.re), wave_sum_fast(v.im)}; }
// tree over <= 16 wave sums held by lanes 0..15 (other lanes hold +0.0): the two cross-row steps add zeros
__device__ __forceinline__ double wave16_tree(double t) {
  t += 0.0;
  t += 0.0;
  return row_tree(t);
}
__device__ __forceinline__ cplx wave16_tree(cplx t) { return cplx{wave16_tree(t.re), wave16_tree(t.im)}; }
__device__ __forceinline__ double first_lane(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                          __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ cplx first_lane(cplx v) { return cplx{first_lane(v.re), first_lane(v.im)}; }
// ONE barrier per reduction: wave sums go to a double-buffered LDS slot array (`red` holds 2 x 16 elements, `phase`
// alternates), then EVERY wave folds the 16 sums itself. Reuse of a buffer is two reductions later, i.e. behind the
// barrier of the reduction in between, which all waves only reach after their reads of this one.
template <typename T> __device__ __forceinline__ T block_sum_fast(T v, T* red, int& phase) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  v = wave_sum_fast(v);
  T* buf = red + 16 * (phase & 1);
  ++phase;
  if (lane == 0) buf[wave] = v;
  __syncthreads();
  T t = (lane < nwaves) ? buf[lane] : scalar_traits<T>::zero();
  return first_lane(wave16_tree(t));
}

// dot over inner nodes, pass 1: grid-stride over rows, one partial per block (fixed grid => deterministic)
template <typename T, int DIM>
__global__ void __launch_bounds__(256) evomg_dot_partial(Geo g, const T* __restrict__ a, const T* __restrict__ b,
                                                         T* __restrict__ partial) {
  __shared__ T red[17];
  const int m = g.n - 2;
  const long rows = (DIM == 2) ? m : (long)m * m;
  T s = scalar_traits<T>::zero();
  for (long r = blockIdx.x; r < rows; r += gridDim.x) {
    const int j = 1 + (int)(r % m), k = (DIM == 3) ? 1 + (int)(r / m) : 0;
    const long base = gidx<DIM>(g, 0, j, k);
    for (int i = 1 + threadIdx.x; i <= m; i += blockDim.x) s = sc_add(s, sc_mul(a[base + i], b[base + i]));
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// The solver loop's residual + r . r on the levels no norm-carrying kernel serves (round 4): evomg_apply<.., true>'s residual of
// every inner node (written) and, in the same pass, its square added up in evomg_dot_partial's order — one launch instead of two,
// the same bits (the residual of a node is the value the dot kernel would have read back).
template <typename T, int DIM>
__global__ void __launch_bounds__(256) evomg_residual_dot_partial(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f,
                                                                  T* __restrict__ r_out, T* __restrict__ partial) {
  __shared__ T red[17];
  const int m = g.n - 2;
  const long rows = (DIM == 2) ? m : (long)m * m;
  T s = scalar_traits<T>::zero();
  for (long r = blockIdx.x; r < rows; r += gridDim.x) {
    const int j = 1 + (int)(r % m), k = (DIM == 3) ? 1 + (int)(r / m) : 0;
    const long base = gidx<DIM>(g, 0, j, k);
    for (int i = 1 + threadIdx.x; i <= m; i += blockDim.x) {
      const long c = base + i;
      const T v = sc_sub(f[c], apply_stencil<T, DIM>(st, x, g, c));
      r_out[c] = v;
      s = sc_add(s, sc_mul(v, v));
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// pass 2: one block sums the partials in a fixed order
template <typename T>
__global__ void __launch_bounds__(256) evomg_dot_final(const T* __restrict__ partial, int count, T* __restrict__ out) {
  prio_latency();
  __shared__ T red[17];
  T s = scalar_traits<T>::zero();
  // eight loads in flight per thread, added in the order they would have been added one by one: with candidates in flight a
  // load takes microseconds, and 36 of them in a dependent chain made this 12 us kernel an 75 us one (profiles/r05_throughput.md)
  for (int i0 = threadIdx.x; i0 < count; i0 += 8 * blockDim.x) {
    T v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + j * blockDim.x;
      v[j] = partial[i < count ? i : i0];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (i0 + j * (int)blockDim.x < count) s = sc_add(s, v[j]);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

// ---- fused passes of the outer (preconditioned BiCGStab) solver ------------------------------------------------------
// Both kernels walk the inner nodes in evomg_dot_partial's order (block b: rows b, b + gridDim, ...; thread i: columns
// i, i + 256, ...) and accumulate their dot products exactly as that kernel does (s = s + a * b, unfused), so
// partial sums — and with evomg_dot_final the dot products — carry the same bits as the separate passes they replace.
// 2D. partial: [NDOT][gridDim.x].
// v = A x (stored); dot 0 = w0 . v; dot 1 (NDOT == 2) = v . v
// The hardware deals consecutive workgroups to the eight XCDs in turn; the rows b - 1, b, b + 1 a stencil reads would then be
// fetched by three XCDs, each through its own L2 (measured at 2049^2 complex: 264 MB read for 134 MB of operands,
// profiles/r05_pmc_c3_tables.txt). `row_block` renumbers the workgroups so that one XCD walks one contiguous band of rows: the
// neighbouring rows come out of its L2. Which workgroup sums which rows — partial[row_block] — is evomg_dot_partial's order.
template <typename T, int NDOT>
__global__ void __launch_bounds__(256) evomg_apply_dots(Geo g, Stencil<T> st, const T* __restrict__ x, T* __restrict__ v,
                                                        const T* __restrict__ w0, T* __restrict__ partial) {
  __shared__ T red[17];
  const int m = g.n - 2;
  const int row_block = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  T s0 = scalar_traits<T>::zero(), s1 = scalar_traits<T>::zero();
  for (long r = row_block; r < m; r += gridDim.x) {
    const long base = gidx<2>(g, 0, 1 + (int)r, 0);
    for (int i = 1 + threadIdx.x; i <= m; i += blockDim.x) {
      const long c = base + i;
      const T ax = apply_stencil<T, 2>(st, x, g, c);
      v[c] = ax;
      s0 = sc_add(s0, sc_mul(w0[c], ax));
      if (NDOT == 2) s1 = sc_add(s1, sc_mul(ax, ax));
    }
  }
  s0 = block_sum(s0, red);
  if (threadIdx.x == 0) partial[row_block] = s0;
  if (NDOT == 2) {
    s1 = block_sum(s1, red);
    if (threadIdx.x == 0) partial[gridDim.x + row_block] = s1;
  }
}
// z = x + a y (stored); dot 0 = z . z; dot 1 = w . z
template <typename T>
__global__ void __launch_bounds__(256) evomg_axpy_dots(Geo g, T* __restrict__ z, const T* __restrict__ x, T a, const T* __restrict__ y,
                                                       const T* __restrict__ w, T* __restrict__ partial) {
  __shared__ T red[17];
  const int m = g.n - 2;
  T s0 = scalar_traits<T>::zero(), s1 = scalar_traits<T>::zero();
  for (long r = blockIdx.x; r < m; r += gridDim.x) {
    const long base = gidx<2>(g, 0, 1 + (int)r, 0);
    for (int i = 1 + threadIdx.x; i <= m; i += blockDim.x) {
      const long c = base + i;
      const T zv = sc_add(x[c], sc_mul(a, y[c]));
      z[c] = zv;
      s0 = sc_add(s0, sc_mul(zv, zv));
      s1 = sc_add(s1, sc_mul(w[c], zv));
    }
  }
  s0 = block_sum(s0, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s0;
  s1 = block_sum(s1, red);
  if (threadIdx.x == 0) partial[gridDim.x + blockIdx.x] = s1;
}
// ---- the outer BiCGStab's step lengths formed and consumed on the device ------------------------------------------------
// alpha = rho' / (rhat . A u) and omega = (t . s) / (t . t) are needed only by the very next vector updates: a final
// reduction that also forms the quotient (evomg_dot_final_quot), and updates that read it from device memory, take the
// host out of two of the three round trips of an iteration. Same operations on the same values as the host loop's
// sc_div: the same bits. scal[0 .. 1]: the dot product(s); scal[slot]: the quotient.
//   ndot == 1: scal[slot] = num / dot0          ndot == 2: scal[slot] = dot0 / dot1
// num_slot >= 0: the numerator is scal[num_slot] instead of `num`
template <typename T>
__global__ void __launch_bounds__(256) evomg_dot_final_quot(const T* __restrict__ partial, int count, int ndot, T num, T* __restrict__ scal, int slot,
                                                            int num_slot = -1) {
  prio_latency();
  __shared__ T red[17];
  T d[2] = {scalar_traits<T>::zero(), scalar_traits<T>::zero()};
  for (int k = 0; k < ndot; ++k) {
    T s = scalar_traits<T>::zero();
    for (int i = threadIdx.x; i < count; i += blockDim.x) s = sc_add(s, partial[(long)k * count + i]);
    d[k] = block_sum(s, red);
  }
  if (threadIdx.x == 0) {
    scal[0] = d[0];
    if (ndot > 1) scal[1] = d[1];
    if (num_slot >= 0) num = scal[num_slot];
    scal[slot] = ndot == 1 ? sc_div(num, d[0]) : sc_div(d[0], d[1]);
  }
}
// the head of a BiCGStab iteration on the device: rho = scal[4] (the previous iteration's rho'), rho' = scal[1] (formed
// with the residual norm at the end of the previous iteration), alpha = scal[2], omega = scal[3]:
//   scal[5] = beta = (rho' / rho) * (alpha / omega) ; scal[4] = rho'
// so that the next iteration can be queued before the host has seen the norm of this one (the host's sc_mul / sc_div on
// the same values: the same bits)
template <typename T> __global__ void evomg_bicg_beta(T* __restrict__ scal) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const T rho = scal[4], rho_new = scal[1];
    scal[5] = sc_mul(sc_div(rho_new, rho), sc_div(scal[2], scal[3]));
    scal[4] = rho_new;
  }
}
// evomg_axpy2 with a = -scal[3] (omega) and b = scal[5] (beta)
template <typename T>
__global__ void __launch_bounds__(256) evomg_axpy2_dev(Geo g, T* __restrict__ z, const T* __restrict__ x1, const T* __restrict__ y1,
                                                       const T* __restrict__ x2, const T* __restrict__ scal) {
  int i, j, k;
  if (!node_of_thread<2>(g, i, j, k)) return;
  if (!is_inner<2>(g, i, j, k)) return;
  const long c = gidx<2>(g, i, j, k);
  const T a = sc_neg(scal[3]), b = scal[5];
  const T t = sc_add(x1[c], sc_mul(a, y1[c]));
  z[c] = sc_add(x2[c], sc_mul(b, t));
}
// z1 = x1 + a y1 ; z2 = x2 - a y2 on inner nodes, a = *ap (two independent axpy steps in one pass)
template <typename T>
__global__ void __launch_bounds__(256) evomg_axpy_pair_dev(Geo g, T* __restrict__ z1, const T* __restrict__ x1, const T* __restrict__ y1,
                                                           T* __restrict__ z2, const T* __restrict__ x2, const T* __restrict__ y2,
                                                           const T* __restrict__ ap) {
  int i, j, k;
  if (!node_of_thread<2>(g, i, j, k)) return;
  if (!is_inner<2>(g, i, j, k)) return;
  const long c = gidx<2>(g, i, j, k);
  const T a = *ap;
  z1[c] = sc_add(x1[c], sc_mul(a, y1[c]));
  z2[c] = sc_add(x2[c], sc_mul(sc_neg(a), y2[c]));
}
// z = x + a y on inner nodes, a = *ap
// ROBIN (Helmholtz columns x_b = x_in / (1 - i k h), evomg_robin_bc's division): the thread of an inner node beside a Robin column
// writes that column's node as well — the boundary pass behind the update is gone
template <typename T, bool ROBIN = false>
__global__ void __launch_bounds__(256) evomg_axpy_dev(Geo g, T* __restrict__ z, const T* __restrict__ x, const T* __restrict__ ap,
                                                      const T* __restrict__ y, double kh = 0.0) {
  int i, j, k;
  if (!node_of_thread<2>(g, i, j, k)) return;
  if (!is_inner<2>(g, i, j, k)) return;
  const long c = gidx<2>(g, i, j, k);
  const T v = sc_add(x[c], sc_mul(*ap, y[c]));
  z[c] = v;
  if constexpr (ROBIN && scalar_traits<T>::is_complex) {
    const cplx den{1.0, -kh};
    if (i == 1) z[c - 1] = sc_div(v, den);
    if (i == g.n - 2) z[c + 1] = sc_div(v, den);
  }
}
// evomg_axpy_dots with a = -(*ap)
template <typename T>
__global__ void __launch_bounds__(256) evomg_axpy_dots_dev(Geo g, T* __restrict__ z, const T* __restrict__ x, const T* __restrict__ ap,
                                                           const T* __restrict__ y, const T* __restrict__ w, T* __restrict__ partial) {
  __shared__ T red[17];
  const int m = g.n - 2;
  const T a = sc_neg(*ap);
  T s0 = scalar_traits<T>::zero(), s1 = scalar_traits<T>::zero();
  for (long r = blockIdx.x; r < m; r += gridDim.x) {
    const long base = gidx<2>(g, 0, 1 + (int)r, 0);
    for (int i = 1 + threadIdx.x; i <= m; i += blockDim.x) {
      const long c = base + i;
      const T zv = sc_add(x[c], sc_mul(a, y[c]));
      z[c] = zv;
      s0 = sc_add(s0, sc_mul(zv, zv));
      s1 = sc_add(s1, sc_mul(w[c], zv));
    }
  }
  s0 = block_sum(s0, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s0;
  s1 = block_sum(s1, red);
  if (threadIdx.x == 0) partial[gridDim.x + blockIdx.x] = s1;
}
// z = x2 + b (x1 + a y1) on inner nodes: two consecutive axpy steps (z = x1 + a y1, then z = x2 + b z) in one pass
template <typename T>
__global__ void __launch_bounds__(256) evomg_axpy2(Geo g, T* __restrict__ z, const T* __restrict__ x1, T a, const T* __restrict__ y1,
                                                   const T* __restrict__ x2, T b) {
  int i, j, k;
  if (!node_of_thread<2>(g, i, j, k)) return;
  if (!is_inner<2>(g, i, j, k)) return;
  const long c = gidx<2>(g, i, j, k);
  const T t = sc_add(x1[c], sc_mul(a, y1[c]));
  z[c] = sc_add(x2[c], sc_mul(b, t));
}
// up to two final reductions in one launch: out[d] = sum of partial[d][0 .. count); host_out: the first sum once more, in pinned
// host memory (the host reads it behind an event without a copy launch)
// beta_step (the outer BiCGStab's device-side scalars, out = scal): evomg_bicg_beta's two lines behind the sums, by the thread
// that has just written scal[1] — the launch of its own between two iterations is gone
template <typename T>
__global__ void __launch_bounds__(256) evomg_dot_final2(const T* __restrict__ partial, int count, int ndot, T* __restrict__ out,
                                                        T* __restrict__ host_out = nullptr, int beta_step = 0) {
  prio_latency();
  __shared__ T red[17];
  for (int d = 0; d < ndot; ++d) {
    T s = scalar_traits<T>::zero();
    for (int i = threadIdx.x; i < count; i += blockDim.x) s = sc_add(s, partial[(long)d * count + i]);
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
      out[d] = s;
      if (d == 0 && host_out) host_out[0] = s;
      if (beta_step && d == 1) {
        const T rho = out[4];
        out[5] = sc_mul(sc_div(s, rho), sc_div(out[2], out[3]));
        out[4] = s;
      }
    }
  }
}

// z = x + a*y on inner nodes
template <typename T, int DIM>
__global__ void __launch_bounds__(256) evomg_axpy(Geo g, T* __restrict__ z, const T* __restrict__ x, T a,
                                                  const T* __restrict__ y) {
  int i, j, k;
  if (!node_of_thread<DIM>(g, i, j, k)) return;
  if (!is_inner<DIM>(g, i, j, k)) return;
  const long c = gidx<DIM>(g, i, j, k);
  z[c] = sc_add(x[c], sc_mul(a, y[c]));
}

// info[2] += *src: the launch-per-iteration Krylov solvers keep their iteration count in device state
static __global__ void evomg_accumulate_iterations(int* __restrict__ info, const int* __restrict__ src) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { info[0] = *src; info[2] += *src; }
}
