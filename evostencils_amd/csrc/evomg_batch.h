// evomg_batch.h — coarse-level op batches (round 2; SURVEY 7.6 "persistent coarse sub-cycle").
//
// On grids up to 129^2 a kernel launch costs more than the work it carries: 5-7 us per launch for <= 16 641 nodes
// (profiles/r02_tutorial_size.md). The interpreter therefore collects consecutive ops on such levels into a list of
// micro-ops and runs the list in ONE launch of ONE 1024-thread workgroup: op after op, a workgroup barrier in between,
// fields in global memory (they sit in this compute unit's L1 / the L2). 2D, real, Dirichlet.
// STATUS: built, bit-identical, and measured SLOWER than what it replaces — off unless EVOMG_BATCH=1. One compute unit moves
// 64 B/clk: an op on 129^2 nodes (6 loads + 1 store of 8 B per node) takes the workgroup ~6 us, what a launch costs, and
// batching breaks the op-level fusions (smoother + residual + restriction, prolongation + smoother) that already run the
// coarse levels on many units. Tutorial-size solve (513^2, levels 5..9, 12 V(1,1) cycles): 3.02 ms batched up to 129^2,
// 2.48 ms up to 65^2, 2.32 ms up to 33^2, 2.31 ms without (profiles/r02_tutorial_size.md).
// Every micro-op evaluates the formula of the corresponding generic kernel of evomg_kernels.h (relax_point,
// apply_stencil, evomg_restrict, interpolate) node by node: bit-identical to the separate launches, to the strip
// kernels and to the oracle.
#pragma once
#include "evomg_kernels.h"

enum { MB_JACOBI = 0, MB_RBGS = 1, MB_RESIDUAL = 2, MB_RESTRICT = 3, MB_PROLONG_ADD = 4, MB_PROLONG_SET = 5, MB_ZERO = 6, MB_COPY = 7 };
#define EVOMG_BATCH_MAX 64          // micro-ops per launch (kernel arguments: 4 KB)
#define EVOMG_BATCH_LEVELS 8        // levels 0 .. 7 (129^2)
struct MicroOp {
  int kind, level;                  // level: of the written field
  const void* a;                    // x (sweeps, residual) | source field (restriction: fine, prolongation: coarse, copy)
  const void* b;                    // right-hand side
  void* c;                          // written field (Jacobi: y; red-black: null, in place on a)
  double om;
};
struct MicroBatch {
  int n, pad;
  MicroOp op[EVOMG_BATCH_MAX];
};
template <typename T> struct BatchLevels {
  Geo g[EVOMG_BATCH_LEVELS];
  Stencil<T> st[EVOMG_BATCH_LEVELS];
};

template <typename T>
__device__ __forceinline__ T relax_point_cd(const Stencil<T>& st, const ConstDiv<T>& cd, const T* x, const T* f, const Geo& g, long c,
                                            double om, double omm) {
  T s = f[c];
  s = sc_sub(s, sc_mul(st.c[1], x[c - 1]));
  s = sc_sub(s, sc_mul(st.c[2], x[c + 1]));
  s = sc_sub(s, sc_mul(st.c[3], x[c - g.pitch]));
  s = sc_sub(s, sc_mul(st.c[4], x[c + g.pitch]));
  const T xs = const_div(s, cd);                   // = s / st.c[0], bit for bit (evomg_scalar.h)
  return sc_add(sc_rmul(omm, x[c]), sc_rmul(om, xs));
}

// One pass over a thread's inner nodes in groups of U: all loads of a group are issued before its first store, so the
// L2 round trips of U nodes overlap whatever the compiler can prove about aliasing (in-place colour passes, prolongation
// onto the field it corrects). Thread (tx, ty) of the 64 x 16 workgroup owns the nodes (i0 + tx + 64 b, 1 + ty + 16 a);
// `stride`: 1, or 2 for the nodes of one colour (i = 2 t + parity of the row). load(i, j) -> values, store(i, j, values).
template <int U, typename LoadF, typename StoreF>
__device__ __forceinline__ void batch_pass(int n, int tx, int ty, int colour, LoadF load, StoreF store) {
  const int cols = (colour < 0) ? n - 2 : (n - 1) / 2;                    // candidates per row (t = 0 .. (n-1)/2 - 1)
  const int nb = (cols + 63) / 64, na = (n - 2 + 15) / 16, total = na * nb;
  for (int k0 = 0; k0 < total; k0 += U) {
    decltype(load(1, 1)) v[U];
    int ii[U], jj[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u, a = k / nb, b = k - a * nb;
      const int j = 1 + ty + 16 * a, t = tx + 64 * b;
      const int i = (colour < 0) ? 1 + t : 2 * t + ((j + colour) & 1);
      ok[u] = k < total && j <= n - 2 && i >= 1 && i <= n - 2;
      ii[u] = ok[u] ? i : 1;
      jj[u] = ok[u] ? j : 1;
      v[u] = load(ii[u], jj[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (ok[u]) store(ii[u], jj[u], v[u]);
  }
}
template <typename T> struct In5 { T c, w, e, s, n, f; };

template <typename T>
__global__ void __launch_bounds__(1024) evomg_coarse_batch(BatchLevels<T> lvls, MicroBatch mb) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 64 columns x 16 rows per pass
  constexpr int U = 4;
  for (int q = 0; q < mb.n; ++q) {
    const MicroOp m = mb.op[q];
    const Geo g = lvls.g[m.level];
    const Stencil<T> st = lvls.st[m.level];
    const int n = g.n;
    const long p = g.pitch;
    const double om = m.om, omm = 1.0 - m.om;
    auto relaxed = [&](const In5<T>& v, const ConstDiv<T>& cd) -> T {   // = relax_point
      T s = v.f;
      s = sc_sub(s, sc_mul(st.c[1], v.w));
      s = sc_sub(s, sc_mul(st.c[2], v.e));
      s = sc_sub(s, sc_mul(st.c[3], v.s));
      s = sc_sub(s, sc_mul(st.c[4], v.n));
      const T xs = const_div(s, cd);                             // = s / st.c[0], bit for bit (evomg_scalar.h)
      return sc_add(sc_rmul(omm, v.c), sc_rmul(om, xs));
    };
    switch (m.kind) {
      case MB_JACOBI:
      case MB_RBGS: {
        const T* x = static_cast<const T*>(m.a);
        const T* f = static_cast<const T*>(m.b);
        T* y = (m.kind == MB_JACOBI) ? static_cast<T*>(m.c) : static_cast<T*>(const_cast<void*>(m.a));
        const ConstDiv<T> cd = const_div_make(st.c[0]);
        auto load = [&](int i, int j) { const long c = i + p * j; return In5<T>{x[c], x[c - 1], x[c + 1], x[c - p], x[c + p], f[c]}; };
        auto store = [&](int i, int j, const In5<T>& v) { y[i + p * j] = relaxed(v, cd); };
        if (m.kind == MB_JACOBI) {
          batch_pass<U>(n, tx, ty, -1, load, store);
          // out of place: y also gets x's boundary values (evomg_jacobi)
          for (int t = threadIdx.x; t < 4 * n; t += 1024) {
            const int side = t / n, r = t - side * n;
            const long c = (side == 0) ? r : (side == 1) ? r + p * (n - 1) : (side == 2) ? p * r : (n - 1) + p * r;
            y[c] = x[c];
          }
        } else {
          batch_pass<U>(n, tx, ty, 0, load, store);
          __syncthreads();                                       // colour 1 reads the new colour-0 values
          batch_pass<U>(n, tx, ty, 1, load, store);
        }
        break;
      }
      case MB_RESIDUAL: {
        const T* x = static_cast<const T*>(m.a);
        const T* f = static_cast<const T*>(m.b);
        T* r = static_cast<T*>(m.c);
        auto load = [&](int i, int j) { const long c = i + p * j; return In5<T>{x[c], x[c - 1], x[c + 1], x[c - p], x[c + p], f[c]}; };
        auto store = [&](int i, int j, const In5<T>& v) {          // = f - apply_stencil
          T a = sc_mul(st.c[0], v.c);
          a = sc_add(a, sc_mul(st.c[1], v.w));
          a = sc_add(a, sc_mul(st.c[2], v.e));
          a = sc_add(a, sc_mul(st.c[3], v.s));
          a = sc_add(a, sc_mul(st.c[4], v.n));
          r[i + p * j] = sc_sub(v.f, a);
        };
        batch_pass<U>(n, tx, ty, -1, load, store);
        break;
      }
      case MB_RESTRICT: {                                      // g: coarse; fine = level + 1
        const Geo gf = lvls.g[m.level + 1];
        const T* fine = static_cast<const T*>(m.a);
        T* coarse = static_cast<T*>(m.c);
        const long pf = gf.pitch;
        struct In9 { T c, e, d; };
        auto load = [&](int I, int J) {                          // = evomg_restrict<T, 2>
          const long c = 2 * I + pf * (2 * J);
          const T e = sc_add(sc_add(fine[c - 1], fine[c + 1]), sc_add(fine[c - pf], fine[c + pf]));
          const T d = sc_add(sc_add(fine[c - pf - 1], fine[c - pf + 1]), sc_add(fine[c + pf - 1], fine[c + pf + 1]));
          return In9{fine[c], e, d};
        };
        auto store = [&](int I, int J, const In9& v) {
          T o = sc_rmul(0.25, v.c);
          o = sc_add(o, sc_rmul(0.125, v.e));
          o = sc_add(o, sc_rmul(0.0625, v.d));
          coarse[I + p * J] = o;
        };
        batch_pass<U>(n, tx, ty, -1, load, store);
        break;
      }
      case MB_PROLONG_ADD:
      case MB_PROLONG_SET: {                                   // g: fine; coarse = level - 1
        const Geo gc = lvls.g[m.level - 1];
        const T* cz = static_cast<const T*>(m.a);
        T* fine = static_cast<T*>(m.c);
        const bool acc = m.kind == MB_PROLONG_ADD;
        struct InP { T old, v; };
        auto load = [&](int i, int j) { return InP{acc ? fine[i + p * j] : scalar_traits<T>::zero(), interpolate<T, 2>(gc, cz, i, j, 0)}; };
        auto store = [&](int i, int j, const InP& v) { fine[i + p * j] = acc ? sc_add(v.old, sc_rmul(om, v.v)) : v.v; };
        batch_pass<U>(n, tx, ty, -1, load, store);
        break;
      }
      case MB_ZERO: {
        T* x = static_cast<T*>(m.c);
        for (int j = 1 + ty; j < n - 1; j += 16)
          for (int i = 1 + tx; i < n - 1; i += 64) x[i + p * j] = scalar_traits<T>::zero();
        break;
      }
      case MB_COPY: {
        const T* s = static_cast<const T*>(m.a);
        T* d = static_cast<T*>(m.c);
        auto load = [&](int i, int j) { return s[i + p * j]; };
        auto store = [&](int i, int j, const T& v) { d[i + p * j] = v; };
        batch_pass<U>(n, tx, ty, -1, load, store);
        for (int t = threadIdx.x; t < 4 * n; t += 1024) {
          const int side = t / n, r = t - side * n;
          const long c = (side == 0) ? r : (side == 1) ? r + p * (n - 1) : (side == 2) ? p * r : (n - 1) + p * r;
          d[c] = s[c];
        }
        break;
      }
    }
    __syncthreads();                                           // the next micro-op reads what this one wrote
  }
}
