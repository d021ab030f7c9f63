// evomg_krylov.h — coarse-grid solvers (K9) as ONE persistent workgroup per solve.
//
// gen_mgCycle@minLevel is a Krylov iteration on a small grid (129^2 for the 4096^2 / 6-level config): hundreds of
// iterations of SpMV + 2..5 dot products on ~1e4 nodes. As separate launches every dot product would be a kernel
// boundary (~1.5 us) plus a host round trip for the stopping test; instead the whole solve is one launch of one
// 1024-thread workgroup (16 wave64s on one CU), dot products are wave-level trees + one LDS hop, and the stopping
// test never leaves the device.
//
// Algorithms: CG as configured in Poisson/2D_FD_Poisson_fromL2.exa3:12-14; BiCGStab literally as written in
// Helmholtz/2D_FD_Helmholtz_fromL3.exa3:396-433. Node ownership, reduction order and the fused-multiply-add
// arithmetic are the ones the oracle restates (oracle/evomg_oracle_impl.h, "Krylov solvers on the coarsest grid"),
// so every variant below (global-memory, register-resident, multi-workgroup) produces the same bits as the oracle.
#pragma once
#include <type_traits>
#include "evomg_kernels.h"

#define EVOMG_KRYLOV_THREADS 1024

template <int DIM> __device__ __forceinline__ long inner_index(const Geo& g, long q, int m) {
  const int i = 1 + (int)(q % m);
  const long t = q / m;
  if (DIM == 2) return gidx<2>(g, i, 1 + (int)t, 0);
  return gidx<3>(g, i, 1 + (int)(t % m), 1 + (int)(t / m));
}

template <typename T, int DIM>
__device__ __forceinline__ T apply_stencil_fma(const Stencil<T>& st, const T* __restrict__ x, const Geo& g, long c) {
  T a = sc_mul(st.c[0], x[c]);
  a = sc_fma(st.c[1], x[c - 1], a);
  a = sc_fma(st.c[2], x[c + 1], a);
  a = sc_fma(st.c[3], x[c - g.pitch], a);
  a = sc_fma(st.c[4], x[c + g.pitch], a);
  if (DIM == 3) {
    a = sc_fma(st.c[5], x[c - g.plane], a);
    a = sc_fma(st.c[6], x[c + g.plane], a);
  }
  return a;
}

// ---- canonical node ownership ------------------------------------------------------------------------------------
// 2D "tile": wave w (of lanes/64) covers the 64 columns of column block w % cb and the R rows of band w / cb; lane = column,
// rows ascending. 2D "pair" tile (two column blocks): wave w = band w, lane l owns columns 2l and 2l+1, visited row by
// row (row i: column 2l, then 2l+1). Otherwise "strided": thread t owns q = t, t+1024, ... (q x-fastest over the inner nodes).
// 3D "column" (when m*m <= 1024): thread t owns the z-column (ix, iy) = (t % m, t / m), iz ascending.
// Workgroup size ("lanes"): 512 threads (8 waves, 256 VGPRs each) for 2D grids up to two column blocks — the LDS
// kernel then keeps r, x AND A p of 32 rows per thread in registers — 1024 threads otherwise.
struct KrylovOwn {
  int tile, pair, column, m, cb, bands, R, lanes, wave;
  long total;
};
#define EVOMG_CG_WAVE_ROWS 16   // rows per lane of the one-wave CG (evomg_cg_wave2d): two half-waves x 16 rows cover m <= 31
// allow_pair: real-valued CG only (the pair tiles exist for the register-resident CG; the global-memory BiCGStab
// and complex kernels keep one column per lane, whose 8-byte accesses are contiguous across the wave)
template <int DIM> __host__ __device__ inline KrylovOwn krylov_own(int n, bool allow_pair = false) {
  KrylovOwn o;
  o.m = n - 2;
  o.total = (DIM == 2) ? (long)o.m * o.m : (long)o.m * o.m * o.m;
  o.cb = (o.m + 63) / 64;
  o.tile = (DIM == 2 && o.cb <= 16) ? 1 : 0;
  // "pair" tiles (65 <= m <= 128, i.e. 129^2): a lane owns the column PAIR (2l, 2l+1), a wave spans the whole width
  // and is one band. The horizontal neighbours of a node are then the lane's other element or ONE DPP shift away,
  // which is what lets the register-resident CG below run without touching memory for them.
  o.pair = (allow_pair && o.tile && o.cb == 2) ? 1 : 0;
  o.lanes = (o.tile && o.cb <= 2) ? 512 : EVOMG_KRYLOV_THREADS;
  o.column = (DIM == 3 && (long)o.m * o.m <= o.lanes) ? 1 : 0;
  o.bands = o.tile ? (o.pair ? o.lanes / 64 : (o.lanes / 64) / o.cb) : 0;
  o.R = o.tile ? (o.m + o.bands - 1) / o.bands : 0;
  // "wave" ownership (real-valued CG on 2D grids up to 33^2 — the coarsest grid of the reference's default level range
  // 5..9): ONE wave owns the whole grid, lane l = (half h = l >> 5, column l & 31) owns the rows 16 h .. 16 h + 15 of its
  // column, rows ascending. Lane 31 / 63 never has a column (m <= 31): the zero it holds is the halo between the halves.
  o.wave = (allow_pair && DIM == 2 && o.m <= 31) ? 1 : 0;
  if (o.wave) { o.tile = 0; o.pair = 0; o.lanes = 64; o.bands = 2; o.R = EVOMG_CG_WAVE_ROWS; }
  return o;
}
// The thread's node sequence in canonical order as (first index, stride, count); `strided` sequences compute the
// index per node instead.
struct OwnWalk {
  long c0, stride;
  int count, strided, ncol;    // ncol: nodes per step of `stride` (2 for a lane that owns a column pair)
};
template <int DIM> __device__ __forceinline__ OwnWalk own_walk(const Geo& g, const KrylovOwn& o) {
  OwnWalk wk{0, 0, 0, 0, 1};
  if (DIM == 2 && o.wave) {
    const int h = threadIdx.x >> 5, ix = threadIdx.x & 31;
    const int iy0 = h * o.R;
    if (threadIdx.x >= 64 || ix >= o.m || iy0 >= o.m) return wk;
    wk.count = min(o.R, o.m - iy0);
    wk.c0 = gidx<2>(g, ix + 1, iy0 + 1, 0);
    wk.stride = g.pitch;
  } else if (DIM == 2 && o.tile) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int band = o.pair ? w : w / o.cb;
    const int ix = o.pair ? 2 * l : 64 * (w % o.cb) + l;
    if (band >= o.bands || ix >= o.m) return wk;
    const int iy0 = band * o.R;
    wk.ncol = (o.pair && ix + 1 < o.m) ? 2 : 1;
    wk.count = max(0, min(o.R, o.m - iy0)) * wk.ncol;
    wk.c0 = gidx<2>(g, ix + 1, iy0 + 1, 0);
    wk.stride = g.pitch;
  } else if (DIM == 3 && o.column) {
    const int t = threadIdx.x;
    if (t >= o.m * o.m) return wk;
    wk.count = o.m;
    wk.c0 = gidx<3>(g, t % o.m + 1, t / o.m + 1, 1);
    wk.stride = g.plane;
  } else {
    wk.strided = 1;
    wk.count = (int)((o.total - threadIdx.x + o.lanes - 1) / o.lanes);
    if ((long)threadIdx.x >= o.total) wk.count = 0;
  }
  return wk;
}
template <int DIM> __device__ __forceinline__ long own_node_at(const Geo& g, const KrylovOwn& o, const OwnWalk& wk, int i) {
  if (wk.strided) return inner_index<DIM>(g, threadIdx.x + (long)o.lanes * i, o.m);
  if (wk.ncol == 2) return wk.c0 + wk.stride * (i >> 1) + (i & 1);
  return wk.c0 + wk.stride * i;
}

// Batched visit: the whole solve runs on ONE compute unit, so what limits it is memory latency (L2, ~1 us), not
// bandwidth. `load(c)` (pure loads) is called for a batch of nodes before `use(c, loaded)` consumes them in
// canonical order: the loads of a batch are all in flight together instead of one node's loads at a time. Slots
// past the end of the sequence re-load the last node (branch-free, so the compiler keeps counted vmcnt waits) and
// are skipped by `use`.
// Batch length: about 32 doubles of loaded values in flight per thread (the 1024-thread workgroup leaves 128 VGPRs).
template <typename V> struct krylov_batch {
  static constexpr int doubles = (int)(sizeof(V) / sizeof(double));
  static constexpr int value = 32 / doubles < 2 ? 2 : (32 / doubles > 8 ? 8 : 32 / doubles);
};
template <int DIM, typename V, typename L, typename F>
__device__ __forceinline__ void own_batched(const Geo& g, const KrylovOwn& o, L&& load, F&& use) {
  constexpr int B = krylov_batch<V>::value;
  const OwnWalk wk = own_walk<DIM>(g, o);
  for (int i0 = 0; i0 < wk.count; i0 += B) {
    V v[B];
#pragma unroll
    for (int j = 0; j < B; ++j) v[j] = load(own_node_at<DIM>(g, o, wk, min(i0 + j, wk.count - 1)));
#pragma unroll
    for (int j = 0; j < B; ++j)
      if (i0 + j < wk.count) use(own_node_at<DIM>(g, o, wk, i0 + j), v[j]);
  }
}
// visits the global indices of this thread's nodes in canonical order
template <int DIM, typename F> __device__ __forceinline__ void own_each(const Geo& g, const KrylovOwn& o, F&& f) {
  const OwnWalk wk = own_walk<DIM>(g, o);
  for (int i = 0; i < wk.count; ++i) f(own_node_at<DIM>(g, o, wk, i));
}
template <typename T> struct KPair { T a, b; };
template <typename T> struct KTriple { T a, b, c; };
template <typename T> struct KQuad { T a, b, c, d; };
template <typename T, int DIM> struct StencilLoads { T v[2 * DIM + 1]; };
// y = A p at this thread's nodes in canonical order, f(c, (A p)[c]); same operation order as apply_stencil_fma
template <typename T, int DIM, typename F>
__device__ __forceinline__ void own_spmv(const Geo& g, const KrylovOwn& o, const Stencil<T>& st,
                                         const T* __restrict__ p, F&& f) {
  own_batched<DIM, StencilLoads<T, DIM>>(g, o,
    [&](long c) {
      StencilLoads<T, DIM> s;
      s.v[0] = p[c];
      s.v[1] = p[c - 1];
      s.v[2] = p[c + 1];
      s.v[3] = p[c - g.pitch];
      s.v[4] = p[c + g.pitch];
      if constexpr (DIM == 3) {
        s.v[5] = p[c - g.plane];
        s.v[6] = p[c + g.plane];
      }
      return s;
    },
    [&](long c, const StencilLoads<T, DIM>& s) {
      T a = sc_mul(st.c[0], s.v[0]);
#pragma unroll
      for (int e = 1; e < 2 * DIM + 1; ++e) a = sc_fma(st.c[e], s.v[e], a);
      f(c, a, s.v[0]);
    });
}

template <typename T, bool ROBIN>
__device__ __forceinline__ void wg_apply_bc(const Geo& g, T* __restrict__ v, double kh) {
  if constexpr (ROBIN) {
    __syncthreads();
    const cplx den{1.0, -kh};
    for (int j = 1 + threadIdx.x; j < g.n - 1; j += blockDim.x) {
      const long row = (long)g.pitch * j;
      v[row] = sc_div(v[row + 1], den);
      v[row + g.n - 1] = sc_div(v[row + g.n - 2], den);
    }
  }
  __syncthreads();
}

// ---- CG, vectors in global memory (L2-resident): any grid size ----------------------------------------------------
// info[0] = iterations performed; info[2] accumulates them over the solves of one evaluation (time model: evomg_time_model)
template <typename T, int DIM>
__global__ void __launch_bounds__(EVOMG_KRYLOV_THREADS)
evomg_cg_single_wg(Geo g, Stencil<T> st, T* __restrict__ x, const T* __restrict__ b, T* __restrict__ r,
                   T* __restrict__ p, T* __restrict__ ap, int maxit, double tol, int* __restrict__ info) {
  prio_latency();
  __shared__ T red[17];
  const KrylovOwn own = krylov_own<DIM>(g.n, !scalar_traits<T>::is_complex);
  const T zero = scalar_traits<T>::zero();
  T loc = zero;
  own_each<DIM>(g, own, [&](long c) {
    const T v = sc_sub(b[c], apply_stencil_fma<T, DIM>(st, x, g, c));
    r[c] = v;
    p[c] = v;
    loc = sc_fma(v, v, loc);
  });
  T rr = block_sum(loc, red);
  const double res0 = sc_norm_from_dot(rr);
  if (res0 == 0.0) {
    if (threadIdx.x == 0) info[0] = 0;
    return;
  }
  __syncthreads();
  int it = 0;
  for (; it < maxit; ++it) {
    loc = zero;
    own_spmv<T, DIM>(g, own, st, p, [&](long c, T v, T pc) {
      ap[c] = v;
      loc = sc_fma(pc, v, loc);
    });
    const T pap = block_sum(loc, red);
    const T alpha = sc_div(rr, pap);
    const T nalpha = sc_neg(alpha);
    loc = zero;
    own_batched<DIM, KQuad<T>>(g, own,
      [&](long c) { return KQuad<T>{p[c], x[c], ap[c], r[c]}; },
      [&](long c, const KQuad<T>& q) {
        x[c] = sc_fma(alpha, q.a, q.b);
        const T rv = sc_fma(nalpha, q.c, q.d);
        r[c] = rv;
        loc = sc_fma(rv, rv, loc);
      });
    const T rrn = block_sum(loc, red);
    const double res = sc_norm_from_dot(rrn);
    if (!(res > tol * res0)) { ++it; break; }
    const T beta = sc_div(rrn, rr);
    own_batched<DIM, KPair<T>>(g, own,
      [&](long c) { return KPair<T>{p[c], r[c]}; },
      [&](long c, const KPair<T>& q) { p[c] = sc_fma(beta, q.a, q.b); });
    rr = rrn;
    __syncthreads();
  }
  if (threadIdx.x == 0) { info[0] = it; info[2] += it; }      // [2]: iterations of all coarse solves since the host last cleared it
}

template <typename T, int DIM>
__device__ __forceinline__ T wg_dot(const Geo& g, const KrylovOwn& own, const T* __restrict__ a,
                                    const T* __restrict__ b, T* red) {
  T loc = scalar_traits<T>::zero();
  own_batched<DIM, KPair<T>>(g, own,
    [&](long c) { return KPair<T>{a[c], b[c]}; },
    [&](long, const KPair<T>& q) { loc = sc_fma(q.a, q.b, loc); });
  return block_sum(loc, red);
}

// ---- BiCGStab, vectors in global memory; ROBIN: Helmholtz boundary handling of p, s, x, r ---------------------------
template <typename T, int DIM, bool ROBIN>
__global__ void __launch_bounds__(EVOMG_KRYLOV_THREADS)
evomg_bicgstab_single_wg(Geo g, Stencil<T> st, T* __restrict__ x, const T* __restrict__ f, T* __restrict__ r,
                         T* __restrict__ p, T* __restrict__ nu, T* __restrict__ hh, T* __restrict__ s,
                         T* __restrict__ t, T* __restrict__ rhat, int maxit, double tol, double kh,
                         int* __restrict__ info) {
  prio_latency();
  __shared__ T red[17];
  const KrylovOwn own = krylov_own<DIM>(g.n);
  const T zero = scalar_traits<T>::zero();
  T loc = zero;
  own_each<DIM>(g, own, [&](long c) {
    const T v = sc_sub(f[c], apply_stencil_fma<T, DIM>(st, x, g, c));
    r[c] = v;
    rhat[c] = v;
    nu[c] = zero;
    p[c] = zero;
    loc = sc_fma(v, v, loc);
  });
  const double res0 = sc_norm_from_dot(block_sum(loc, red));
  wg_apply_bc<T, ROBIN>(g, r, kh);
  // rhat is a plain copy of r including its boundary values; p and nu start as zero everywhere
  if constexpr (ROBIN) {
    for (int j = 1 + threadIdx.x; j < g.n - 1; j += blockDim.x) {
      const long row = (long)g.pitch * j;
      rhat[row] = r[row];
      rhat[row + g.n - 1] = r[row + g.n - 1];
      p[row] = zero; p[row + g.n - 1] = zero;
      nu[row] = zero; nu[row + g.n - 1] = zero;
    }
    __syncthreads();
  }
  if (res0 == 0.0) {
    if (threadIdx.x == 0) info[0] = 0;
    return;
  }
  T alpha = scalar_traits<T>::from_real(1.0), beta, rho, rho_new = scalar_traits<T>::from_real(1.0),
    omega = scalar_traits<T>::from_real(1.0);
  int it = 0;
  for (; it < maxit; ++it) {
    rho = rho_new;
    rho_new = wg_dot<T, DIM>(g, own, rhat, r, red);
    beta = sc_mul(sc_div(rho_new, rho), sc_div(alpha, omega));
    const T nomega = sc_neg(omega);
    own_batched<DIM, KTriple<T>>(g, own,
      [&](long c) { return KTriple<T>{nu[c], p[c], r[c]}; },
      [&](long c, const KTriple<T>& q) {
        const T pv = sc_fma(nomega, q.a, q.b);
        p[c] = sc_fma(beta, pv, q.c);
      });
    wg_apply_bc<T, ROBIN>(g, p, kh);
    own_spmv<T, DIM>(g, own, st, p, [&](long c, T v, T) { nu[c] = v; });
    __syncthreads();
    alpha = sc_div(rho_new, wg_dot<T, DIM>(g, own, rhat, nu, red));
    const T nalpha = sc_neg(alpha);
    own_batched<DIM, KQuad<T>>(g, own,
      [&](long c) { return KQuad<T>{p[c], x[c], nu[c], r[c]}; },
      [&](long c, const KQuad<T>& q) {
        hh[c] = sc_fma(alpha, q.a, q.b);
        s[c] = sc_fma(nalpha, q.c, q.d);
      });
    wg_apply_bc<T, ROBIN>(g, s, kh);
    own_spmv<T, DIM>(g, own, st, s, [&](long c, T v, T) { t[c] = v; });
    __syncthreads();
    const T ts = wg_dot<T, DIM>(g, own, t, s, red);
    const T tt = wg_dot<T, DIM>(g, own, t, t, red);
    omega = sc_div(ts, tt);
    const T nomega2 = sc_neg(omega);
    loc = zero;
    own_batched<DIM, KTriple<T>>(g, own,
      [&](long c) { return KTriple<T>{s[c], hh[c], t[c]}; },
      [&](long c, const KTriple<T>& q) {
        x[c] = sc_fma(omega, q.a, q.b);
        const T rv = sc_fma(nomega2, q.c, q.a);
        r[c] = rv;
        loc = sc_fma(rv, rv, loc);
      });
    const double res = sc_norm_from_dot(block_sum(loc, red));
    wg_apply_bc<T, ROBIN>(g, x, kh);
    wg_apply_bc<T, ROBIN>(g, r, kh);
    if (res < tol * res0) { ++it; break; }
    if (!(res == res)) { ++it; break; }
  }
  if (threadIdx.x == 0) { info[0] = it; info[2] += it; }      // [2]: iterations of all coarse solves since the host last cleared it
}

// ---- BiCGStab on 2D grids up to 9^2 in ONE wave ------------------------------------------------------------------------
// The reference's Helmholtz example ends on 9^2 (knowledge: levels 3..7): 49 unknowns, two solves per outer iteration; the
// 512-thread kernel above spends 42 us per solve on barriers and L2 round trips. Canonical ownership there (tile, m <= 8):
// thread (wave w, lane l) owns node (column l, row w), ONE node; sums = tree over a wave's columns, then tree over the
// wave sums. Here the same nodes sit in one wave as lane = 8 * column + row: the wave tree's offsets 32, 16, 8 then add
// columns 4, 2, 1 apart and its offsets 4, 2, 1 rows 4, 2, 1 apart — exactly that order (the zeros the big kernel adds
// are exact), so nothing changes for the oracle. All eight vectors are one register value per lane; west / east are
// the lanes 8 away (the Robin columns are the own value over (1 - i k h)), south / north the adjacent lanes (row 7 of a
// column never holds a node: its zero is the Dirichlet row).
template <typename T> __device__ __forceinline__ T lane_value(T v, int src);
template <> __device__ __forceinline__ double lane_value<double>(double v, int src) { return __shfl(v, src, 64); }
template <> __device__ __forceinline__ cplx lane_value<cplx>(cplx v, int src) { return cplx{__shfl(v.re, src, 64), __shfl(v.im, src, 64)}; }

// XZERO (the COPY to the solver's right-hand side and the SET_ZERO of x in front of the solve folded in, as in
// evomg_cg_wave2d): f is the copy's source, the copy itself is written to copy_dst, x is not read — its inner nodes and
// Robin columns are zero by the SET_ZERO, its Dirichlet rows by the caller's bookkeeping — so r = f.
// (the body is a device function: evomg_lower2d.h runs it on LDS images from the first wave of its workgroup)
template <typename T, bool ROBIN, bool XZERO = false>
__device__ __forceinline__ void
evomg_bicgstab_wave2d_body(const Geo& g, const Stencil<T>& st, T* __restrict__ x, const T* __restrict__ f, T* __restrict__ r_out, int maxit,
                           double tol, double kh, int* __restrict__ info, T* __restrict__ copy_dst = nullptr) {
  const int m = g.n - 2;                         // host-checked: 1 <= m <= 7
  const int l = threadIdx.x, col = l >> 3, row = l & 7;
  const bool own = col < m && row < m;
  const long c = gidx<2>(g, min(col, m - 1) + 1, min(row, m - 1) + 1, 0);
  const T zero = scalar_traits<T>::zero();
  const cplx den{1.0, -kh};
  // value of a neighbouring column's node (same row); beyond the first / last inner column: the boundary value
  auto robin = [&](T own_value) -> T {
    if constexpr (ROBIN && scalar_traits<T>::is_complex) return sc_div(own_value, den);
    else return zero;
  };
  auto apply = [&](T pc) -> T {                  // (A p) at the lane's node, operation order of apply_stencil_fma
    const T from_w = lane_value<T>(pc, (l + 56) & 63), from_e = lane_value<T>(pc, (l + 8) & 63);
    const T from_s = lane_value<T>(pc, (l + 63) & 63), from_n = lane_value<T>(pc, (l + 1) & 63);
    const T pw = col == 0 ? robin(pc) : from_w;
    const T pe = col == m - 1 ? robin(pc) : from_e;
    const T ps = row == 0 ? zero : from_s;
    const T pn = row == m - 1 ? zero : from_n;
    T a = sc_mul(st.c[0], pc);
    a = sc_fma(st.c[1], pw, a);
    a = sc_fma(st.c[2], pe, a);
    a = sc_fma(st.c[3], ps, a);
    a = sc_fma(st.c[4], pn, a);
    return own ? a : zero;
  };
  auto dot = [&](T a, T b) -> T { return first_lane(wave_sum_fast(sc_fma(a, b, zero))); };
  T xv = zero, r;
  if constexpr (XZERO) {
    for (int q = l; q < g.n * g.n; q += 64) {
      const long cq = gidx<2>(g, q % g.n, q / g.n, 0);
      copy_dst[cq] = f[cq];
    }
    r = own ? f[c] : zero;
  } else {
    xv = own ? x[c] : zero;
    r = own ? sc_sub(f[c], apply_stencil_fma<T, 2>(st, x, g, c)) : zero;
  }
  const T rhat = r;
  T nu = zero, p = zero, hh = zero, s = zero, tt = zero;
  const double res0 = sc_norm_from_dot(dot(r, r));
  int it = 0;
  if (res0 != 0.0) {
    T alpha = scalar_traits<T>::from_real(1.0), beta, rho, rho_new = scalar_traits<T>::from_real(1.0),
      omega = scalar_traits<T>::from_real(1.0);
    for (; it < maxit; ++it) {
      rho = rho_new;
      rho_new = dot(rhat, r);
      beta = sc_mul(sc_div(rho_new, rho), sc_div(alpha, omega));
      const T pv = sc_fma(sc_neg(omega), nu, p);
      p = own ? sc_fma(beta, pv, r) : zero;
      nu = apply(p);
      alpha = sc_div(rho_new, dot(rhat, nu));
      hh = sc_fma(alpha, p, xv);
      s = own ? sc_fma(sc_neg(alpha), nu, r) : zero;
      tt = apply(s);
      const T ts = dot(tt, s);
      const T t2 = dot(tt, tt);
      omega = sc_div(ts, t2);
      xv = sc_fma(omega, s, hh);
      r = own ? sc_fma(sc_neg(omega), tt, s) : zero;
      const double res = sc_norm_from_dot(dot(r, r));
      if (res < tol * res0) { ++it; break; }
      if (!(res == res)) { ++it; break; }
    }
  }
  if (own) {
    x[c] = xv;
    r_out[c] = r;
    if constexpr (ROBIN && scalar_traits<T>::is_complex) {
      // the boundary columns of r and — once an iteration has run — of x, as wg_apply_bc leaves them
      // (XZERO without an iteration: the SET_ZERO's own boundary pass left zeros there)
      if (col == 0) { r_out[c - 1] = sc_div(r, den); if (it > 0) x[c - 1] = sc_div(xv, den); else if (XZERO) x[c - 1] = zero; }
      if (col == m - 1) { r_out[c + 1] = sc_div(r, den); if (it > 0) x[c + 1] = sc_div(xv, den); else if (XZERO) x[c + 1] = zero; }
    }
  }
  if (l == 0) { info[0] = it; info[2] += it; }      // [2]: iterations of all coarse solves since the host last cleared it
}
template <typename T, bool ROBIN, bool XZERO = false>
__global__ void __launch_bounds__(64)
evomg_bicgstab_wave2d(Geo g, Stencil<T> st, T* __restrict__ x, const T* __restrict__ f, T* __restrict__ r_out, int maxit, double tol,
                      double kh, int* __restrict__ info, T* __restrict__ copy_dst = nullptr) {
  prio_latency();
  evomg_bicgstab_wave2d_body<T, ROBIN, XZERO>(g, st, x, f, r_out, maxit, tol, kh, info, copy_dst);
}

// ---- CG held in registers (2D tile ownership) ------------------------------------------------------------------------
// For coarsest grids up to 129^2 (real) one 512-thread workgroup (8 waves = 8 bands of rows) runs the whole solve out
// of registers:
//   * a thread owns R rows of E columns (E = 2: the column pair of a "pair" tile, R = 16 at 129^2; E = 1 up to 65^2)
//     and keeps p, r and A p of all of them in registers (192 of its 256 VGPRs); south / north neighbours are the
//     thread's own previous / next row;
//   * west / east neighbours are the lane's other element or live in the adjacent lane: ONE DPP wave shift per
//     element pair and direction, no memory access (an LDS-resident p was bound by LDS bandwidth: 6 accesses x 8 B
//     per node and iteration = 2.5 us of a 3.4 us iteration; single-column lanes with DPP were VALU bound by the
//     shifts). What is left in LDS:
//       - x, thread-private, touched once per iteration (x += alpha p): LDS as spill space, conflict-free layout. Its
//         round trip (ds_write is 6 LDS cycles per 8 bytes and lane) stood alone in the r update as 30 % of an iteration;
//         it now rides one row ahead of the stencil in the A p pass of the NEXT iteration together with p = r + beta p
//         (both need the old p), where the arithmetic hides it: 129^2 2.6 -> 2.1 us per iteration (tools/cg129_time.py;
//         one row of lead and groups of four rows measured best; rows split between the two passes, x in registers
//         with A p in LDS instead (spills) measured worse);
//       - the band seams: every wave publishes its first and last row of r before the r . r reduction and reads the
//         neighbouring bands' rows after it (2 KB); the neighbours' p rows are recomputed, not exchanged, so an
//         iteration has exactly two barriers (the two reductions);
//   * column-less elements and slack rows keep p = r = 0 by construction: their alpha is zero (per-lane copy of the
//     step length), and A p of slack rows is forced to zero; nothing else needs masking;
//   * reductions: permlane-swap / DPP wave trees and one barrier each (block_sum_fast).
// Arithmetic and summation order are those of evomg_cg_single_wg with the same ownership, and of the oracle.
// LDS (elements of T): 32 reduction slots | row seams [8 waves][first, last][E][64] | x [R][512][E].
__host__ __device__ inline long cg_reg2d_lds_elems(int rt, int e) { return 32 + 8L * 2 * e * 64 + (long)rt * e * 512; }

// MASK_FROM: first unrolled row that may lie outside the thread's band (needs masking); rows below it are always
// rows of the grid for every wave (host-checked: own.R == R and the last band keeps at least MASK_FROM rows).
// POW2 (real T; host: cg_reg2d_pow2): the four off-centre coefficients are one value cc = +-2^k and the centre is
// c0 = q0 cc with q0 = +-2^j (every Poisson level with h = 2^-L: cc = -4^L, q0 = -4). Then the stencil sum's roundings
// commute with the scaling by cc — c0 p is exact, round(cc w + c0 p) = cc round(w + q0 p), and so on — so the pass keeps
// t = (A p) / cc instead of A p: one fma and three additions per node instead of a product and four fmas. Everything
// downstream scales exactly too: the partial sums of p . t are those of p . A p divided by cc (every fma and every
// addition of the reduction tree rounds the scaled operands to the scaled result), rr / (p . t) = cc alpha is the
// correctly rounded quotient scaled, alpha itself is recovered by one exact product per thread, and r - (cc alpha) t
// is the very real number r - alpha (A p). Bit-identical to the general form (barring overflow / underflow of values
// near 1e+-300, which a residual norm never reaches); 32 of ~550 vector instructions per wave and iteration fewer.
// XZERO / b_copy (round 4): the COPY `gen_rhs = RHS` and the SET_ZERO of the solution in front of the solve, folded in as in
// the one-wave kernels below — b is the copy's source, the workgroup writes the copy (the whole array, padding included),
// the solve starts from x = 0 without reading x (r = b): two launches per cycle fewer.
template <typename T, int R, int E, int MASK_FROM, bool POW2 = false, bool XZERO = false>
__global__ void __launch_bounds__(512)
evomg_cg_reg2d(Geo g, Stencil<T> st, T* __restrict__ x, const T* __restrict__ b, T* __restrict__ r_out, int maxit,
               double tol, int* __restrict__ info, T* __restrict__ b_copy = nullptr) {
  prio_latency();
  extern __shared__ __align__(16) unsigned char lds_raw[];
  T* red = reinterpret_cast<T*>(lds_raw);          // 2 x 16 wave sums (double-buffered)
  T* row_seam = red + 32;                          // [wave][0: first row, 1: row R-1][element][lane]
  T* xs = row_seam + 8 * 2 * E * 64;               // [row * E + element][thread]
  const KrylovOwn own = krylov_own<2>(g.n, !scalar_traits<T>::is_complex);
  const T zero = scalar_traits<T>::zero();
  const int tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63;
  const int band = w;                              // host-checked: own.bands == 8 waves, E == (own.pair ? 2 : 1)
  const int iy0 = band * own.R;
  const int rows = max(0, min(own.R, own.m - iy0));   // wave-uniform
  const int ix0 = E * l;
  bool ok_e[E];
#pragma unroll
  for (int e = 0; e < E; ++e) ok_e[e] = ix0 + e < own.m;
  const bool has_south = band > 0, has_north = band + 1 < own.bands;       // wave-uniform
  const T* south_src = row_seam + (((w - 1) * 2 + 1) * E) * 64 + l;          // row R-1 of the band below
  const T* north_src = row_seam + (((w + 1) * 2) * E) * 64 + l;              // first row of the band above
  T* const my_seam = row_seam + ((w * 2) * E) * 64 + l;
  // a thread's x values of one row sit side by side: with E = 2 real elements one 16-byte LDS access per row and
  // direction (ds_read_b128 is 4 LDS cycles where two ds_read_b64 issued as ds_read2st64_b64 are 8; the store 13 either way)
  struct alignas(sizeof(T) * E) XRow { T v[E]; };
  XRow* const myx = reinterpret_cast<XRow*>(xs) + tid;     // [row][thread]
  const long c0 = gidx<2>(g, min(ix0, own.m - 1) + 1, min(iy0, own.m - 1) + 1, 0);

  if (b_copy) {
    const long elems = (long)g.pitch * g.n;
    for (long q = tid; q < elems; q += 512) b_copy[q] = b[q];
  }
  T r[R][E], p[R][E], ap[R][E];
  T loc = zero;
  // branch-free: slots that are not nodes of this thread read a clamped (valid) address and are masked to zero
#pragma unroll
  for (int i = 0; i < R; ++i) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const bool ok = ok_e[e] && i < rows;
      const long c = c0 + (long)g.pitch * (i < rows ? i : 0) + (ok_e[e] ? e : 0);
      T v;
      if constexpr (XZERO) {
        v = ok ? b[c] : zero;
        myx[i * 512].v[e] = zero;
      } else {
        v = ok ? sc_sub(b[c], apply_stencil_fma<T, 2>(st, x, g, c)) : zero;
        myx[i * 512].v[e] = x[c];
      }
      r[i][e] = v;
      p[i][e] = v;
      loc = sc_fma(v, v, loc);
    }
    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  // Band seams: what crosses waves is the first and the last row of r, published before the r . r reduction (its
  // barrier makes them visible). The neighbouring bands' rows of p are never exchanged: p_new = r + beta p_old is
  // recomputed from the neighbour's r row and the previous neighbour p row with the very fused multiply-add the
  // owner executes (bit-identical), which saves the barrier a p exchange would need after every update of p.
  auto publish_seams = [&]() {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      my_seam[e * 64] = r[0][e];
      my_seam[(E + e) * 64] = r[R - 1][e];
    }
  };
  T p_south[E], p_north[E], r_south[E], r_north[E];
  auto read_seams = [&]() {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      r_south[e] = has_south ? south_src[e * 64] : zero;
      r_north[e] = has_north ? north_src[e * 64] : zero;
    }
  };
  publish_seams();
  int red_phase = 0;
  T rr = block_sum_fast(loc, red, red_phase);      // its barrier also publishes the seams
  read_seams();
#pragma unroll
  for (int e = 0; e < E; ++e) { p_south[e] = r_south[e]; p_north[e] = r_north[e]; }
  const double res0 = sc_norm_from_dot(rr);
  T alpha_e[E], nalpha_e[E];                       // zero step for column-less elements: their r, p and x never move
  T q0 = zero, inv_cc = zero;
  if constexpr (POW2) { q0 = st.c[0] / st.c[1]; inv_cc = 1.0 / st.c[1]; }      // exact: powers of two
  // A p (kept in registers) and p . A p. With UPD the two updates that use the OLD p ride along, one row ahead of the
  // stencil: x += alpha p (the LDS round trip of x hides behind the stencil's arithmetic instead of standing alone in
  // the r update: it was 30 % of the iteration there) and p = r + beta p, own rows and the neighbouring bands' seam rows.
  auto apply_pass = [&](auto upd, T beta) {
    constexpr bool UPD = decltype(upd)::value;
    loc = zero;
    auto update_row = [&](int j) {
      XRow xv = myx[j * 512];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        xv.v[e] = sc_fma(alpha_e[e], p[j][e], xv.v[e]);
        p[j][e] = sc_fma(beta, p[j][e], r[j][e]);
      }
      myx[j * 512] = xv;
    };
    if constexpr (UPD) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        p_south[e] = sc_fma(beta, p_south[e], r_south[e]);
        p_north[e] = sc_fma(beta, p_north[e], r_north[e]);
      }
      update_row(0);
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if constexpr (UPD) {
        if (i + 1 < R) update_row(i + 1);
      }
      const T from_w = from_left(p[i][E - 1]);      // west of the lane's first element
      const T from_e = from_right(p[i][0]);         // east of the lane's last element
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const T pc = p[i][e];
        const T pw = (e == 0) ? from_w : p[i][e > 0 ? e - 1 : 0];
        const T pe = (e == E - 1) ? from_e : p[i][e < E - 1 ? e + 1 : e];
        const T ps = (i == 0) ? p_south[e] : p[i > 0 ? i - 1 : 0][e];
        const T pn = (i == R - 1) ? p_north[e] : p[i < R - 1 ? i + 1 : i][e];
        T a;
        if constexpr (POW2) {
          a = sc_fma(q0, pc, pw);                        // (A p) / cc, see above
          a = sc_add(a, pe);
          a = sc_add(a, ps);
          a = sc_add(a, pn);
        } else {
          a = sc_mul(st.c[0], pc);
          a = sc_fma(st.c[1], pw, a);
          a = sc_fma(st.c[2], pe, a);
          a = sc_fma(st.c[3], ps, a);
          a = sc_fma(st.c[4], pn, a);
        }
        if (i >= MASK_FROM) a = (i < rows) ? a : zero;   // slack row: p is zero there but the stencil sum is not
        ap[i][e] = a;
        loc = sc_fma(pc, a, loc);                        // column-less elements: p = 0, contributes exactly 0
      }
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the scheduler look-ahead (VGPR budget)
    }
  };
  int it = 0;
  if (res0 != 0.0 && maxit > 0) {
    apply_pass(std::false_type{}, zero);
    for (;;) {
      const T pap = block_sum_fast(loc, red, red_phase);
      const T alpha_s = sc_div(rr, pap);                 // POW2: cc alpha (ap holds (A p) / cc)
      T alpha = alpha_s;
      if constexpr (POW2) alpha = sc_mul(alpha_s, inv_cc);
      const T nalpha = sc_neg(alpha_s);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        alpha_e[e] = ok_e[e] ? alpha : zero;
        nalpha_e[e] = ok_e[e] ? nalpha : zero;
      }
      // ---- r -= alpha A p ; r . r ----
      loc = zero;
#pragma unroll
      for (int i = 0; i < R; ++i) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const T rv = sc_fma(nalpha_e[e], ap[i][e], r[i][e]);     // slack rows: A p = 0, r stays 0
          r[i][e] = rv;
          loc = sc_fma(rv, rv, loc);
        }
      }
      // the previous r seams were read right after the previous r . r reduction, i.e. before every wave passed the
      // p . A p barrier of this iteration: overwriting them here is safe
      publish_seams();
      const T rrn = block_sum_fast(loc, red, red_phase);    // its barrier publishes the seams
      read_seams();
      const double res = sc_norm_from_dot(rrn);
      ++it;
      if (!(res > tol * res0) || it >= maxit) {
        // the last step of x (the iterations before it took theirs inside apply_pass)
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
          for (int e = 0; e < E; ++e) myx[i * 512].v[e] = sc_fma(alpha_e[e], p[i][e], myx[i * 512].v[e]);
        break;
      }
      // ---- x += alpha p ; p = r + beta p ; A p ; p . A p ----
      apply_pass(std::true_type{}, sc_div(rrn, rr));
      rr = rrn;
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (ok_e[e] && i < rows) {
        const long c = c0 + (long)g.pitch * i + e;
        x[c] = myx[i * 512].v[e];
        r_out[c] = r[i][e];
      }
  if (threadIdx.x == 0) { info[0] = it; info[2] += it; }      // [2]: iterations of all coarse solves since the host last cleared it
}

// ---- CG on grids up to 33^2 in ONE wave -------------------------------------------------------------------------------
// The reference's default level range (knowledge: 5..9) ends on 33^2: 961 unknowns, ~120 iterations per solve, 12 solves
// per candidate. With eight waves an iteration is a chain of two workgroup reductions (wave tree, LDS, barrier, tree of the
// wave sums), a square root and two divisions around 4 rows of work per thread: 0.92 us, nearly all of it latency. Here
// the grid lives in ONE wave: lane (h, column) keeps p, r, A p and x of its 16 rows in registers, west / east are one
// DPP wave shift (lanes 31 and 63 hold the zero column between and after the halves), the row seam between the halves is
// one v_permlane32_swap per iteration, and a reduction is the wave tree alone — no LDS, no barrier.
// XZERO: the solve starts from x = 0 (the SET_ZERO in front of it is folded in): r = b, nothing of x is read.
// b_copy: the COPY `gen_rhs = RHS` in front of the solve, folded in (the whole array, padding included, as the copy op).
// Ownership / summation order: krylov_own().wave, restated by the oracle (own_make: wave).
// MASK_FROM: first unrolled row that may be a slack row in either half (host: min over the halves' row counts, or 0);
// rows below it are rows of the grid in both halves and need no masking (33^2: 15 — only the upper half's last row is
// slack). Column-less lanes need none either: their stencil coefficients are zeroed, so A p = 0 there by construction.
// v = s * v + a in place (one v_fma_f64 with v as destination and second factor). For a live addend the compiler prefers
// "copy a, then v_fmac" into a fresh register and copies every row back at the loop head: 52 moves per iteration of the
// one-wave CG. Same instruction, same rounding as sc_fma(s, v, a).
__device__ __forceinline__ void scale_add_in_place(double& v, double s, double a) {
  asm("v_fma_f64 %0, %1, %0, %2" : "+v"(v) : "v"(s), "v"(a));
}
__device__ __forceinline__ void scale_add_in_place(cplx& v, cplx s, cplx a) { v = sc_fma(s, v, a); }

// POW2: the power-of-two form of evomg_cg_reg2d (A p kept divided by the off-centre coefficient; the step lengths of
// column-less lanes are zeroed instead of their coefficients) — same bits, 16 of 342 vector instructions per iteration fewer.
template <typename T, bool XZERO, int MASK_FROM, bool POW2 = false>
__device__ __forceinline__ void
evomg_cg_wave2d_body(const Geo& g, const Stencil<T>& st, T* __restrict__ x, const T* __restrict__ b, T* __restrict__ r_out, T* __restrict__ b_copy,
                     int maxit, double tol, int* __restrict__ info) {
  constexpr int R = EVOMG_CG_WAVE_ROWS;
  const int l = threadIdx.x, h = l >> 5, ix = l & 31;
  const int m = g.n - 2;
  const bool col = ix < m;
  const int iy0 = h * R;
  const int rows = max(0, min(R, m - iy0));
  const T zero = scalar_traits<T>::zero();
  if (b_copy) {
    const long elems = (long)g.pitch * g.n;
    for (long q = l; q < elems; q += 64) b_copy[q] = b[q];
  }
  const long c0 = gidx<2>(g, min(ix, m - 1) + 1, min(iy0, m - 1) + 1, 0);
  Stencil<T> sl;                                   // this lane's coefficients: zero where the lane has no column
#pragma unroll
  for (int e = 0; e < 5; ++e) sl.c[e] = col ? st.c[e] : zero;
  T q0 = zero, inv_cc = zero;
  if constexpr (POW2) { q0 = st.c[0] / st.c[1]; inv_cc = 1.0 / st.c[1]; }      // exact: powers of two
  T r[R], p[R], ap[R], xv[R];
  T loc = zero;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const bool ok = col && i < rows;
    const long c = c0 + (long)g.pitch * (i < rows ? i : 0);
    T v;
    if constexpr (XZERO) {
      v = ok ? b[c] : zero;
      xv[i] = zero;
    } else {
      v = ok ? sc_sub(b[c], apply_stencil_fma<T, 2>(st, x, g, c)) : zero;
      xv[i] = x[c];
    }
    r[i] = v;
    p[i] = v;
    loc = sc_fma(v, v, loc);
  }
  T rr = first_lane(wave_sum_fast(loc));
  const double res0 = sc_norm_from_dot(rr);
  int it = 0;
  if (res0 != 0.0 && maxit > 0) {
    // p = r + beta p at the TOP of the next iteration (not behind the convergence test of this one): definition and uses of
    // p then share a block and the update happens in place — at the bottom the compiler wrote it to fresh registers and
    // copied all sixteen rows back at the loop head (52 moves of 387 vector instructions per iteration)
    T beta = zero;
    bool have_beta = false;
    for (;;) {
      if (have_beta) {
#pragma unroll
        for (int i = 0; i < R; ++i) scale_add_in_place(p[i], beta, r[i]);
      }
      // the row across the seam: the lower half needs the upper half's first row, the upper half the lower half's last
      const T mine = h ? p[0] : p[R - 1];
      const T theirs = lanes_swap32(mine);
      const T p_south = h ? theirs : zero, p_north = h ? zero : theirs;
      loc = zero;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const T pc = p[i];
        const T pw = from_left(pc), pe = from_right(pc);
        const T ps = (i == 0) ? p_south : p[i > 0 ? i - 1 : 0];
        const T pn = (i == R - 1) ? p_north : p[i < R - 1 ? i + 1 : i];
        T a;
        if constexpr (POW2) {
          a = sc_fma(q0, pc, pw);                        // (A p) / cc: see evomg_cg_reg2d
          a = sc_add(a, pe);
          a = sc_add(a, ps);
          a = sc_add(a, pn);
        } else {
          a = sc_mul(sl.c[0], pc);
          a = sc_fma(sl.c[1], pw, a);
          a = sc_fma(sl.c[2], pe, a);
          a = sc_fma(sl.c[3], ps, a);
          a = sc_fma(sl.c[4], pn, a);
        }
        if (i >= MASK_FROM) a = (i < rows) ? a : zero;   // slack rows: p = r = 0 for good (column-less lanes: sl = 0)
        ap[i] = a;
        loc = sc_fma(pc, a, loc);
      }
      const T pap = first_lane(wave_sum_fast(loc));
      const T alpha_s = sc_div(rr, pap);                 // POW2: cc alpha
      T alpha = alpha_s, nalpha = sc_neg(alpha_s);
      if constexpr (POW2) {
        // column-less lanes: p = 0 there, so their A p never enters the sums; a zero step keeps their r at zero
        alpha = sc_mul(alpha_s, inv_cc);
        nalpha = col ? nalpha : zero;
      }
      loc = zero;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        xv[i] = sc_fma(alpha, p[i], xv[i]);
        const T rv = sc_fma(nalpha, ap[i], r[i]);
        r[i] = rv;
        loc = sc_fma(rv, rv, loc);
      }
      const T rrn = first_lane(wave_sum_fast(loc));
      const double res = sc_norm_from_dot(rrn);
      ++it;
      if (!(res > tol * res0) || it >= maxit) break;
      beta = sc_div(rrn, rr);
      have_beta = true;
      rr = rrn;
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i)
    if (col && i < rows) {
      const long c = c0 + (long)g.pitch * i;
      x[c] = xv[i];
      r_out[c] = r[i];
    }
  if (l == 0) { info[0] = it; info[2] += it; }      // [2]: iterations of all coarse solves since the host last cleared it
}
template <typename T, bool XZERO, int MASK_FROM, bool POW2 = false>
__global__ void __launch_bounds__(64)
evomg_cg_wave2d(Geo g, Stencil<T> st, T* __restrict__ x, const T* __restrict__ b, T* __restrict__ r_out, T* __restrict__ b_copy,
                int maxit, double tol, int* __restrict__ info) {
  prio_latency();
  evomg_cg_wave2d_body<T, XZERO, MASK_FROM, POW2>(g, st, x, b, r_out, b_copy, maxit, tol, info);
}

// ---- CG on 3D grids up to 9^3 in ONE wave ---------------------------------------------------------------------------
// The reference's 3D example ends on 5^3 (knowledge: levels 2..6): 27 unknowns, for which the 1024-thread kernel above
// spends 30 us per solve on barriers and L2 round trips. Here lane t < m^2 owns the z-column (t % m, t / m) — the canonical
// "column" ownership, so nothing changes for the oracle: with m^2 <= 64 all partial sums live in wave 0 and the 16-wave
// tree of the big kernel only adds exact zeros — and keeps r, x, p, A p of its (up to 8) nodes in registers; in-plane
// neighbours come from an LDS image of p with a zero halo (one wave: the barrier is a formality), z neighbours are the
// lane's own registers; reductions are the wave tree alone.
#define EVOMG_CG_WAVE3D_MAX_M 8
template <typename T>
__global__ void __launch_bounds__(64)
evomg_cg_wave3d(Geo g, Stencil<T> st, T* __restrict__ x, const T* __restrict__ b, T* __restrict__ r_out, int maxit, double tol,
                int* __restrict__ info) {
  prio_latency();
  constexpr int S = EVOMG_CG_WAVE3D_MAX_M;
  __shared__ T img[S * (S + 2) * (S + 2)];
  const int m = g.n - 2, W = m + 2, PS = W * W;
  const int t = threadIdx.x;
  const bool col = t < m * m;
  const int ix = col ? t % m : 0, iy = col ? t / m : 0;
  const int o0 = col ? (iy + 1) * W + ix + 1 : 0;          // column-less lanes sit on the halo corner (always zero)
  const long c0 = gidx<3>(g, ix + 1, iy + 1, 1);
  const T zero = scalar_traits<T>::zero();
  for (int i = t; i < S * PS; i += 64) img[i] = zero;
  __syncthreads();
  T r[S], xr[S], pr[S], ap[S];
  T loc = zero;
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const bool ok = col && i < m;
    const long c = c0 + g.plane * (i < m ? i : 0);
    const T v = ok ? sc_sub(b[c], apply_stencil_fma<T, 3>(st, x, g, c)) : zero;
    r[i] = v;
    pr[i] = v;
    xr[i] = x[c];
    loc = sc_fma(v, v, loc);
  }
  T rr = first_lane(wave_sum_fast(loc));
  const double res0 = sc_norm_from_dot(rr);
  int it = 0;
  if (res0 != 0.0) {
    for (; it < maxit; ++it) {
#pragma unroll
      for (int i = 0; i < S; ++i)
        if (col && i < m) img[i * PS + o0] = pr[i];
      __syncthreads();
      loc = zero;
#pragma unroll
      for (int i = 0; i < S; ++i) {
        const bool ok = col && i < m;
        const T* q = img + (i < m ? i : 0) * PS + o0;
        const T pc = pr[i];
        const T pd = (i == 0) ? zero : pr[i > 0 ? i - 1 : 0];
        const T pu = (i + 1 < m && i + 1 < S) ? pr[i + 1 < S ? i + 1 : i] : zero;
        T a = sc_mul(st.c[0], pc);
        a = sc_fma(st.c[1], q[-1], a);
        a = sc_fma(st.c[2], q[1], a);
        a = sc_fma(st.c[3], q[-W], a);
        a = sc_fma(st.c[4], q[W], a);
        a = sc_fma(st.c[5], pd, a);
        a = sc_fma(st.c[6], pu, a);
        a = ok ? a : zero;
        ap[i] = a;
        loc = sc_fma(pc, a, loc);
      }
      const T pap = first_lane(wave_sum_fast(loc));
      const T alpha = sc_div(rr, pap);
      const T nalpha = sc_neg(alpha);
      loc = zero;
#pragma unroll
      for (int i = 0; i < S; ++i) {
        xr[i] = sc_fma(alpha, pr[i], xr[i]);
        const T rv = sc_fma(nalpha, ap[i], r[i]);
        r[i] = rv;
        loc = sc_fma(rv, rv, loc);
      }
      const T rrn = first_lane(wave_sum_fast(loc));
      const double res = sc_norm_from_dot(rrn);
      if (!(res > tol * res0)) { ++it; break; }
      const T beta = sc_div(rrn, rr);
#pragma unroll
      for (int i = 0; i < S; ++i) pr[i] = sc_fma(beta, pr[i], r[i]);
      rr = rrn;
      __syncthreads();                       // all reads of the image done before the next iteration overwrites it
    }
  }
#pragma unroll
  for (int i = 0; i < S; ++i)
    if (col && i < m) {
      const long c = c0 + g.plane * i;
      x[c] = xr[i];
      r_out[c] = r[i];
    }
  if (t == 0) { info[0] = it; info[2] += it; }      // [2]: iterations of all coarse solves since the host last cleared it
}

// ---- CG on LARGE coarsest grids as a sequence of many-unit launches ------------------------------------------------------
// A cycle over few levels leaves a big coarsest grid (2049^2 over four levels: 257^2; 257^3 over three: 65^3). One
// workgroup then streams every vector through ONE compute unit's 64 B/clk, hundreds of iterations long: 62 ms per
// solve at 257^2, 174 ms at 65^3 (tools/coarse_big.py) — a cliff, not a slope. From 257^2 / 65^3 up the solve is three
// ordinary launches per iteration instead, all compute units busy:
//   apply:   A p -> ap, partial sums of p . A p                                   (kcg_apply)
//   update:  every block reduces those partials itself (same order as evomg_dot_final), alpha = rr / pAp,
//            x += alpha p, r -= alpha A p, partial sums of r . r                  (kcg_update)
//   direct:  every block reduces them, tests convergence, beta = rr' / rr, p = r + beta p; block 0 publishes the next
//            state (rr, iteration count, done flag)                               (kcg_direct)
// The scalars never visit the host: they live in a two-slot state (slot = iteration parity, so that a block still reading
// rr is never overtaken by block 0 writing the next one); the host queues 16 iterations at a time and looks at the done
// flag between batches; kernels of iterations past convergence return at once. Arithmetic: the Krylov kernels' (fma
// SpMV, fma updates); dot products: fma accumulation in evomg_dot_partial's node order, then the block trees, then
// evomg_dot_final's order over the partials — restated by the oracle (FN(dot_blocked_fma)).
struct KcgState {
  double rr, res0;
  int it, done;
};
template <int DIM> __host__ __device__ inline bool krylov_cg_per_launch(int n) { return DIM == 2 ? n >= 257 : n >= 65; }

template <typename T> __device__ __forceinline__ T kcg_reduce(const T* __restrict__ partial, int count, T* red) {
  T s = scalar_traits<T>::zero();
  for (int i = threadIdx.x; i < count; i += blockDim.x) s = sc_add(s, partial[i]);
  return block_sum(s, red);
}
// visits this block's inner nodes in evomg_dot_partial's order
template <int DIM, typename F> __device__ __forceinline__ void kcg_nodes(const Geo& g, F&& f) {
  const int m = g.n - 2;
  const long rows = (DIM == 2) ? m : (long)m * m;
  for (long r = blockIdx.x; r < rows; r += gridDim.x) {
    const int j = 1 + (int)(r % m), k = (DIM == 3) ? 1 + (int)(r / m) : 0;
    const long base = gidx<DIM>(g, 0, j, k);
    for (int i = 1 + threadIdx.x; i <= m; i += blockDim.x) f(base + i);
  }
}
template <typename T, int DIM>
__global__ void __launch_bounds__(256) kcg_init(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ b, T* __restrict__ r,
                                                T* __restrict__ p, T* __restrict__ partial) {
  __shared__ T red[17];
  T s = scalar_traits<T>::zero();
  kcg_nodes<DIM>(g, [&](long c) {
    const T v = sc_sub(b[c], apply_stencil_fma<T, DIM>(st, x, g, c));
    r[c] = v;
    p[c] = v;
    s = sc_fma(v, v, s);
  });
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
template <typename T>
__global__ void __launch_bounds__(256) kcg_start(const T* __restrict__ partial, int count, KcgState* state) {
  __shared__ T red[17];
  const T rr = kcg_reduce(partial, count, red);
  if (threadIdx.x == 0) {
    const double res0 = sc_norm_from_dot(rr);
    state[0] = KcgState{rr, res0, 0, res0 == 0.0 ? 1 : 0};
  }
}
template <typename T, int DIM>
__global__ void __launch_bounds__(256) kcg_apply(Geo g, Stencil<T> st, const T* __restrict__ p, T* __restrict__ ap, T* __restrict__ partial,
                                                 const KcgState* state, int cur) {
  __shared__ T red[17];
  if (state[cur].done) return;
  T s = scalar_traits<T>::zero();
  kcg_nodes<DIM>(g, [&](long c) {
    const T v = apply_stencil_fma<T, DIM>(st, p, g, c);
    ap[c] = v;
    s = sc_fma(p[c], v, s);
  });
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
template <typename T, int DIM>
__global__ void __launch_bounds__(256) kcg_update(Geo g, T* __restrict__ x, T* __restrict__ r, const T* __restrict__ p, const T* __restrict__ ap,
                                                  const T* __restrict__ partial_in, int count, T* __restrict__ partial_out,
                                                  const KcgState* state, int cur) {
  __shared__ T red[17];
  if (state[cur].done) return;
  const T pap = kcg_reduce(partial_in, count, red);
  const T alpha = sc_div(state[cur].rr, pap);
  const T nalpha = sc_neg(alpha);
  T s = scalar_traits<T>::zero();
  kcg_nodes<DIM>(g, [&](long c) {
    x[c] = sc_fma(alpha, p[c], x[c]);
    const T rv = sc_fma(nalpha, ap[c], r[c]);
    r[c] = rv;
    s = sc_fma(rv, rv, s);
  });
  __syncthreads();
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial_out[blockIdx.x] = s;
}
template <typename T, int DIM>
__global__ void __launch_bounds__(256) kcg_direct(Geo g, const T* __restrict__ r, T* __restrict__ p, const T* __restrict__ partial_in, int count,
                                                  KcgState* state, int cur, double tol) {
  __shared__ T red[17];
  const KcgState st0 = state[cur];
  if (st0.done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) state[cur ^ 1] = st0;       // hand the final state on
    return;
  }
  const T rrn = kcg_reduce(partial_in, count, red);
  const double res = sc_norm_from_dot(rrn);
  const bool converged = !(res > tol * st0.res0);
  if (!converged) {
    const T beta = sc_div(rrn, st0.rr);
    kcg_nodes<DIM>(g, [&](long c) { p[c] = sc_fma(beta, p[c], r[c]); });
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) state[cur ^ 1] = KcgState{rrn, st0.res0, st0.it + 1, converged ? 1 : 0};
}

// ---- BiCGStab on larger coarsest grids as a sequence of many-unit launches (2D) -----------------------------------------
// Same idea as kcg_* for the BiCGStab of Helmholtz/2D_FD_Helmholtz_fromL3.exa3:396-433 (complex, Robin columns) from
// 129^2 up — C3's coarsest grid: two iterations there, 260 us on one compute unit (45 passes over 266 KB vectors at 64 B/clk),
// and a cliff on anything larger. Six launches per iteration, every scalar formed by every block from the partial sums of
// the launch before (evomg_dot_final's order) — kb_p: rho' = rhat . r, beta, p = r + beta (p - omega nu) [+ Robin columns];
// kb_spmv<1>: nu = A p, rhat . nu; kb_hs: alpha, h = x + alpha p, s = r - alpha nu [+ Robin]; kb_spmv<2>: t = A s, t . s, t . t;
// kb_xr: omega, x = h + omega s, r = s - omega t [+ Robin], r . r, rhat . r; kb_check (one block): convergence, next state.
// The Robin column next to an inner column is written by the thread that wrote the inner node. Dot products: fused
// accumulation in evomg_dot_partial's node order with the operand order of the single-workgroup kernel; restated by the
// oracle (coarse_bicgstab with FN(dot_blocked)). scal: [0] alpha, [1] omega, [2] rho'.
template <typename T> struct KbState {
  T rho;
  double res0;
  int it, done;
};
__host__ __device__ inline bool krylov_bicgstab_per_launch(int n) { return n >= 129; }

template <typename F> __device__ __forceinline__ void kb_nodes(const Geo& g, F&& f) {       // f(index, column)
  const int m = g.n - 2;
  for (long r = blockIdx.x; r < m; r += gridDim.x) {
    const long base = gidx<2>(g, 0, 1 + (int)r, 0);
    for (int i = 1 + threadIdx.x; i <= m; i += blockDim.x) f(base + i, i);
  }
}
// the Robin column beside an inner column: x_b = x_in / (1 - i k h)
template <typename T, bool ROBIN> __device__ __forceinline__ void kb_robin(const Geo& g, T* __restrict__ v, long c, int i, T value, double kh) {
  if constexpr (ROBIN && scalar_traits<T>::is_complex) {
    const cplx den{1.0, -kh};
    if (i == 1) v[c - 1] = sc_div(value, den);
    if (i == g.n - 2) v[c + 1] = sc_div(value, den);
  }
}
template <typename T, bool ROBIN>
__global__ void __launch_bounds__(256) kb_init(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f, T* __restrict__ r,
                                               T* __restrict__ rhat, T* __restrict__ nu, T* __restrict__ p, T* __restrict__ partial, double kh) {
  __shared__ T red[17];
  const T zero = scalar_traits<T>::zero();
  T s0 = zero, s1 = zero;
  kb_nodes(g, [&](long c, int i) {
    const T v = sc_sub(f[c], apply_stencil_fma<T, 2>(st, x, g, c));
    r[c] = v;
    rhat[c] = v;
    nu[c] = zero;
    p[c] = zero;
    s0 = sc_fma(v, v, s0);
    s1 = sc_fma(v, v, s1);                      // rhat . r with rhat = r
    kb_robin<T, ROBIN>(g, r, c, i, v, kh);
    kb_robin<T, ROBIN>(g, rhat, c, i, v, kh);
    if constexpr (ROBIN) {
      if (i == 1) { p[c - 1] = zero; nu[c - 1] = zero; }
      if (i == g.n - 2) { p[c + 1] = zero; nu[c + 1] = zero; }
    }
  });
  s0 = block_sum(s0, red);
  s1 = block_sum(s1, red);
  if (threadIdx.x == 0) { partial[blockIdx.x] = s0; partial[gridDim.x + blockIdx.x] = s1; }
}
template <typename T>
__global__ void __launch_bounds__(256) kb_start(const T* __restrict__ partial, int count, KbState<T>* state, T* scal) {
  __shared__ T red[17];
  const T rr = kcg_reduce(partial, count, red);
  if (threadIdx.x == 0) {
    const double res0 = sc_norm_from_dot(rr);
    const T one = scalar_traits<T>::from_real(1.0);
    state[0] = KbState<T>{one, res0, 0, res0 == 0.0 ? 1 : 0};
    scal[0] = one;
    scal[1] = one;
  }
}
template <typename T, bool ROBIN>
__global__ void __launch_bounds__(256) kb_p(Geo g, const T* __restrict__ r, const T* __restrict__ nu, T* __restrict__ p,
                                            const T* __restrict__ partial_rhat_r, int count, const KbState<T>* state, int cur, T* scal, double kh) {
  __shared__ T red[17];
  if (state[cur].done) return;
  const T rho_new = kcg_reduce(partial_rhat_r, count, red);
  const T alpha = scal[0], omega = scal[1];
  const T beta = sc_mul(sc_div(rho_new, state[cur].rho), sc_div(alpha, omega));
  const T nomega = sc_neg(omega);
  kb_nodes(g, [&](long c, int i) {
    const T pv = sc_fma(nomega, nu[c], p[c]);
    const T v = sc_fma(beta, pv, r[c]);
    p[c] = v;
    kb_robin<T, ROBIN>(g, p, c, i, v, kh);
  });
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[2] = rho_new;
}
// out = A in; NDOT == 1: partial = w . out; NDOT == 2: partial[0] = out . w, partial[1] = out . out
template <typename T, int NDOT>
__global__ void __launch_bounds__(256) kb_spmv(Geo g, Stencil<T> st, const T* __restrict__ in, T* __restrict__ out, const T* __restrict__ w,
                                               T* __restrict__ partial, const KbState<T>* state, int cur) {
  __shared__ T red[17];
  if (state[cur].done) return;
  const T zero = scalar_traits<T>::zero();
  T s0 = zero, s1 = zero;
  kb_nodes(g, [&](long c, int) {
    const T v = apply_stencil_fma<T, 2>(st, in, g, c);
    out[c] = v;
    if (NDOT == 1) s0 = sc_fma(w[c], v, s0);
    else { s0 = sc_fma(v, w[c], s0); s1 = sc_fma(v, v, s1); }
  });
  s0 = block_sum(s0, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s0;
  if (NDOT == 2) {
    s1 = block_sum(s1, red);
    if (threadIdx.x == 0) partial[gridDim.x + blockIdx.x] = s1;
  }
}
template <typename T, bool ROBIN>
__global__ void __launch_bounds__(256) kb_hs(Geo g, const T* __restrict__ x, const T* __restrict__ r, const T* __restrict__ p, const T* __restrict__ nu,
                                             T* __restrict__ hh, T* __restrict__ s, const T* __restrict__ partial_in, int count,
                                             const KbState<T>* state, int cur, T* scal, double kh) {
  __shared__ T red[17];
  if (state[cur].done) return;
  const T alpha = sc_div(scal[2], kcg_reduce(partial_in, count, red));
  const T nalpha = sc_neg(alpha);
  kb_nodes(g, [&](long c, int i) {
    hh[c] = sc_fma(alpha, p[c], x[c]);
    const T v = sc_fma(nalpha, nu[c], r[c]);
    s[c] = v;
    kb_robin<T, ROBIN>(g, s, c, i, v, kh);
  });
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[0] = alpha;
}
template <typename T, bool ROBIN>
__global__ void __launch_bounds__(256) kb_xr(Geo g, T* __restrict__ x, T* __restrict__ r, const T* __restrict__ s, const T* __restrict__ hh,
                                             const T* __restrict__ t, const T* __restrict__ rhat, const T* __restrict__ partial_in, int count,
                                             T* __restrict__ partial_out, const KbState<T>* state, int cur, T* scal, double kh) {
  __shared__ T red[17];
  if (state[cur].done) return;
  const T ts = kcg_reduce(partial_in, count, red);
  const T tt = kcg_reduce(partial_in + count, count, red);
  const T omega = sc_div(ts, tt);
  const T nomega = sc_neg(omega);
  const T zero = scalar_traits<T>::zero();
  T s0 = zero, s1 = zero;
  kb_nodes(g, [&](long c, int i) {
    const T xv = sc_fma(omega, s[c], hh[c]);
    x[c] = xv;
    const T rv = sc_fma(nomega, t[c], s[c]);
    r[c] = rv;
    s0 = sc_fma(rv, rv, s0);
    s1 = sc_fma(rhat[c], rv, s1);
    kb_robin<T, ROBIN>(g, x, c, i, xv, kh);
    kb_robin<T, ROBIN>(g, r, c, i, rv, kh);
  });
  s0 = block_sum(s0, red);
  s1 = block_sum(s1, red);
  if (threadIdx.x == 0) { partial_out[blockIdx.x] = s0; partial_out[count + blockIdx.x] = s1; }
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[1] = omega;
}
template <typename T>
__global__ void __launch_bounds__(256) kb_check(const T* __restrict__ partial_rr, int count, KbState<T>* state, int cur, const T* scal, double tol) {
  __shared__ T red[17];
  const KbState<T> st0 = state[cur];
  if (st0.done) {
    if (threadIdx.x == 0) state[cur ^ 1] = st0;
    return;
  }
  const double res = sc_norm_from_dot(kcg_reduce(partial_rr, count, red));
  if (threadIdx.x == 0) state[cur ^ 1] = KbState<T>{scal[2], st0.res0, st0.it + 1, (res < tol * st0.res0 || !(res == res)) ? 1 : 0};
}

// ---- the same BiCGStab in THREE launches per iteration (round 5) ---------------------------------------------------------------
// A launch of these kernels is 3.5-4.5 us however little it does (127 workgroups at 129^2), six of them per iteration were a
// sixth of a C3 outer iteration. Two of the six only prepared a vector for the stencil application behind them, and the vector is
// an elementwise formula: kb2_ps applies the stencil to p_new = r + beta (p - omega nu) EVALUATED at the five nodes it reads
// (same fused multiply-adds as kb_p: same bits; the Robin column next to an inner column by the same division) — p and nu are
// double-buffered so that no workgroup reads what another one is overwriting —, kb2_hst does the same with s = r - alpha nu
// (kb_hs + kb_spmv<2>). kb_check's test moves to the head of the NEXT iteration's first launch (every workgroup forms it from
// the same partial sums), kb_start's into the first one: state[k & 1] is WRITTEN by kb2_ps of iteration k (rho' of this
// iteration, the count of completed iterations, done) and read by the other launches of that iteration; kb2_finish closes a
// batch (the test for its last iteration) in the slot the host reads. Dot products, node order, scalars: those of kb_*.
template <typename T, bool ROBIN, bool FIRST>
__global__ void __launch_bounds__(256) kb2_ps(Geo g, Stencil<T> st, const T* __restrict__ r, const T* __restrict__ rhat, const T* __restrict__ nu_in,
                                              T* __restrict__ nu_out, const T* __restrict__ p_in, T* __restrict__ p_out,
                                              const T* __restrict__ partial_r, int count, T* __restrict__ partial_nu, KbState<T>* state, int cur,
                                              const T* __restrict__ scal, double kh, double tol) {
  __shared__ T red[17];
  const T one = scalar_traits<T>::from_real(1.0), zero = scalar_traits<T>::zero();
  KbState<T> prev;
  T alpha = one, omega = one;
  if constexpr (FIRST) {
    const double res0 = sc_norm_from_dot(kcg_reduce(partial_r, count, red));
    prev = KbState<T>{one, res0, 0, res0 == 0.0 ? 1 : 0};
    if (prev.done) {
      if (blockIdx.x == 0 && threadIdx.x == 0) state[cur] = prev;
      return;
    }
  } else {
    prev = state[cur ^ 1];
    if (prev.done) {
      if (blockIdx.x == 0 && threadIdx.x == 0) state[cur] = prev;
      return;
    }
    const double res = sc_norm_from_dot(kcg_reduce(partial_r, count, red));        // kb_check of the iteration before
    if (res < tol * prev.res0 || !(res == res)) {
      if (blockIdx.x == 0 && threadIdx.x == 0) state[cur] = KbState<T>{prev.rho, prev.res0, prev.it + 1, 1};
      return;
    }
    prev.it += 1;
    alpha = scal[0];
    omega = scal[1];
  }
  const T rho_new = kcg_reduce(partial_r + count, count, red);
  const T beta = sc_mul(sc_div(rho_new, prev.rho), sc_div(alpha, omega));
  const T nomega = sc_neg(omega);
  const int n = g.n;
  auto p_inner = [&](long q) -> T { return sc_fma(beta, sc_fma(nomega, nu_in[q], p_in[q]), r[q]); };
  auto p_at = [&](long q, int i, int j) -> T {            // p_new at a node the stencil reads: (i, j) with i or j possibly on the boundary
    if (j < 1 || j > n - 2) return p_in[q];                // boundary rows are never written: both buffers hold the same values
    if (i >= 1 && i <= n - 2) return p_inner(q);
    if constexpr (ROBIN && scalar_traits<T>::is_complex) return sc_div(p_inner(i == 0 ? q + 1 : q - 1), cplx{1.0, -kh});
    else return p_in[q];
  };
  T s0 = zero;
  const int m = n - 2;
  for (long row = blockIdx.x; row < m; row += gridDim.x) {
    const int j = 1 + (int)row;
    const long base = gidx<2>(g, 0, j, 0);
    for (int i = 1 + threadIdx.x; i <= m; i += blockDim.x) {
      const long c = base + i;
      const T pc = p_inner(c);
      p_out[c] = pc;
      kb_robin<T, ROBIN>(g, p_out, c, i, pc, kh);
      T a = sc_mul(st.c[0], pc);                           // apply_stencil_fma's order
      a = sc_fma(st.c[1], p_at(c - 1, i - 1, j), a);
      a = sc_fma(st.c[2], p_at(c + 1, i + 1, j), a);
      a = sc_fma(st.c[3], p_at(c - g.pitch, i, j - 1), a);
      a = sc_fma(st.c[4], p_at(c + g.pitch, i, j + 1), a);
      nu_out[c] = a;
      s0 = sc_fma(rhat[c], a, s0);
    }
  }
  s0 = block_sum(s0, red);
  if (threadIdx.x == 0) partial_nu[blockIdx.x] = s0;
  if (blockIdx.x == 0 && threadIdx.x == 0) state[cur] = KbState<T>{rho_new, prev.res0, prev.it, 0};
}
// alpha ; h = x + alpha p ; s = r - alpha nu (stored) ; t = A s with s evaluated at the five nodes ; t . s, t . t
template <typename T, bool ROBIN>
__global__ void __launch_bounds__(256) kb2_hst(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ r, const T* __restrict__ p,
                                               const T* __restrict__ nu, T* __restrict__ hh, T* __restrict__ s, T* __restrict__ t,
                                               const T* __restrict__ partial_nu, int count, T* __restrict__ partial_t, const KbState<T>* state, int cur,
                                               T* scal, double kh) {
  __shared__ T red[17];
  if (state[cur].done) return;
  const T alpha = sc_div(state[cur].rho, kcg_reduce(partial_nu, count, red));
  const T nalpha = sc_neg(alpha);
  const T zero = scalar_traits<T>::zero();
  const int n = g.n;
  auto s_inner = [&](long q) -> T { return sc_fma(nalpha, nu[q], r[q]); };
  auto s_at = [&](long q, int i, int j) -> T {
    if (j < 1 || j > n - 2) return s[q];                   // boundary rows: never written
    if (i >= 1 && i <= n - 2) return s_inner(q);
    if constexpr (ROBIN && scalar_traits<T>::is_complex) return sc_div(s_inner(i == 0 ? q + 1 : q - 1), cplx{1.0, -kh});
    else return s[q];                                      // Dirichlet columns: never written
  };
  T s0 = zero, s1 = zero;
  const int m = n - 2;
  for (long row = blockIdx.x; row < m; row += gridDim.x) {
    const int j = 1 + (int)row;
    const long base = gidx<2>(g, 0, j, 0);
    for (int i = 1 + threadIdx.x; i <= m; i += blockDim.x) {
      const long c = base + i;
      hh[c] = sc_fma(alpha, p[c], x[c]);
      const T sc = s_inner(c);
      s[c] = sc;                                           // no launch-mate reads a stored inner or Robin value of s: they form it
      kb_robin<T, ROBIN>(g, s, c, i, sc, kh);
      T a = sc_mul(st.c[0], sc);
      a = sc_fma(st.c[1], s_at(c - 1, i - 1, j), a);
      a = sc_fma(st.c[2], s_at(c + 1, i + 1, j), a);
      a = sc_fma(st.c[3], s_at(c - g.pitch, i, j - 1), a);
      a = sc_fma(st.c[4], s_at(c + g.pitch, i, j + 1), a);
      t[c] = a;
      s0 = sc_fma(a, sc, s0);
      s1 = sc_fma(a, a, s1);
    }
  }
  s0 = block_sum(s0, red);
  if (threadIdx.x == 0) partial_t[blockIdx.x] = s0;
  s1 = block_sum(s1, red);
  if (threadIdx.x == 0) partial_t[gridDim.x + blockIdx.x] = s1;
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[0] = alpha;
}
// closes a batch: the convergence test for its last iteration (state[cur], written by that iteration's kb2_ps) into state[cur ^ 1],
// which the host reads and the next batch's first kb2_ps overwrites
// host_out: the same state once more, in pinned host memory — the host reads it behind the stream without a copy launch
template <typename T>
__global__ void __launch_bounds__(256) kb2_finish(const T* __restrict__ partial_r, int count, KbState<T>* state, int cur, double tol,
                                                  KbState<T>* __restrict__ host_out) {
  __shared__ T red[17];
  const KbState<T> st0 = state[cur];
  if (st0.done) {
    if (threadIdx.x == 0) { state[cur ^ 1] = st0; *host_out = st0; }
    return;
  }
  const double res = sc_norm_from_dot(kcg_reduce(partial_r, count, red));
  if (threadIdx.x == 0) {
    const KbState<T> next{st0.rho, st0.res0, st0.it + 1, (res < tol * st0.res0 || !(res == res)) ? 1 : 0};
    state[cur ^ 1] = next;
    *host_out = next;
  }
}

// ---- CG on 33^3 across EVOMG_CG_GROUPS compute units (3D column ownership, m >= 16) ------------------------------------
// One CU moves 64 B/clk: with 29 791 unknowns and 17 vector accesses per unknown and iteration the single-workgroup
// kernel above needs ~60 us per iteration, and the 3D configuration spends half its time there. Here the grid is cut
// into G slabs of z planes, one 1024-thread workgroup each: thread t owns the column (t % m, t / m) of its slab
// (<= 4 nodes), so r, x, p and A p of the whole slab live in registers and p additionally in LDS (zero halo image,
// in-plane neighbours). What crosses workgroups per iteration:
//   exchange A: the G partial sums of p . A p;
//   exchange B: the G partial sums of r . r  +  the slab's first and last plane of r.
// The neighbour planes of p are never sent: p_new = r + beta p_old is recomputed from the neighbour's r plane and the
// previous neighbour p plane with the same fused multiply-add the owner executes, so it is bit-identical — two
// grid-wide synchronisations per iteration instead of three.
// Synchronisation: every exchanged double travels as two "LL" words that carry their own sequence number (ll_store /
// ll_load below); consumers poll the words themselves (bounded: a workgroup that never arrives turns into status
// info[1] = 1, not a hang). Every workgroup forms the global sums in the same order (group 0, 1, ..., G-1), so all of
// them take the same branch at the stopping test and leave the loop together. The G workgroups must be co-resident:
// G = 8 of 256 CUs.
// Reduction order restated by the oracle: FN(dot_canon) with groups (oracle/evomg_oracle_impl.h).
#define EVOMG_CG_GROUPS 8
template <int DIM> __host__ __device__ inline int krylov_cg_groups(int n) {
  if (DIM != 3) return 1;
  const KrylovOwn o = krylov_own<3>(n);
  return (o.column && o.m >= 2 * EVOMG_CG_GROUPS) ? EVOMG_CG_GROUPS : 1;
}
#define EVOMG_CG_SLAB 4   // planes per slab: ceil(31 / 8)

__device__ __forceinline__ void krylov_opaque(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void krylov_opaque(cplx& v) { asm volatile("" : "+v"(v.re), "+v"(v.im)); }

// Exchange protocol ("LL" words, round 4; profiles/r04_rendezvous.md): a double travels as two 8-byte words
// {32 data bits, 32-bit sequence number}, each written by ONE relaxed agent-scope 64-bit atomic store (single-copy atomic);
// the consumer polls the two words themselves until both carry the sequence number it waits for. No flag after the data
// (no store acknowledgement to wait for) and no data load after the flag: one store trip and one poll trip per exchange
// instead of four (1.79 -> 1.22 us at G = 8 on eight XCDs, 0.82 us on one). Slots are double-buffered by the parity of
// the exchange counter: a peer can overwrite a slot only two exchanges later, i.e. after it has seen everybody's next
// publication, which everybody makes after reading this one. Sequence numbers never repeat across launches (the host
// passes a running base), so words left by an earlier solve can never match.
typedef unsigned long long evomg_u64;
__device__ __forceinline__ void ll_store(evomg_u64* slot, double v, unsigned seq) {
  const evomg_u64 bits = (evomg_u64)__double_as_longlong(v);
  __hip_atomic_store(slot, ((evomg_u64)seq << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(slot + 1, ((evomg_u64)seq << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// bounded: false when the words never arrived (a peer that is not resident); *v is then unspecified
__device__ __forceinline__ bool ll_load(const evomg_u64* slot, unsigned seq, double* v) {
  int budget = 1 << 22;
  evomg_u64 a, b;
  for (;;) {
    a = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(a >> 32) == seq && (unsigned)(b >> 32) == seq) break;
    if (--budget == 0) { *v = 0.0; return false; }
  }
  *v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
  return true;
}
// The same in two halves, so that the trip of the first attempt hides behind other work: ll_peek issues the two loads
// (nothing waits for them), ll_finish looks at what they brought and only polls on if the words had not arrived yet.
struct LLPeek { evomg_u64 a, b; };
__device__ __forceinline__ LLPeek ll_peek(const evomg_u64* slot) {
  LLPeek p;
  p.a = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  p.b = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return p;
}
__device__ __forceinline__ bool ll_finish(const LLPeek& p, const evomg_u64* slot, unsigned seq, double* v) {
  if ((unsigned)(p.a >> 32) == seq && (unsigned)(p.b >> 32) == seq) {
    *v = __longlong_as_double((long long)((p.a & 0xffffffffull) | (p.b << 32)));
    return true;
  }
  return ll_load(slot, seq, v);
}
// The same words as ONE 16-byte access each way (round 5; hardware guide: an 8-byte sc1 store is a fabric write of its own and
// costs 2.7x a 16-byte one per byte, 8-byte sc1 loads run at 0.54-0.70x the 16-byte rate): global_store_dwordx4 / global_load_dwordx4
// with the sc1 bit, i.e. what the relaxed agent-scope atomics above lower to, twice as wide. Should the 16 bytes ever travel as
// two halves, nothing changes: each half is one of the self-validating 8-byte words. The poll of K slots issues all K loads
// before it waits for any (one trip, not K); slots a lane does not need may point at any readable word (its own slot).
typedef unsigned int evomg_u32x4 __attribute__((ext_vector_type(4)));
// same_xcd: every workgroup that reads the word runs on the XCD of the writer (the caller has established that: ll_same_xcd).
// Then the store may be a PLAIN one: it stays in that XCD's L2 (write-through L1), where the readers' sc1 loads — which
// bypass their L1 and are served by the L2 — find it: the exchange never leaves the XCD (129^2 CG: 1.85 -> 1.56 us per
// iteration). Across XCDs a plain store is invisible until the kernel ends (measured: the poll times out), hence the flag.
__device__ __forceinline__ void ll_store16(evomg_u64* slot, double v, unsigned seq, bool same_xcd = false) {
  const evomg_u64 bits = (evomg_u64)__double_as_longlong(v);
  const evomg_u32x4 w = {(unsigned)bits, seq, (unsigned)(bits >> 32), seq};
  // (the s_nop: a store of more than 64 bits reads its data registers one wait state after it issues; the compiler inserts that
  // state for its own stores, it cannot see into this one — without it the next instruction may overwrite the words under way)
  if (same_xcd) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(slot), "v"(w) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(slot), "v"(w) : "memory");
}
// the XCD (accelerator complex die) this wave runs on
__device__ __forceinline__ unsigned ll_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
template <int K> struct LLWords { evomg_u32x4 w[K]; };
template <int K> __device__ __forceinline__ LLWords<K> ll_fetch16(const evomg_u64* const (&slot)[K]) {
  static_assert(K >= 1 && K <= 5, "slots per poll");
  LLWords<K> r;
  if constexpr (K == 1)
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r.w[0]) : "v"(slot[0]) : "memory");
  else if constexpr (K == 2)
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(r.w[0]), "=&v"(r.w[1]) : "v"(slot[0]), "v"(slot[1]) : "memory");
  else if constexpr (K == 3)
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\t"
                 "s_waitcnt vmcnt(0)" : "=&v"(r.w[0]), "=&v"(r.w[1]), "=&v"(r.w[2]) : "v"(slot[0]), "v"(slot[1]), "v"(slot[2]) : "memory");
  else if constexpr (K == 4)
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(r.w[0]), "=&v"(r.w[1]), "=&v"(r.w[2]), "=&v"(r.w[3]) : "v"(slot[0]), "v"(slot[1]), "v"(slot[2]), "v"(slot[3]) : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
                 "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(r.w[0]), "=&v"(r.w[1]), "=&v"(r.w[2]), "=&v"(r.w[3]), "=&v"(r.w[4])
                 : "v"(slot[0]), "v"(slot[1]), "v"(slot[2]), "v"(slot[3]), "v"(slot[4]) : "memory");
  return r;
}
// polls K slots until every NEEDED one carries `seq`; bounded (false: a peer never arrived; v is then unspecified).
// The loop is WAVE-UNIFORM (round 6): every lane fetches again until every lane of the wave has its words — one ballot and a
// scalar branch per round instead of a divergent loop's exec-mask bookkeeping (59 scalar instructions per round in the ISA of
// the 129^2 CG: a failed look cost 400 cycles before the next one was issued). A lane that already had its words fetches
// them once more: they stay valid until the slot is written again two exchanges later.
#ifdef EVOMG_CG1_PROFILE
__device__ unsigned long long evomg_ll_poll_rounds[2];      // developer builds: [0] polls, [1] fetch rounds they took
#endif
template <int K> __device__ __forceinline__ bool ll_poll16(const evomg_u64* const (&slot)[K], const bool (&need)[K], unsigned seq, double (&v)[K]) {
  int budget = 1 << 21;
  LLWords<K> r;
  unsigned bad;
  for (;;) {
    r = ll_fetch16<K>(slot);
    bad = 0u;
#pragma unroll
    for (int k = 0; k < K; ++k) bad |= need[k] ? ((r.w[k].y ^ seq) | (r.w[k].w ^ seq)) : 0u;
    if (__builtin_amdgcn_ballot_w64(bad != 0u) == 0ull) break;
    if (--budget == 0) break;
  }
#ifdef EVOMG_CG1_PROFILE
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&evomg_ll_poll_rounds[0], 1ull);
    atomicAdd(&evomg_ll_poll_rounds[1], (unsigned long long)((1 << 21) - budget + 1));
  }
#endif
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = __longlong_as_double((long long)(((evomg_u64)r.w[k].z << 32) | (evomg_u64)r.w[k].x));
  return bad == 0u;
}

// all threads call it: threads q < groups fetch group q's partial sum of this exchange into lds_part (one buffer per
// exchange kind: the other kind's barrier separates a buffer's readers from its next writers); after the barrier every
// thread adds the partial sums in the order 0 .. G-1. lds_flag is sticky (set once, never cleared).
__device__ __forceinline__ bool grid_collect_ll(const evomg_u64* slots, int groups, unsigned seq, double* lds_part, int* lds_flag,
                                                double* sum) {
  const int q = (int)threadIdx.x;
  if (q < groups) {
    double v;
    if (!ll_load(slots + 2 * q, seq, &v)) *lds_flag = 1;
    lds_part[q] = v;
  }
  __syncthreads();
  double s = lds_part[0];
  for (int q = 1; q < groups; ++q) s = s + lds_part[q];
  *sum = s;
  return *lds_flag == 0;
}

// ---- the flag-after-data protocol (evomg_fas.h still uses it: whole tiles travel there, not single doubles) ----
__device__ __forceinline__ void xchg_store(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double xchg_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xchg_store(cplx* p, cplx v) { xchg_store(&p->re, v.re); xchg_store(&p->im, v.im); }
__device__ __forceinline__ cplx xchg_load(const cplx* p) { return cplx{xchg_load(&p->re), xchg_load(&p->im)}; }
//   publish: every published value has been written with an agent-scope atomic store; the workgroup barrier orders
//            those stores before thread 0's agent-scope RELEASE store of the sequence flag; consumers poll the flag with
//            agent-scope ACQUIRE loads, then fetch the data with agent-scope atomic loads.
__device__ __forceinline__ void grid_publish(unsigned* flags, int w, unsigned seq) {
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flags + w, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ll: [2 kinds A | B][2 parities][G][2 words] partial sums, then [2 parities][G][first | last plane][1024][2 words] faces of r
// (EVOMG_CG_LL_WORDS 8-byte words, zeroed once by the host); seq_base: sequence numbers of this launch start above it.
// stride: only every stride-th workgroup of the grid takes part (stride 8 puts all G of them on ONE XCD — workgroups are
// dealt to the XCDs round-robin — where a rendezvous costs 0.82 instead of 1.22 us); the others leave at once.
#define EVOMG_CG_LL_PART_WORDS (2 * 2 * EVOMG_CG_GROUPS * 2)
#define EVOMG_CG_LL_WORDS (EVOMG_CG_LL_PART_WORDS + 2 * EVOMG_CG_GROUPS * 2 * 1024 * 2)
template <typename T>
__global__ void __launch_bounds__(1024)
evomg_cg_multi_wg3d(Geo g, Stencil<T> st, T* __restrict__ x, const T* __restrict__ b, T* __restrict__ r_out, int maxit,
                    double tol, int* __restrict__ info, evomg_u64* ll, unsigned seq_base, int stride) {
  static_assert(!scalar_traits<T>::is_complex, "the LL words carry one double");
  constexpr int G = EVOMG_CG_GROUPS, S = EVOMG_CG_SLAB;
  if ((int)blockIdx.x % stride != 0) return;
  prio_latency();
  const int w = (int)blockIdx.x / stride;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  T* red = reinterpret_cast<T*>(lds_raw);              // 2 x 16 wave sums
  int* lds_flag = reinterpret_cast<int*>(red + 32);
  T* lds_partA = red + 34;                             // G partial sums of the current exchange, one buffer per kind
  T* lds_partB = red + 34 + G;
  T* img = red + 34 + 2 * G;                           // S planes x (m+2) x (m+2), zero halo
  const KrylovOwn own = krylov_own<3>(g.n);
  const int m = own.m, W = m + 2, PS = W * W;
  const int zs = (w * m) / G, ze = ((w + 1) * m) / G, cnt = ze - zs;      // 1 <= cnt <= S (m >= 2 G)
  const T zero = scalar_traits<T>::zero();
  for (int i = threadIdx.x; i < S * PS; i += blockDim.x) img[i] = zero;
  const int t = threadIdx.x;
  if (t == 0) *lds_flag = 0;
  const bool col = t < m * m;
  const int ix = col ? t % m : 0, iy = col ? t / m : 0;
  const int o0 = col ? (iy + 1) * W + ix + 1 : 0;      // column-less threads sit on the halo corner (always zero)
  const long c0 = gidx<3>(g, ix + 1, iy + 1, zs + 1);
  auto part_slots = [&](int kind, unsigned seq) { return ll + ((kind * 2 + (int)(seq & 1u)) * G) * 2; };
  auto face_slot = [&](unsigned seq, int group, int which) {
    return ll + EVOMG_CG_LL_PART_WORDS + ((((size_t)(seq & 1u) * G + group) * 2 + which) * 1024 + t) * 2;
  };
  __syncthreads();

  T r[S], xr[S], pr[S], ap[S];
  T loc = zero;
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const bool ok = col && i < cnt;
    const long c = c0 + g.plane * (i < cnt ? i : 0);
    const T v = ok ? sc_sub(b[c], apply_stencil_fma<T, 3>(st, x, g, c)) : zero;
    r[i] = v;
    xr[i] = x[c];
    loc = sc_fma(v, v, loc);
  }
  int red_phase = 0;
  // a[cnt - 1] as a chain of selects over registers; the opaque copies keep the compiler from turning the chain
  // back into a dynamically indexed (scratch) array
  auto last_of = [&](const T (&a)[S]) -> T {
    T v = a[0];
#pragma unroll
    for (int i = 1; i < S; ++i) {
      T cand = a[i];
      krylov_opaque(cand);
      v = (i < cnt) ? cand : v;
    }
    return v;
  };
  // ---- exchange B #0: r . r and the r faces ----
  unsigned seqA = seq_base, seqB = seq_base;
  bool alive = true;
  bool faces_ok = true;                                 // a face that never arrived: reported (info[1]), not waited for again
  T part = block_sum_fast(loc, red, red_phase);
  ++seqB;
  if (t == 0) ll_store(part_slots(1, seqB) + 2 * w, part, seqB);
  if (col) { ll_store(face_slot(seqB, w, 0), r[0], seqB); ll_store(face_slot(seqB, w, 1), last_of(r), seqB); }
  T rr;
  alive = grid_collect_ll(part_slots(1, seqB), G, seqB, lds_partB, lds_flag, &rr);
  T rn_lo = zero, rn_hi = zero;
  if (alive && col && w > 0) faces_ok &= ll_load(face_slot(seqB, w - 1, 1), seqB, &rn_lo);
  if (alive && col && w < G - 1) faces_ok &= ll_load(face_slot(seqB, w + 1, 0), seqB, &rn_hi);
  T p_lo = rn_lo, p_hi = rn_hi;
#pragma unroll
  for (int i = 0; i < S; ++i) {
    pr[i] = r[i];
    if (col && i < cnt) img[i * PS + o0] = r[i];
  }
  __syncthreads();
  const double res0 = sc_norm_from_dot(rr);
  int it = 0;
  if (alive && res0 != 0.0) {
    for (; it < maxit; ++it) {
      loc = zero;
#pragma unroll
      for (int i = 0; i < S; ++i) {
        const bool ok = col && i < cnt;
        const T* q = img + i * PS + o0;
        const T pc = pr[i];
        const T pd = (i == 0) ? p_lo : pr[i - 1];
        const T pu = (i + 1 < cnt) ? pr[(i + 1 < S) ? i + 1 : i] : p_hi;
        T a = sc_mul(st.c[0], pc);
        a = sc_fma(st.c[1], q[-1], a);
        a = sc_fma(st.c[2], q[1], a);
        a = sc_fma(st.c[3], q[-W], a);
        a = sc_fma(st.c[4], q[W], a);
        a = sc_fma(st.c[5], pd, a);
        a = sc_fma(st.c[6], pu, a);
        a = ok ? a : zero;
        ap[i] = a;
        loc = sc_fma(pc, a, loc);
      }
      part = block_sum_fast(loc, red, red_phase);
      ++seqA;
      if (t == 0) ll_store(part_slots(0, seqA) + 2 * w, part, seqA);
      T pap;
      if (!grid_collect_ll(part_slots(0, seqA), G, seqA, lds_partA, lds_flag, &pap)) { alive = false; break; }
      const T alpha = sc_div(rr, pap);
      const T nalpha = sc_neg(alpha);
      loc = zero;
#pragma unroll
      for (int i = 0; i < S; ++i) {
        xr[i] = sc_fma(alpha, pr[i], xr[i]);
        const T rv = sc_fma(nalpha, ap[i], r[i]);
        r[i] = rv;
        loc = sc_fma(rv, rv, loc);
      }
      part = block_sum_fast(loc, red, red_phase);
      ++seqB;
      if (t == 0) ll_store(part_slots(1, seqB) + 2 * w, part, seqB);
      if (col) { ll_store(face_slot(seqB, w, 0), r[0], seqB); ll_store(face_slot(seqB, w, 1), last_of(r), seqB); }
      T rrn;
      // (polling the faces in the same phase as the sums was slower: a thousand spinning threads delay the eight that matter)
      if (!grid_collect_ll(part_slots(1, seqB), G, seqB, lds_partB, lds_flag, &rrn)) { alive = false; break; }
      const double res = sc_norm_from_dot(rrn);
      if (!(res > tol * res0)) { ++it; break; }
      // the neighbouring slabs' faces of r: asked for here, looked at after this slab's own update of p (their trip, 0.4 -
      // 0.8 us when waited for on the spot, hides behind it)
      const evomg_u64* slot_lo = face_slot(seqB, w > 0 ? w - 1 : 0, 1);
      const evomg_u64* slot_hi = face_slot(seqB, w < G - 1 ? w + 1 : G - 1, 0);
      const LLPeek peek_lo = ll_peek(slot_lo), peek_hi = ll_peek(slot_hi);
      const T beta = sc_div(rrn, rr);
#pragma unroll
      for (int i = 0; i < S; ++i) {
        pr[i] = sc_fma(beta, pr[i], r[i]);
        if (col && i < cnt) img[i * PS + o0] = pr[i];
      }
      rr = rrn;
      __syncthreads();
      if (col && w > 0) faces_ok &= ll_finish(peek_lo, slot_lo, seqB, &rn_lo);
      if (col && w < G - 1) faces_ok &= ll_finish(peek_hi, slot_hi, seqB, &rn_hi);
      p_lo = sc_fma(beta, p_lo, rn_lo);
      p_hi = sc_fma(beta, p_hi, rn_hi);
    }
  }
#pragma unroll
  for (int i = 0; i < S; ++i)
    if (col && i < cnt) {
      const long c = c0 + g.plane * i;
      x[c] = xr[i];
      r_out[c] = r[i];
    }
  // info[1] is sticky (only ever set here, cleared by the host before an evaluation): a later successful solve must not
  // hide an earlier missed rendezvous, and any group may be the one that saw it
  if (t == 0) {
    if (w == 0) { info[0] = it; info[2] += it; }      // [2]: iterations of all coarse solves since the host last cleared it
    if (!alive) info[1] = 1;
  }
  if (!faces_ok) info[1] = 1;
}
