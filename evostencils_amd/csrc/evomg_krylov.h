// evomg_krylov.h — coarse-grid solvers (K9) as ONE persistent workgroup per solve.
//
// gen_mgCycle@minLevel is a Krylov iteration on a small grid (129^2 for the 4096^2 / 6-level config): hundreds of
// iterations of SpMV + 2..5 dot products on ~1e4 points. Run as separate launches every dot product would be a
// kernel boundary (~1.5 us) plus a host round trip for the stopping test; instead the whole solve is one launch of
// one 1024-thread workgroup (16 wave64s on one CU): vectors live in global memory but stay in that CU's L1/L2,
// dot products are wave-shuffle + LDS reductions, the stopping test never leaves the device.
//
// Algorithms: CG as configured in Poisson/2D_FD_Poisson_fromL2.exa3:12-14; BiCGStab literally as written in
// Helmholtz/2D_FD_Helmholtz_fromL3.exa3:396-433 (same statement order as oracle FN(coarse_cg)/FN(coarse_bicgstab)).
#pragma once
#include "evomg_kernels.h"

#define EVOMG_KRYLOV_THREADS 1024

template <int DIM> __device__ __forceinline__ long inner_index(const Geo& g, long q, int m) {
  const int i = 1 + (int)(q % m);
  const long t = q / m;
  if (DIM == 2) return gidx<2>(g, i, 1 + (int)t, 0);
  return gidx<3>(g, i, 1 + (int)(t % m), 1 + (int)(t / m));
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

// info[0] = iterations performed
template <typename T, int DIM>
__global__ void __launch_bounds__(EVOMG_KRYLOV_THREADS)
evomg_cg_single_wg(Geo g, Stencil<T> st, T* __restrict__ x, const T* __restrict__ b, T* __restrict__ r,
                   T* __restrict__ p, T* __restrict__ ap, int maxit, double tol, int* __restrict__ info) {
  __shared__ T red[17];
  const int m = g.n - 2;
  const long total = (DIM == 2) ? (long)m * m : (long)m * m * m;
  const T zero = scalar_traits<T>::zero();
  T loc = zero;
  for (long q = threadIdx.x; q < total; q += blockDim.x) {
    const long c = inner_index<DIM>(g, q, m);
    const T v = sc_sub(b[c], apply_stencil<T, DIM>(st, x, g, c));
    r[c] = v;
    p[c] = v;
    loc = sc_add(loc, sc_mul(v, v));
  }
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
    for (long q = threadIdx.x; q < total; q += blockDim.x) {
      const long c = inner_index<DIM>(g, q, m);
      const T v = apply_stencil<T, DIM>(st, p, g, c);
      ap[c] = v;
      loc = sc_add(loc, sc_mul(p[c], v));
    }
    const T pap = block_sum(loc, red);
    const T alpha = sc_div(rr, pap);
    const T nalpha = sc_neg(alpha);
    loc = zero;
    for (long q = threadIdx.x; q < total; q += blockDim.x) {
      const long c = inner_index<DIM>(g, q, m);
      x[c] = sc_add(x[c], sc_mul(alpha, p[c]));
      const T rv = sc_add(r[c], sc_mul(nalpha, ap[c]));
      r[c] = rv;
      loc = sc_add(loc, sc_mul(rv, rv));
    }
    const T rrn = block_sum(loc, red);
    const double res = sc_norm_from_dot(rrn);
    if (!(res > tol * res0)) { ++it; break; }
    const T beta = sc_div(rrn, rr);
    for (long q = threadIdx.x; q < total; q += blockDim.x) {
      const long c = inner_index<DIM>(g, q, m);
      p[c] = sc_add(r[c], sc_mul(beta, p[c]));
    }
    rr = rrn;
    __syncthreads();
  }
  if (threadIdx.x == 0) info[0] = it;
}

template <typename T, int DIM>
__device__ __forceinline__ T wg_dot(const Geo& g, const T* __restrict__ a, const T* __restrict__ b, T* red) {
  const int m = g.n - 2;
  const long total = (DIM == 2) ? (long)m * m : (long)m * m * m;
  T loc = scalar_traits<T>::zero();
  for (long q = threadIdx.x; q < total; q += blockDim.x) {
    const long c = inner_index<DIM>(g, q, m);
    loc = sc_add(loc, sc_mul(a[c], b[c]));
  }
  return block_sum(loc, red);
}

template <typename T, int DIM, bool ROBIN>
__global__ void __launch_bounds__(EVOMG_KRYLOV_THREADS)
evomg_bicgstab_single_wg(Geo g, Stencil<T> st, T* __restrict__ x, const T* __restrict__ f, T* __restrict__ r,
                         T* __restrict__ p, T* __restrict__ nu, T* __restrict__ hh, T* __restrict__ s,
                         T* __restrict__ t, T* __restrict__ rhat, int maxit, double tol, double kh,
                         int* __restrict__ info) {
  __shared__ T red[17];
  const int m = g.n - 2;
  const long total = (DIM == 2) ? (long)m * m : (long)m * m * m;
  const T zero = scalar_traits<T>::zero();
  T loc = zero;
  for (long q = threadIdx.x; q < total; q += blockDim.x) {
    const long c = inner_index<DIM>(g, q, m);
    const T v = sc_sub(f[c], apply_stencil<T, DIM>(st, x, g, c));
    r[c] = v;
    rhat[c] = v;
    nu[c] = zero;
    p[c] = zero;
    loc = sc_add(loc, sc_mul(v, v));
  }
  const double res0 = sc_norm_from_dot(block_sum(loc, red));
  wg_apply_bc<T, ROBIN>(g, r, kh);   // residual_st applies the field's BC
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
    rho_new = wg_dot<T, DIM>(g, rhat, r, red);
    beta = sc_mul(sc_div(rho_new, rho), sc_div(alpha, omega));
    const T nomega = sc_neg(omega);
    for (long q = threadIdx.x; q < total; q += blockDim.x) {
      const long c = inner_index<DIM>(g, q, m);
      const T pv = sc_add(p[c], sc_mul(nomega, nu[c]));
      p[c] = sc_add(r[c], sc_mul(beta, pv));
    }
    wg_apply_bc<T, ROBIN>(g, p, kh);
    for (long q = threadIdx.x; q < total; q += blockDim.x) {
      const long c = inner_index<DIM>(g, q, m);
      nu[c] = apply_stencil<T, DIM>(st, p, g, c);
    }
    __syncthreads();
    alpha = sc_div(rho_new, wg_dot<T, DIM>(g, rhat, nu, red));
    const T nalpha = sc_neg(alpha);
    for (long q = threadIdx.x; q < total; q += blockDim.x) {
      const long c = inner_index<DIM>(g, q, m);
      hh[c] = sc_add(x[c], sc_mul(alpha, p[c]));
      s[c] = sc_add(r[c], sc_mul(nalpha, nu[c]));
    }
    wg_apply_bc<T, ROBIN>(g, s, kh);
    for (long q = threadIdx.x; q < total; q += blockDim.x) {
      const long c = inner_index<DIM>(g, q, m);
      t[c] = apply_stencil<T, DIM>(st, s, g, c);
    }
    __syncthreads();
    const T ts = wg_dot<T, DIM>(g, t, s, red);
    const T tt = wg_dot<T, DIM>(g, t, t, red);
    omega = sc_div(ts, tt);
    const T nomega2 = sc_neg(omega);
    loc = zero;
    for (long q = threadIdx.x; q < total; q += blockDim.x) {
      const long c = inner_index<DIM>(g, q, m);
      x[c] = sc_add(hh[c], sc_mul(omega, s[c]));
      const T rv = sc_add(s[c], sc_mul(nomega2, t[c]));
      r[c] = rv;
      loc = sc_add(loc, sc_mul(rv, rv));
    }
    const double res = sc_norm_from_dot(block_sum(loc, red));
    wg_apply_bc<T, ROBIN>(g, x, kh);
    wg_apply_bc<T, ROBIN>(g, r, kh);
    if (res < tol * res0) { ++it; break; }
    if (!(res == res)) { ++it; break; }
  }
  if (threadIdx.x == 0) info[0] = it;
}

// ---- CG with the search direction resident in LDS -------------------------------------------------------------
// For coarsest grids whose inner nodes fit the CU's 160 KB LDS (2D up to 129^2, 3D up to 17^3 in FP64) the
// whole iteration runs out of LDS + registers: p sits in LDS surrounded by a zero halo (the homogeneous boundary
// of an error-type field), so the 5/7-point SpMV needs no boundary tests; every thread keeps the residual r and
// the iterate x of its ITEMS nodes in registers; A p is recomputed for the residual update instead of being
// stored (LDS reads are cheaper than 32 more VGPRs at 1024 threads/CU). Node ownership q = t + 1024 i and the
// reduction tree are those of evomg_cg_single_wg, i.e. the canonical order the oracle restates: results are
// bit-identical to the global-memory kernel and to the oracle.
template <typename T, int DIM>
__device__ __forceinline__ T lds_apply(const Stencil<T>& st, const T* __restrict__ p, int o, int row, int plane) {
  T a = sc_mul(st.c[0], p[o]);
  a = sc_add(a, sc_mul(st.c[1], p[o - 1]));
  a = sc_add(a, sc_mul(st.c[2], p[o + 1]));
  a = sc_add(a, sc_mul(st.c[3], p[o - row]));
  a = sc_add(a, sc_mul(st.c[4], p[o + row]));
  if (DIM == 3) {
    a = sc_add(a, sc_mul(st.c[5], p[o - plane]));
    a = sc_add(a, sc_mul(st.c[6], p[o + plane]));
  }
  return a;
}

// padded LDS image of p: two spare zero rows in front (the dummy cell of ItemWalk::each_all sits in the second one,
// so that its own south neighbour is still inside the image), then the halo'ed grid
template <int DIM> __host__ __device__ inline long cg_lds_elems(int n) {
  const long m = n - 2, row = m + 1;
  if (DIM == 2) return (m + 2) * row + 2 * row;
  return (m + 2) * row * (m + 1) + 3 * row;
}

// Walks the ITEMS nodes q = t, t+1024, ... owned by a thread, yielding each node's offset in the padded LDS image
// without storing 16 offsets per thread: the (ix, iy) carry arithmetic of "+1024" is replayed in every pass.
template <int DIM> struct ItemWalk {
  int m, row, plane;
  int inc, cx, cy;          // q += 1024  ==>  ix += cx, iy += cy (+ carries), off += inc (+ carries)
  int ix0, iy0, off0, cnt;
  __device__ __forceinline__ ItemWalk(const Geo& g, long total) {
    m = g.n - 2;
    row = m + 1;
    plane = row * (m + 1);
    const int t = threadIdx.x;
    const int T0 = EVOMG_KRYLOV_THREADS;
    cx = T0 % m;
    const int rest = T0 / m;
    cy = (DIM == 2) ? rest : rest % m;
    const int cz = (DIM == 2) ? 0 : rest / m;
    inc = (DIM == 2) ? rest * row + cx : cz * plane + cy * row + cx;
    ix0 = t % m;
    const int r0 = t / m;
    iy0 = (DIM == 2) ? r0 : r0 % m;
    const int iz0 = (DIM == 2) ? 0 : r0 / m;
    off0 = 2 * row + ((DIM == 2) ? (iy0 + 1) * row + ix0 + 1 : (iz0 + 1) * plane + (iy0 + 1) * row + ix0 + 1);
    cnt = (t < total) ? (int)((total - t + T0 - 1) / T0) : 0;
  }
  // Branch-free pass over all ITEMS slots: slots beyond this thread's node count are pointed at a dummy LDS cell
  // (second spare front row, column 1) whose value and neighbours are zero and which no real node ever reads, so they
  // contribute exact zeros. Without per-slot branches the 5 LDS reads of several nodes are in flight together.
  template <int ITEMS, int GROUP, typename F> __device__ __forceinline__ void each_all(F&& f) const {
    int off = off0;
    int ix = ix0, iy = iy0;
    asm volatile("" : "+v"(off), "+v"(ix), "+v"(iy));
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      f(i, (i < cnt) ? off : row + 1);
      if ((i & (GROUP - 1)) == GROUP - 1) __builtin_amdgcn_sched_barrier(0);   // bound the scheduler's look-ahead
      off += inc;
      ix += cx;
      iy += cy;
      if (ix >= m) { ix -= m; off += 1; iy += 1; }
      if (DIM == 3 && iy >= m) { iy -= m; off += row; }
    }
  }
  template <int ITEMS, typename F> __device__ __forceinline__ void each(F&& f) const {
    int off = off0, ix = ix0, iy = iy0;
    // opaque to the optimiser: otherwise LICM hoists all ITEMS offsets out of the CG iteration loop and keeps
    // them live (that, not the LDS reads, is what spilled at 16 items per thread)
    asm volatile("" : "+v"(off), "+v"(ix), "+v"(iy));
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (i < cnt) f(i, off, ix, iy);
      // keep the scheduler from hoisting every item's LDS reads to the top (it would spill at 16 items/thread)
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      off += inc;
      ix += cx;
      iy += cy;
      if (ix >= m) { ix -= m; off += 1; iy += 1; }          // x wraps into the next row: row - m = 1
      if (DIM == 3 && iy >= m) { iy -= m; off += row; }     // y wraps into the next plane: plane - m*row = row
    }
  }
};

template <typename T, int DIM, int ITEMS>
__global__ void __launch_bounds__(EVOMG_KRYLOV_THREADS)
evomg_cg_lds(Geo g, Stencil<T> st, T* __restrict__ x, const T* __restrict__ b, T* __restrict__ r_out, int maxit,
             double tol, int* __restrict__ info) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  T* red = reinterpret_cast<T*>(lds_raw);          // 2 x 16 wave sums (double-buffered)
  T* p = red + 32;
  const int m = g.n - 2;
  const long total = (DIM == 2) ? (long)m * m : (long)m * m * m;
  const long lds_elems = cg_lds_elems<DIM>(g.n);
  const T zero = scalar_traits<T>::zero();
  for (long i = threadIdx.x; i < lds_elems; i += blockDim.x) p[i] = zero;
  __syncthreads();
  const ItemWalk<DIM> walk(g, total);
  const int row = walk.row, plane = walk.plane;
  // global index of the node at padded LDS offset `off` (2D: needs ix, iy only)
  auto gindex = [&](int off, int ix, int iy) -> long {
    if (DIM == 2) return gidx<2>(g, ix + 1, iy + 1, 0);
    const int iz = (off - 2 * row) / plane - 1;
    return gidx<3>(g, ix + 1, iy + 1, iz + 1);
  };

  T r[ITEMS], xr[ITEMS];
  T loc = zero;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) { r[i] = zero; xr[i] = zero; }
  walk.template each<ITEMS>([&](int i, int off, int ix, int iy) {
    const long c = gindex(off, ix, iy);
    const T v = sc_sub(b[c], apply_stencil<T, DIM>(st, x, g, c));
    r[i] = v;
    xr[i] = x[c];
    p[off] = v;
    loc = sc_add(loc, sc_mul(v, v));
  });
  int red_phase = 0;
  T rr = block_sum_fast(loc, red, red_phase);
  const double res0 = sc_norm_from_dot(rr);
  int it = 0;
  if (res0 != 0.0) {
    for (; it < maxit; ++it) {
      // block_sum / the trailing barrier of the previous iteration made every p entry visible
      loc = zero;
      walk.template each_all<ITEMS, (ITEMS >= 16 ? 2 : 4)>([&](int i, int off) {
        loc = sc_add(loc, sc_mul(p[off], lds_apply<T, DIM>(st, p, off, row, plane)));
      });
      const T pap = block_sum_fast(loc, red, red_phase);
      const T alpha = sc_div(rr, pap);
      const T nalpha = sc_neg(alpha);
      loc = zero;
      walk.template each_all<ITEMS, (ITEMS >= 16 ? 2 : 4)>([&](int i, int off) {
        const T pv = p[off];
        const T apv = lds_apply<T, DIM>(st, p, off, row, plane);
        xr[i] = sc_add(xr[i], sc_mul(alpha, pv));
        const T rv = sc_add(r[i], sc_mul(nalpha, apv));
        r[i] = rv;
        loc = sc_add(loc, sc_mul(rv, rv));
      });
      const T rrn = block_sum_fast(loc, red, red_phase);   // its barrier: all reads of p are done before p is rewritten
      const double res = sc_norm_from_dot(rrn);
      if (!(res > tol * res0)) { ++it; break; }
      const T beta = sc_div(rrn, rr);
      walk.template each_all<ITEMS, (ITEMS >= 16 ? 2 : 4)>([&](int i, int off) { p[off] = sc_add(r[i], sc_mul(beta, p[off])); });
      rr = rrn;
      __syncthreads();
    }
  }
  walk.template each<ITEMS>([&](int i, int off, int ix, int iy) {
    const long c = gindex(off, ix, iy);
    x[c] = xr[i];
    r_out[c] = r[i];
  });
  if (threadIdx.x == 0) info[0] = it;
}
