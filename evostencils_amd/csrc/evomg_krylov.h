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
