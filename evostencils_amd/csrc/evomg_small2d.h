// evomg_small2d.h — point smoothers on 2D grids up to 33^2 with the whole field in LDS.
//
// On small levels a strip sweep (evomg_strip.h) is a latency chain: a handful of waves each march eight to ten rows, every
// row step waiting for the relaxation (a complex division on Helmholtz problems) of the one before — 5.6 us for a plain
// sweep and 13.2 us for sweep + residual + restriction on a 17^2 ... 65^2 complex grid, whatever its size; at the
// reference's own level ranges (Helmholtz 3..7, tutorial 5..9) most launches are of this kind. Here ONE 1024-thread
// workgroup holds the field in LDS and runs the same op fusions as phases separated by barriers, every phase over all
// rows at once (a thread owns the nodes tid, tid + 1024: at most two). Measured back to back (tools/bench_small.py, complex
// RB-GS sweep): 17^2 6.6 -> 3.6 us, 33^2 6.7 -> 4.9 us; at 65^2 (five nodes per thread) 10.7 us against the strips' 6.7 us,
// so the strips keep that size:
//   load x (PRO: x + om P(coarse), then the Robin columns of the corrected rows) -> red -> black (Robin between the
//   colours) -> Robin columns of the result -> store y -> (TAIL_RESTRICT: residual -> coarse right-hand side + its Robin columns)
// Per-node formulas, their operation order and the boundary handling are those of evomg_strip2d (relax5, residual5, the
// prolongation / restriction expressions), which are elementwise: results are bit-identical (tests: the strip form via
// EVOMG_NO_SMALL2D, and the oracle). The residual-norm tail (TAIL_NORM) stays with the strip kernel: its partial sums
// have a canonical order tied to strips and only the finest level carries it.
// dynamic LDS: n * n elements of T, twice for Jacobi (out of place) or TAIL_RESTRICT (residual image).
#pragma once
#include "evomg_strip.h"

template <typename T, int MODE, bool PRO, bool TAIL_RESTRICT, bool XZERO>
__global__ void __launch_bounds__(1024)
evomg_small2d(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f, T* __restrict__ y, double om, double omm,
              double robin_kh, int robin_out, ProlongIn<T> pro) {
  static_assert(MODE == STRIP_JACOBI || MODE == STRIP_RBGS, "smoother modes only");
  extern __shared__ __align__(16) unsigned char small_lds[];
  constexpr int MAXK = 2;                                   // 33^2 nodes / 1024 threads (host-checked: n <= 33)
  const int n = g.n, nn = n * n;
  T* A = reinterpret_cast<T*>(small_lds);
  T* B = A + nn;
  const T zero = scalar_traits<T>::zero();
  const int tid = (int)threadIdx.x;
  constexpr bool CPLX = scalar_traits<T>::is_complex;
  const bool robin = CPLX && robin_kh != 0.0;
  // x_b = x_in / (1 - i k h)
  auto robin_div = [&](T v, double kh) -> T {
    if constexpr (CPLX) return sc_div(v, cplx{1.0, -kh});
    else return v;
  };
  auto robin_columns = [&](T* img, double kh) {             // first and last column of every inner row
    for (int j = 1 + tid; j < n - 1; j += 1024) {
      img[j * n] = robin_div(img[j * n + 1], kh);
      img[j * n + n - 1] = robin_div(img[j * n + n - 2], kh);
    }
  };
  T fv[MAXK];
  int qi[MAXK], qj[MAXK];
  // ---- load (and correct) the input ----
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const int q = tid + 1024 * k;
    const bool ok = q < nn;
    const int i = ok ? q % n : 0, j = ok ? q / n : 0;
    qi[k] = ok ? i : -1;
    qj[k] = j;
    const long c = gidx<2>(g, i, j, 0);
    fv[k] = f[c];
    T v = zero;
    if constexpr (!XZERO) v = x[c];
    if constexpr (PRO) {
      if (ok && i >= 1 && i <= n - 2 && j >= 1 && j <= n - 2) {
        const T* cr = pro.coarse + (long)pro.gc.pitch * (j >> 1) + (i >> 1);
        const long cp = pro.gc.pitch;
        T p;
        if (!(j & 1)) p = (i & 1) ? sc_rmul(0.5, sc_add(cr[0], cr[1])) : cr[0];
        else p = (i & 1) ? sc_rmul(0.25, sc_add(sc_add(cr[0], cr[1]), sc_add(cr[cp], cr[cp + 1]))) : sc_rmul(0.5, sc_add(cr[0], cr[cp]));
        v = sc_add(v, sc_rmul(pro.om, p));
      }
    }
    if (ok) A[q] = v;
  }
  __syncthreads();
  if constexpr (PRO && CPLX) {
    if (robin) {                                            // the prolongation op's apply_bc on the corrected field
      robin_columns(A, robin_kh);
      __syncthreads();
    }
  }
  T* cur = A;
  if constexpr (MODE == STRIP_RBGS) {
    // ---- red (even index sum): all neighbours are black, still old ----
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int i = qi[k], j = qj[k], q = j * n + i;
      if (i >= 1 && i <= n - 2 && j >= 1 && j <= n - 2 && !((i + j) & 1))
        A[q] = relax5(st, A[q], A[q - 1], A[q + 1], A[q - n], A[q + n], fv[k], om, omm, pro.inv_c0);
    }
    __syncthreads();
    // ---- black; Helmholtz: the two-pass form re-applies the Robin condition between the colours, so a black node next
    // to the left / right boundary sees the boundary value derived from its own (still old) value ----
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int i = qi[k], j = qj[k], q = j * n + i;
      if (i >= 1 && i <= n - 2 && j >= 1 && j <= n - 2 && ((i + j) & 1)) {
        T w = A[q - 1], e = A[q + 1];
        if constexpr (CPLX) {
          if (robin) {
            if (i == 1) w = robin_div(A[q], robin_kh);
            if (i == n - 2) e = robin_div(A[q], robin_kh);
          }
        }
        A[q] = relax5(st, A[q], w, e, A[q - n], A[q + n], fv[k], om, omm, pro.inv_c0);
      }
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int i = qi[k], j = qj[k], q = j * n + i;
      if (i >= 0) {
        const bool inner = i >= 1 && i <= n - 2 && j >= 1 && j <= n - 2;
        B[q] = inner ? relax5(st, A[q], A[q - 1], A[q + 1], A[q - n], A[q + n], fv[k], om, omm, pro.inv_c0) : A[q];
      }
    }
    __syncthreads();
    cur = B;
  }
  if constexpr (CPLX) {
    if (robin && robin_out) {
      robin_columns(cur, robin_kh);
      __syncthreads();
    }
  }
  // ---- the sweep's result, boundary included ----
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const int i = qi[k], j = qj[k];
    if (i >= 0) y[gidx<2>(g, i, j, 0)] = cur[j * n + i];
  }
  if constexpr (TAIL_RESTRICT) {
    // ---- residual of the result (zero outside the inner nodes) and its full-weighting restriction ----
    T* res = (cur == A) ? B : A;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int i = qi[k], j = qj[k], q = j * n + i;
      if (i >= 0) {
        const bool inner = i >= 1 && i <= n - 2 && j >= 1 && j <= n - 2;
        res[q] = inner ? residual5(st, cur[q], cur[q - 1], cur[q + 1], cur[q - n], cur[q + n], fv[k]) : zero;
      }
    }
    __syncthreads();
    const int nc = pro.gc.n;
    auto coarse_value = [&](int I, int J) -> T {
      const T* r = res + (2 * J) * n + 2 * I;
      const T e4 = sc_add(sc_add(r[-1], r[1]), sc_add(r[-n], r[n]));
      const T q4 = sc_add(sc_add(r[-n - 1], r[-n + 1]), sc_add(r[n - 1], r[n + 1]));
      T v = sc_rmul(0.25, r[0]);
      v = sc_add(v, sc_rmul(0.125, e4));
      v = sc_add(v, sc_rmul(0.0625, q4));
      return v;
    };
    const bool robin_c = CPLX && pro.robin_kh_coarse != 0.0;
    for (int qc = tid; qc < nc * nc; qc += 1024) {
      const int I = qc % nc, J = qc / nc;
      if (J < 1 || J > nc - 2) continue;
      T* dst = pro.coarse_out + (long)pro.gc.pitch * J + I;
      if (I >= 1 && I <= nc - 2) *dst = coarse_value(I, J);
      else if (robin_c) *dst = robin_div(coarse_value(I == 0 ? 1 : nc - 2, J), pro.robin_kh_coarse);
    }
  }
}
