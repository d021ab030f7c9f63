// evomg_small2d.h — point smoothers on small and medium 2D grids with the field (or a tile of it) in LDS.
//
// On small levels a strip sweep (evomg_strip.h) is a latency chain: a handful of waves each march eight to ten rows, every
// row step waiting for the relaxation (a complex division on Helmholtz problems) of the one before — 6.6 us for a plain
// sweep and 13 us for sweep + residual + restriction on a complex grid, whatever its size; at the reference's own level
// ranges (Helmholtz 3..7, tutorial 5..9) most launches are of this kind. Here a 1024-thread workgroup holds a tile of at
// most 33 x 33 nodes plus a halo of four in LDS (grids up to 33^2: the whole field, one workgroup) and runs the same op
// fusions as phases separated by barriers, every phase over all rows at once (a thread owns at most two nodes):
//   load x (PRO: x + om P(coarse), then the Robin columns of the corrected rows) -> red -> black (Robin between the
//   colours) -> Robin columns of the result -> store y -> (TAIL_RESTRICT: residual -> coarse right-hand side + its Robin columns)
// Tiles are independent: each phase is valid one node further inside the extended tile than the one before (red 1, black 2,
// residual 3 nodes from an edge that is not the domain's; the owned tile and the 3 x 3 neighbourhoods of its coarse nodes
// lie inside that), the halo is recomputed, x and y are different arrays (host). Per-node formulas, their operation order
// and the boundary handling are those of evomg_strip2d (relax5, residual5, the prolongation / restriction expressions),
// which are elementwise: results are bit-identical (tests: the strip form via EVOMG_NO_SMALL2D, and the oracle). The
// residual-norm tail (TAIL_NORM) stays with the strip kernel: its partial sums have a canonical order tied to strips.
// dynamic LDS: one image of the extended tile, two for Jacobi (out of place) or TAIL_RESTRICT (residual image).
#pragma once
#include "evomg_strip.h"

#define EVOMG_SMALL2D_TILE 33
#define EVOMG_SMALL2D_HALO 4
#define EVOMG_SMALL2D_HALO2 6       // two sweeps before the tail on a grid of several tiles (round 6): every sweep needs two more nodes
__host__ __device__ inline int small2d_tiles(int n) { return (n - 1 + 31) / 32; }                    // per direction
__host__ __device__ inline int small2d_tile_size(int n) { const int t = small2d_tiles(n); return (n + t - 1) / t; }
__host__ __device__ inline int small2d_image_elems(int n, int halo = EVOMG_SMALL2D_HALO) {
  const int w = small2d_tiles(n) == 1 ? n : small2d_tile_size(n) + 2 * halo;
  return w * w;
}

template <typename T, int MODE, bool PRO, bool TAIL_RESTRICT, bool XZERO>
__global__ void __launch_bounds__(1024)
evomg_small2d(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f, T* __restrict__ y, double om, double omm,
              double robin_kh, int robin_out, ProlongIn<T> pro, int nsweeps) {
  // nsweeps > 1 (red-black; host-checked: 1 or 2): that many identical sweeps in place before the tail, each followed by the
  // Robin columns as the separate ops would leave them. On a grid of several tiles the second sweep is valid two nodes further
  // inside the extended tile than the first (red 3, black 4, residual 5 nodes from an edge that is not the domain's): the halo
  // is six nodes then (round 6; 45 x 45 nodes, still two per thread)
  static_assert(MODE == STRIP_JACOBI || MODE == STRIP_RBGS, "smoother modes only");
  prio_latency();
  extern __shared__ __align__(16) unsigned char small_lds[];
  constexpr int MAXK = 2;                                   // (33 + 8)^2 nodes / 1024 threads
  constexpr int BIG = 1 << 20;
  const int n = g.n;
  const int tiles = small2d_tiles(n), ts = small2d_tile_size(n);
  const int H = (nsweeps > 1 && tiles > 1) ? EVOMG_SMALL2D_HALO2 : EVOMG_SMALL2D_HALO;
  const int tx = (int)blockIdx.x % tiles, ty = (int)blockIdx.x / tiles;
  // owned nodes [ox0, ox1) x [oy0, oy1); extended tile [ex0, ex1) x [ey0, ey1), clipped to the grid
  const int ox0 = tx * ts, ox1 = min(ox0 + ts, n), oy0 = ty * ts, oy1 = min(oy0 + ts, n);
  const int ex0 = max(ox0 - H, 0), ex1 = min(ox1 + H, n), ey0 = max(oy0 - H, 0), ey1 = min(oy1 + H, n);
  const int ew = ex1 - ex0, eh = ey1 - ey0, en = ew * eh;
  T* A = reinterpret_cast<T*>(small_lds);
  T* B = A + en;
  const double inv_c0k = relax_div_const(st, pro.inv_c0);    // relax5's divisor-side constant
  const T zero = scalar_traits<T>::zero();
  const int tid = (int)threadIdx.x;
  constexpr bool CPLX = scalar_traits<T>::is_complex;
  const bool robin = CPLX && robin_kh != 0.0;
  // x_b = x_in / (1 - i k h)
  auto robin_div = [&](T v, double kh) -> T {
    if constexpr (CPLX) return sc_div(v, cplx{1.0, -kh});
    else return v;
  };
  // nodes away from the edges of the extended tile that are not edges of the grid
  auto depth = [&](int i, int j) -> int {
    const int dl = ex0 == 0 ? BIG : i - ex0, dr = ex1 == n ? BIG : ex1 - 1 - i;
    const int db = ey0 == 0 ? BIG : j - ey0, dt = ey1 == n ? BIG : ey1 - 1 - j;
    return min(min(dl, dr), min(db, dt));
  };
  // first and last column of every inner row of the extended tile at least `need` rows inside (where the tile has them)
  auto robin_columns = [&](T* img, double kh, int need) {
    for (int lj = tid; lj < eh; lj += 1024) {
      const int j = ey0 + lj;
      if (j < 1 || j > n - 2) continue;
      const int dv = min(ey0 == 0 ? BIG : j - ey0, ey1 == n ? BIG : ey1 - 1 - j);
      if (dv < need) continue;
      if (ex0 == 0) img[lj * ew] = robin_div(img[lj * ew + 1], kh);
      if (ex1 == n) img[lj * ew + ew - 1] = robin_div(img[lj * ew + ew - 2], kh);
    }
  };
  T fv[MAXK];
  int qi[MAXK], qj[MAXK], ql[MAXK];                         // node (global), image index; qi < 0: no node
  // ---- load (and correct) the input ----
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const int q = tid + 1024 * k;
    const bool ok = q < en;
    const int li = ok ? q % ew : 0, lj = ok ? q / ew : 0;
    const int i = ex0 + li, j = ey0 + lj;
    qi[k] = ok ? i : -1;
    qj[k] = j;
    ql[k] = lj * ew + li;
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
    if (ok) A[ql[k]] = v;
  }
  __syncthreads();
  if constexpr (PRO && CPLX) {
    if (robin) {                                            // the prolongation op's apply_bc on the corrected field
      robin_columns(A, robin_kh, 0);
      __syncthreads();
    }
  }
  T* cur = A;
  if constexpr (MODE == STRIP_RBGS) {
   for (int sweep = 0; sweep < nsweeps; ++sweep) {
    if (sweep > 0) {
      if constexpr (CPLX) {
        if (robin && robin_out) {
          robin_columns(A, robin_kh, 2 * sweep);
          __syncthreads();
        }
      }
    }
    // ---- red (even index sum): all neighbours are black, still old ----
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int i = qi[k], j = qj[k], q = ql[k];
      if (i >= 1 && i <= n - 2 && j >= 1 && j <= n - 2 && !((i + j) & 1) && depth(i, j) >= 2 * sweep + 1)
        A[q] = relax5(st, A[q], A[q - 1], A[q + 1], A[q - ew], A[q + ew], fv[k], om, omm, inv_c0k);
    }
    __syncthreads();
    // ---- black; Helmholtz: the two-pass form re-applies the Robin condition between the colours, so a black node next
    // to the left / right boundary sees the boundary value derived from its own (still old) value ----
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int i = qi[k], j = qj[k], q = ql[k];
      if (i >= 1 && i <= n - 2 && j >= 1 && j <= n - 2 && ((i + j) & 1) && depth(i, j) >= 2 * sweep + 2) {
        T w = A[q - 1], e = A[q + 1];
        if constexpr (CPLX) {
          if (robin) {
            if (i == 1) w = robin_div(A[q], robin_kh);
            if (i == n - 2) e = robin_div(A[q], robin_kh);
          }
        }
        A[q] = relax5(st, A[q], w, e, A[q - ew], A[q + ew], fv[k], om, omm, inv_c0k);
      }
    }
    __syncthreads();
   }
  } else {
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int i = qi[k], j = qj[k], q = ql[k];
      if (i >= 0) {
        const bool inner = i >= 1 && i <= n - 2 && j >= 1 && j <= n - 2 && depth(i, j) >= 1;
        B[q] = inner ? relax5(st, A[q], A[q - 1], A[q + 1], A[q - ew], A[q + ew], fv[k], om, omm, inv_c0k) : A[q];
      }
    }
    __syncthreads();
    cur = B;
  }
  if constexpr (CPLX) {
    if (robin && robin_out) {
      robin_columns(cur, robin_kh, MODE == STRIP_RBGS ? 2 * nsweeps : 2);
      __syncthreads();
    }
  }
  // ---- the sweep's result on the owned nodes, boundary included ----
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const int i = qi[k], j = qj[k];
    if (i >= ox0 && i < ox1 && j >= oy0 && j < oy1) y[gidx<2>(g, i, j, 0)] = cur[ql[k]];
  }
  if constexpr (TAIL_RESTRICT) {
    // ---- residual of the result (zero outside the inner nodes) and its full-weighting restriction ----
    T* res = (cur == A) ? B : A;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int i = qi[k], j = qj[k], q = ql[k];
      if (i >= 0) {
        const bool inner = i >= 1 && i <= n - 2 && j >= 1 && j <= n - 2 && depth(i, j) >= (MODE == STRIP_RBGS ? 2 * nsweeps + 1 : 3);
        res[q] = inner ? residual5(st, cur[q], cur[q - 1], cur[q + 1], cur[q - ew], cur[q + ew], fv[k]) : zero;
      }
    }
    __syncthreads();
    const int nc = pro.gc.n;
    auto coarse_value = [&](int I, int J) -> T {
      const T* r = res + (2 * J - ey0) * ew + (2 * I - ex0);
      const T e4 = sc_add(sc_add(r[-1], r[1]), sc_add(r[-ew], r[ew]));
      const T q4 = sc_add(sc_add(r[-ew - 1], r[-ew + 1]), sc_add(r[ew - 1], r[ew + 1]));
      T v = sc_rmul(0.25, r[0]);
      v = sc_add(v, sc_rmul(0.125, e4));
      v = sc_add(v, sc_rmul(0.0625, q4));
      return v;
    };
    const bool robin_c = CPLX && pro.robin_kh_coarse != 0.0;
    // the coarse nodes whose fine twin (2I, 2J) this tile owns
    const int I0 = (ox0 + 1) >> 1, I1 = (ox1 + 1) >> 1, J0 = (oy0 + 1) >> 1, J1 = (oy1 + 1) >> 1, cw = I1 - I0;
    for (int qc = tid; qc < cw * (J1 - J0); qc += 1024) {
      const int I = I0 + qc % cw, J = J0 + qc / cw;
      if (J < 1 || J > nc - 2) continue;
      T* dst = pro.coarse_out + (long)pro.gc.pitch * J + I;
      if (I >= 1 && I <= nc - 2) *dst = coarse_value(I, J);
      else if (robin_c) *dst = robin_div(coarse_value(I == 0 ? 1 : nc - 2, J), pro.robin_kh_coarse);
    }
  }
}
