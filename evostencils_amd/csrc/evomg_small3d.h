// evomg_small3d.h — one visit of a small 3D level (up to 17^3) in ONE launch with the field in LDS.
//
// Below 129^3 the 3D levels run one thread-per-node launch per op and colour: the reference's 3D example at its own level
// range (2..6, V(2,1) red-black) is 48 launches per cycle at the 3.1 us launch floor, and the marching tiles are slower
// still there (a handful of waves marching planes). On the levels that fit one compute unit's LDS whole — 17^3 doubles are
// 39 KB — one 1024-thread workgroup runs a whole visit of the level as phases separated by barriers:
//   [u = 0 | u += w P(coarse)]  ->  k red-black sweeps in place  ->  store u  ->  [r = f - A u  ->  store r  ->  R r to the coarse level]
// with the device functions of the generic kernels (relax_point, apply_stencil, interpolate, the 27-point weighting of
// evomg_restrict) applied to the LDS image — same formulas, same operation order, elementwise: bit-identical to the
// separate launches (tests: EVOMG_NO_SMALL3D, and the oracle through every small 3D parity test).
// dynamic LDS: u and f images (n^3 each), and the residual image when the tail is present.
#pragma once
#include "evomg_kernels.h"

static __global__ void __launch_bounds__(1024)      // static: the header is part of two translation units
evomg_small3d_chain(Geo g, Stencil<double> st, double* __restrict__ x, const double* __restrict__ f, double om, double omm, int nsweeps,
                    int zero_first, const double* __restrict__ coarse_in, Geo gc, double om_p, double* __restrict__ res_out,
                    double* __restrict__ coarse_out) {
  extern __shared__ __align__(16) double small3d_lds[];
  const int n = g.n, n2 = n * n, nn = n2 * n;
  double* A = small3d_lds;
  double* F = A + nn;
  double* R = F + nn;
  Geo gl;                                                   // the LDS images: rows and planes packed
  gl.n = n;
  gl.pitch = n;
  gl.plane = n2;
  const int tid = (int)threadIdx.x;
  for (int q = tid; q < nn; q += 1024) {
    const int i = q % n, j = (q / n) % n, k = q / n2;
    const long c = gidx<3>(g, i, j, k);
    const bool inner = is_inner<3>(g, i, j, k);
    double v = (zero_first && inner) ? 0.0 : x[c];
    if (coarse_in && inner) v = v + om_p * interpolate<double, 3>(gc, coarse_in, i, j, k);
    A[q] = v;
    F[q] = f[c];
  }
  __syncthreads();
  // n is odd, so the colour of node q = i + n j + n^2 k is q & 1: the threads enumerate the nodes of one colour only
  // (a checkerboard mask would idle half of every wave through the division); the quotient by the stencil centre in
  // three operations with the bits of the division (evomg_scalar.h: const_div, as in the 3D tile kernels)
  const ConstDiv<double> cd0 = const_div_make(st.c[0]);
  for (int s = 0; s < nsweeps; ++s)
    for (int colour = 0; colour < 2; ++colour) {
      for (int q = 2 * tid + colour; q < nn; q += 2048) {
        const int i = q % n, j = (q / n) % n, k = q / n2;
        if (!is_inner<3>(g, i, j, k)) continue;
        double r = F[q];                                    // relax_point<double, 3>, operation for operation
        r = r - st.c[1] * A[q - 1];
        r = r - st.c[2] * A[q + 1];
        r = r - st.c[3] * A[q - n];
        r = r - st.c[4] * A[q + n];
        r = r - st.c[5] * A[q - n2];
        r = r - st.c[6] * A[q + n2];
        const double xs = const_div(r, cd0);
        A[q] = omm * A[q] + om * xs;
      }
      __syncthreads();
    }
  for (int q = tid; q < nn; q += 1024) {
    const int i = q % n, j = (q / n) % n, k = q / n2;
    if (is_inner<3>(g, i, j, k)) x[gidx<3>(g, i, j, k)] = A[q];
  }
  if (!coarse_out) return;
  for (int q = tid; q < nn; q += 1024) {
    const int i = q % n, j = (q / n) % n, k = q / n2;
    double r = 0.0;
    if (is_inner<3>(g, i, j, k)) {
      r = F[q] - apply_stencil<double, 3>(st, A, gl, q);
      res_out[gidx<3>(g, i, j, k)] = r;
    }
    R[q] = r;
  }
  __syncthreads();
  const int nc = gc.n, nc2 = nc * nc;
  for (int qc = tid; qc < nc2 * nc; qc += 1024) {
    const int I = qc % nc, J = (qc / nc) % nc, K = qc / nc2;
    if (!is_inner<3>(gc, I, J, K)) continue;
    const int i = 2 * I, j = 2 * J, k = 2 * K;
    double face = 0.0, edge = 0.0, corner = 0.0;
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int nz = (dx != 0) + (dy != 0) + (dz != 0);
          if (nz == 0) continue;
          const double v = R[gidx<3>(gl, i + dx, j + dy, k + dz)];
          if (nz == 1) face = face + v;
          else if (nz == 2) edge = edge + v;
          else corner = corner + v;
        }
    double v = 0.125 * R[gidx<3>(gl, i, j, k)];
    v = v + 0.0625 * face;
    v = v + 0.03125 * edge;
    v = v + 0.015625 * corner;
    coarse_out[gidx<3>(gc, I, J, K)] = v;
  }
}

// ---- fused two-colour RB-GS on the 3D levels below the tile threshold (9^3 .. 65^3), out of place (round 4) -------------
// There a red-black sweep was two thread-per-node launches (evomg_rbgs_colour, in place): the reference's 3D example at its
// own level range spends a third of its launches on them. Here ONE launch does both colours without a grid-wide dependency:
// a 1024-thread workgroup owns a block of 16 x 8 x 8 inner nodes (one node per thread and phase: the launch is a latency chain), loads it with two layers of halo into LDS, relaxes the reds of
// the block AND of one layer around it (the neighbouring blocks compute those reds too, from the same old blacks: the same
// bits), then the blacks of the block from those reds, and writes the block to the spare array (a pointer swap makes it
// current, as for every out-of-place smoother; the boundary faces: ensure_faces). Red work 1.76x, one launch instead of two.
// Arithmetic = relax_point (the quotient by the centre as const_div, as in evomg_small3d_chain and the tiles): bit-identical.
// The SET_ZERO or the PROLONG_ADD in front of the sweep ride on the block's load (zero_first / coarse_in): one more launch less.
#define EVOMG_RB3D_BX 16
#define EVOMG_RB3D_BY 8
#define EVOMG_RB3D_BZ 8
#define EVOMG_RB3D_THREADS 1024
static __global__ void __launch_bounds__(EVOMG_RB3D_THREADS)
evomg_rbgs3d_block(Geo g, Stencil<double> st, const double* __restrict__ x, const double* __restrict__ f, double* __restrict__ y,
                   double om, double omm, int nbx, int nby, int zero_first, const double* __restrict__ coarse_in, Geo gc, double om_p) {
  constexpr int BX = EVOMG_RB3D_BX, BY = EVOMG_RB3D_BY, BZ = EVOMG_RB3D_BZ;
  constexpr int TX = BX + 4, TY = BY + 4, TZ = BZ + 4;
  __shared__ double U[TZ * TY * TX];
  const int n = g.n, tid = (int)threadIdx.x;
  const int bi = (int)blockIdx.x % nbx, bj = ((int)blockIdx.x / nbx) % nby, bk = (int)blockIdx.x / (nbx * nby);
  const int i0 = 1 + bi * BX, j0 = 1 + bj * BY, k0 = 1 + bk * BZ;        // first inner node of the block; U[0] sits at (i0-2, j0-2, k0-2)
  for (int q = tid; q < TX * TY * TZ; q += EVOMG_RB3D_THREADS) {
    const int gi = i0 - 2 + q % TX, gj = j0 - 2 + (q / TX) % TY, gk = k0 - 2 + q / (TX * TY);
    const bool in = gi >= 0 && gi < n && gj >= 0 && gj < n && gk >= 0 && gk < n;
    const bool inner = gi >= 1 && gi <= n - 2 && gj >= 1 && gj <= n - 2 && gk >= 1 && gk <= n - 2;
    // zero_first: the SET_ZERO in front of the sweep, elided (the field's boundary is zero as well: host-checked);
    // coarse_in: the PROLONG_ADD in front of the sweep, folded in as evomg_small3d_chain folds it (u + w P(coarse) on inner nodes)
    double v = (in && !zero_first) ? x[gidx<3>(g, gi, gj, gk)] : 0.0;
    if (coarse_in && inner) v = v + om_p * interpolate<double, 3>(gc, coarse_in, gi, gj, gk);
    U[q] = v;
  }
  __syncthreads();
  const ConstDiv<double> cd0 = const_div_make(st.c[0]);
  auto relax = [&](int li, int lj, int lk, int gi, int gj, int gk) {
    const int q = (lk * TY + lj) * TX + li;
    double r = f[gidx<3>(g, gi, gj, gk)];                    // relax_point<double, 3>, operation for operation
    r = r - st.c[1] * U[q - 1];
    r = r - st.c[2] * U[q + 1];
    r = r - st.c[3] * U[q - TX];
    r = r - st.c[4] * U[q + TX];
    r = r - st.c[5] * U[q - TX * TY];
    r = r - st.c[6] * U[q + TX * TY];
    const double xs = const_div(r, cd0);
    U[q] = omm * U[q] + om * xs;
  };
  // colour 0 (even index sum) on the block and one layer around it: local indices 1 .. B + 2
  {
    constexpr int HX = (BX + 2) / 2, RY = BY + 2, RZ = BZ + 2;
    for (int q = tid; q < HX * RY * RZ; q += EVOMG_RB3D_THREADS) {
      const int lj = 1 + (q / HX) % RY, lk = 1 + q / (HX * RY);
      const int gj = j0 - 2 + lj, gk = k0 - 2 + lk;
      const int li = 1 + 2 * (q % HX) + ((i0 - 1 + gj + gk) & 1);
      const int gi = i0 - 2 + li;
      if (gi >= 1 && gi <= n - 2 && gj >= 1 && gj <= n - 2 && gk >= 1 && gk <= n - 2) relax(li, lj, lk, gi, gj, gk);
    }
  }
  __syncthreads();
  // colour 1 on the block: local indices 2 .. B + 1
  {
    constexpr int HX = BX / 2;
    for (int q = tid; q < HX * BY * BZ; q += EVOMG_RB3D_THREADS) {
      const int lj = 2 + (q / HX) % BY, lk = 2 + q / (HX * BY);
      const int gj = j0 - 2 + lj, gk = k0 - 2 + lk;
      const int li = 2 + 2 * (q % HX) + ((i0 + gj + gk + 1) & 1);
      const int gi = i0 - 2 + li;
      if (gi <= n - 2 && gj <= n - 2 && gk <= n - 2) relax(li, lj, lk, gi, gj, gk);
    }
  }
  __syncthreads();
  for (int q = tid; q < BX * BY * BZ; q += EVOMG_RB3D_THREADS) {
    const int li = 2 + q % BX, lj = 2 + (q / BX) % BY, lk = 2 + q / (BX * BY);
    const int gi = i0 - 2 + li, gj = j0 - 2 + lj, gk = k0 - 2 + lk;
    if (gi <= n - 2 && gj <= n - 2 && gk <= n - 2) y[gidx<3>(g, gi, gj, gk)] = U[(lk * TY + lj) * TX + li];
  }
}

// ---- residual + full weighting in ONE launch on the 3D levels below the tile threshold (round 4) ---------------------------
// RESIDUAL@L, RESTRICT@(L-1) with the residual field dead afterwards were two thread-per-node launches there. Here a 1024-thread
// workgroup owns a block of 8 x 4 x 4 coarse nodes: it loads the 19 x 11 x 11 fine values of u around them into LDS, forms the
// residual on the 17 x 9 x 9 fine nodes the weighting reads (evomg_apply's arithmetic: f - apply_stencil) and then the 27-point
// sums in evomg_restrict<3>'s order (as evomg_small3d_chain does for a whole level): bit-identical, fields and all.
#define EVOMG_RR3D_CX 8
#define EVOMG_RR3D_CY 4
#define EVOMG_RR3D_CZ 4
static __global__ void __launch_bounds__(1024)
evomg_residual_restrict3d_block(Geo g, Geo gc, Stencil<double> st, const double* __restrict__ x, const double* __restrict__ f,
                                double* __restrict__ res_out, double* __restrict__ coarse, int nbx, int nby, int nbz) {
  constexpr int CX = EVOMG_RR3D_CX, CY = EVOMG_RR3D_CY, CZ = EVOMG_RR3D_CZ;
  constexpr int FX = 2 * CX + 1, FY = 2 * CY + 1, FZ = 2 * CZ + 1;        // fine nodes whose residual is needed
  constexpr int UX = FX + 2, UY = FY + 2, UZ = FZ + 2;                    // ... and the values of u around them
  __shared__ double U[UZ * UY * UX];
  __shared__ double R[FZ * FY * FX];
  const int n = g.n, nc = gc.n, tid = (int)threadIdx.x;
  const int bi = (int)blockIdx.x % nbx, bj = ((int)blockIdx.x / nbx) % nby, bk = (int)blockIdx.x / (nbx * nby);
  const int I0 = 1 + bi * CX, J0 = 1 + bj * CY, K0 = 1 + bk * CZ;         // first inner coarse node of the block
  const int i0 = 2 * I0 - 1, j0 = 2 * J0 - 1, k0 = 2 * K0 - 1;            // fine node of R[0]; U[0] sits one further out
  for (int q = tid; q < UX * UY * UZ; q += 1024) {
    const int gi = i0 - 1 + q % UX, gj = j0 - 1 + (q / UX) % UY, gk = k0 - 1 + q / (UX * UY);
    const bool in = gi >= 0 && gi < n && gj >= 0 && gj < n && gk >= 0 && gk < n;
    U[q] = in ? x[gidx<3>(g, gi, gj, gk)] : 0.0;
  }
  __syncthreads();
  Geo gl;                                                                   // the LDS image of u: rows and planes packed
  gl.n = UX;
  gl.pitch = UX;
  gl.plane = UX * UY;
  for (int q = tid; q < FX * FY * FZ; q += 1024) {
    const int li = q % FX, lj = (q / FX) % FY, lk = q / (FX * FY);
    const int gi = i0 + li, gj = j0 + lj, gk = k0 + lk;
    double r = 0.0;
    if (gi >= 1 && gi <= n - 2 && gj >= 1 && gj <= n - 2 && gk >= 1 && gk <= n - 2) {
      r = f[gidx<3>(g, gi, gj, gk)] - apply_stencil<double, 3>(st, U, gl, gidx<3>(gl, li + 1, lj + 1, lk + 1));
      // the residual field as the separate launch leaves it (evomg_small3d_chain writes it too): every fine node by ONE block —
      // the first 2 C layers of its region, the last block of a direction also the layer behind them
      if (res_out && (li < 2 * CX || bi == nbx - 1) && (lj < 2 * CY || bj == nby - 1) && (lk < 2 * CZ || bk == nbz - 1))
        res_out[gidx<3>(g, gi, gj, gk)] = r;
    }
    R[q] = r;
  }
  __syncthreads();
  for (int qc = tid; qc < CX * CY * CZ; qc += 1024) {
    const int I = I0 + qc % CX, J = J0 + (qc / CX) % CY, K = K0 + qc / (CX * CY);
    if (I > nc - 2 || J > nc - 2 || K > nc - 2) continue;
    const int li = 2 * (I - I0) + 1, lj = 2 * (J - J0) + 1, lk = 2 * (K - K0) + 1;     // the coarse node's fine node in R
    double face = 0.0, edge = 0.0, corner = 0.0;
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int nz = (dx != 0) + (dy != 0) + (dz != 0);
          if (nz == 0) continue;
          const double v = R[((lk + dz) * FY + (lj + dy)) * FX + li + dx];
          if (nz == 1) face = face + v;
          else if (nz == 2) edge = edge + v;
          else corner = corner + v;
        }
    double v = 0.125 * R[(lk * FY + lj) * FX + li];
    v = v + 0.0625 * face;
    v = v + 0.03125 * edge;
    v = v + 0.015625 * corner;
    coarse[gidx<3>(gc, I, J, K)] = v;
  }
}
