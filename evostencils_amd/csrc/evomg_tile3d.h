// evomg_tile3d.h — 3D marching tiles whose waves share their y halo through LDS (round 2).
//
// The round-1 3D kernels (evomg_strip3d*) gave every wave a private tile of 128 columns x 4 rows and let it load its own
// halo rows: (TY+2)/TY = 1.5x the x traffic for Jacobi / residual, 2x for the fused red-black sweep, and at 513^3 those
// re-reads reach the fabric (rocprofv3 FETCH_SIZE 1.3-1.4x the algorithmic bytes, profiles/r01_pmc_3d.md) — neither
// L1/L2 nor lock-stepping the waves merged them. Here a workgroup of NW waves owns 128 columns x NW*4 consecutive rows
// and marches along z in lock step:
//   * lane = column pair (16-byte accesses), x neighbours by DPP wave shifts, z neighbours = the wave's own registers
//     (ring of the planes k-1, k, k+1 and the prefetched k+2) — as before;
//   * y neighbours across waves: each wave parks the first and last row of its plane k+1 in LDS one step before that
//     plane becomes the centre plane; after the step's single barrier the neighbouring waves pick them up. Only the two
//     outermost waves of the workgroup still load a halo row from memory: (NW*4+2)/(NW*4) = 1.06x at NW = 8;
//   * halo rows are only ever needed for the centre plane (the 7-point stencil couples y within a plane), so the ring
//     holds the wave's own rows only.
// The kernels write INNER nodes only; the caller makes y's boundary faces equal to x's (the ping-pong twin already
// has them — Level::alt_mirror — or evomg_copy_faces runs first), exactly as for the block smoothers.
// Operation order = relax7 / residual7 (evomg_strip.h) = relax_point / apply_stencil = the oracle's FN(relax3) /
// FN(apply_st3): results are bit-identical to every other 3D Jacobi / residual kernel of this library.
#pragma once
#include "evomg_strip.h"

#define EVOMG_TILE3D_TY 4
// two-cell block-Jacobi shapes as tile modes (K3): the tile's 2 x 2 local system is eliminated once on the host
// (make_block_plan, the oracle's order) and replayed here on the right-hand sides, exactly as
// evomg_block_jacobi_fast does — bit-identical — but with the tile kernels' memory access.
//   (2,1,1): tiles are the column pairs (1,2), (3,4), ... = element b of a lane and element a of its right neighbour:
//            each lane forms the right-hand side of its b as first cell and of its a as second cell, trades them with
//            its neighbours by one DPP shift each way and replays the elimination of both tiles it takes part in;
//   (1,2,1): tiles are the row pairs (row0, row0+1), (row0+2, row0+3) of a wave: no exchange at all.
// The last inner column / row has no partner (2^L - 1 inner nodes): a one-cell tile, i.e. the point relaxation.
enum { TILE_BLOCK211 = 4, TILE_BLOCK121 = 5 };
template <typename T> struct TwoCellPlan {
  T fac, upper, pivot0, pivot1;      // BlockPlan: fac[1][0], upper[0][1], pivot[0], pivot[1]
};

// grid: n_blocks (rounded up to a multiple of 8) workgroups of 64*NW threads; block -> (y group fastest, strip, z chunk)
template <typename T, int MODE, int NW, bool NT_F>
__global__ void __launch_bounds__(64 * NW)
evomg_tile3d(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f, T* __restrict__ y, double om,
             double omm, int planes_per_chunk, int n_strips, int n_ygroups, int n_blocks, TwoCellPlan<T> pl) {
  static_assert(MODE == STRIP_JACOBI || MODE == STRIP_RESIDUAL || MODE == TILE_BLOCK211 || MODE == TILE_BLOCK121,
                "3D tiles: Jacobi, residual, two-cell block-Jacobi");
  constexpr int TY = EVOMG_TILE3D_TY;
  __shared__ __attribute__((aligned(16))) T edge[2][NW][2][128];   // [plane parity][wave][first | last row][column]
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (logical_block >= n_blocks) return;                   // whole workgroup: no barrier is skipped by a part of it
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ygroup = logical_block % n_ygroups;
  const int strip = (logical_block / n_ygroups) % n_strips;
  const int chunk = logical_block / (n_ygroups * n_strips);
  const int n = g.n;
  const long pitch = g.pitch, plane = g.plane;
  const int c0 = strip * EVOMG_STRIP_COLS - 2 + 2 * lane;
  const bool col_ok = c0 >= 0 && c0 < n;
  const int row0 = 1 + (ygroup * NW + w) * TY;              // first row of this wave (inner rows start at 1)
  const int z0 = 1 + chunk * planes_per_chunk;              // first plane of this chunk (inner planes start at 1)
  const int z1 = min(z0 + planes_per_chunk, n - 1);         // one past the last plane computed
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
  const bool store_lane = lane >= 1 && lane <= 62 && col_ok && (inner_a || inner_b);
  const int c_clamped = min(max(c0, 0), n - 1);
  // branch-free clamped loads (see evomg_strip2d): whatever an out-of-domain slot delivers only feeds non-inner nodes
  auto load_x = [&](int row, int pl) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl, 0), n - 1);      // wave-uniform
    return load_pair<T>(x + plane * p + pitch * r + c_clamped);
  };
  auto load_f = [&](int row, int pl) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl, 0), n - 1);
    if constexpr (NT_F) return load_pair_nt<T>(f + plane * p + pitch * r + c_clamped);
    else return load_pair<T>(f + plane * p + pitch * r + c_clamped);
  };
  auto park = [&](int parity, const Pair<T>& first, const Pair<T>& last) {
    *reinterpret_cast<Pair<T>*>(&edge[parity][w][0][2 * lane]) = first;
    *reinterpret_cast<Pair<T>*>(&edge[parity][w][1][2 * lane]) = last;
  };
  auto pick = [&](int parity, int wave, int which) -> Pair<T> {
    return *reinterpret_cast<const Pair<T>*>(&edge[parity][wave][which][2 * lane]);
  };

  Pair<T> xr[4][TY];          // ring slot (k - z0 + 1) & 3 holds plane k (own rows row0 .. row0+TY-1)
  Pair<T> fr[2][TY];          // ring slot (k - z0) & 1 holds f of plane k
  Pair<T> hs, hn;             // rows row0-1 and row0+TY of the centre plane
  Pair<T> gs, gn;             // the same rows of the next centre plane, when they come from memory (outermost waves)
#pragma unroll
  for (int t = 0; t < TY; ++t) {
    xr[0][t] = load_x(row0 + t, z0 - 1);
    xr[1][t] = load_x(row0 + t, z0);
    xr[2][t] = load_x(row0 + t, z0 + 1);
    fr[0][t] = load_f(row0 + t, z0);
  }
  hs = load_x(row0 - 1, z0);
  hn = load_x(row0 + TY, z0);
  if (w > 0 || w < NW - 1) {
    // halo rows of the first centre plane from the neighbouring waves (the outermost waves keep their own loads)
    park(z0 & 1, xr[1][0], xr[1][TY - 1]);
  }
  __syncthreads();
  if (w > 0) hs = pick(z0 & 1, w - 1, 1);
  if (w < NW - 1) hn = pick(z0 & 1, w + 1, 0);

  for (int kb = z0; kb < z1; kb += 4) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int k = kb + d;
      if (k < z1) {                                            // uniform over the workgroup (same chunk)
        // prefetch plane k+2 of x (own rows), plane k+1 of f, and — outermost waves — the halo rows of plane k+1
#pragma unroll
        for (int t = 0; t < TY; ++t) xr[(d + 3) & 3][t] = load_x(row0 + t, k + 2);
#pragma unroll
        for (int t = 0; t < TY; ++t) fr[(d + 1) & 1][t] = load_f(row0 + t, k + 1);
        if (w == 0) gs = load_x(row0 - 1, k + 1);
        if (w == NW - 1) gn = load_x(row0 + TY, k + 1);
        const Pair<T>* below = xr[d & 3];
        const Pair<T>* here = xr[(d + 1) & 3];
        const Pair<T>* above = xr[(d + 2) & 3];
        const Pair<T>* fk = fr[d & 1];
        // plane k+1 is about to become the centre plane: park its first and last row for the neighbours
        park((k + 1) & 1, above[0], above[TY - 1]);
        if constexpr (MODE == TILE_BLOCK121) {
#pragma unroll
          for (int tp = 0; tp < TY; tp += 2) {
            const int rA = row0 + tp;                            // first cell of the tile; rA + 1 the second
            const Pair<T> cA = here[tp], cB = here[tp + 1];
            const Pair<T> sA = (tp == 0) ? hs : here[tp > 0 ? tp - 1 : 0], nB = (tp + 1 == TY - 1) ? hn : here[tp + 2 < TY ? tp + 2 : tp];
            const T wA = from_left(cA.b), eA = from_right(cA.a), wB = from_left(cB.b), eB = from_right(cB.a);
            auto cell = [&](T f0, T w_, T e_, T yo, T d_, T u_, bool second) -> T {
              T r = f0;
              r = sc_sub(r, sc_mul(st.c[1], w_));
              r = sc_sub(r, sc_mul(st.c[2], e_));
              r = sc_sub(r, sc_mul(st.c[second ? 4 : 3], yo));
              r = sc_sub(r, sc_mul(st.c[5], d_));
              r = sc_sub(r, sc_mul(st.c[6], u_));
              return r;
            };
            auto solve = [&](T r0, T r1, T x0, T x1, T& o0, T& o1) {
              r1 = sc_sub(r1, sc_mul(pl.fac, r0));
              r1 = sc_div(r1, pl.pivot1);
              r0 = sc_div(sc_sub(r0, sc_mul(pl.upper, r1)), pl.pivot0);
              o0 = sc_add(sc_rmul(omm, x0), sc_rmul(om, r0));
              o1 = sc_add(sc_rmul(omm, x1), sc_rmul(om, r1));
            };
            Pair<T> oA = cA, oB = cB;
            if (rA + 1 <= n - 2) {                               // full tile (wave-uniform)
              T o0, o1;
              solve(cell(fk[tp].a, wA, cA.b, sA.a, below[tp].a, above[tp].a, false),
                    cell(fk[tp + 1].a, wB, cB.b, nB.a, below[tp + 1].a, above[tp + 1].a, true), cA.a, cB.a, o0, o1);
              if (inner_a) { oA.a = o0; oB.a = o1; }
              solve(cell(fk[tp].b, cA.a, eA, sA.b, below[tp].b, above[tp].b, false),
                    cell(fk[tp + 1].b, cB.a, eB, nB.b, below[tp + 1].b, above[tp + 1].b, true), cA.b, cB.b, o0, o1);
              if (inner_b) { oA.b = o0; oB.b = o1; }
              if (store_lane) {
                store_pair(y + plane * k + pitch * rA + c0, oA);
                store_pair(y + plane * k + pitch * (rA + 1) + c0, oB);
              }
            } else if (rA == n - 2) {                            // the last inner row: a one-cell tile
              if (inner_a) oA.a = relax7(st, cA.a, wA, cA.b, sA.a, cB.a, below[tp].a, above[tp].a, fk[tp].a, om, omm);
              if (inner_b) oA.b = relax7(st, cA.b, cA.a, eA, sA.b, cB.b, below[tp].b, above[tp].b, fk[tp].b, om, omm);
              if (store_lane) store_pair(y + plane * k + pitch * rA + c0, oA);
            }
          }
        } else {
#pragma unroll
        for (int t = 0; t < TY; ++t) {
          const int row = row0 + t;
          const Pair<T> c = here[t], s = (t == 0) ? hs : here[t > 0 ? t - 1 : 0], nn = (t == TY - 1) ? hn : here[t < TY - 1 ? t + 1 : t];
          const Pair<T> dn = below[t], up = above[t];
          const T wa = from_left(c.b), eb = from_right(c.a);
          if constexpr (MODE == TILE_BLOCK211) {
            // b as first cell of its tile (west neighbour a is outside), a as second cell (east neighbour b is outside)
            T rb0 = fk[t].b;
            rb0 = sc_sub(rb0, sc_mul(st.c[1], c.a));
            rb0 = sc_sub(rb0, sc_mul(st.c[3], s.b));
            rb0 = sc_sub(rb0, sc_mul(st.c[4], nn.b));
            rb0 = sc_sub(rb0, sc_mul(st.c[5], dn.b));
            rb0 = sc_sub(rb0, sc_mul(st.c[6], up.b));
            T ra1 = fk[t].a;
            ra1 = sc_sub(ra1, sc_mul(st.c[2], c.b));
            ra1 = sc_sub(ra1, sc_mul(st.c[3], s.a));
            ra1 = sc_sub(ra1, sc_mul(st.c[4], nn.a));
            ra1 = sc_sub(ra1, sc_mul(st.c[5], dn.a));
            ra1 = sc_sub(ra1, sc_mul(st.c[6], up.a));
            const T r0_left = from_left(rb0), r1_right = from_right(ra1);          // whole wave
            // tile (left b, a): a is the second unknown
            const T xa = sc_div(sc_sub(ra1, sc_mul(pl.fac, r0_left)), pl.pivot1);
            // tile (b, right a): b is the first unknown
            const T x1 = sc_div(sc_sub(r1_right, sc_mul(pl.fac, rb0)), pl.pivot1);
            const T xb = sc_div(sc_sub(rb0, sc_mul(pl.upper, x1)), pl.pivot0);
            if (row <= n - 2 && store_lane) {
              Pair<T> out = c;
              if (inner_a) out.a = sc_add(sc_rmul(omm, c.a), sc_rmul(om, xa));
              if (inner_b) out.b = (c0 + 1 == n - 2) ? relax7(st, c.b, c.a, eb, s.b, nn.b, dn.b, up.b, fk[t].b, om, omm)
                                                     : sc_add(sc_rmul(omm, c.b), sc_rmul(om, xb));
              store_pair(y + plane * k + pitch * row + c0, out);
            }
          } else
          if (row <= n - 2 && store_lane) {                    // row: wave-uniform
            T* dst = y + plane * k + pitch * row + c0;
            Pair<T> out;
            if constexpr (MODE == STRIP_JACOBI) {
              out.a = inner_a ? relax7(st, c.a, wa, c.b, s.a, nn.a, dn.a, up.a, fk[t].a, om, omm) : c.a;
              out.b = inner_b ? relax7(st, c.b, c.a, eb, s.b, nn.b, dn.b, up.b, fk[t].b, om, omm) : c.b;
              store_pair(dst, out);
            } else {
              out.a = residual7(st, c.a, wa, c.b, s.a, nn.a, dn.a, up.a, fk[t].a);
              out.b = residual7(st, c.b, c.a, eb, s.b, nn.b, dn.b, up.b, fk[t].b);
              if (inner_a && inner_b) store_pair(dst, out);
              else if (inner_a) dst[0] = out.a;
              else dst[1] = out.b;
            }
          }
        }
        }
        __syncthreads();
        hs = (w > 0) ? pick((k + 1) & 1, w - 1, 1) : gs;
        hn = (w < NW - 1) ? pick((k + 1) & 1, w + 1, 0) : gn;
      }
    }
  }
}
