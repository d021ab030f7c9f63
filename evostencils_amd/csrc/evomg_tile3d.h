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
#include <type_traits>

#define EVOMG_TILE3D_TY 4
// two-cell block-Jacobi shapes as tile modes (K3): the tile's 2 x 2 local system is eliminated once on the host
// (make_block_plan, the oracle's order) and replayed here on the right-hand sides, exactly as
// evomg_block_jacobi_fast does — bit-identical — but with the tile kernels' memory access.
//   (2,1,1): tiles are the column pairs (1,2), (3,4), ... = element b of a lane and element a of its right neighbour:
//            each lane forms the right-hand side of its b as first cell and of its a as second cell, trades them with
//            its neighbours by one DPP shift each way and replays the elimination of both tiles it takes part in;
//   (1,2,1): tiles are the row pairs (row0, row0+1), (row0+2, row0+3) of a wave: no exchange at all.
// The last inner column / row has no partner (2^L - 1 inner nodes): a one-cell tile, i.e. the point relaxation.
enum { TILE_BLOCK211 = 4, TILE_BLOCK121 = 5, TILE_RESNORM = 6, TILE_RESNORM_W = 7 };
// TILE_RESNORM(_W): the solver loop's residual + norm in one pass (K4 + K9): the squared residuals of every inner row
// (strip s, row j, plane k) are summed over the wave — element a, then element b per lane, then the wave64 tree — into
// aux[((k-1)*m + (j-1))*n_strips + s]; a second, fixed-order reduction over that array gives the norm. The order depends
// on the strip width only (not on NW, chunking or workgroup order): it is the canonical summation order of the 3D solver
// loop from 129^3 up and is restated by the oracle (FN(norm_tile3d)). _W also writes the residual field; without it the
// pass moves 16 instead of 24 + 8 B/node (the loop's residual field is usually dead: the cycle recomputes it).
template <typename T> struct TwoCellPlan {
  T fac, upper, pivot0, pivot1;      // BlockPlan: fac[1][0], upper[0][1], pivot[0], pivot[1]
};

// grid: n_blocks (rounded up to a multiple of 8) workgroups of 64*NW threads; block -> (y group fastest, strip, z chunk)
// build knob for A/B only: -DEVOMG_TILE3D_WPE=4 forces 128 VGPRs (two workgroups per compute unit, 36-132 B of scratch per
// thread) — measured slower at 513^3 (Jacobi 666 -> 739 us), hence off
#ifndef EVOMG_TILE3D_WPE
#define EVOMG_TILE3D_WPE 0
#endif
template <typename T, int MODE, int NW, int NT, bool POW2 = false>
__global__ void __launch_bounds__(64 * NW)
#if EVOMG_TILE3D_WPE > 0
__attribute__((amdgpu_waves_per_eu(EVOMG_TILE3D_WPE, EVOMG_TILE3D_WPE)))
#endif
evomg_tile3d(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f, T* __restrict__ y, double om,
             double omm, int planes_per_chunk, int n_strips, int n_ygroups, int n_blocks, TwoCellPlan<T> pl, int order,
             T* __restrict__ aux, int xpitch) {
  static_assert(MODE == STRIP_JACOBI || MODE == STRIP_RESIDUAL || MODE == TILE_BLOCK211 || MODE == TILE_BLOCK121 ||
                    MODE == TILE_RESNORM || MODE == TILE_RESNORM_W,
                "3D tiles: Jacobi, residual (+ norm), two-cell block-Jacobi");
  constexpr int TY = EVOMG_TILE3D_TY;
  __shared__ __attribute__((aligned(16))) T edge[2][NW][2][128];   // [plane parity][wave][first | last row][column]
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (logical_block >= n_blocks) return;                   // whole workgroup: no barrier is skipped by a part of it
  prio_level(g);
  const int lane = threadIdx.x & 63;
  const ConstDiv<T> cd0 = const_div_make(st.c[0]);     // quotients by the stencil centre (evomg_scalar.h)
  const ConstDiv<T> cdp0 = const_div_make(pl.pivot0), cdp1 = const_div_make(pl.pivot1);
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // order 0: y group fastest (y neighbours adjacent in time); order 1: strip fastest — workgroups that run together then
  // cover whole rows, i.e. one contiguous span of NW*4 rows per plane
  const int ygroup = order ? (logical_block / n_strips) % n_ygroups : logical_block % n_ygroups;
  const int strip = order ? logical_block % n_strips : (logical_block / n_ygroups) % n_strips;
  const int chunk = logical_block / (n_ygroups * n_strips);
  const bool last_strip = strip == n_strips - 1;            // uniform
  const int n = g.n;
  const long pitch = g.pitch, plane = g.plane;
  // xpitch (even, <= EVOMG_STRIP_COLS): columns a strip stores — the host spreads the columns evenly over the strips
  // (tile3d_xpitch), so that no workgroup of a launch is lighter than the others; lanes past the strip's halo lane load
  // what that lane loads (one address: no traffic of their own)
  const int c0 = strip * xpitch - 2 + 2 * lane;
  const bool col_ok = c0 >= 0 && c0 < n;
  const int row0 = 1 + (ygroup * NW + w) * TY;              // first row of this wave (inner rows start at 1)
  const int z0 = 1 + chunk * planes_per_chunk;              // first plane of this chunk (inner planes start at 1)
  const int z1 = min(z0 + planes_per_chunk, n - 1);         // one past the last plane computed
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
  const bool store_lane = lane >= 1 && 2 * lane <= xpitch && col_ok && (inner_a || inner_b);
  const int c_clamped = min(max(c0, 0), min(n - 1, strip * xpitch + xpitch));
  // branch-free clamped loads (see evomg_strip2d): whatever an out-of-domain slot delivers only feeds non-inner nodes
  auto load_x = [&](int row, int pl) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl, 0), n - 1);      // wave-uniform
    if constexpr ((NT & 2) != 0) return load_pair_nt<T>(x + plane * p + pitch * r + c_clamped);
    else return load_pair<T>(x + plane * p + pitch * r + c_clamped);
  };
  auto store_y = [&](T* dst, const Pair<T>& v) {
    if constexpr ((NT & 4) != 0) __builtin_nontemporal_store(evomg_f64x2{v.a, v.b}, reinterpret_cast<evomg_f64x2*>(dst));
    else store_pair(dst, v);
  };
  auto load_f = [&](int row, int pl) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl, 0), n - 1);
    if constexpr ((NT & 1) != 0) return load_pair_nt<T>(f + plane * p + pitch * r + c_clamped);
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
              r = msub<POW2>(r, st.c[1], w_);
              r = msub<POW2>(r, st.c[2], e_);
              r = msub<POW2>(r, st.c[second ? 4 : 3], yo);
              r = msub<POW2>(r, st.c[5], d_);
              r = msub<POW2>(r, st.c[6], u_);
              return r;
            };
            auto solve = [&](T r0, T r1, T x0, T x1, T& o0, T& o1) {
              r1 = sc_sub(r1, sc_mul(pl.fac, r0));
              r1 = const_div(r1, cdp1);
              r0 = const_div(sc_sub(r0, sc_mul(pl.upper, r1)), cdp0);
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
                store_y(y + plane * k + pitch * rA + c0, oA);
                store_y(y + plane * k + pitch * (rA + 1) + c0, oB);
              }
            } else if (rA == n - 2) {                            // the last inner row: a one-cell tile
              if (inner_a) oA.a = relax7p<POW2>(st, cd0, cA.a, wA, cA.b, sA.a, cB.a, below[tp].a, above[tp].a, fk[tp].a, om, omm);
              if (inner_b) oA.b = relax7p<POW2>(st, cd0, cA.b, cA.a, eA, sA.b, cB.b, below[tp].b, above[tp].b, fk[tp].b, om, omm);
              if (store_lane) store_y(y + plane * k + pitch * rA + c0, oA);
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
            rb0 = msub<POW2>(rb0, st.c[1], c.a);
            rb0 = msub<POW2>(rb0, st.c[3], s.b);
            rb0 = msub<POW2>(rb0, st.c[4], nn.b);
            rb0 = msub<POW2>(rb0, st.c[5], dn.b);
            rb0 = msub<POW2>(rb0, st.c[6], up.b);
            T ra1 = fk[t].a;
            ra1 = msub<POW2>(ra1, st.c[2], c.b);
            ra1 = msub<POW2>(ra1, st.c[3], s.a);
            ra1 = msub<POW2>(ra1, st.c[4], nn.a);
            ra1 = msub<POW2>(ra1, st.c[5], dn.a);
            ra1 = msub<POW2>(ra1, st.c[6], up.a);
            const T r0_left = from_left(rb0);                                      // whole wave
            // tile (left b, a): a is the second unknown
            const T xa = const_div(sc_sub(ra1, sc_mul(pl.fac, r0_left)), cdp1);
            // tile (b, right a): b is the first unknown; the second unknown of that tile is the right lane's xa
            // (same operands, same operations: the value, not a recomputation of it, crosses the lanes)
            const T x1 = from_right(xa);                                           // whole wave
            const T xb = const_div(sc_sub(rb0, sc_mul(pl.upper, x1)), cdp0);
            if (row <= n - 2 && store_lane) {
              Pair<T> out = c;
              if (inner_a) out.a = sc_add(sc_rmul(omm, c.a), sc_rmul(om, xa));
              if (inner_b) out.b = sc_add(sc_rmul(omm, c.b), sc_rmul(om, xb));
              // the last inner column has no partner: a one-cell tile (only the last strip has such a lane)
              if (last_strip && c0 + 1 == n - 2) out.b = relax7p<POW2>(st, cd0, c.b, c.a, eb, s.b, nn.b, dn.b, up.b, fk[t].b, om, omm);
              store_y(y + plane * k + pitch * row + c0, out);
            }
          } else if constexpr (MODE == TILE_RESNORM || MODE == TILE_RESNORM_W) {
            const bool valid = row <= n - 2 && store_lane;     // (row: wave-uniform)
            const T ra = residual7p<POW2>(st, c.a, wa, c.b, s.a, nn.a, dn.a, up.a, fk[t].a);
            const T rb = residual7p<POW2>(st, c.b, c.a, eb, s.b, nn.b, dn.b, up.b, fk[t].b);
            T sq = scalar_traits<T>::zero();
            if (valid && inner_a) sq = sc_add(sq, sc_mul(ra, ra));
            if (valid && inner_b) sq = sc_add(sq, sc_mul(rb, rb));
            sq = wave_sum_fast(sq);                                  // whole wave
            if (row <= n - 2) {
              if (lane == 0) aux[((long)(k - 1) * (n - 2) + (row - 1)) * n_strips + strip] = sq;
              if constexpr (MODE == TILE_RESNORM_W) {
                if (valid) {
                  T* dst = y + plane * k + pitch * row + c0;
                  if (inner_a && inner_b) store_pair(dst, Pair<T>{ra, rb});
                  else if (inner_a) dst[0] = ra;
                  else dst[1] = rb;
                }
              }
            }
          } else
          if (row <= n - 2 && store_lane) {                    // row: wave-uniform
            T* dst = y + plane * k + pitch * row + c0;
            Pair<T> out;
            if constexpr (MODE == STRIP_JACOBI) {
              out.a = inner_a ? relax7p<POW2>(st, cd0, c.a, wa, c.b, s.a, nn.a, dn.a, up.a, fk[t].a, om, omm) : c.a;
              out.b = inner_b ? relax7p<POW2>(st, cd0, c.b, c.a, eb, s.b, nn.b, dn.b, up.b, fk[t].b, om, omm) : c.b;
              store_y(dst, out);
            } else {
              out.a = residual7p<POW2>(st, c.a, wa, c.b, s.a, nn.a, dn.a, up.a, fk[t].a);
              out.b = residual7p<POW2>(st, c.b, c.a, eb, s.b, nn.b, dn.b, up.b, fk[t].b);
              if (inner_a && inner_b) store_y(dst, out);
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

// ---- 3D fused two-colour red-black Gauss-Seidel (K2) on LDS-sharing tiles, out of place -------------------------------
// The lag scheme of evomg_strip3d_rbgs (step k: reds of plane k from old values, then blacks of plane k-1 from the new
// reds of planes k-2, k-1, k) with the y direction shared inside a workgroup instead of recomputed per wave:
//   * old x: as in evomg_tile3d, each wave parks the edge rows of plane k+1 one step before it becomes the centre plane;
//   * new reds: after relaxing the reds of plane k a wave parks those of its first and last row (one value per lane:
//     which element of a pair is red alternates with the row); the neighbours need them one step later, for the blacks
//     of plane k. Both exchanges ride on the step's single barrier.
//   * the workgroup relaxes reds on all its H = NW*4 rows but can finish blacks only where both neighbouring red rows
//     exist: it stores rows R0+1 .. R0+H-2 and consecutive workgroups overlap by two rows (x: (H+2)/(H-2), f: H/(H-2):
//     1.13 / 1.07 at NW = 8, against 2.0 / 1.5 per private 4-row tile). The topmost workgroup also finishes its last row:
//     the row above it is the boundary, whose "reds" are just the old values.
// Colour 0 = even index sum. c0, R0 (a multiple of H-2) and the first plane of a chunk are even, so which element of a
// pair is red is static inside the twice-unrolled z loop. Inner nodes only are written (see evomg_tile3d).
// Operation order = relax7 = relax_point: bit-identical to the two-pass form (evomg_rbgs_colour x 2) and to the oracle.
template <typename T, int NW, int NT, bool POW2 = false>
__global__ void __launch_bounds__(64 * NW)
evomg_tile3d_rbgs(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f, T* __restrict__ y, double om,
                  double omm, int planes_per_chunk, int n_strips, int n_ygroups, int n_blocks, int xpitch) {
  constexpr int TY = EVOMG_TILE3D_TY, H = NW * TY;
  static_assert((H - 2) % 2 == 0 && TY % 2 == 0, "row parity must be static");
  __shared__ __attribute__((aligned(16))) T edge_x[2][NW][2][128];   // old x: [plane parity][wave][first | last row][column]
  __shared__ __attribute__((aligned(16))) T edge_r[2][NW][2][64];    // new reds: one value per lane
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (logical_block >= n_blocks) return;
  prio_level(g);
  const int lane = threadIdx.x & 63;
  const ConstDiv<T> cd0 = const_div_make(st.c[0]);     // quotients by the stencil centre (evomg_scalar.h)
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int strip = logical_block % n_strips;                 // strip fastest: concurrent workgroups cover whole rows
  const int ygroup = (logical_block / n_strips) % n_ygroups;
  const int chunk = logical_block / (n_ygroups * n_strips);
  const int n = g.n;
  const long pitch = g.pitch, plane = g.plane;
  const int c0 = strip * xpitch - 2 + 2 * lane;               // xpitch: see evomg_tile3d
  const bool col_ok = c0 >= 0 && c0 < n;
  const int R0 = ygroup * (H - 2);                            // first red row of the workgroup (even)
  const int row0 = R0 + w * TY;
  const bool top_group = R0 + H - 1 >= n - 2;                 // the row above the workgroup is the boundary (or beyond)
  const int z0 = 1 + chunk * planes_per_chunk;                // stored planes [z0, z1)
  const int z1 = min(z0 + planes_per_chunk, n - 1);
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
  const bool store_lane = lane >= 1 && 2 * lane <= xpitch && col_ok && (inner_a || inner_b);
  const int c_clamped = min(max(c0, 0), min(n - 1, strip * xpitch + xpitch));
  const T zero = scalar_traits<T>::zero();
  auto load_x = [&](int row, int pl) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl, 0), n - 1);
    return load_pair<T>(x + plane * p + pitch * r + c_clamped);
  };
  auto load_f = [&](int row, int pl) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl, 0), n - 1);
    if constexpr ((NT & 1) != 0) return load_pair_nt<T>(f + plane * p + pitch * r + c_clamped);
    else return load_pair<T>(f + plane * p + pitch * r + c_clamped);
  };
  auto park_x = [&](int parity, const Pair<T>& first, const Pair<T>& last) {
    *reinterpret_cast<Pair<T>*>(&edge_x[parity][w][0][2 * lane]) = first;
    *reinterpret_cast<Pair<T>*>(&edge_x[parity][w][1][2 * lane]) = last;
  };
  auto pick_x = [&](int parity, int wave, int which) -> Pair<T> {
    return *reinterpret_cast<const Pair<T>*>(&edge_x[parity][wave][which][2 * lane]);
  };

  const int ks = z0 - 1;                 // first plane whose reds are relaxed (even)
  const int ke = z1 + 1;                 // one past the last
  Pair<T> xb[TY], xc[TY], xa[TY], xq[TY];      // old x, planes k-1, k, k+1, k+2 (prefetch); own rows
  Pair<T> fb[TY], fc[TY], fq[TY];              // f, planes k-1, k, k+1 (prefetch)
  T r2[TY], r1[TY], rc[TY];                    // new reds of planes k-2, k-1, k
  Pair<T> hs, hn, gs, gn;                      // old-x halo rows of the centre plane / of the next one (outermost waves)
  Pair<T> hn_prev;                             // old-x row above the wave in plane k-1 (top group: the boundary row)
  T rs1 = zero, rn1 = zero;                    // new reds of plane k-1 in the rows below / above the wave
#pragma unroll
  for (int t = 0; t < TY; ++t) {
    xb[t] = load_x(row0 + t, ks - 1);
    xc[t] = load_x(row0 + t, ks);
    xa[t] = load_x(row0 + t, ks + 1);
    fc[t] = load_f(row0 + t, ks);
    fb[t] = Pair<T>{zero, zero};
    r2[t] = zero;
    r1[t] = zero;
  }
  hs = load_x(row0 - 1, ks);
  hn = load_x(row0 + TY, ks);
  hn_prev = hn;
  park_x(ks & 1, xc[0], xc[TY - 1]);
  __syncthreads();
  if (w > 0) hs = pick_x(ks & 1, w - 1, 1);
  if (w < NW - 1) hn = pick_x(ks & 1, w + 1, 0);

  for (int kb = ks; kb < ke; kb += 2) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int k = kb + d;                                    // d == 0: k even, d == 1: k odd (ks is even)
      if (k < ke) {
#pragma unroll
        for (int t = 0; t < TY; ++t) xq[t] = load_x(row0 + t, k + 2);
#pragma unroll
        for (int t = 0; t < TY; ++t) fq[t] = load_f(row0 + t, k + 1);
        if (w == 0) gs = load_x(row0 - 1, k + 1);
        if (w == NW - 1) gn = load_x(row0 + TY, k + 1);
        park_x((k + 1) & 1, xa[0], xa[TY - 1]);
        const bool plane_inner = k >= 1 && k <= n - 2;
        // ---- reds of plane k, own rows ----
#pragma unroll
        for (int t = 0; t < TY; ++t) {
          const int row = row0 + t;
          const bool row_inner = plane_inner && row >= 1 && row <= n - 2;
          const Pair<T> c = xc[t], s = (t == 0) ? hs : xc[t > 0 ? t - 1 : 0], nn = (t == TY - 1) ? hn : xc[t < TY - 1 ? t + 1 : t];
          const Pair<T> dn = xb[t], up = xa[t];
          // row0 even => row has the parity of t; k has the parity of d: a is red iff t + d is even
          if (((t + d) & 1) == 0) {
            const T we = from_left(c.b);
            rc[t] = (row_inner && inner_a) ? relax7p<POW2>(st, cd0, c.a, we, c.b, s.a, nn.a, dn.a, up.a, fc[t].a, om, omm) : c.a;
          } else {
            const T ea = from_right(c.a);
            rc[t] = (row_inner && inner_b) ? relax7p<POW2>(st, cd0, c.b, c.a, ea, s.b, nn.b, dn.b, up.b, fc[t].b, om, omm) : c.b;
          }
        }
        edge_r[k & 1][w][0][lane] = rc[0];
        edge_r[k & 1][w][1][lane] = rc[TY - 1];
        // ---- blacks of plane k-1, own rows, from the new reds of planes k-2, k-1, k ----
        const int kbk = k - 1;
        const bool bplane_inner = kbk >= 1 && kbk <= n - 2;
        const bool bplane_stored = kbk >= z0 && kbk < z1;
        // the boundary row above the topmost workgroup never changes: its "new reds" are its old values (plane k-1)
        const bool own_top = (w == NW - 1);
#pragma unroll
        for (int t = 0; t < TY; ++t) {
          const int row = row0 + t;
          const bool row_inner = bplane_inner && row >= 1 && row <= n - 2;
          const Pair<T> old = xb[t];
          const T red = r1[t];
          // plane k-1 has the parity of d + 1: a is red iff t + d + 1 is even; the rows above / below hold their red in
          // the OTHER element
          const bool a_red = ((t + d + 1) & 1) == 0;
          const T top_old = a_red ? hn_prev.b : hn_prev.a;      // red of the row above, were it a boundary row
          const T rs = (t == 0) ? rs1 : r1[t > 0 ? t - 1 : 0];
          const T rn = (t == TY - 1) ? ((own_top && top_group) ? top_old : rn1) : r1[t < TY - 1 ? t + 1 : t];
          Pair<T> out;
          if (a_red) {                                         // a red, b black
            const T e = from_right(red);
            out.a = red;
            out.b = (row_inner && inner_b) ? relax7p<POW2>(st, cd0, old.b, red, e, rs, rn, r2[t], rc[t], fb[t].b, om, omm) : old.b;
          } else {                                             // b red, a black
            const T we = from_left(red);
            out.a = (row_inner && inner_a) ? relax7p<POW2>(st, cd0, old.a, we, red, rs, rn, r2[t], rc[t], fb[t].a, om, omm) : old.a;
            out.b = red;
          }
          const bool row_stored = row >= 1 && row <= n - 2 && row >= R0 + 1 && (row <= R0 + H - 2 || top_group);
          if (store_lane && bplane_stored && row_stored) store_pair(y + plane * kbk + pitch * row + c0, out);
        }
        __syncthreads();
        hn_prev = hn;                                          // halo of plane k: next step's plane k-1
        hs = (w > 0) ? pick_x((k + 1) & 1, w - 1, 1) : gs;
        hn = (w < NW - 1) ? pick_x((k + 1) & 1, w + 1, 0) : gn;
        rs1 = (w > 0) ? edge_r[k & 1][w - 1][1][lane] : zero;
        rn1 = (w < NW - 1) ? edge_r[k & 1][w + 1][0][lane] : zero;
        // ---- advance one plane ----
#pragma unroll
        for (int t = 0; t < TY; ++t) { r2[t] = r1[t]; r1[t] = rc[t]; fb[t] = fc[t]; fc[t] = fq[t]; xb[t] = xc[t]; xc[t] = xa[t]; xa[t] = xq[t]; }
      }
    }
  }
}

// first stage of the fixed-order reduction of a long array of partial sums: block b of gridDim.x, thread t adds the
// elements b*256 + t, + gridDim.x*256, ...; then the block_sum trees; evomg_dot_final adds the block sums
template <typename T>
__global__ void __launch_bounds__(256) evomg_sum_stage1(const T* __restrict__ p, long count, T* __restrict__ out) {
  prio_latency();
  __shared__ T red[17];
  T s = scalar_traits<T>::zero();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) s = sc_add(s, p[i]);
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// ---- TWO consecutive Jacobi-type sweeps in one pass (temporal fusion) --------------------------------------------------
// V(2,2)-type candidates relax the same field twice in a row with the same smoother (BASELINE config C4: block-Jacobi
// (2,1,1) twice before and twice after the coarse-grid correction). The red-black kernel above is already a two-stage
// pipeline along z; here both stages are full sweeps: at step k the wave forms y1 = S(x) on plane k from old values and
// then y2 = S(y1) on plane k-1 from y1 on the planes k-2, k-1, k (registers) — y1 never reaches memory: x and f are read
// once, y2 is written once, 24 instead of 48 B/node for the pair. y direction as in evomg_tile3d_rbgs: waves trade the
// old-x edge rows of plane k+1 and the y1 edge rows of plane k through LDS on the step's single barrier; a workgroup
// forms y1 on its 32 rows, finishes y2 on 30, consecutive workgroups overlap by two rows; the topmost workgroup finishes
// its last row against the boundary row (y1 = x there). Sweep arithmetic = the one-sweep tile kernel's, applied twice:
// bit-identical to two launches.
// PRO (round 2): the sweeps' input is x + om_p * P(coarse) instead of x — the PROLONG_ADD that precedes the post-smoothing
// sweeps of a V-cycle — formed on the fly exactly as evomg_prolong_cell3d forms it (sums from zero in dz, dy, dx order,
// one scaling, then x + om_p * v on inner nodes): the corrected field is never written nor read back (a 1.1 GB
// read-modify-write pass at 513^3 less). Every row of x the kernel loads is corrected before anything uses it:
//   * own rows and the outermost waves' halo rows of plane k+2 at the END of step k (they were loaded at its start), from
//     the four coarse rows (row0/2 - 1 .. row0/2 + 2) x two coarse planes the step loaded beside them (eight 16-byte
//     loads per lane, L1/L2 hits: the coarse field is 1/8 of the fine one);
//   * the prologue's three planes with loads of their own.
template <typename T> struct TwiceProlong {
  const T* coarse;
  Geo gc;
  double om;
};
// NORM0 (round 4): the kernel also leaves the canonical partial sums of || f - A x ||^2 of its INPUT x in norm_aux (the
// layout and order of evomg_tile3d<TILE_RESNORM>: one wave sum per strip, row and plane; every inner node counted once, by the
// workgroup that stores it) — the solver loop's residual norm of the iterate rides on the first (double) sweep of the NEXT
// cycle, launched speculatively (impl_solvers.inl: norm_on_head), instead of a pass of its own over x and f. The residual
// comes from the rows the first sweep holds anyway (residual7p: the arithmetic of the residual kernels), seven more
// multiply-adds per node on a launch that waits for memory.
// XZ (round 4): the sweeps' input is the zero field (the SET_ZERO in front of them, elided: boundary zero as well, host-checked)
// — nothing of x is loaded; the arithmetic is the general form's on literal zeros.
template <typename T, int MODE, int NW, int NT, bool PRO = false, bool POW2 = false, bool NORM0 = false, bool XZ = false>
__global__ void __launch_bounds__(64 * NW)
evomg_tile3d_twice(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f, T* __restrict__ y, double om,
                   double omm, int planes_per_chunk, int n_strips, int n_ygroups, int n_blocks, TwoCellPlan<T> pl,
                   TwiceProlong<T> pro, int xpitch, T* __restrict__ norm_aux) {
  static_assert(MODE == STRIP_JACOBI || MODE == TILE_BLOCK211, "two fused sweeps: point Jacobi and (2,1,1) blocks");
  static_assert(!(PRO && NORM0), "the norm is that of the sweeps' input as it lies in memory");
  static_assert(!(XZ && (PRO || NORM0)), "zero input: plain double sweep only");
  constexpr int TY = EVOMG_TILE3D_TY, H = NW * TY;
  __shared__ __attribute__((aligned(16))) T edge_x[2][NW][2][128];   // old x of the next centre plane
  __shared__ __attribute__((aligned(16))) T edge_y[2][NW][2][128];   // y1 of the current centre plane
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (logical_block >= n_blocks) return;
  prio_level(g);
  const int lane = threadIdx.x & 63;
  const ConstDiv<T> cd0 = const_div_make(st.c[0]);     // quotients by the stencil centre (evomg_scalar.h)
  const ConstDiv<T> cdp0 = const_div_make(pl.pivot0), cdp1 = const_div_make(pl.pivot1);
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int strip = logical_block % n_strips;
  const int ygroup = (logical_block / n_strips) % n_ygroups;
  const int chunk = logical_block / (n_ygroups * n_strips);
  const int n = g.n;
  const long pitch = g.pitch, plane = g.plane;
  // halo lanes per side: one for point Jacobi; two for the (2,1,1) blocks, whose tiles pair element b of a lane with
  // element a of its right neighbour — the second sweep's tile at the strip's edge reads a first-sweep value that the
  // outermost lane cannot form (its own tile partner lies outside the wave)
  const bool last_strip = strip == n_strips - 1;              // uniform
  constexpr int HALO = (MODE == TILE_BLOCK211) ? 2 : 1;
  const int c0 = strip * xpitch - 2 * HALO + 2 * lane;        // xpitch <= 128 - 4 * HALO: see evomg_tile3d
  const bool col_ok = c0 >= 0 && c0 < n;
  const int R0 = ygroup * (H - 2);
  const int row0 = R0 + w * TY;
  const bool top_group = R0 + H - 1 >= n - 2;
  const int z0 = 1 + chunk * planes_per_chunk;                // stored planes [z0, z1)
  const int z1 = min(z0 + planes_per_chunk, n - 1);
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
  const bool store_lane = lane >= HALO && 2 * (lane - HALO) < xpitch && col_ok && (inner_a || inner_b);
  const int c_clamped = min(max(c0, 0), min(n - 1, strip * xpitch + xpitch + 2 * HALO - 2));
  const T zero = scalar_traits<T>::zero();
  const Pair<T> zpair{zero, zero};
  auto load_x = [&](int row, int pl_) -> Pair<T> {
    if constexpr (XZ) return zpair;
    const int r = min(max(row, 0), n - 1), p = min(max(pl_, 0), n - 1);
    return load_pair<T>(x + plane * p + pitch * r + c_clamped);
  };
  auto load_f = [&](int row, int pl_) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl_, 0), n - 1);
    if constexpr ((NT & 1) != 0) return load_pair_nt<T>(f + plane * p + pitch * r + c_clamped);
    else return load_pair<T>(f + plane * p + pitch * r + c_clamped);
  };
  auto park = [&](T (*edge)[NW][2][128], int parity, const Pair<T>& first, const Pair<T>& last) {
    *reinterpret_cast<Pair<T>*>(&edge[parity][w][0][2 * lane]) = first;
    *reinterpret_cast<Pair<T>*>(&edge[parity][w][1][2 * lane]) = last;
  };
  auto pick = [&](T (*edge)[NW][2][128], int parity, int wave, int which) -> Pair<T> {
    return *reinterpret_cast<const Pair<T>*>(&edge[parity][wave][which][2 * lane]);
  };
  // ---- PRO: trilinear prolongation of the coarse correction onto a loaded row ----
  const int nc = PRO ? pro.gc.n : 1;
  const int Ic = min(c_clamped >> 1, nc - 1);               // coarse column of element a; b lies between Ic and Ic + 1
  auto coarse_line = [&](int J, int K) -> Pair<T> {           // e[K][J][Ic], e[K][J][Ic + 1] (8-byte aligned 16-byte load)
    const int jj = min(max(J, 0), nc - 1), kk = min(max(K, 0), nc - 1);
    return load_pair<T>(pro.coarse + pro.gc.plane * kk + (long)pro.gc.pitch * jj + Ic);
  };
  // x + om_p * P(e) at (row, pl_): l00 = coarse row J = row/2 of coarse plane K = pl_/2, l01 = row J+1, l10 / l11 = plane K+1
  auto corrected = [&](const Pair<T>& raw, int row, int pl_, const Pair<T>& l00, const Pair<T>& l01, const Pair<T>& l10,
                       const Pair<T>& l11) -> Pair<T> {
    const bool in = row >= 1 && row <= n - 2 && pl_ >= 1 && pl_ <= n - 2;          // uniform
    const bool oy = (row & 1) != 0, oz = (pl_ & 1) != 0;                           // uniform
    T s0 = sc_add(zero, l00.a), s1 = sc_add(sc_add(zero, l00.a), l00.b);
    if (oy) { s0 = sc_add(s0, l01.a); s1 = sc_add(sc_add(s1, l01.a), l01.b); }
    if (oz) {
      s0 = sc_add(s0, l10.a); s1 = sc_add(sc_add(s1, l10.a), l10.b);
      if (oy) { s0 = sc_add(s0, l11.a); s1 = sc_add(sc_add(s1, l11.a), l11.b); }
    }
    const double w0 = (oy && oz) ? 0.25 : (oy || oz) ? 0.5 : 1.0;
    const T v0 = sc_rmul(w0, s0), v1 = sc_rmul(0.5 * w0, s1);
    Pair<T> out = raw;
    if (in && inner_a) out.a = sc_add(raw.a, sc_rmul(pro.om, v0));
    if (in && inner_b) out.b = sc_add(raw.b, sc_rmul(pro.om, v1));
    return out;
  };
  // a row loaded and corrected with coarse loads of its own (prologue)
  auto load_x_pro = [&](int row, int pl_) -> Pair<T> {
    const Pair<T> raw = load_x(row, pl_);
    if constexpr (!PRO) return raw;
    else {
      const int r = min(max(row, 0), n - 1), p_ = min(max(pl_, 0), n - 1);
      const int J = r >> 1, K = p_ >> 1;
      return corrected(raw, row, pl_, coarse_line(J, K), coarse_line(J + 1, K), coarse_line(J, K + 1), coarse_line(J + 1, K + 1));
    }
  };
  // one sweep at one row: centre pair c with its south / north / down / up pairs and right-hand side; non-inner nodes keep
  // their value. Whole wave (DPP shifts inside).
  auto sweep = [&](const Pair<T>& c, const Pair<T>& s, const Pair<T>& nn, const Pair<T>& dn, const Pair<T>& up,
                   const Pair<T>& fk, bool node_rows_inner, auto want_residual, Pair<T>& res) -> Pair<T> {
    Pair<T> out = c;
    const T wa = from_left(c.b), eb = from_right(c.a);
    if constexpr (decltype(want_residual)::value) {
      res.a = residual7p<POW2>(st, c.a, wa, c.b, s.a, nn.a, dn.a, up.a, fk.a);
      res.b = residual7p<POW2>(st, c.b, c.a, eb, s.b, nn.b, dn.b, up.b, fk.b);
    }
    if constexpr (MODE == STRIP_JACOBI) {
      const T ra = relax7p<POW2>(st, cd0, c.a, wa, c.b, s.a, nn.a, dn.a, up.a, fk.a, om, omm);
      const T rb = relax7p<POW2>(st, cd0, c.b, c.a, eb, s.b, nn.b, dn.b, up.b, fk.b, om, omm);
      if (node_rows_inner && inner_a) out.a = ra;
      if (node_rows_inner && inner_b) out.b = rb;
    } else {
      T rb0 = fk.b;
      rb0 = msub<POW2>(rb0, st.c[1], c.a);
      rb0 = msub<POW2>(rb0, st.c[3], s.b);
      rb0 = msub<POW2>(rb0, st.c[4], nn.b);
      rb0 = msub<POW2>(rb0, st.c[5], dn.b);
      rb0 = msub<POW2>(rb0, st.c[6], up.b);
      T ra1 = fk.a;
      ra1 = msub<POW2>(ra1, st.c[2], c.b);
      ra1 = msub<POW2>(ra1, st.c[3], s.a);
      ra1 = msub<POW2>(ra1, st.c[4], nn.a);
      ra1 = msub<POW2>(ra1, st.c[5], dn.a);
      ra1 = msub<POW2>(ra1, st.c[6], up.a);
      const T r0_left = from_left(rb0);
      const T xa = const_div(sc_sub(ra1, sc_mul(pl.fac, r0_left)), cdp1);
      const T x1 = from_right(xa);             // second unknown of the tile (b, right a) = the right lane's xa
      const T xb = const_div(sc_sub(rb0, sc_mul(pl.upper, x1)), cdp0);
      if (node_rows_inner && inner_a) out.a = sc_add(sc_rmul(omm, c.a), sc_rmul(om, xa));
      if (node_rows_inner && inner_b) out.b = sc_add(sc_rmul(omm, c.b), sc_rmul(om, xb));
      // last inner column: a one-cell tile (only the last strip has such a lane)
      if (last_strip && node_rows_inner && c0 + 1 == n - 2) out.b = relax7p<POW2>(st, cd0, c.b, c.a, eb, s.b, nn.b, dn.b, up.b, fk.b, om, omm);
    }
    return out;
  };

  const int ks = z0 - 1;
  const int ke = z1 + 1;
  Pair<T> xb[TY], xc[TY], xa[TY], xq[TY];      // old x, planes k-1, k, k+1, k+2 (prefetch); own rows
  Pair<T> fb[TY], fc[TY], fq[TY];              // f, planes k-1, k, k+1 (prefetch)
  Pair<T> p2[TY], p1[TY], pc[TY];              // y1 on planes k-2, k-1, k
  Pair<T> hs, hn, gs = zpair, gn = zpair;      // old-x halo rows of the centre plane / of the next one (outermost waves)
  Pair<T> hn_prev;                             // old-x row above the wave in plane k-1 (top group: the boundary row)
  Pair<T> ys1 = zpair, yn1 = zpair;            // y1 of plane k-1 in the rows below / above the wave
#pragma unroll
  for (int t = 0; t < TY; ++t) {
    xb[t] = load_x_pro(row0 + t, ks - 1);
    xc[t] = load_x_pro(row0 + t, ks);
    xa[t] = load_x_pro(row0 + t, ks + 1);
    fc[t] = load_f(row0 + t, ks);
    fb[t] = zpair;
    p2[t] = zpair;
    p1[t] = zpair;
  }
  hs = load_x_pro(row0 - 1, ks);
  hn = load_x_pro(row0 + TY, ks);
  hn_prev = hn;
  // PRO: the outermost waves' halo rows are fetched one plane further ahead (beside xq, so that the step's coarse rows serve
  // them too); gs / gn enter the loop holding plane ks + 1
  Pair<T> gqs = zpair, gqn = zpair;
  Pair<T> cl[2][4];                            // coarse rows row0/2 - 1 .. row0/2 + 2 of the coarse planes (k+2)/2, (k+2)/2 + 1
  if constexpr (PRO) {
    if (w == 0) gs = load_x_pro(row0 - 1, ks + 1);
    if (w == NW - 1) gn = load_x_pro(row0 + TY, ks + 1);
  }
  park(edge_x, ks & 1, xc[0], xc[TY - 1]);
  __syncthreads();
  if (w > 0) hs = pick(edge_x, ks & 1, w - 1, 1);
  if (w < NW - 1) hn = pick(edge_x, ks & 1, w + 1, 0);

  for (int k = ks; k < ke; ++k) {
#pragma unroll
    for (int t = 0; t < TY; ++t) xq[t] = load_x(row0 + t, k + 2);
#pragma unroll
    for (int t = 0; t < TY; ++t) fq[t] = load_f(row0 + t, k + 1);
    if constexpr (PRO) {
      if (w == 0) gqs = load_x(row0 - 1, k + 2);
      if (w == NW - 1) gqn = load_x(row0 + TY, k + 2);
      const int Jb = (row0 >> 1) - 1, Kq = min(max(k + 2, 0), n - 1) >> 1;
      // the upper coarse plane only for odd fine planes, the row below the workgroup only for its first wave (uniform)
      const bool odd_plane = ((k + 2) & 1) != 0;
#pragma unroll
      for (int dz = 0; dz < 2; ++dz)
        if (dz == 0 || odd_plane) {
          if (w == 0) cl[dz][0] = coarse_line(Jb, Kq + dz);
#pragma unroll
          for (int jl = 1; jl < 4; ++jl) cl[dz][jl] = coarse_line(Jb + jl, Kq + dz);
        }
    } else {
      if (w == 0) gs = load_x(row0 - 1, k + 1);
      if (w == NW - 1) gn = load_x(row0 + TY, k + 1);
    }
    park(edge_x, (k + 1) & 1, xa[0], xa[TY - 1]);
    const bool plane_inner = k >= 1 && k <= n - 2;
    // ---- first sweep: y1 on plane k, own rows ----
#pragma unroll
    for (int t = 0; t < TY; ++t) {
      const int row = row0 + t;
      const Pair<T> s = (t == 0) ? hs : xc[t > 0 ? t - 1 : 0], nn = (t == TY - 1) ? hn : xc[t < TY - 1 ? t + 1 : t];
      Pair<T> res = zpair;
      pc[t] = sweep(xc[t], s, nn, xb[t], xa[t], fc[t], plane_inner && row >= 1 && row <= n - 2, std::integral_constant<bool, NORM0>{}, res);
      if constexpr (NORM0) {
        // (row, plane) counted by this workgroup: the ones it stores (wave-uniform)
        const bool counts = k >= z0 && k < z1 && row >= 1 && row <= n - 2 && row >= R0 + 1 && (row <= R0 + H - 2 || top_group);
        if (counts) {
          T sq = zero;
          if (store_lane && inner_a) sq = sc_add(sq, sc_mul(res.a, res.a));
          if (store_lane && inner_b) sq = sc_add(sq, sc_mul(res.b, res.b));
          // the canonical wave holds column pair c0 = strip * xpitch - 2 + 2 l in lane l: with two halo lanes this wave's
          // lanes sit one to the right of it
          if constexpr (HALO == 2) sq = from_right(sq);
          sq = wave_sum_fast(sq);
          if (lane == 0) norm_aux[((long)(k - 1) * (n - 2) + (row - 1)) * n_strips + strip] = sq;
        }
      }
    }
    park(edge_y, k & 1, pc[0], pc[TY - 1]);
    // ---- second sweep: y2 on plane k-1, own rows, from y1 on the planes k-2, k-1, k ----
    const int kb2 = k - 1;
    const bool bplane_inner = kb2 >= 1 && kb2 <= n - 2;
    const bool bplane_stored = kb2 >= z0 && kb2 < z1;
#pragma unroll
    for (int t = 0; t < TY; ++t) {
      const int row = row0 + t;
      const Pair<T> s = (t == 0) ? ys1 : p1[t > 0 ? t - 1 : 0];
      // the boundary row above the topmost workgroup: y1 = x there
      const Pair<T> nn = (t == TY - 1) ? ((w == NW - 1 && top_group) ? hn_prev : yn1) : p1[t < TY - 1 ? t + 1 : t];
      Pair<T> unused;
      const Pair<T> out = sweep(p1[t], s, nn, p2[t], pc[t], fb[t], bplane_inner && row >= 1 && row <= n - 2, std::false_type{}, unused);
      const bool row_stored = row >= 1 && row <= n - 2 && row >= R0 + 1 && (row <= R0 + H - 2 || top_group);
      if (store_lane && bplane_stored && row_stored) store_pair(y + plane * kb2 + pitch * row + c0, out);
    }
    __syncthreads();
    hn_prev = hn;
    hs = (w > 0) ? pick(edge_x, (k + 1) & 1, w - 1, 1) : gs;
    hn = (w < NW - 1) ? pick(edge_x, (k + 1) & 1, w + 1, 0) : gn;
    ys1 = (w > 0) ? pick(edge_y, k & 1, w - 1, 1) : zpair;
    yn1 = (w < NW - 1) ? pick(edge_y, k & 1, w + 1, 0) : zpair;
    if constexpr (PRO) {
      // plane k+2 arrives: correct its rows before they become `xa`. Row row0 + t (row0 even): coarse row index
      // jl = 1 + floor(t / 2) of the four loaded, the next one as well when t is odd
#pragma unroll
      for (int t = 0; t < TY; ++t) {
        const int jl = 1 + (t >> 1);
        xq[t] = corrected(xq[t], row0 + t, k + 2, cl[0][jl], cl[0][jl + 1 < 4 ? jl + 1 : 3], cl[1][jl], cl[1][jl + 1 < 4 ? jl + 1 : 3]);
      }
      if (w == 0) gs = corrected(gqs, row0 - 1, k + 2, cl[0][0], cl[0][1], cl[1][0], cl[1][1]);          // row0 - 1: odd
      if (w == NW - 1) gn = corrected(gqn, row0 + TY, k + 2, cl[0][3], cl[0][3], cl[1][3], cl[1][3]);     // row0 + 4: even
    }
#pragma unroll
    for (int t = 0; t < TY; ++t) { p2[t] = p1[t]; p1[t] = pc[t]; fb[t] = fc[t]; fc[t] = fq[t]; xb[t] = xc[t]; xc[t] = xa[t]; xa[t] = xq[t]; }
  }
}

// ---- 3D residual + full-weighting restriction in one pass (K4 + K5) ---------------------------------------------------
// coarse(I,J,K) = sum of the 27 fine residuals around (2I,2J,2K), weighted 1/8, 1/16, 1/32, 1/64 — without the residual
// field ever reaching memory (24 B/node -> 16 B/node + the coarse field; the separate kernels move 24 + 8 + 1 B/node).
// Geometry of the red-black tiles: a workgroup computes the residual on H = NW*4 rows x 128 columns and marches along z;
//   * lane = column pair (c0 even, c0 + 1): the coarse column c0/2 sits on element a; its x neighbours are the lane's own
//     b and the left lane's b (one DPP shift per row); lanes 1..xpitch/2 produce coarse nodes (at most 62 per strip);
//   * coarse rows sit on the wave's local rows 0 and 2 (row0 is even): row 2 needs the wave's rows 1..3, row 0 needs the
//     LAST residual row of the wave below — parked in LDS, riding on the step's single barrier; the workgroup's first row
//     has no such neighbour, so it finishes the coarse rows at local rows 2, 4, .., H-2 and consecutive workgroups overlap
//     by two rows, as the red-black tiles do;
//   * planes: fine plane k = 2K is the centre plane of coarse plane K, the odd planes around it feed two coarse planes
//     each. A chunk of C coarse planes computes the residual on 2C + 1 fine planes.
// Summation order = evomg_restrict<T,3> (loop order dz, dy, dx; face / edge / corner sums carried across the three planes
// in that order) on residuals computed by residual7 = evomg_apply: bit-identical to the two separate kernels.
template <typename T, int NW, int NT, bool POW2 = false>
__global__ void __launch_bounds__(64 * NW)
evomg_tile3d_residual_restrict(Geo g, Geo gc, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f,
                               T* __restrict__ coarse, int cplanes_per_chunk, int n_strips, int n_ygroups, int n_blocks,
                               int xpitch) {
  constexpr int TY = EVOMG_TILE3D_TY, H = NW * TY;
  static_assert(TY == 4 && (H - 2) % 2 == 0, "coarse rows on local rows 0 and 2 of every wave");
  __shared__ __attribute__((aligned(16))) T edge_x[2][NW][2][128];   // x: [plane parity][wave][first | last row][column]
  __shared__ __attribute__((aligned(16))) T edge_r[2][NW][128];      // residual: last row of the wave
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (logical_block >= n_blocks) return;
  prio_level(g);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int strip = logical_block % n_strips;                 // strip fastest: concurrent workgroups cover whole rows
  const int ygroup = (logical_block / n_strips) % n_ygroups;
  const int chunk = logical_block / (n_ygroups * n_strips);
  const int n = g.n, nc = gc.n;
  const long pitch = g.pitch, plane = g.plane;
  const int c0 = strip * xpitch - 2 + 2 * lane;                // even (xpitch: see evomg_tile3d)
  const int R0 = ygroup * (H - 2);                            // even
  const int row0 = R0 + w * TY;
  const int K0 = 1 + chunk * cplanes_per_chunk;               // coarse planes [K0, K1)
  const int K1 = min(K0 + cplanes_per_chunk, nc - 1);
  const int ks = 2 * K0 - 1, ke = 2 * K1;                     // fine planes [ks, ke): odd first and last plane
  const int c_clamped = min(max(c0, 0), min(n - 1, strip * xpitch + xpitch));
  const T zero = scalar_traits<T>::zero();
  const Pair<T> zpair{zero, zero};
  auto load_x = [&](int row, int pl_) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl_, 0), n - 1);
    return load_pair<T>(x + plane * p + pitch * r + c_clamped);
  };
  auto load_f = [&](int row, int pl_) -> Pair<T> {
    const int r = min(max(row, 0), n - 1), p = min(max(pl_, 0), n - 1);
    if constexpr ((NT & 1) != 0) return load_pair_nt<T>(f + plane * p + pitch * r + c_clamped);
    else return load_pair<T>(f + plane * p + pitch * r + c_clamped);
  };
  auto park_x = [&](int parity, const Pair<T>& first, const Pair<T>& last) {
    *reinterpret_cast<Pair<T>*>(&edge_x[parity][w][0][2 * lane]) = first;
    *reinterpret_cast<Pair<T>*>(&edge_x[parity][w][1][2 * lane]) = last;
  };
  auto pick_x = [&](int parity, int wave, int which) -> Pair<T> {
    return *reinterpret_cast<const Pair<T>*>(&edge_x[parity][wave][which][2 * lane]);
  };
  // the two coarse nodes of this lane: column c0 / 2, rows (row0 + 0) / 2 and (row0 + 2) / 2
  const int I = c0 >> 1;
  const bool col_valid = lane >= 1 && 2 * lane <= xpitch && I >= 1 && I <= nc - 2;
  bool valid[2];
  long cidx[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int J = (row0 + 2 * q) >> 1;
    valid[q] = col_valid && J >= 1 && J <= nc - 2 && !(w == 0 && q == 0);
    cidx[q] = (long)gc.pitch * min(max(J, 0), nc - 1) + min(max(I, 0), nc - 1);
  }

  Pair<T> xb[TY], xc[TY], xa[TY], xq[TY];      // x, planes k-1, k, k+1, k+2 (prefetch); own rows
  Pair<T> fc[TY], fq[TY];                      // f, planes k, k+1 (prefetch)
  Pair<T> hs, hn, gs = zpair, gn = zpair;      // x halo rows of the centre plane / of the next one (outermost waves)
  T face[2], edg[2], cor[2], cen[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) face[q] = edg[q] = cor[q] = cen[q] = zero;
#pragma unroll
  for (int t = 0; t < TY; ++t) {
    xb[t] = load_x(row0 + t, ks - 1);
    xc[t] = load_x(row0 + t, ks);
    xa[t] = load_x(row0 + t, ks + 1);
    fc[t] = load_f(row0 + t, ks);
  }
  hs = load_x(row0 - 1, ks);
  hn = load_x(row0 + TY, ks);
  park_x(ks & 1, xc[0], xc[TY - 1]);
  __syncthreads();
  if (w > 0) hs = pick_x(ks & 1, w - 1, 1);
  if (w < NW - 1) hn = pick_x(ks & 1, w + 1, 0);

  for (int k = ks; k < ke; ++k) {
#pragma unroll
    for (int t = 0; t < TY; ++t) xq[t] = load_x(row0 + t, k + 2);
#pragma unroll
    for (int t = 0; t < TY; ++t) fq[t] = load_f(row0 + t, k + 1);
    if (w == 0) gs = load_x(row0 - 1, k + 1);
    if (w == NW - 1) gn = load_x(row0 + TY, k + 1);
    park_x((k + 1) & 1, xa[0], xa[TY - 1]);
    // ---- residual of plane k, own rows (whole wave: DPP shifts inside) ----
    Pair<T> rk[TY];
#pragma unroll
    for (int t = 0; t < TY; ++t) {
      const Pair<T> c = xc[t], s = (t == 0) ? hs : xc[t > 0 ? t - 1 : 0], nn = (t == TY - 1) ? hn : xc[t < TY - 1 ? t + 1 : t];
      const T wa = from_left(c.b), eb = from_right(c.a);
      rk[t].a = residual7p<POW2>(st, c.a, wa, c.b, s.a, nn.a, xb[t].a, xa[t].a, fc[t].a);
      rk[t].b = residual7p<POW2>(st, c.b, c.a, eb, s.b, nn.b, xb[t].b, xa[t].b, fc[t].b);
    }
    *reinterpret_cast<Pair<T>*>(&edge_r[k & 1][w][2 * lane]) = rk[TY - 1];
    __syncthreads();
    hs = (w > 0) ? pick_x((k + 1) & 1, w - 1, 1) : gs;
    hn = (w < NW - 1) ? pick_x((k + 1) & 1, w + 1, 0) : gn;
    const Pair<T> rs = (w > 0) ? *reinterpret_cast<const Pair<T>*>(&edge_r[k & 1][w - 1][2 * lane]) : zpair;
    // ---- the contributions of plane k to the coarse nodes ----
    const T lbs = from_left(rs.b);
    T lb[TY];
#pragma unroll
    for (int t = 0; t < TY; ++t) lb[t] = from_left(rk[t].b);
    const bool odd = (k & 1) != 0;                           // uniform
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = 2 * q;
      const Pair<T> south = (q == 0) ? rs : rk[1], mid = rk[j], north = rk[j + 1];
      const T lsouth = (q == 0) ? lbs : lb[1], lmid = lb[j], lnorth = lb[j + 1];
      const T C = mid.a, W = lmid, E = mid.b, S = south.a, N = north.a;
      const T SW = lsouth, SE = south.b, NW_ = lnorth, NE = north.b;
      if (odd) {
        // upper plane of coarse plane (k-1)/2 ...
        const T fa = sc_add(face[q], C);
        const T ed = sc_add(sc_add(sc_add(sc_add(edg[q], S), W), E), N);
        const T co = sc_add(sc_add(sc_add(sc_add(cor[q], SW), SE), NW_), NE);
        T v = sc_rmul(0.125, cen[q]);
        v = sc_add(v, sc_rmul(0.0625, fa));
        v = sc_add(v, sc_rmul(0.03125, ed));
        v = sc_add(v, sc_rmul(0.015625, co));
        if (k > ks && valid[q]) coarse[(long)gc.plane * ((k - 1) >> 1) + cidx[q]] = v;
        // ... and lower plane of coarse plane (k+1)/2
        face[q] = sc_add(zero, C);
        edg[q] = sc_add(sc_add(sc_add(sc_add(zero, S), W), E), N);
        cor[q] = sc_add(sc_add(sc_add(sc_add(zero, SW), SE), NW_), NE);
      } else {
        face[q] = sc_add(sc_add(sc_add(sc_add(face[q], S), W), E), N);
        edg[q] = sc_add(sc_add(sc_add(sc_add(edg[q], SW), SE), NW_), NE);
        cen[q] = C;
      }
    }
#pragma unroll
    for (int t = 0; t < TY; ++t) { fc[t] = fq[t]; xb[t] = xc[t]; xc[t] = xa[t]; xa[t] = xq[t]; }
  }
}
