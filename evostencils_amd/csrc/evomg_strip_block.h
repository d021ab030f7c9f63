// evomg_strip_block.h — 2D collective block-Jacobi with (2,1), (1,2) and (2,2) tiles on the strip layout (K3; round 2).
//
// evomg_block_jacobi_fast gives one thread a tile: its accesses are 16-byte pieces with re-read halo columns, and at
// 4097^2 it reaches 54-63 % of the HBM peak where the point sweeps reach 75 %. Here the point sweeps' layout does the
// block shapes: one wave64 per 128-column strip, lane = column pair (c0 even), marching down a chunk of rows in registers.
// Tiles are anchored at the first inner node, so in x a tile is (element b of lane l, element a of lane l + 1) and in y
// the row pair (j0, j0 + 1), j0 odd:
//   * every lane forms the right-hand sides of its own cells (what lies outside the tile goes to the right-hand side, in
//     the order W, E, S, N of evomg_block_jacobi_fast);
//   * BX = 2: the lane owning the tile's SECOND column (lane l + 1) receives the first column's right-hand sides by one
//     DPP shift each, replays the host-side elimination (make_block_plan: the oracle's order) on all cells of the tile,
//     keeps its own unknowns and hands the first column's back by one DPP shift each;
//   * BX = 1: two independent tiles per lane (column a, column b), no exchange.
// Tiles cut by the far boundary (2^L - 1 inner nodes per dimension: the last column and the last row have no partner): of a
// two-cell shape a single node is left, of a (2,2) tile a (1,2) tile (last column), a (2,1) tile (last row) or a single node
// (corner): extra waves of the same launch solve those, one thread each, as the generic kernel does for cut tiles.
// Quotients by the pivots: const_div (evomg_scalar.h). Real and complex fields (the Robin columns of a Helmholtz field are
// re-applied by the caller after the sweep, as for every block smoother). Bit-identical to evomg_block_jacobi_fast and the oracle.
// grid: ceil(strips * chunks / 4) blocks of 256 threads (4 independent waves), rows_per_chunk even.
#pragma once
#include "evomg_strip.h"

// elimination of a two-cell tile (BlockPlan of a (2,1) or (1,2) shape): fac[1][0], upper[0][1], pivot[0], pivot[1]
template <typename T> struct TwoCell2D {
  T fac, upper, pivot0, pivot1;
};

// cut_col / cut_row: the eliminations of the (1,2) / (2,1) systems a (2,2) tile leaves at the far boundary
template <typename T, int BX, int BY, bool NT_XY>
__global__ void __launch_bounds__(256)
evomg_strip2d_block(Geo g, Stencil<T> st, BlockPlan<T> pl, const T* __restrict__ x, const T* __restrict__ f,
                    T* __restrict__ y, double om, double omm, int rows_per_chunk, int n_strips, int n_waves,
                    TwoCell2D<T> cut_col, TwoCell2D<T> cut_row) {
  static_assert((BX == 1 || BX == 2) && (BY == 1 || BY == 2) && BX * BY > 1, "two- and four-cell tiles");
  constexpr int M = BX * BY;
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int wave = __builtin_amdgcn_readfirstlane((int)(logical_block * 4 + (threadIdx.x >> 6)));
  const int lane = threadIdx.x & 63;
  prio_level(g);
  if (wave >= n_waves) {
    // two-cell shapes: what a cut tile leaves is a single node, i.e. a point relaxation (as evomg_block_jacobi_fast does).
    // The waves behind the strip waves walk that line — the last inner column (BX = 2) or row (BY = 2) — 64 nodes each,
    // so that the op stays one launch and the strip waves carry no code for it.
    if constexpr (M == 2) {
      const int t = 1 + (wave - n_waves) * 64 + lane;
      if (t <= g.n - 2) {
        const long c = (BX == 2) ? gidx<2>(g, g.n - 2, t, 0) : gidx<2>(g, t, g.n - 2, 0);
        y[c] = relax_point<T, 2>(st, x, f, g, c, om, omm);
      }
    } else {
      // (2,2): the last column holds (1,2) tiles, the last row (2,1) tiles, the corner a single node — one thread each,
      // right-hand side and elimination as the generic kernel forms them for a cut tile (W, E, S, N; make_block_plan)
      const int m2 = g.n - 2, half = m2 / 2;                    // m2 is odd: `half` full pairs + the partner-less line
      const int idx = (wave - n_waves) * 64 + lane;
      const long p = g.pitch;
      auto finish = [&](long c0_, long c1_, T r0, T r1, const TwoCell2D<T>& tc) {
        r1 = sc_sub(r1, sc_mul(tc.fac, r0));
        r1 = sc_div(r1, tc.pivot1);
        r0 = sc_div(sc_sub(r0, sc_mul(tc.upper, r1)), tc.pivot0);
        y[c0_] = sc_add(sc_rmul(omm, x[c0_]), sc_rmul(om, r0));
        y[c1_] = sc_add(sc_rmul(omm, x[c1_]), sc_rmul(om, r1));
      };
      if (idx < half) {                                         // column n-2, rows (1 + 2 idx, 2 + 2 idx)
        const long ca = gidx<2>(g, m2, 1 + 2 * idx, 0), cb = ca + p;
        T r0 = f[ca];
        r0 = sc_sub(r0, sc_mul(st.c[1], x[ca - 1]));
        r0 = sc_sub(r0, sc_mul(st.c[2], x[ca + 1]));
        r0 = sc_sub(r0, sc_mul(st.c[3], x[ca - p]));
        T r1 = f[cb];
        r1 = sc_sub(r1, sc_mul(st.c[1], x[cb - 1]));
        r1 = sc_sub(r1, sc_mul(st.c[2], x[cb + 1]));
        r1 = sc_sub(r1, sc_mul(st.c[4], x[cb + p]));
        finish(ca, cb, r0, r1, cut_col);
      } else if (idx < 2 * half) {                              // row n-2, columns (1 + 2 k, 2 + 2 k)
        const int k = idx - half;
        const long ca = gidx<2>(g, 1 + 2 * k, m2, 0), cb = ca + 1;
        T r0 = f[ca];
        r0 = sc_sub(r0, sc_mul(st.c[1], x[ca - 1]));
        r0 = sc_sub(r0, sc_mul(st.c[3], x[ca - p]));
        r0 = sc_sub(r0, sc_mul(st.c[4], x[ca + p]));
        T r1 = f[cb];
        r1 = sc_sub(r1, sc_mul(st.c[2], x[cb + 1]));
        r1 = sc_sub(r1, sc_mul(st.c[3], x[cb - p]));
        r1 = sc_sub(r1, sc_mul(st.c[4], x[cb + p]));
        finish(ca, cb, r0, r1, cut_row);
      } else if (idx == 2 * half) {                             // the corner node
        const long c = gidx<2>(g, m2, m2, 0);
        y[c] = relax_point<T, 2>(st, x, f, g, c, om, omm);
      }
    }
    return;
  }
  const int strip = wave % n_strips;
  const int chunk = wave / n_strips;
  const int n = g.n;
  const long pitch = g.pitch;
  const int c0 = strip * EVOMG_STRIP_COLS - 2 + 2 * lane;       // even; element a = column c0, b = c0 + 1
  const int c_clamped = min(max(c0, 0), n - 1);
  const bool store_lane = lane >= 1 && lane <= 62 && c0 >= 0 && c0 < n;
  // full tiles only. x: tile (b of this lane, a of the right lane) = columns (c0 + 1, c0 + 2); this lane's a belongs to
  // the tile (c0 - 1, c0). BX = 1: every inner column is its own tile column.
  const bool a_full = (BX == 2) ? (c0 - 1 >= 1 && c0 <= n - 2) : (c0 >= 1 && c0 <= n - 2);
  const bool b_full = (BX == 2) ? (c0 + 1 >= 1 && c0 + 2 <= n - 2) : (c0 + 1 >= 1 && c0 + 1 <= n - 2);
  ConstDiv<T> cdp[M];
#pragma unroll
  for (int p = 0; p < M; ++p) cdp[p] = const_div_make(pl.pivot[p]);
  // NT_XY: as in evomg_strip2d — x and y streamed with the non-temporal hint, the right-hand side kept cacheable
  auto load_x = [&](int row) -> Pair<T> {
    const int r = min(max(row, 0), n - 1);
    if constexpr (NT_XY) return load_pair_nt<T>(x + pitch * r + c_clamped);
    else return load_pair<T>(x + pitch * r + c_clamped);
  };
  auto load_f = [&](int row) -> Pair<T> {
    const int r = min(max(row, 0), n - 1);
    return load_pair<T>(f + pitch * r + c_clamped);
  };
  // the replay of make_block_plan's elimination on one tile's right-hand sides (cells: i0 outermost), in place
  auto solve = [&](T (&r)[M]) {
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
      r[p] = const_div(s, cdp[p]);
    }
  };
  // tile rows of this chunk: first rows j0 = 1 + chunk * rows + BY * t (rows_per_chunk is a multiple of BY)
  const int y0 = 1 + chunk * rows_per_chunk;
  const int y1 = min(y0 + rows_per_chunk, n - 1);              // one past the last row this chunk may write
  Pair<T> below = load_x(y0 - 1);
  Pair<T> cur[BY], nxt[BY], fc[BY], fn[BY];
#pragma unroll
  for (int t = 0; t < BY; ++t) { cur[t] = load_x(y0 + t); fc[t] = load_f(y0 + t); }
  for (int j0 = y0; j0 < y1; j0 += BY) {
#pragma unroll
    for (int t = 0; t < BY; ++t) { nxt[t] = load_x(j0 + BY + t); fn[t] = load_f(j0 + BY + t); }
    const bool rows_full = j0 + BY - 1 <= n - 2;               // wave-uniform
    // right-hand sides of this lane's cells, rows cj = 0 .. BY-1: ra[cj] (element a), rb[cj] (element b)
    T ra[BY], rb[BY];
#pragma unroll
    for (int cj = 0; cj < BY; ++cj) {
      const Pair<T> c = cur[cj];
      const Pair<T> s = (cj == 0) ? below : cur[cj > 0 ? cj - 1 : 0];
      const Pair<T> nn = (cj == BY - 1) ? nxt[0] : cur[cj + 1 < BY ? cj + 1 : cj];
      const bool s_out = cj == 0, n_out = cj == BY - 1;        // south / north neighbour outside the tile
      T a = fc[cj].a, b = fc[cj].b;
      if constexpr (BX == 2) {
        // a is the tile's second column: west inside, east (own b) outside; b the first column: west (own a) outside
        a = sc_sub(a, sc_mul(st.c[2], c.b));
        b = sc_sub(b, sc_mul(st.c[1], c.a));
      } else {
        const T wa = from_left(c.b), eb = from_right(c.a);     // whole wave
        a = sc_sub(a, sc_mul(st.c[1], wa));
        a = sc_sub(a, sc_mul(st.c[2], c.b));
        b = sc_sub(b, sc_mul(st.c[1], c.a));
        b = sc_sub(b, sc_mul(st.c[2], eb));
      }
      if (s_out) { a = sc_sub(a, sc_mul(st.c[3], s.a)); b = sc_sub(b, sc_mul(st.c[3], s.b)); }
      if (n_out) { a = sc_sub(a, sc_mul(st.c[4], nn.a)); b = sc_sub(b, sc_mul(st.c[4], nn.b)); }
      ra[cj] = a;
      rb[cj] = b;
    }
    T xa[BY], xb[BY];                                          // the local solutions of this lane's cells
    if constexpr (BX == 2) {
      // this lane solves the tile whose second column is its a: cells 0 .. BY-1 = the left lane's b, BY .. 2BY-1 = own a
      T r[M];
#pragma unroll
      for (int cj = 0; cj < BY; ++cj) { r[cj] = from_left(rb[cj]); r[BY + cj] = ra[cj]; }     // whole wave
      solve(r);
#pragma unroll
      for (int cj = 0; cj < BY; ++cj) { xa[cj] = r[BY + cj]; xb[cj] = from_right(r[cj]); }    // own b: solved by the right lane
    } else {
      T r[M];
#pragma unroll
      for (int cj = 0; cj < BY; ++cj) r[cj] = ra[cj];
      solve(r);
#pragma unroll
      for (int cj = 0; cj < BY; ++cj) xa[cj] = r[cj];
#pragma unroll
      for (int cj = 0; cj < BY; ++cj) r[cj] = rb[cj];
      solve(r);
#pragma unroll
      for (int cj = 0; cj < BY; ++cj) xb[cj] = r[cj];
    }
    if (store_lane) {
#pragma unroll
      for (int cj = 0; cj < BY; ++cj) {
        const Pair<T> c = cur[cj];
        T* dst = y + pitch * (j0 + cj) + c0;
        const T va = sc_add(sc_rmul(omm, c.a), sc_rmul(om, xa[cj]));
        const T vb = sc_add(sc_rmul(omm, c.b), sc_rmul(om, xb[cj]));
        const bool wa_ = rows_full && a_full, wb_ = rows_full && b_full;
        if (wa_ && wb_) store_pair<NT_XY>(dst, Pair<T>{va, vb});
        else if (wa_) dst[0] = va;
        else if (wb_) dst[1] = vb;
      }
    }
    below = cur[BY - 1];
#pragma unroll
    for (int t = 0; t < BY; ++t) { cur[t] = nxt[t]; fc[t] = fn[t]; }
  }
}
