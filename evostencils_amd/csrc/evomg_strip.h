// evomg_strip.h — the 2D "marching strip" kernels: Jacobi (K1), fused two-colour red-black Gauss-Seidel (K2) and
// residual (K4) for 5-point stencils at the HBM roofline.
//
// Work decomposition (no LDS, no barriers, no atomics):
//   * one wave64 owns a vertical strip of 128 consecutive columns: lane l holds the column PAIR (2l, 2l+1) of the
//     strip, so every global access is one aligned 16-byte load/store per lane (1 KiB per wave instruction);
//   * the wave marches down its chunk of rows with a three-row register window (rows j-1, j, j+1 of x), so each
//     x row is fetched once per strip; rows are prefetched PF steps ahead through a register ring to keep
//     (PF+1) x 2 KiB per wave in flight;
//   * horizontal neighbours live in the adjacent lanes: one DPP wave shift (v_mov_b32_dpp wave_shr:1 / wave_shl:1)
//     per 32-bit half, no LDS round trip. Lanes 0 and 63 are halo lanes: only the 62 interior pairs (124 columns)
//     are stored, which is what makes the strip self-contained (3% redundant column loads, served by L2);
//   * red-black Gauss-Seidel runs BOTH colours in the same pass, out of place: at step j the wave relaxes the red
//     nodes of row j from old values, then the black nodes of row j-1 from the new reds of rows j-2, j-1, j held in
//     registers. Algorithmic traffic 24 B/node (x, f read once; y written once) instead of 48 B/node for two
//     in-place colour passes; the result is bit-identical to the two-pass form (same formula, same operands).
//
// Vertical halo: a chunk of H rows re-reads 2 (Jacobi/residual) or 4 (RB-GS) rows of x: (H+4)/H.
#pragma once
#include "evomg_kernels.h"

#define EVOMG_STRIP_COLS 124   // stored columns per strip (62 interior lane pairs)

enum { STRIP_JACOBI = 0, STRIP_RBGS = 1, STRIP_RESIDUAL = 2 };

template <typename T> struct Pair {
  T a, b;
};

__device__ __forceinline__ int dpp_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }
// value held by lane-1 (lane 0 receives 0)
__device__ __forceinline__ double from_left(double v) {
  return __hiloint2double(dpp_shr1(__double2hiint(v)), dpp_shr1(__double2loint(v)));
}
// value held by lane+1 (lane 63 receives 0)
__device__ __forceinline__ double from_right(double v) {
  return __hiloint2double(dpp_shl1(__double2hiint(v)), dpp_shl1(__double2loint(v)));
}
__device__ __forceinline__ cplx from_left(cplx v) { return cplx{from_left(v.re), from_left(v.im)}; }
__device__ __forceinline__ cplx from_right(cplx v) { return cplx{from_right(v.re), from_right(v.im)}; }

template <typename T> __device__ __forceinline__ Pair<T> load_pair(const T* __restrict__ p);
template <> __device__ __forceinline__ Pair<double> load_pair<double>(const double* __restrict__ p) {
  const double2 v = *reinterpret_cast<const double2*>(p);
  return Pair<double>{v.x, v.y};
}
template <> __device__ __forceinline__ Pair<cplx> load_pair<cplx>(const cplx* __restrict__ p) {
  const double2 v0 = reinterpret_cast<const double2*>(p)[0];
  const double2 v1 = reinterpret_cast<const double2*>(p)[1];
  return Pair<cplx>{cplx{v0.x, v0.y}, cplx{v1.x, v1.y}};
}
typedef double evomg_f64x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ Pair<T> load_pair_nt(const T* __restrict__ p);
template <> __device__ __forceinline__ Pair<double> load_pair_nt<double>(const double* __restrict__ p) {
  const evomg_f64x2 v = __builtin_nontemporal_load(reinterpret_cast<const evomg_f64x2*>(p));
  return Pair<double>{v.x, v.y};
}
template <> __device__ __forceinline__ Pair<cplx> load_pair_nt<cplx>(const cplx* __restrict__ p) {
  const evomg_f64x2 v0 = __builtin_nontemporal_load(reinterpret_cast<const evomg_f64x2*>(p));
  const evomg_f64x2 v1 = __builtin_nontemporal_load(reinterpret_cast<const evomg_f64x2*>(p) + 1);
  return Pair<cplx>{cplx{v0.x, v0.y}, cplx{v1.x, v1.y}};
}
__device__ __forceinline__ void store_pair(double* __restrict__ p, const Pair<double>& v) {
  *reinterpret_cast<double2*>(p) = make_double2(v.a, v.b);
}
__device__ __forceinline__ void store_pair(cplx* __restrict__ p, const Pair<cplx>& v) {
  reinterpret_cast<double2*>(p)[0] = make_double2(v.a.re, v.a.im);
  reinterpret_cast<double2*>(p)[1] = make_double2(v.b.re, v.b.im);
}

// same operation order as relax_point / apply_stencil (evomg_kernels.h) and the oracle's FN(relax2)/FN(apply_st2)
template <typename T>
__device__ __forceinline__ T relax5(const Stencil<T>& st, T c, T w, T e, T s, T n, T f, double om, double omm) {
  T r = f;
  r = sc_sub(r, sc_mul(st.c[1], w));
  r = sc_sub(r, sc_mul(st.c[2], e));
  r = sc_sub(r, sc_mul(st.c[3], s));
  r = sc_sub(r, sc_mul(st.c[4], n));
  const T xs = sc_div(r, st.c[0]);
  return sc_add(sc_rmul(omm, c), sc_rmul(om, xs));
}
template <typename T>
__device__ __forceinline__ T residual5(const Stencil<T>& st, T c, T w, T e, T s, T n, T f) {
  T a = sc_mul(st.c[0], c);
  a = sc_add(a, sc_mul(st.c[1], w));
  a = sc_add(a, sc_mul(st.c[2], e));
  a = sc_add(a, sc_mul(st.c[3], s));
  a = sc_add(a, sc_mul(st.c[4], n));
  return sc_sub(f, a);
}

// grid: ceil(strips * chunks / 4) blocks of 256 threads (4 independent waves); rows_per_chunk = H
template <typename T, int MODE, int PF, bool NT_F>
__global__ void __launch_bounds__(256)
evomg_strip2d(Geo g, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f, T* __restrict__ y,
              double om, double omm, int rows_per_chunk, int n_strips, int n_waves, double robin_kh) {
  // XCD-aware block order: the dispatcher deals blocks round-robin over the 8 XCDs (block b -> XCD b % 8), each
  // with a private L2. Remap so that every XCD works on one contiguous band of rows: vertically adjacent chunks
  // then share their halo rows through that XCD's L2 instead of re-reading them from HBM. gridDim.x is a multiple
  // of 8 (host side), so the map is a bijection; placement only affects speed, never results.
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int wave = __builtin_amdgcn_readfirstlane((int)(logical_block * 4 + (threadIdx.x >> 6)));
  if (wave >= n_waves) return;
  const int lane = threadIdx.x & 63;
  const int strip = wave % n_strips;
  const int chunk = wave / n_strips;
  const int n = g.n;
  const long pitch = g.pitch;
  const int c0 = strip * EVOMG_STRIP_COLS - 2 + 2 * lane;   // first column of this lane's pair (even)
  const bool col_ok = c0 >= 0 && c0 < n;                    // c0+1 <= n < pitch: the pair load stays inside the row
  const int y0 = chunk * rows_per_chunk;
  const int y1 = min(y0 + rows_per_chunk, n);               // stored rows [y0, y1)
  const T zero = scalar_traits<T>::zero();
  const Pair<T> zpair{zero, zero};
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
  const bool store_lane = lane >= 1 && lane <= 62 && col_ok;

  // Branch-free loads: out-of-domain rows / columns are clamped to the nearest valid pair instead of being
  // predicated off. Whatever they deliver only ever feeds non-inner nodes, whose result is discarded in favour of
  // the old value — and straight-line code is what lets the compiler keep the PF-deep prefetch ring in flight
  // with counted s_waitcnt vmcnt(N) instead of draining to vmcnt(0) at every predicated block.
  const int c_clamped = min(max(c0, 0), n - 1);             // n-1 is even; (n-1, n) is still inside the row pitch
  auto load_row = [&](const T* __restrict__ base, int row) -> Pair<T> {
    const int r = min(max(row, 0), n - 1);                  // wave-uniform
    return load_pair<T>(base + pitch * r + c_clamped);
  };
  // right-hand side: read exactly once per sweep -> optional non-temporal (streaming) load so that it does not
  // evict x / y from the 256 MiB Infinity Cache between back-to-back sweeps
  auto load_row_f = [&](const T* __restrict__ base, int row) -> Pair<T> {
    const int r = min(max(row, 0), n - 1);
    if constexpr (NT_F) return load_pair_nt<T>(base + pitch * r + c_clamped);
    else return load_pair<T>(base + pitch * r + c_clamped);
  };

  constexpr int LEAD = (MODE == STRIP_RBGS) ? 1 : 0;        // RB-GS relaxes red one row ahead of black
  const int js = y0 - LEAD;                                 // first row relaxed (red)
  const int je = y1 + LEAD;                                 // one past the last row relaxed
  Pair<T> xm = load_row(x, js - 1);
  Pair<T> xc = load_row(x, js);
  Pair<T> xq[PF], fq[PF];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    xq[d] = load_row(x, js + 1 + d);
    fq[d] = load_row_f(f, js + d);
  }
  Pair<T> fm = zpair;          // f of row j-1 (black phase)
  T rm2 = zero, rm1 = zero;    // new red of rows j-2, j-1 (this lane's pair)

  for (int base = js; base < je; base += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      const int j = base + d;
      if (j < je) {                                          // wave-uniform
        const Pair<T> xp = xq[d];
        const Pair<T> fc = fq[d];
        xq[d] = load_row(x, j + 1 + PF);
        fq[d] = load_row_f(f, j + PF);
        const bool row_inner = j >= 1 && j <= n - 2;
        // chunk heights are even and js = y0 - LEAD, so with an even prefetch depth the row parity is static
        const int parity = (PF % 2 == 0) ? ((LEAD + d) & 1) : (j & 1);
        if constexpr (MODE == STRIP_JACOBI || MODE == STRIP_RESIDUAL) {
          const T wa = from_left(xc.b), eb = from_right(xc.a);
          Pair<T> out;
          if constexpr (MODE == STRIP_JACOBI) {
            out.a = (row_inner && inner_a) ? relax5(st, xc.a, wa, xc.b, xm.a, xp.a, fc.a, om, omm) : xc.a;
            out.b = (row_inner && inner_b) ? relax5(st, xc.b, xc.a, eb, xm.b, xp.b, fc.b, om, omm) : xc.b;
            if (store_lane && j >= 0 && j < n) store_pair(y + pitch * j + c0, out);
          } else {
            // residual: only inner nodes are written (the boundary of the residual field is never touched)
            if (row_inner && store_lane) {
              out.a = residual5(st, xc.a, wa, xc.b, xm.a, xp.a, fc.a);
              out.b = residual5(st, xc.b, xc.a, eb, xm.b, xp.b, fc.b);
              T* dst = y + pitch * j + c0;
              if (inner_a && inner_b) store_pair(dst, out);
              else if (inner_a) dst[0] = out.a;
              else if (inner_b) dst[1] = out.b;
            }
          }
        } else {
          // ---- red nodes of row j (colour 0: even index sum; c0 is even => element a when j is even) ----
          T rc;
          if (parity == 0) {
            const T w = from_left(xc.b);
            rc = (row_inner && inner_a) ? relax5(st, xc.a, w, xc.b, xm.a, xp.a, fc.a, om, omm) : xc.a;
          } else {
            const T e = from_right(xc.a);
            rc = (row_inner && inner_b) ? relax5(st, xc.b, xc.a, e, xm.b, xp.b, fc.b, om, omm) : xc.b;
          }
          // ---- black nodes of row j-1 from the new reds of rows j-2 (rm2), j-1 (rm1), j (rc) ----
          const int jb = j - 1;
          const bool brow_inner = jb >= 1 && jb <= n - 2;
          Pair<T> out;
          if (parity == 1) {            // jb = j-1 even: red = a, black = b
            T e = from_right(rm1);
            T w = rm1;
            if constexpr (scalar_traits<T>::is_complex) {
              // Helmholtz Robin boundary (K8): the two-pass form re-applies x_b = x_in / (1 - i k h) between the
              // colours, so a black node next to the left/right boundary sees the boundary value derived from
              // its own (still old) value. Identical to the stored value whenever the field is BC-consistent.
              if (robin_kh != 0.0) {
                const cplx den{1.0, -robin_kh};
                if (c0 == 0) w = sc_div(xm.b, den);
                if (c0 + 1 == n - 2) e = sc_div(xm.b, den);
              }
            }
            const T bl = (brow_inner && inner_b) ? relax5(st, xm.b, w, e, rm2, rc, fm.b, om, omm) : xm.b;
            out.a = rm1;
            out.b = bl;
          } else {                      // red = b, black = a
            const T w = from_left(rm1);
            const T bl = (brow_inner && inner_a) ? relax5(st, xm.a, w, rm1, rm2, rc, fm.a, om, omm) : xm.a;
            out.a = bl;
            out.b = rm1;
          }
          if (store_lane && jb >= y0 && jb < y1) store_pair(y + pitch * jb + c0, out);
          rm2 = rm1;
          rm1 = rc;
          fm = fc;
        }
        xm = xc;
        xc = xp;
      }
    }
  }
}

// ---- fused residual + full-weighting restriction (K4 + K5) ------------------------------------------------------
// The residual of fine row j is computed exactly as in STRIP_RESIDUAL and kept in a three-row register window; every
// second row the wave emits one coarse row: lane l's pair starts at the even fine column c0, i.e. at coarse column
// c0/2, whose 3x3 fine neighbourhood is the lane's own pair plus the left lane's second element (one DPP shift per
// row). Consecutive lanes write consecutive coarse nodes. The residual field itself is written only if WRITE_R
// (it is skipped when the op program never reads it again before overwriting it): 16 + 8/4 instead of 24 + 10 B/node.
// Same operation order as evomg_apply<.., true> followed by evomg_restrict => bit-identical coarse right-hand side.
template <typename T, bool WRITE_R, bool NT_F>
__global__ void __launch_bounds__(256)
evomg_strip_residual_restrict(Geo g, Geo gc, Stencil<T> st, const T* __restrict__ x, const T* __restrict__ f,
                              T* __restrict__ res, T* __restrict__ coarse, int rows_per_chunk, int n_strips, int n_waves) {
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int wave = __builtin_amdgcn_readfirstlane((int)(logical_block * 4 + (threadIdx.x >> 6)));
  if (wave >= n_waves) return;
  const int lane = threadIdx.x & 63;
  const int strip = wave % n_strips;
  const int chunk = wave / n_strips;
  const int n = g.n, nc = gc.n;
  const long pitch = g.pitch;
  const int c0 = strip * EVOMG_STRIP_COLS - 2 + 2 * lane;
  const bool col_ok = c0 >= 0 && c0 < n;
  const int y0 = chunk * rows_per_chunk;                     // even
  const int y1 = min(y0 + rows_per_chunk, n);
  const T zero = scalar_traits<T>::zero();
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
  const bool store_lane = lane >= 1 && lane <= 62 && col_ok;
  const int I = c0 >> 1;                                     // coarse column of element a
  const bool coarse_col_inner = I >= 1 && I <= nc - 2;
  const int c_clamped = min(max(c0, 0), n - 1);
  auto load_row = [&](const T* __restrict__ base, int row) -> Pair<T> {
    const int r = min(max(row, 0), n - 1);
    return load_pair<T>(base + pitch * r + c_clamped);
  };
  auto load_row_f = [&](const T* __restrict__ base, int row) -> Pair<T> {
    const int r = min(max(row, 0), n - 1);
    if constexpr (NT_F) return load_pair_nt<T>(base + pitch * r + c_clamped);
    else return load_pair<T>(base + pitch * r + c_clamped);
  };
  constexpr int PF = 2;
  const int js = y0 - 1;            // first residual row needed (south neighbour row of the chunk's first coarse row)
  const int je = y1;                // one past the last
  Pair<T> xm = load_row(x, js - 1);
  Pair<T> xc = load_row(x, js);
  Pair<T> xq[PF], fq[PF];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    xq[d] = load_row(x, js + 1 + d);
    fq[d] = load_row_f(f, js + d);
  }
  Pair<T> r2{zero, zero}, r1{zero, zero};    // residual rows j-2, j-1
  for (int base = js; base < je; base += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      const int j = base + d;
      if (j < je) {
        const Pair<T> xp = xq[d];
        const Pair<T> fc = fq[d];
        xq[d] = load_row(x, j + 1 + PF);
        fq[d] = load_row_f(f, j + PF);
        const bool row_inner = j >= 1 && j <= n - 2;
        const T wa = from_left(xc.b), eb = from_right(xc.a);
        Pair<T> rc;
        rc.a = (row_inner && inner_a) ? residual5(st, xc.a, wa, xc.b, xm.a, xp.a, fc.a) : zero;
        rc.b = (row_inner && inner_b) ? residual5(st, xc.b, xc.a, eb, xm.b, xp.b, fc.b) : zero;
        if constexpr (WRITE_R) {
          if (row_inner && store_lane && j >= y0) {
            T* dst = res + pitch * j + c0;
            if (inner_a && inner_b) store_pair(dst, rc);
            else if (inner_a) dst[0] = rc.a;
            else if (inner_b) dst[1] = rc.b;
          }
        }
        // js = y0 - 1 is odd, PF is even: d == 0 <=> j odd <=> rows j-2, j-1, j are a complete coarse neighbourhood
        if (d == 0) {
          const int J = (j - 1) >> 1;                        // coarse row; fine centre row j-1 = 2J
          const T w = from_left(r1.b), sw = from_left(r2.b), nw = from_left(rc.b);
          const T e4 = sc_add(sc_add(w, r1.b), sc_add(r2.a, rc.a));
          const T q4 = sc_add(sc_add(sw, r2.b), sc_add(nw, rc.b));
          T v = sc_rmul(0.25, r1.a);
          v = sc_add(v, sc_rmul(0.125, e4));
          v = sc_add(v, sc_rmul(0.0625, q4));
          if (store_lane && coarse_col_inner && J >= 1 && J <= nc - 2 && 2 * J >= y0) coarse[(long)gc.pitch * J + I] = v;
        }
        r2 = r1;
        r1 = rc;
        xm = xc;
        xc = xp;
      }
    }
  }
}

// ---- bilinear prolongation (+ scaled correction), fine-row marching (K6 / K6') -------------------------------------
// Lane pair (c0, c0+1) with c0 even: element a sits on coarse column I = c0/2, element b between I and I+1 (the right
// lane's coarse value, one DPP shift). Each coarse row is loaded once per two fine rows. In place: a lane reads and
// writes only its own pair. Operation order of interpolate<T,2> + evomg_prolong => bit-identical.
template <typename T, bool ACC>
__global__ void __launch_bounds__(256)
evomg_strip_prolong(Geo g, Geo gc, const T* __restrict__ coarse, T* __restrict__ fine, double om, int rows_per_chunk,
                    int n_strips, int n_waves) {
  const int per_xcd = gridDim.x >> 3;
  const int logical_block = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int wave = __builtin_amdgcn_readfirstlane((int)(logical_block * 4 + (threadIdx.x >> 6)));
  if (wave >= n_waves) return;
  const int lane = threadIdx.x & 63;
  const int strip = wave % n_strips;
  const int chunk = wave / n_strips;
  const int n = g.n, nc = gc.n;
  const long pitch = g.pitch;
  const int c0 = strip * EVOMG_STRIP_COLS - 2 + 2 * lane;
  const bool col_ok = c0 >= 0 && c0 < n;
  const int y0 = chunk * rows_per_chunk;                     // even
  const int y1 = min(y0 + rows_per_chunk, n);
  const bool inner_a = c0 >= 1 && c0 <= n - 2;
  const bool inner_b = c0 + 1 >= 1 && c0 + 1 <= n - 2;
  const bool store_lane = lane >= 1 && lane <= 62 && col_ok;
  const int c_clamped = min(max(c0, 0), n - 1);
  const int I = min(max(c0 >> 1, 0), nc - 1);
  auto load_fine = [&](int row) -> Pair<T> {
    const int r = min(max(row, 0), n - 1);
    return load_pair<T>(fine + pitch * r + c_clamped);
  };
  auto load_coarse = [&](int J) -> T {
    const int r = min(max(J, 0), nc - 1);
    return coarse[(long)gc.pitch * r + I];
  };
  const int J0 = y0 >> 1;
  T cJ = load_coarse(J0), cJ1 = load_coarse(J0 + 1);
  Pair<T> x0 = load_fine(y0), x1 = load_fine(y0 + 1);
  for (int j = y0; j < y1; j += 2) {
    const int J = j >> 1;
    // prefetch the next coarse row and the next two fine rows
    const T cJ2 = load_coarse(J + 2);
    const Pair<T> x2 = load_fine(j + 2), x3 = load_fine(j + 3);
    const T cJr = from_right(cJ), cJ1r = from_right(cJ1);
    // even fine row j = 2J
    {
      const bool row_inner = j >= 1 && j <= n - 2;
      const T va = cJ;
      const T vb = sc_rmul(0.5, sc_add(cJ, cJr));
      Pair<T> out;
      out.a = (row_inner && inner_a) ? (ACC ? sc_add(x0.a, sc_rmul(om, va)) : va) : x0.a;
      out.b = (row_inner && inner_b) ? (ACC ? sc_add(x0.b, sc_rmul(om, vb)) : vb) : x0.b;
      if (store_lane) store_pair(fine + pitch * j + c0, out);
    }
    // odd fine row j + 1 = 2J + 1
    if (j + 1 < y1) {
      const bool row_inner = j + 1 >= 1 && j + 1 <= n - 2;
      const T va = sc_rmul(0.5, sc_add(cJ, cJ1));
      const T vb = sc_rmul(0.25, sc_add(sc_add(cJ, cJr), sc_add(cJ1, cJ1r)));
      Pair<T> out;
      out.a = (row_inner && inner_a) ? (ACC ? sc_add(x1.a, sc_rmul(om, va)) : va) : x1.a;
      out.b = (row_inner && inner_b) ? (ACC ? sc_add(x1.b, sc_rmul(om, vb)) : vb) : x1.b;
      if (store_lane) store_pair(fine + pitch * (j + 1) + c0, out);
    }
    cJ = cJ1;
    cJ1 = cJ2;
    x0 = x2;
    x1 = x3;
  }
}
