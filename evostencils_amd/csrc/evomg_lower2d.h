// evomg_lower2d.h — the LOWER PART of a cycle in one launch. OFF by default (EVOMG_LOWER2D=1):
// measured not faster than the pipelined launches it replaces — profiles/r04_lower_cycle.md.
//
// At the reference's own level ranges (Helmholtz: levels 3..7, Poisson: 5..9; example_problems/*/*.knowledge:3-4) the levels
// below 65^2 hold a few thousand nodes: every op there is a launch of 3-6 us that does a fraction of a microsecond of work,
// and a V-cycle spends most of its launches on them. Here ONE 1024-thread workgroup keeps every field of the levels up to
// 33^2 (complex) / 65^2 (real) in LDS and interprets the run of consecutive ops that live entirely on those levels — point
// smoothers (Jacobi, two-colour Gauss-Seidel), residual, full weighting, bilinear prolongation (+ correction), zero, copy,
// and the one-wave Krylov coarse solver, executed by the workgroup's first wave — as micro-ops separated by barriers.
//
// Every micro-op applies the device functions of the generic kernels (evomg_kernels.h: relax_point, apply_stencil,
// interpolate, the full-weighting expression of evomg_restrict) to the LDS images, one after the other with the boundary
// handling the interpreter puts behind each op (Helmholtz: the Robin columns after every op that writes a field, and between
// the colours of a red-black sweep) — the same formulas on the same values in the same order, so every field of every level
// carries the bits the per-op launches produce (tests: with and without EVOMG_LOWER2D=1, and the oracle). `with jacobi` sweeps whose slot
// is never advanced (EVOMG_JACOBI_EXASTENCILS_V11) are computed into the level's spare image and dropped, as in the launches.
// All fields of the run's levels are loaded at the start and the ones the run writes are stored at the end.
#pragma once
#include "evomg_krylov.h"

#define EVOMG_LOWER_MAX_OPS 64
#define EVOMG_LOWER_MAX_LEVELS 6
enum { LOWER_SMOOTH = 1, LOWER_RESIDUAL, LOWER_RESTRICT, LOWER_PROLONG_ADD, LOWER_PROLONG_SET, LOWER_ZERO, LOWER_COPY, LOWER_COARSE };
enum { LOWER_SLOT_SOL = 0, LOWER_SLOT_RHS, LOWER_SLOT_RES, LOWER_SLOT_CGS, LOWER_SLOT_ALT, LOWER_SLOTS };

struct LowerOp {
  signed char kind, level, dst, src0, src1, part;   // level: index into the run's level table (0 = coarsest of the run); slots: LOWER_SLOT_*
  signed char lost, pad;                            // SMOOTH: the result is dropped (`with jacobi` slot never advanced)
  double om;
};

template <typename T> struct LowerProgram {
  int n_ops, n_levels;
  int robin;                       // Helmholtz Robin columns
  int cgs_kind, cgs_maxit, cg_mask_full;   // coarse solver: EVOMG_CGS_*, iteration cap, evomg_cg_wave2d's MASK_FROM choice
  double cgs_tol;
  int* info;
  Geo g[EVOMG_LOWER_MAX_LEVELS];                    // global layout (n, pitch)
  double kh[EVOMG_LOWER_MAX_LEVELS];
  Stencil<T> st[EVOMG_LOWER_MAX_LEVELS];
  T* field[EVOMG_LOWER_MAX_LEVELS][LOWER_SLOT_ALT]; // global arrays (nullptr: the level has no such field)
  int offset[EVOMG_LOWER_MAX_LEVELS][LOWER_SLOTS];  // element offset of the image in LDS (-1: none)
  unsigned char store[EVOMG_LOWER_MAX_LEVELS][LOWER_SLOT_ALT];   // written by the run: stored at the end
  LowerOp op[EVOMG_LOWER_MAX_OPS];
};

template <typename T>
__global__ void __launch_bounds__(1024) evomg_lower2d(LowerProgram<T> P) {
  extern __shared__ __align__(16) unsigned char lower_lds[];
  T* const pool = reinterpret_cast<T*>(lower_lds);
  __shared__ int img_off[EVOMG_LOWER_MAX_LEVELS][LOWER_SLOTS];      // current image of (level, slot): Jacobi swaps SOL-like slots with ALT
  const int tid = (int)threadIdx.x;
  constexpr bool CPLX = scalar_traits<T>::is_complex;
  const T zero = scalar_traits<T>::zero();
  if (tid < EVOMG_LOWER_MAX_LEVELS * LOWER_SLOTS) img_off[tid / LOWER_SLOTS][tid % LOWER_SLOTS] = P.offset[tid / LOWER_SLOTS][tid % LOWER_SLOTS];
  // ---- load every field of every level of the run ----
  for (int L = 0; L < P.n_levels; ++L) {
    const int n = P.g[L].n, nn = n * n;
    for (int s = 0; s < LOWER_SLOT_ALT; ++s) {
      const T* src = P.field[L][s];
      const int off = P.offset[L][s];
      if (!src || off < 0) continue;
      for (int q = tid; q < nn; q += 1024) pool[off + q] = src[gidx<2>(P.g[L], q % n, q / n, 0)];
    }
  }
  __syncthreads();
  auto geo_lds = [&](int L) { Geo gl; gl.n = P.g[L].n; gl.pitch = P.g[L].n; gl.plane = 0; return gl; };
  // Robin columns of an image: x_b = x_in / (1 - i k h) on the inner rows (evomg_robin_bc)
  auto robin_bc = [&](T* img, int L) {
    if constexpr (CPLX) {
      if (P.robin) {
        const int n = P.g[L].n;
        const cplx den{1.0, -P.kh[L]};
        for (int j = 1 + tid; j < n - 1; j += 1024) {
          img[j * n] = sc_div(img[j * n + 1], den);
          img[j * n + n - 1] = sc_div(img[j * n + n - 2], den);
        }
        __syncthreads();
      }
    }
  };
  for (int k = 0; k < P.n_ops; ++k) {
    const LowerOp op = P.op[k];
    const int L = op.level;
    const Geo gl = geo_lds(L);
    const int n = gl.n, nn = n * n;
    switch (op.kind) {                                       // uniform
      case LOWER_SMOOTH: {
        T* x = pool + img_off[L][op.dst];
        const T* f = pool + img_off[L][op.src1];
        const double om = op.om, omm = 1.0 - op.om;
        if (op.part == EVOMG_PART_RED_BLACK) {
          // n is odd: the colour of node q = i + n j is q & 1; every thread relaxes nodes of ONE colour (evomg_rbgs_colour)
          for (int colour = 0; colour < 2; ++colour) {
            for (int q = 2 * tid + colour; q < nn; q += 2048) {
              const int i = q % n, j = q / n;
              if (is_inner<2>(gl, i, j, 0)) x[q] = relax_point<T, 2>(P.st[L], x, f, gl, q, om, omm);
            }
            __syncthreads();
            robin_bc(x, L);
          }
        } else {
          T* y = pool + img_off[L][LOWER_SLOT_ALT];
          for (int q = tid; q < nn; q += 1024) {
            const int i = q % n, j = q / n;
            y[q] = is_inner<2>(gl, i, j, 0) ? relax_point<T, 2>(P.st[L], x, f, gl, q, om, omm) : x[q];
          }
          __syncthreads();
          robin_bc(y, L);
          if (!op.lost) {                                    // advance: the spare image becomes the field
            if (tid == 0) {
              const int t = img_off[L][op.dst];
              img_off[L][op.dst] = img_off[L][LOWER_SLOT_ALT];
              img_off[L][LOWER_SLOT_ALT] = t;
            }
            __syncthreads();
          }
        }
        break;
      }
      case LOWER_RESIDUAL: {
        const T* x = pool + img_off[L][op.src0];
        const T* f = pool + img_off[L][op.src1];
        T* r = pool + img_off[L][op.dst];
        for (int q = tid; q < nn; q += 1024) {
          const int i = q % n, j = q / n;
          if (is_inner<2>(gl, i, j, 0)) r[q] = sc_sub(f[q], apply_stencil<T, 2>(P.st[L], x, gl, q));
        }
        __syncthreads();
        robin_bc(r, L);
        break;
      }
      case LOWER_RESTRICT: {                                 // dst@L = R * src0@(L+1)   (evomg_restrict)
        const Geo gf = geo_lds(L + 1);
        const T* fine = pool + img_off[L + 1][op.src0];
        T* coarse = pool + img_off[L][op.dst];
        const long p = gf.pitch;
        for (int q = tid; q < nn; q += 1024) {
          const int I = q % n, J = q / n;
          if (!is_inner<2>(gl, I, J, 0)) continue;
          const long c = gidx<2>(gf, 2 * I, 2 * J, 0);
          const T e = sc_add(sc_add(fine[c - 1], fine[c + 1]), sc_add(fine[c - p], fine[c + p]));
          const T qq = sc_add(sc_add(fine[c - p - 1], fine[c - p + 1]), sc_add(fine[c + p - 1], fine[c + p + 1]));
          T v = sc_rmul(0.25, fine[c]);
          v = sc_add(v, sc_rmul(0.125, e));
          v = sc_add(v, sc_rmul(0.0625, qq));
          coarse[q] = v;
        }
        __syncthreads();
        robin_bc(coarse, L);
        break;
      }
      case LOWER_PROLONG_ADD:
      case LOWER_PROLONG_SET: {                              // dst@L (+)= w * P * src0@(L-1)   (evomg_prolong)
        const Geo gc = geo_lds(L - 1);
        const T* cz = pool + img_off[L - 1][op.src0];
        T* fz = pool + img_off[L][op.dst];
        for (int q = tid; q < nn; q += 1024) {
          const int i = q % n, j = q / n;
          if (!is_inner<2>(gl, i, j, 0)) continue;
          const T v = interpolate<T, 2>(gc, cz, i, j, 0);
          fz[q] = op.kind == LOWER_PROLONG_ADD ? sc_add(fz[q], sc_rmul(op.om, v)) : v;
        }
        __syncthreads();
        robin_bc(fz, L);
        break;
      }
      case LOWER_ZERO: {                                     // inner nodes (and, Helmholtz, the Robin columns of a zero field)
        T* x = pool + img_off[L][op.dst];
        const bool side = CPLX && P.robin;
        for (int q = tid; q < nn; q += 1024) {
          const int i = q % n, j = q / n;
          if (is_inner<2>(gl, i, j, 0) || (side && j >= 1 && j <= n - 2)) x[q] = zero;
        }
        __syncthreads();
        break;
      }
      case LOWER_COPY: {                                     // the whole array, as the copy op
        const T* s = pool + img_off[L][op.src0];
        T* d = pool + img_off[L][op.dst];
        if (s != d)
          for (int q = tid; q < nn; q += 1024) d[q] = s[q];
        __syncthreads();
        break;
      }
      case LOWER_COARSE: {
        T* x = pool + img_off[L][LOWER_SLOT_SOL];
        const T* b = pool + img_off[L][LOWER_SLOT_CGS];
        T* r = pool + img_off[L][LOWER_SLOT_RES];
        if (tid < 64) {                                      // the one-wave solvers of evomg_krylov.h on the LDS images
          if (P.cgs_kind == EVOMG_CGS_BICGSTAB) {
            if (CPLX && P.robin) evomg_bicgstab_wave2d_body<T, true>(gl, P.st[L], x, b, r, P.cgs_maxit, P.cgs_tol, P.kh[L], P.info);
            else evomg_bicgstab_wave2d_body<T, false>(gl, P.st[L], x, b, r, P.cgs_maxit, P.cgs_tol, 0.0, P.info);
          } else {
            if constexpr (!CPLX) {
              if (P.cg_mask_full) evomg_cg_wave2d_body<T, false, EVOMG_CG_WAVE_ROWS - 1>(gl, P.st[L], x, b, r, (T*)nullptr, P.cgs_maxit, P.cgs_tol, P.info);
              else evomg_cg_wave2d_body<T, false, 0>(gl, P.st[L], x, b, r, (T*)nullptr, P.cgs_maxit, P.cgs_tol, P.info);
            }
          }
        }
        __syncthreads();
        if (P.cgs_kind != EVOMG_CGS_BICGSTAB) robin_bc(x, L);      // the CG path's apply_bc; the BiCGStab kernel writes its own
        break;
      }
      default: break;
    }
  }
  // ---- store what the run wrote ----
  for (int L = 0; L < P.n_levels; ++L) {
    const int n = P.g[L].n, nn = n * n;
    for (int s = 0; s < LOWER_SLOT_ALT; ++s) {
      T* dst = P.field[L][s];
      if (!dst || !P.store[L][s] || img_off[L][s] < 0) continue;
      const T* img = pool + img_off[L][s];
      for (int q = tid; q < nn; q += 1024) dst[gidx<2>(P.g[L], q % n, q / n, 0)] = img[q];
    }
  }
}
