/*
 * evomg_oracle_impl.h — scalar-generic body of the CPU oracle. TEST INFRASTRUCTURE ONLY.
 *
 * Included twice by evomg_oracle.c with
 *   SC   = scalar type (double | cplx)
 *   FN(x) = name mangler (x##_r | x##_c)
 * and the scalar primitives sc_* defined by the includer.
 *
 * Every function is a restatement of what ExaStencils' generated C++ computes for
 * one ExaSlang statement kind that the reference's lowering emits
 * (evostencils/code_generation/exastencils.py:684-925). ExaStencils itself is not
 * in /root/reference (third-party, tag v1.1); semantics marked [EXTERNAL] follow
 * SURVEY.md Appendix B; the recorded run notebooks/tutorial.ipynb:3373 pins them (header of evomg_oracle.c).
 *
 * Arithmetic is written as explicit, fixed-order two-operand operations and is
 * compiled with -ffp-contract=off, so that the HIP kernels (same order, same flag)
 * reproduce every pointwise result bit for bit.
 */

typedef struct {
  int L;            /* absolute level */
  int n;            /* nodes per dimension = 2^L + 1 */
  size_t count;     /* n^d */
  SC* f[EVOMG_F_COUNT];
  SC* init[EVOMG_F_COUNT];   /* initial contents (NULL = zero) */
  SC* alt;          /* ping-pong twin for out-of-place smoothers */
  SC* tmp[6];       /* Krylov work vectors (allocated on demand) */
  SC  st[EVOMG_STENCIL_ENTRIES];
} FN(level_t);

typedef struct {
  evomg_problem_t prob;
  int dim;
  FN(level_t) lv[EVOMG_MAX_LEVELS];
  SC outer_st[EVOMG_STENCIL_ENTRIES];
  SC robin;         /* 1 / (1 - i k h) per level is computed on the fly */
  char err[256];
} FN(ctx_t);

#define IDX2(n, i, j) ((size_t)(i) + (size_t)(n) * (size_t)(j))
#define IDX3(n, i, j, k) ((size_t)(i) + (size_t)(n) * ((size_t)(j) + (size_t)(n) * (size_t)(k)))
#ifndef OMP_WORTH
/* small levels run serially: a parallel region costs more than the loop */
#define OMP_WORTH(n) ((c->dim == 3) ? ((n) >= 33) : ((n) >= 257))
#endif

static SC* FN(zalloc)(size_t count) {
  SC* p = (SC*)calloc(count, sizeof(SC));
  return p;
}

static SC* FN(field)(FN(ctx_t)* c, int level, int field) {
  /* gen_error_u@L and the solution of level L < max are the same array (exastencils.py:296-301). */
  if (field == EVOMG_F_CORR && level < c->prob.max_level) field = EVOMG_F_SOL;
  return c->lv[level].f[field];
}
static void FN(set_fieldptr)(FN(ctx_t)* c, int level, int field, SC* p) {
  if (field == EVOMG_F_CORR && level < c->prob.max_level) field = EVOMG_F_SOL;
  c->lv[level].f[field] = p;
}

/* ---- boundary handling (K8) [EXTERNAL: SURVEY App. B.2] ------------------------------------ */
/* Dirichlet: boundary nodes are simply never written. Helmholtz: Robin at x = 0, 1:
 * x_b = x_in / (1 - i k h)   (example_problems/Helmholtz/2D_FD_Helmholtz_fromL3.exa4:43-59),
 * Dirichlet 0 at y = 0, 1 (already zero and never written). */
static void FN(apply_bc)(FN(ctx_t)* c, int level, SC* x) {
#if SC_IS_COMPLEX
  if (c->prob.bc_kind != EVOMG_BC_HELMHOLTZ_ROBIN) return;
  const int n = c->lv[level].n;
  const double h = 1.0 / (double)(n - 1);
  cplx den; den.re = 1.0; den.im = -(c->prob.wavenumber * h);
  /* rows j = 1 .. n-2 only: the corner/top/bottom nodes are Dirichlet 0 */
  for (int j = 1; j < n - 1; ++j) {
    x[IDX2(n, 0, j)] = sc_div(x[IDX2(n, 1, j)], den);
    x[IDX2(n, n - 1, j)] = sc_div(x[IDX2(n, n - 2, j)], den);
  }
#else
  (void)c; (void)level; (void)x;
#endif
}

/* ---- stencil application at one inner node ------------------------------------------------- */
static inline SC FN(apply_st2)(const SC* st, const SC* x, int n, int i, int j) {
  SC a = sc_mul(st[0], x[IDX2(n, i, j)]);
  a = sc_add(a, sc_mul(st[1], x[IDX2(n, i - 1, j)]));
  a = sc_add(a, sc_mul(st[2], x[IDX2(n, i + 1, j)]));
  a = sc_add(a, sc_mul(st[3], x[IDX2(n, i, j - 1)]));
  a = sc_add(a, sc_mul(st[4], x[IDX2(n, i, j + 1)]));
  return a;
}
static inline SC FN(apply_st3)(const SC* st, const SC* x, int n, int i, int j, int k) {
  SC a = sc_mul(st[0], x[IDX3(n, i, j, k)]);
  a = sc_add(a, sc_mul(st[1], x[IDX3(n, i - 1, j, k)]));
  a = sc_add(a, sc_mul(st[2], x[IDX3(n, i + 1, j, k)]));
  a = sc_add(a, sc_mul(st[3], x[IDX3(n, i, j - 1, k)]));
  a = sc_add(a, sc_mul(st[4], x[IDX3(n, i, j + 1, k)]));
  a = sc_add(a, sc_mul(st[5], x[IDX3(n, i, j, k - 1)]));
  a = sc_add(a, sc_mul(st[6], x[IDX3(n, i, j, k + 1)]));
  return a;
}
/* point relaxation: x* = (f - sum_{o != 0} a_o x_o) / a_0 ; x <- (1-w) x + w x*
 * ("solve locally ... relax w" with one unknown, exastencils.py:608-633) [EXTERNAL: App. B.5] */
static inline SC FN(relax2)(const SC* st, const SC* x, const SC* f, int n, int i, int j, double om, double omm) {
  SC s = f[IDX2(n, i, j)];
  s = sc_sub(s, sc_mul(st[1], x[IDX2(n, i - 1, j)]));
  s = sc_sub(s, sc_mul(st[2], x[IDX2(n, i + 1, j)]));
  s = sc_sub(s, sc_mul(st[3], x[IDX2(n, i, j - 1)]));
  s = sc_sub(s, sc_mul(st[4], x[IDX2(n, i, j + 1)]));
  SC xs = sc_div(s, st[0]);
  return sc_add(sc_rmul(omm, x[IDX2(n, i, j)]), sc_rmul(om, xs));
}
static inline SC FN(relax3)(const SC* st, const SC* x, const SC* f, int n, int i, int j, int k, double om, double omm) {
  SC s = f[IDX3(n, i, j, k)];
  s = sc_sub(s, sc_mul(st[1], x[IDX3(n, i - 1, j, k)]));
  s = sc_sub(s, sc_mul(st[2], x[IDX3(n, i + 1, j, k)]));
  s = sc_sub(s, sc_mul(st[3], x[IDX3(n, i, j - 1, k)]));
  s = sc_sub(s, sc_mul(st[4], x[IDX3(n, i, j + 1, k)]));
  s = sc_sub(s, sc_mul(st[5], x[IDX3(n, i, j, k - 1)]));
  s = sc_sub(s, sc_mul(st[6], x[IDX3(n, i, j, k + 1)]));
  SC xs = sc_div(s, st[0]);
  return sc_add(sc_rmul(omm, x[IDX3(n, i, j, k)]), sc_rmul(om, xs));
}

/* ---- `with jacobi` slot handling. The sweep has written the next slot y from the active slot x.
 * TEXTBOOK: one advance, y becomes the field (what exastencils.py:338-359 intends by inserting `advance`).
 * EXASTENCILS_V11: no advance ever happens, the field keeps x and y stays the spare slot — the behaviour of the
 * reference's pipeline with ExaStencils v1.1, pinned by notebooks/tutorial.ipynb:3373 (oracle/APPB_MATRIX.md). */
static void FN(advance_slot)(FN(ctx_t)* c, int level, int dstf, SC* x, SC* y) {
  if (c->prob.jacobi_semantics == EVOMG_JACOBI_EXASTENCILS_V11) return;
  c->lv[level].alt = x;
  FN(set_fieldptr)(c, level, dstf, y);
}

/* ---- K1: damped Jacobi ("with jacobi": writes the next slot, then advance; exastencils.py:338-359) */
static void FN(smooth_jacobi)(FN(ctx_t)* c, int level, int dstf, int rhsf, double om) {
  FN(level_t)* lv = &c->lv[level];
  const int n = lv->n;
  const SC* x = FN(field)(c, level, dstf);
  const SC* f = FN(field)(c, level, rhsf);
  SC* y = lv->alt;
  const double omm = 1.0 - om;
  if (c->dim == 2) {
#pragma omp parallel for schedule(static) if (OMP_WORTH(n))
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        if (i == 0 || j == 0 || i == n - 1 || j == n - 1) y[IDX2(n, i, j)] = x[IDX2(n, i, j)];
        else y[IDX2(n, i, j)] = FN(relax2)(lv->st, x, f, n, i, j, om, omm);
      }
  } else {
#pragma omp parallel for schedule(static) if (OMP_WORTH(n))
    for (int k = 0; k < n; ++k)
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          if (i == 0 || j == 0 || k == 0 || i == n - 1 || j == n - 1 || k == n - 1)
            y[IDX3(n, i, j, k)] = x[IDX3(n, i, j, k)];
          else y[IDX3(n, i, j, k)] = FN(relax3)(lv->st, x, f, n, i, j, k, om, omm);
        }
  }
  FN(apply_bc)(c, level, y);
  FN(advance_slot)(c, level, dstf, (SC*)x, y);
}

/* ---- K2: red-black Gauss-Seidel: colour 0 = even index sum first, then colour 1, in place,
 * boundary handling in between (exastencils.py:659-667,785-798) [EXTERNAL: App. B.7] */
static void FN(smooth_rbgs)(FN(ctx_t)* c, int level, int dstf, int rhsf, double om) {
  FN(level_t)* lv = &c->lv[level];
  const int n = lv->n;
  SC* x = FN(field)(c, level, dstf);
  const SC* f = FN(field)(c, level, rhsf);
  const double omm = 1.0 - om;
  for (int colour = 0; colour < 2; ++colour) {
    if (c->dim == 2) {
#pragma omp parallel for schedule(static) if (OMP_WORTH(n))
      for (int j = 1; j < n - 1; ++j)
        for (int i = 1; i < n - 1; ++i)
          if (((i + j) & 1) == colour) x[IDX2(n, i, j)] = FN(relax2)(lv->st, x, f, n, i, j, om, omm);
    } else {
#pragma omp parallel for schedule(static) if (OMP_WORTH(n))
      for (int k = 1; k < n - 1; ++k)
        for (int j = 1; j < n - 1; ++j)
          for (int i = 1; i < n - 1; ++i)
            if (((i + j + k) & 1) == colour)
              x[IDX3(n, i, j, k)] = FN(relax3)(lv->st, x, f, n, i, j, k, om, omm);
    }
    FN(apply_bc)(c, level, x);
  }
}

/* ---- K3: collective block-Jacobi. Non-overlapping tiles of shape `b` anchored at the first inner
 * node; tiles cut by the far boundary solve the smaller system (IR meaning:
 * evostencils/stencils/multiple.py:204-217; SURVEY App. B.6 decision). Unknown order inside a tile:
 * i0 outermost, as the reference lists them (notebooks/tutorial.ipynb:3432-3447). Dense solve =
 * Gaussian elimination without pivoting in that order. */
static void FN(smooth_block)(FN(ctx_t)* c, int level, int dstf, int rhsf, double om, const int* b) {
  FN(level_t)* lv = &c->lv[level];
  const int n = lv->n, dim = c->dim;
  const SC* x = FN(field)(c, level, dstf);
  const SC* f = FN(field)(c, level, rhsf);
  SC* y = lv->alt;
  const SC* st = lv->st;
  const double omm = 1.0 - om;
  const int bx = b[0], by = b[1], bz = (dim == 3) ? b[2] : 1;
  const int inner = n - 2;
  const int tx = (inner + bx - 1) / bx, ty = (inner + by - 1) / by, tz = (dim == 3) ? (inner + bz - 1) / bz : 1;
  memcpy(y, x, lv->count * sizeof(SC));
  static const int off[7][3] = {{0,0,0},{-1,0,0},{1,0,0},{0,-1,0},{0,1,0},{0,0,-1},{0,0,1}};
  const int nnb = 1 + 2 * dim;
#pragma omp parallel for schedule(static) collapse(2) if (OMP_WORTH(n))
  for (int tk = 0; tk < tz; ++tk)
    for (int tj = 0; tj < ty; ++tj)
      for (int ti = 0; ti < tx; ++ti) {
        const int i0 = 1 + ti * bx, j0 = 1 + tj * by, k0 = (dim == 3) ? 1 + tk * bz : 0;
        const int mx = (i0 + bx <= n - 1) ? bx : (n - 1 - i0);
        const int my = (j0 + by <= n - 1) ? by : (n - 1 - j0);
        const int mz = (dim == 3) ? ((k0 + bz <= n - 1) ? bz : (n - 1 - k0)) : 1;
        const int m = mx * my * mz;
        SC A[EVOMG_MAX_BLOCK_CELLS][EVOMG_MAX_BLOCK_CELLS];
        SC r[EVOMG_MAX_BLOCK_CELLS];
        for (int p = 0; p < m; ++p)
          for (int q = 0; q < m; ++q) A[p][q] = sc_zero();
        /* assemble: unknown index p = (ci*my + cj)*mz + ck  (i0 outermost) */
        for (int ci = 0; ci < mx; ++ci)
          for (int cj = 0; cj < my; ++cj)
            for (int ck = 0; ck < mz; ++ck) {
              const int p = (ci * my + cj) * mz + ck;
              const int gi = i0 + ci, gj = j0 + cj, gk = k0 + ck;
              const size_t gc = (dim == 3) ? IDX3(n, gi, gj, gk) : IDX2(n, gi, gj);
              SC s = f[gc];
              A[p][p] = st[0];
              for (int e = 1; e < nnb; ++e) {
                const int di = ci + off[e][0], dj = cj + off[e][1], dk = ck + off[e][2];
                if (di >= 0 && di < mx && dj >= 0 && dj < my && dk >= 0 && dk < mz) {
                  A[p][(di * my + dj) * mz + dk] = st[e];
                } else {
                  const size_t gn = (dim == 3) ? IDX3(n, gi + off[e][0], gj + off[e][1], gk + off[e][2])
                                               : IDX2(n, gi + off[e][0], gj + off[e][1]);
                  s = sc_sub(s, sc_mul(st[e], x[gn]));
                }
              }
              r[p] = s;
            }
        /* forward elimination */
        for (int p = 0; p < m; ++p) {
          for (int q = p + 1; q < m; ++q) {
            if (sc_iszero(A[q][p])) continue;
            const SC fac = sc_div(A[q][p], A[p][p]);
            for (int t = p + 1; t < m; ++t) A[q][t] = sc_sub(A[q][t], sc_mul(fac, A[p][t]));
            r[q] = sc_sub(r[q], sc_mul(fac, r[p]));
          }
        }
        /* back substitution */
        for (int p = m - 1; p >= 0; --p) {
          SC s = r[p];
          for (int t = p + 1; t < m; ++t) s = sc_sub(s, sc_mul(A[p][t], r[t]));
          r[p] = sc_div(s, A[p][p]);
        }
        for (int ci = 0; ci < mx; ++ci)
          for (int cj = 0; cj < my; ++cj)
            for (int ck = 0; ck < mz; ++ck) {
              const int p = (ci * my + cj) * mz + ck;
              const size_t gc = (dim == 3) ? IDX3(n, i0 + ci, j0 + cj, k0 + ck) : IDX2(n, i0 + ci, j0 + cj);
              y[gc] = sc_add(sc_rmul(omm, x[gc]), sc_rmul(om, r[p]));
            }
      }
  FN(apply_bc)(c, level, y);
  FN(advance_slot)(c, level, dstf, (SC*)x, y);
}

/* ---- generic y = A x on inner nodes (boundary of y untouched) -------------------------------- */
static void FN(apply_op)(FN(ctx_t)* c, int level, const SC* st, const SC* x, SC* y) {
  const int n = c->lv[level].n;
  if (c->dim == 2) {
#pragma omp parallel for schedule(static) if (OMP_WORTH(n))
    for (int j = 1; j < n - 1; ++j)
      for (int i = 1; i < n - 1; ++i) y[IDX2(n, i, j)] = FN(apply_st2)(st, x, n, i, j);
  } else {
#pragma omp parallel for schedule(static) if (OMP_WORTH(n))
    for (int k = 1; k < n - 1; ++k)
      for (int j = 1; j < n - 1; ++j)
        for (int i = 1; i < n - 1; ++i) y[IDX3(n, i, j, k)] = FN(apply_st3)(st, x, n, i, j, k);
  }
}

/* ---- K4: residual  r = f - (A x)   (exastencils.py:837-853) ------------------------------------ */
static void FN(residual_st)(FN(ctx_t)* c, int level, const SC* st, const SC* x, const SC* f, SC* r) {
  const int n = c->lv[level].n;
  if (c->dim == 2) {
#pragma omp parallel for schedule(static) if (OMP_WORTH(n))
    for (int j = 1; j < n - 1; ++j)
      for (int i = 1; i < n - 1; ++i)
        r[IDX2(n, i, j)] = sc_sub(f[IDX2(n, i, j)], FN(apply_st2)(st, x, n, i, j));
  } else {
#pragma omp parallel for schedule(static) if (OMP_WORTH(n))
    for (int k = 1; k < n - 1; ++k)
      for (int j = 1; j < n - 1; ++j)
        for (int i = 1; i < n - 1; ++i)
          r[IDX3(n, i, j, k)] = sc_sub(f[IDX3(n, i, j, k)], FN(apply_st3)(st, x, n, i, j, k));
  }
  FN(apply_bc)(c, level, r);
}

/* ---- K5: full-weighting restriction, coarse-point driven [EXTERNAL: App. B.3]
 * weights [1/4 1/2 1/4]^(x)d (consistent with stencils/gallery.py:188-219) --------------------- */
static void FN(restrict_fw)(FN(ctx_t)* c, int coarse_level, const SC* fine, SC* coarse) {
  const int nc = c->lv[coarse_level].n, nf = c->lv[coarse_level + 1].n;
  if (c->dim == 2) {
#pragma omp parallel for schedule(static) if (OMP_WORTH(nf))
    for (int J = 1; J < nc - 1; ++J)
      for (int I = 1; I < nc - 1; ++I) {
        const int i = 2 * I, j = 2 * J;
        SC e = sc_add(sc_add(fine[IDX2(nf, i - 1, j)], fine[IDX2(nf, i + 1, j)]),
                      sc_add(fine[IDX2(nf, i, j - 1)], fine[IDX2(nf, i, j + 1)]));
        SC k = sc_add(sc_add(fine[IDX2(nf, i - 1, j - 1)], fine[IDX2(nf, i + 1, j - 1)]),
                      sc_add(fine[IDX2(nf, i - 1, j + 1)], fine[IDX2(nf, i + 1, j + 1)]));
        SC v = sc_rmul(0.25, fine[IDX2(nf, i, j)]);
        v = sc_add(v, sc_rmul(0.125, e));
        v = sc_add(v, sc_rmul(0.0625, k));
        coarse[IDX2(nc, I, J)] = v;
      }
  } else {
#pragma omp parallel for schedule(static) if (OMP_WORTH(nf))
    for (int K = 1; K < nc - 1; ++K)
      for (int J = 1; J < nc - 1; ++J)
        for (int I = 1; I < nc - 1; ++I) {
          const int i = 2 * I, j = 2 * J, k = 2 * K;
          SC face = sc_zero(), edge = sc_zero(), corner = sc_zero();
          /* fixed order: dz outermost, then dy, then dx */
          for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
              for (int dx = -1; dx <= 1; ++dx) {
                const int nz = (dx != 0) + (dy != 0) + (dz != 0);
                const SC v = fine[IDX3(nf, i + dx, j + dy, k + dz)];
                if (nz == 1) face = sc_add(face, v);
                else if (nz == 2) edge = sc_add(edge, v);
                else if (nz == 3) corner = sc_add(corner, v);
              }
          SC v = sc_rmul(0.125, fine[IDX3(nf, i, j, k)]);
          v = sc_add(v, sc_rmul(0.0625, face));
          v = sc_add(v, sc_rmul(0.03125, edge));
          v = sc_add(v, sc_rmul(0.015625, corner));
          coarse[IDX3(nc, I, J, K)] = v;
        }
  }
  FN(apply_bc)(c, coarse_level, coarse);
}

/* ---- K6/K6': bi-/tri-linear prolongation of the coarse field to fine inner node (i,j[,k]) ------- */
static inline SC FN(interp2)(const SC* cz, int nc, int i, int j) {
  const int I = i >> 1, J = j >> 1;
  if (!(i & 1) && !(j & 1)) return cz[IDX2(nc, I, J)];
  if ((i & 1) && !(j & 1)) return sc_rmul(0.5, sc_add(cz[IDX2(nc, I, J)], cz[IDX2(nc, I + 1, J)]));
  if (!(i & 1) && (j & 1)) return sc_rmul(0.5, sc_add(cz[IDX2(nc, I, J)], cz[IDX2(nc, I, J + 1)]));
  return sc_rmul(0.25, sc_add(sc_add(cz[IDX2(nc, I, J)], cz[IDX2(nc, I + 1, J)]),
                              sc_add(cz[IDX2(nc, I, J + 1)], cz[IDX2(nc, I + 1, J + 1)])));
}
static inline SC FN(interp3)(const SC* cz, int nc, int i, int j, int k) {
  const int I = i >> 1, J = j >> 1, K = k >> 1;
  const int ox = i & 1, oy = j & 1, oz = k & 1;
  /* sum over the 2^(#odd) coarse parents, fixed order dz, dy, dx; weight 1/2^(#odd) */
  SC s = sc_zero();
  for (int dz = 0; dz <= oz; ++dz)
    for (int dy = 0; dy <= oy; ++dy)
      for (int dx = 0; dx <= ox; ++dx) s = sc_add(s, cz[IDX3(nc, I + dx, J + dy, K + dz)]);
  const int nodd = ox + oy + oz;
  const double w = (nodd == 0) ? 1.0 : (nodd == 1) ? 0.5 : (nodd == 2) ? 0.25 : 0.125;
  return sc_rmul(w, s);
}
static void FN(prolong)(FN(ctx_t)* c, int fine_level, const SC* coarse, SC* fine, int accumulate, double om) {
  const int nf = c->lv[fine_level].n, nc = c->lv[fine_level - 1].n;
  if (c->dim == 2) {
#pragma omp parallel for schedule(static) if (OMP_WORTH(nf))
    for (int j = 1; j < nf - 1; ++j)
      for (int i = 1; i < nf - 1; ++i) {
        const SC v = FN(interp2)(coarse, nc, i, j);
        if (accumulate) fine[IDX2(nf, i, j)] = sc_add(fine[IDX2(nf, i, j)], sc_rmul(om, v));
        else fine[IDX2(nf, i, j)] = v;
      }
  } else {
#pragma omp parallel for schedule(static) if (OMP_WORTH(nf))
    for (int k = 1; k < nf - 1; ++k)
      for (int j = 1; j < nf - 1; ++j)
        for (int i = 1; i < nf - 1; ++i) {
          const SC v = FN(interp3)(coarse, nc, i, j, k);
          if (accumulate) fine[IDX3(nf, i, j, k)] = sc_add(fine[IDX3(nf, i, j, k)], sc_rmul(om, v));
          else fine[IDX3(nf, i, j, k)] = v;
        }
  }
  FN(apply_bc)(c, fine_level, fine);
}

/* ---- reductions over inner nodes. dot(a,b) = sum a_i * b_i WITHOUT conjugation: literal reading of the DSL's
 * `dot ( resHat, Residual )` typed Complex<Real> (Helmholtz/2D_FD_Helmholtz_fromL3.exa3:86-88,158) [EXTERNAL].
 *
 * Summation order = the device's two-pass reduction (csrc/evomg_kernels.h evomg_dot_partial / evomg_dot_final),
 * restated so that Krylov recurrences fed by these sums (preconditioned BiCGStab on the finest level) follow the
 * same trajectory on both sides: pass 1 — B = min(rows, 1024) blocks of 256 threads, block b takes the inner rows
 * b, b+B, ..., thread t the nodes i = 1+t, 1+t+256, ... of each row, then a wave64 shuffle tree per wave and the
 * same tree over the 4 wave sums; pass 2 — 256 threads take the partials t, t+256, ..., same trees. */
#define DOT_BLOCKS 1024
#define DOT_THREADS 256
static SC FN(tree64)(SC* v);
static SC FN(block_tree)(SC* th) {
  SC waves[64];
  for (int w = 0; w < 64; ++w) waves[w] = sc_zero();
  for (int w = 0; w < DOT_THREADS / 64; ++w) waves[w] = FN(tree64)(&th[64 * w]);
  return FN(tree64)(waves);
}
static SC FN(dot_blocked)(FN(ctx_t)* c, int level, const SC* a, const SC* b, int fused);
static SC FN(dot)(FN(ctx_t)* c, int level, const SC* a, const SC* b) { return FN(dot_blocked)(c, level, a, b, 0); }
/* fused: the per-thread accumulation is a fused multiply-add chain (the launch-per-iteration CG on large coarsest grids,
 * csrc/evomg_krylov.h kcg_*); the node order and the trees are the same */
static SC FN(dot_blocked)(FN(ctx_t)* c, int level, const SC* a, const SC* b, int fused) {
  const int n = c->lv[level].n, m = n - 2;
  const long rows = (c->dim == 2) ? m : (long)m * m;
  const int B = (int)(rows < DOT_BLOCKS ? rows : DOT_BLOCKS);
  SC partial[DOT_BLOCKS];
#pragma omp parallel for schedule(static) if (OMP_WORTH(n))
  for (int blk = 0; blk < B; ++blk) {
    SC th[DOT_THREADS];
    for (int t = 0; t < DOT_THREADS; ++t) th[t] = sc_zero();
    for (long r = blk; r < rows; r += B) {
      const int j = 1 + (int)(r % m), k = (c->dim == 3) ? 1 + (int)(r / m) : 0;
      const size_t base = (c->dim == 3) ? IDX3(n, 0, j, k) : IDX2(n, 0, j);
      for (int i = 1; i <= m; ++i) {
        const int t = (i - 1) % DOT_THREADS;
        th[t] = fused ? sc_fma(a[base + i], b[base + i], th[t]) : sc_add(th[t], sc_mul(a[base + i], b[base + i]));
      }
    }
    partial[blk] = FN(block_tree)(th);
  }
  SC th[DOT_THREADS];
  for (int t = 0; t < DOT_THREADS; ++t) th[t] = sc_zero();
  for (int i = 0; i < B; ++i) th[i % DOT_THREADS] = sc_add(th[i % DOT_THREADS], partial[i]);
  return FN(block_tree)(th);
}
/* residual norm of the solver loop: |sqrt(dot(r,r))|  (for real fields: sqrt(sum r^2)) [EXTERNAL: App. B.8] */
static double FN(norm)(FN(ctx_t)* c, int level, const SC* a) { return sc_norm_from_dot(FN(dot)(c, level, a, a)); }

/* ---- Krylov solvers on the coarsest grid: node ownership, reduction order and arithmetic ----------------------
 * The HIP solvers run as ONE workgroup of `lanes` = 512 or 1024 threads = 8 or 16 wave64s (csrc/evomg_krylov.h). To make
 * the oracle bit-identical to them (a Krylov recurrence amplifies any difference in summation order up to the
 * attainable-accuracy floor) this block restates
 *   (i) which inner nodes a thread owns and in which order it accumulates them:
 *       2D "tile" ownership (when cb = ceil(m/64) <= 16): wave w covers the 64 columns of column block w % cb and the
 *       R = ceil(m / bands) rows of band w / cb (bands = (lanes/64) / cb); lane l owns column 64*(w % cb) + l, rows
 *       ascending; lanes = 512 when cb <= 2 (grids up to 129^2), else 1024; with cb = 2 ("pair" tiles) wave w is band w
 *       and lane l owns the columns 2l, 2l+1, row by row;
 *       3D "column" ownership (when m*m <= 1024): thread t owns the z-column (t % m, t / m), iz ascending (CG on 33^3:
 *       split into 8 slabs of planes, one workgroup each, see dot_canon_grouped);
 *       otherwise (large 3D, or very wide 2D grids) "strided" ownership: thread t owns q = t, t+1024, ... with q running
 *       x-fastest over the inner nodes;
 *  (ii) the reduction tree: wave64 tree with offsets 32,16,8,4,2,1, then the same tree over the 16 wave sums;
 * (iii) fused multiply-add arithmetic inside the solvers (SpMV, dots, vector updates), written out explicitly — what
 *       a C++ compiler does to the generated solver at -O3 with contraction on, and what halves the FP64 issue
 *       slots of the device kernels. Pointwise multigrid kernels elsewhere stay unfused. */
#define KRYLOV_LANES 1024
typedef struct {
  int tile, pair, column, m, cb, bands, R, slots, lanes, wave;
  long total;
} FN(own_t);
/* for_cg: the pair tiles are used by the real-valued CG only (BiCGStab and complex solvers: one column per lane) */
static FN(own_t) FN(own_make)(FN(ctx_t)* c, int level, int for_cg) {
  FN(own_t) o;
  o.m = c->lv[level].n - 2;
  o.total = (c->dim == 2) ? (long)o.m * o.m : (long)o.m * o.m * o.m;
  o.cb = (o.m + 63) / 64;
  o.tile = (c->dim == 2 && o.cb <= 16);
  /* workgroup size: 512 threads (8 waves) for 2D grids up to two column blocks, else KRYLOV_LANES = 1024 */
  o.lanes = (o.tile && o.cb <= 2) ? 512 : KRYLOV_LANES;
  /* "pair" tiles (two column blocks, i.e. 129^2): wave w = band w, lane l owns the columns 2l and 2l+1 and visits
   * them row by row (row i: column 2l, then 2l+1) */
  o.pair = (o.tile && o.cb == 2 && for_cg && !SC_IS_COMPLEX);
  o.bands = o.tile ? (o.pair ? o.lanes / 64 : (o.lanes / 64) / o.cb) : 0;
  o.R = o.tile ? (o.m + o.bands - 1) / o.bands : 0;
  /* 3D "column" ownership when m*m <= 1024: thread t owns the z-column (t % m, t / m), iz ascending */
  o.column = (c->dim == 3 && (long)o.m * o.m <= o.lanes);
  o.slots = o.pair ? 2 * o.R : o.tile ? o.R : o.column ? o.m : (int)((o.total + o.lanes - 1) / o.lanes);
  /* "wave" ownership, real-valued CG on 2D grids up to 33^2 (csrc/evomg_krylov.h: evomg_cg_wave2d): ONE wave64, lane l =
   * (half h = l >> 5, column l & 31) owns the rows 16 h .. 16 h + 15 of its column, rows ascending; the reduction is the
   * wave64 tree alone */
  o.wave = (c->dim == 2 && for_cg && !SC_IS_COMPLEX && o.m <= 31);
  if (o.wave) { o.tile = 0; o.pair = 0; o.lanes = 64; o.bands = 2; o.R = 16; o.slots = 16; }
  return o;
}
/* dense index of slot i of thread t, or (size_t)-1 if the slot is empty */
static size_t FN(own_node)(FN(ctx_t)* c, int level, const FN(own_t)* o, int t, int i) {
  const int n = c->lv[level].n, m = o->m;
  if (o->wave) {
    const int h = t >> 5, ix = t & 31, iy = h * o->R + i;
    if (t >= 64 || ix >= m || i >= o->R || iy >= m) return (size_t)-1;
    return IDX2(n, ix + 1, iy + 1);
  }
  if (o->pair) {
    const int w = t >> 6, l = t & 63;
    if (w >= o->bands || 2 * l >= m) return (size_t)-1;
    const int ncol = (2 * l + 1 < m) ? 2 : 1;
    const int row = i / ncol, ix = 2 * l + i % ncol, iy = w * o->R + row;
    if (row >= o->R || iy >= m) return (size_t)-1;
    return IDX2(n, ix + 1, iy + 1);
  }
  if (o->tile) {
    const int w = t >> 6, l = t & 63;
    const int band = w / o->cb, ix = 64 * (w % o->cb) + l, iy = band * o->R + i;
    if (band >= o->bands || ix >= m || iy >= m) return (size_t)-1;
    return IDX2(n, ix + 1, iy + 1);
  }
  if (o->column) {
    if (t >= m * m || i >= m) return (size_t)-1;
    return IDX3(n, t % m + 1, t / m + 1, i + 1);
  }
  const long q = t + (long)o->lanes * i;
  if (q >= o->total) return (size_t)-1;
  const int ix = (int)(q % m);
  const long r = q / m;
  return (c->dim == 2) ? IDX2(n, ix + 1, 1 + (int)r) : IDX3(n, ix + 1, 1 + (int)(r % m), 1 + (int)(r / m));
}
static SC FN(tree64)(SC* v) {
  for (int off = 32; off > 0; off >>= 1)
    for (int l = 0; l < off; ++l) v[l] = sc_add(v[l], v[l + off]);
  return v[0];
}
/* CG on 3D column grids with at least 2 planes per group (33^3) runs as KRYLOV_GROUPS = 8 workgroups, one per slab of
 * z planes [g*m/8, (g+1)*m/8) (csrc/evomg_krylov.h: evomg_cg_multi_wg3d): thread t accumulates its column WITHIN the
 * slab (iz ascending), each workgroup reduces its 1024 partials with the wave/16-wave trees, and the group sums are
 * added in the order 0, 1, ..., 7. */
#define KRYLOV_GROUPS 8
static int FN(cg_groups)(FN(ctx_t)* c, int level) {
  const FN(own_t) o = FN(own_make)(c, level, 1);
  return (c->dim == 3 && o.column && o.m >= 2 * KRYLOV_GROUPS) ? KRYLOV_GROUPS : 1;
}
static SC FN(dot_canon_grouped)(FN(ctx_t)* c, int level, const SC* a, const SC* b) {
  const FN(own_t) o = FN(own_make)(c, level, 1);
  const int n = c->lv[level].n, m = o.m;
  SC total = sc_zero();
  for (int g = 0; g < KRYLOV_GROUPS; ++g) {
    const int zs = (g * m) / KRYLOV_GROUPS, ze = ((g + 1) * m) / KRYLOV_GROUPS;
    SC part[KRYLOV_LANES];
    for (int t = 0; t < KRYLOV_LANES; ++t) {
      SC s = sc_zero();
      if (t < m * m)
        for (int iz = zs; iz < ze; ++iz) {
          const size_t idx = IDX3(n, t % m + 1, t / m + 1, iz + 1);
          s = sc_fma(a[idx], b[idx], s);
        }
      part[t] = s;
    }
    SC waves[64];
    for (int w = 0; w < 64; ++w) waves[w] = sc_zero();
    for (int w = 0; w < KRYLOV_LANES / 64; ++w) waves[w] = FN(tree64)(&part[64 * w]);
    const SC gs = FN(tree64)(waves);
    total = (g == 0) ? gs : sc_add(total, gs);
  }
  return total;
}
static SC FN(dot_canon_for)(FN(ctx_t)* c, int level, const SC* a, const SC* b, int for_cg) {
  const FN(own_t) o = FN(own_make)(c, level, for_cg);
  SC part[KRYLOV_LANES];
  for (int t = 0; t < o.lanes; ++t) {
    SC s = sc_zero();
    for (int i = 0; i < o.slots; ++i) {
      const size_t idx = FN(own_node)(c, level, &o, t, i);
      if (idx != (size_t)-1) s = sc_fma(a[idx], b[idx], s);
    }
    part[t] = s;
  }
  SC waves[64];
  for (int w = 0; w < 64; ++w) waves[w] = sc_zero();
  for (int w = 0; w < o.lanes / 64; ++w) waves[w] = FN(tree64)(&part[64 * w]);
  return FN(tree64)(waves);
}
static SC FN(dot_canon)(FN(ctx_t)* c, int level, const SC* a, const SC* b) { return FN(dot_canon_for)(c, level, a, b, 0); }
/* fused-multiply-add forms used inside the Krylov solvers */
static inline SC FN(apply_fma2)(const SC* st, const SC* x, int n, int i, int j) {
  SC a = sc_mul(st[0], x[IDX2(n, i, j)]);
  a = sc_fma(st[1], x[IDX2(n, i - 1, j)], a);
  a = sc_fma(st[2], x[IDX2(n, i + 1, j)], a);
  a = sc_fma(st[3], x[IDX2(n, i, j - 1)], a);
  a = sc_fma(st[4], x[IDX2(n, i, j + 1)], a);
  return a;
}
static inline SC FN(apply_fma3)(const SC* st, const SC* x, int n, int i, int j, int k) {
  SC a = sc_mul(st[0], x[IDX3(n, i, j, k)]);
  a = sc_fma(st[1], x[IDX3(n, i - 1, j, k)], a);
  a = sc_fma(st[2], x[IDX3(n, i + 1, j, k)], a);
  a = sc_fma(st[3], x[IDX3(n, i, j - 1, k)], a);
  a = sc_fma(st[4], x[IDX3(n, i, j + 1, k)], a);
  a = sc_fma(st[5], x[IDX3(n, i, j, k - 1)], a);
  a = sc_fma(st[6], x[IDX3(n, i, j, k + 1)], a);
  return a;
}
/* y = A x (RES == 0) or y = f - A x (RES == 1) on inner nodes, fma form */
static void FN(krylov_apply)(FN(ctx_t)* c, int level, const SC* st, const SC* x, const SC* f, SC* y, int res) {
  const int n = c->lv[level].n;
  if (c->dim == 2) {
    for (int j = 1; j < n - 1; ++j)
      for (int i = 1; i < n - 1; ++i) {
        const SC a = FN(apply_fma2)(st, x, n, i, j);
        y[IDX2(n, i, j)] = res ? sc_sub(f[IDX2(n, i, j)], a) : a;
      }
  } else {
    for (int k = 1; k < n - 1; ++k)
      for (int j = 1; j < n - 1; ++j)
        for (int i = 1; i < n - 1; ++i) {
          const SC a = FN(apply_fma3)(st, x, n, i, j, k);
          y[IDX3(n, i, j, k)] = res ? sc_sub(f[IDX3(n, i, j, k)], a) : a;
        }
  }
}
/* z = a*y + x on inner nodes (one fma per real product) */
static void FN(krylov_axpy)(FN(ctx_t)* c, int level, SC* z, SC a, const SC* y, const SC* x) {
  const int n = c->lv[level].n;
  const int rows = (c->dim == 2) ? n : n * n;
  for (int r = 0; r < rows; ++r) {
    const int j = (c->dim == 2) ? r : r % n, k = (c->dim == 2) ? 1 : r / n;
    if (j >= 1 && j < n - 1 && k >= 1 && (c->dim == 2 || k < n - 1)) {
      const size_t base = (size_t)r * (size_t)n;
      for (int i = 1; i < n - 1; ++i) z[base + i] = sc_fma(a, y[base + i], x[base + i]);
    }
  }
}

static void FN(zero_inner)(FN(ctx_t)* c, int level, SC* x) {
  const int n = c->lv[level].n;
  if (c->dim == 2) {
    for (int j = 1; j < n - 1; ++j)
      for (int i = 1; i < n - 1; ++i) x[IDX2(n, i, j)] = sc_zero();
  } else {
    for (int k = 1; k < n - 1; ++k)
      for (int j = 1; j < n - 1; ++j)
        for (int i = 1; i < n - 1; ++i) x[IDX3(n, i, j, k)] = sc_zero();
  }
  FN(apply_bc)(c, level, x);
}

static SC* FN(tmpvec)(FN(ctx_t)* c, int level, int which) {
  FN(level_t)* lv = &c->lv[level];
  if (!lv->tmp[which]) lv->tmp[which] = FN(zalloc)(lv->count);
  return lv->tmp[which];
}

/* z = x + a*y on inner nodes (a scalar) */
static void FN(axpy_to)(FN(ctx_t)* c, int level, SC* z, const SC* x, SC a, const SC* y) {
  const int n = c->lv[level].n;
  const int rows = (c->dim == 2) ? n : n * n;
#pragma omp parallel for schedule(static) if (OMP_WORTH(n))
  for (int r = 0; r < rows; ++r) {
    const int j = (c->dim == 2) ? r : r % n, k = (c->dim == 2) ? 1 : r / n;
    if (j >= 1 && j < n - 1 && k >= 1 && (c->dim == 2 || k < n - 1)) {
      const size_t base = (size_t)r * (size_t)n;
      for (int i = 1; i < n - 1; ++i) z[base + i] = sc_add(x[base + i], sc_mul(a, y[base + i]));
    }
  }
}

/* ---- K9a: CG coarse-grid solver = gen_mgCycle@minLevel for Poisson
 * (solver_cgs = "CG", 1000 its, 1e-12: Poisson/2D_FD_Poisson_fromL2.exa3:12-14) [EXTERNAL: App. B.9] */
static int FN(coarse_cg)(FN(ctx_t)* c) {
  const int L = c->prob.min_level;
  FN(level_t)* lv = &c->lv[L];
  SC* x = FN(field)(c, L, EVOMG_F_CORR);
  const SC* b = FN(field)(c, L, EVOMG_F_CGS_RHS);
  SC* r = FN(field)(c, L, EVOMG_F_RES);
  SC* p = FN(tmpvec)(c, L, 0);
  SC* ap = FN(tmpvec)(c, L, 1);
  const int grouped = FN(cg_groups)(c, L) > 1;
  /* LARGE coarsest grids (257^2 / 65^3 and up): the device runs the solve as many-unit launches whose dot products follow
   * the blocked order of the fine-level reductions with fused accumulation (csrc/evomg_krylov.h: krylov_cg_per_launch) */
  const int per_launch = !SC_IS_COMPLEX && ((c->dim == 2) ? lv->n >= 257 : lv->n >= 65);
#define CG_DOT(u, v) (per_launch ? FN(dot_blocked)(c, L, u, v, 1) : grouped ? FN(dot_canon_grouped)(c, L, u, v) : FN(dot_canon_for)(c, L, u, v, 1))
  FN(krylov_apply)(c, L, lv->st, x, b, r, 1);
  SC rr = CG_DOT(r, r);
  const double res0 = sc_norm_from_dot(rr);
  if (res0 == 0.0) return 0;
  /* p = r on the inner nodes; p keeps its homogeneous boundary (an error-type field) */
  FN(krylov_axpy)(c, L, p, sc_zero(), r, r);
  int it = 0;
  for (; it < c->prob.cgs_max_iterations; ++it) {
    FN(krylov_apply)(c, L, lv->st, p, NULL, ap, 0);
    const SC pap = CG_DOT(p, ap);
    const SC alpha = sc_div(rr, pap);
    FN(krylov_axpy)(c, L, x, alpha, p, x);
    FN(krylov_axpy)(c, L, r, sc_neg(alpha), ap, r);
    const SC rr_new = CG_DOT(r, r);
    const double res = sc_norm_from_dot(rr_new);
    if (!(res > c->prob.cgs_target_res_reduction * res0)) { ++it; break; }
    const SC beta = sc_div(rr_new, rr);
    FN(krylov_axpy)(c, L, p, beta, p, r);
    rr = rr_new;
  }
  FN(apply_bc)(c, L, x);
  if (getenv("EVOMG_ORACLE_DEBUG")) fprintf(stderr, "[oracle] coarse CG: %d iterations (cap %d)\n", it, c->prob.cgs_max_iterations);
#undef CG_DOT
  return it;
}

/* ---- K9a': CG in its single-reduction form (Chronopoulos & Gear) — what cgs_kind = EVOMG_CGS_CG runs on real 2D grids up to
 * 129^2 (csrc/evomg_cg1.h: evomg_cg1_band2d). Same configuration as the classical form (Poisson/2D_FD_Poisson_fromL2.exa3:12-14),
 * same stopping rule; both inner products of an iteration are needed at the same point of the recurrence:
 *     r0 = b - A x0, w0 = A r0, g = (r0, r0), d = (w0, r0), den = d, alpha = g / den, beta = 0, p = s = 0
 *     repeat:  p = r + beta p;  s = w + beta s;  x += alpha p;  r -= alpha s;  w = A r;  g' = (r, r);  d = (w, r)
 *              stop when sqrt(g') <= tol sqrt(g0) or at the iteration cap
 *              beta = g' (1 / g);  den = d - beta^2 den;  alpha = g' / den;  g = g'
 * Summation order = the device's band ownership (csrc/evomg_cg1.h: cg1_plan): band k owns the R rows k R .. k R + R - 1; a
 * wave holds H bands of C = 64 / H lanes, lane (h, c) of wave gw accumulates the E columns E c .. of band gw H + h (rows
 * ascending, elements ascending, fused multiply-add), then the wave64 tree, then the tree with offsets 16 .. 1 over the
 * (at most 32) wave sums. */
typedef struct { int R, E, H, waves; } FN(cg1_plan_t);
static FN(cg1_plan_t) FN(cg1_plan)(int n) {
  const int m = n - 2;
  FN(cg1_plan_t) p = {0, 0, 0, 0};
  if (m < 1 || m > 128) return p;
  p.E = m > 64 ? 2 : 1;
  p.H = m >= 32 ? 1 : m >= 16 ? 2 : m >= 8 ? 4 : m >= 4 ? 8 : 16;
  p.R = m >= 64 ? 4 : m >= 32 ? 16 : m >= 16 ? 4 : m >= 8 ? 2 : 1;
  const int bands = (m + p.R - 1) / p.R;
  p.waves = (bands + p.H - 1) / p.H;
  if (p.waves > 32) p.waves = 0;
  return p;
}
static int FN(cg_groups)(FN(ctx_t)* c, int level);
/* which coarsest grids the single-reduction form serves (csrc/impl_coarse.inl): the 2D level grids 3^2 .. 129^2 (bands,
 * evomg_cg1_band2d) and the 3D grid that runs as eight slabs, 33^3 (evomg_cg1_slab3d: the grouped summation order of the
 * classical form, dot_canon_grouped) */
static int FN(cg1_serves)(FN(ctx_t)* c, int level) {
  const int m = c->lv[level].n - 2;
  if (SC_IS_COMPLEX || c->prob.cgs_kind != EVOMG_CGS_CG) return 0;
  if (c->dim == 3) return FN(cg_groups)(c, level) > 1;
  return m == 1 || m == 3 || m == 7 || m == 15 || m == 31 || m == 63 || m == 127;
}
static SC FN(dot_bands)(FN(ctx_t)* c, int level, const SC* a, const SC* b) {
  const int n = c->lv[level].n, m = n - 2;
  const FN(cg1_plan_t) pl = FN(cg1_plan)(n);
  const int C = 64 / pl.H;
  SC waves[64];
  for (int w = 0; w < 64; ++w) waves[w] = sc_zero();
  for (int gw = 0; gw < pl.waves; ++gw) {
    SC lanes[64];
    for (int l = 0; l < 64; ++l) {
      SC s = sc_zero();
      for (int i = 0; i < pl.R; ++i)
        for (int e = 0; e < pl.E; ++e) {
          const int ix = pl.E * (l % C) + e, iy = (gw * pl.H + l / C) * pl.R + i;
          if (ix < m && iy < m) {
            const size_t idx = IDX2(n, ix + 1, iy + 1);
            s = sc_fma(a[idx], b[idx], s);
          }
        }
      lanes[l] = s;
    }
    waves[gw] = FN(tree64)(lanes);
  }
  /* the device's tree runs over 32 slots (offsets 16 .. 1): tree64's first step only adds the zeros of slots 32 .. 63 */
  return FN(tree64)(waves);
}
static int FN(coarse_cg1)(FN(ctx_t)* c) {
  const int L = c->prob.min_level;
  FN(level_t)* lv = &c->lv[L];
  const int n = lv->n;
  SC* x = FN(field)(c, L, EVOMG_F_CORR);
  const SC* b = FN(field)(c, L, EVOMG_F_CGS_RHS);
  SC* rout = FN(field)(c, L, EVOMG_F_RES);
  /* r, p, s, w are vectors of the solver with a homogeneous boundary (the device keeps them in registers): work arrays
   * whose boundary entries stay zero; the residual field receives the inner nodes of r at the end */
  SC* r = FN(tmpvec)(c, L, 0);
  SC* p = FN(tmpvec)(c, L, 1);
  SC* s = FN(tmpvec)(c, L, 2);
  SC* w = FN(tmpvec)(c, L, 3);
  memset(r, 0, lv->count * sizeof(SC));
  memset(p, 0, lv->count * sizeof(SC));
  memset(s, 0, lv->count * sizeof(SC));
  memset(w, 0, lv->count * sizeof(SC));
#define CG1_DOT(u, v) (c->dim == 3 ? FN(dot_canon_grouped)(c, L, u, v) : FN(dot_bands)(c, L, u, v))
  FN(krylov_apply)(c, L, lv->st, x, b, r, 1);
  FN(krylov_apply)(c, L, lv->st, r, NULL, w, 0);
  SC g = CG1_DOT(r, r);
  SC d = CG1_DOT(w, r);
  const double res0 = sc_norm_from_dot(g);
  int it = 0;
  if (res0 != 0.0 && c->prob.cgs_max_iterations > 0) {
    SC den = d, alpha = sc_div(g, d), beta = sc_zero();
    for (;;) {
      const SC inv_g = sc_div(sc_from_real(1.0), g);
      FN(krylov_axpy)(c, L, p, beta, p, r);                 /* p = beta p + r */
      FN(krylov_axpy)(c, L, s, beta, s, w);                 /* s = beta s + w */
      FN(krylov_axpy)(c, L, x, alpha, p, x);                /* x = alpha p + x */
      FN(krylov_axpy)(c, L, r, sc_neg(alpha), s, r);        /* r = -alpha s + r */
      FN(krylov_apply)(c, L, lv->st, r, NULL, w, 0);
      const SC gn = CG1_DOT(r, r);
      d = CG1_DOT(w, r);
      ++it;
      const double res = sc_norm_from_dot(gn);
      if (!(res > c->prob.cgs_target_res_reduction * res0) || it >= c->prob.cgs_max_iterations) break;
      beta = sc_mul(gn, inv_g);
      den = sc_fma(sc_neg(sc_mul(beta, beta)), den, d);
      alpha = sc_div(gn, den);
      g = gn;
    }
  }
#undef CG1_DOT
  if (c->dim == 2) {
    for (int j = 1; j < n - 1; ++j)
      for (int i = 1; i < n - 1; ++i) rout[IDX2(n, i, j)] = r[IDX2(n, i, j)];
  } else {
    for (int k = 1; k < n - 1; ++k)
      for (int j = 1; j < n - 1; ++j)
        for (int i = 1; i < n - 1; ++i) rout[IDX3(n, i, j, k)] = r[IDX3(n, i, j, k)];
  }
  FN(apply_bc)(c, L, x);
  if (getenv("EVOMG_ORACLE_DEBUG")) fprintf(stderr, "[oracle] coarse CG (single reduction): %d iterations (cap %d)\n", it, c->prob.cgs_max_iterations);
  return it;
}

/* ---- K9b: BiCGStab coarse-grid solver: literal restatement of
 * example_problems/Helmholtz/2D_FD_Helmholtz_fromL3.exa3:396-433 (the DSL text IS the spec). */
static int FN(coarse_bicgstab)(FN(ctx_t)* c) {
  const int L = c->prob.min_level;
  FN(level_t)* lv = &c->lv[L];
  SC* x = FN(field)(c, L, EVOMG_F_CORR);        /* gen_error_u */
  const SC* f = FN(field)(c, L, EVOMG_F_CGS_RHS); /* f@coarsest == gen_rhs_u */
  SC* r = FN(field)(c, L, EVOMG_F_RES);         /* gen_residual_u */
  SC* p = FN(tmpvec)(c, L, 0);
  SC* nu = FN(tmpvec)(c, L, 1);
  SC* hh = FN(tmpvec)(c, L, 2);
  SC* s = FN(tmpvec)(c, L, 3);
  SC* t = FN(tmpvec)(c, L, 4);
  SC* rhat = FN(tmpvec)(c, L, 5);
  const SC* st = lv->st;
  /* 129^2 and up (2D): the device runs the solve as many-unit launches whose dot products follow the blocked order of the
   * fine-level reductions with fused accumulation (csrc/evomg_krylov.h: krylov_bicgstab_per_launch) */
  const int per_launch = c->dim == 2 && lv->n >= 129;
#define B_DOT(u, v) (per_launch ? FN(dot_blocked)(c, L, u, v, 1) : FN(dot_canon)(c, L, u, v))
  FN(krylov_apply)(c, L, st, x, f, r, 1);
  const double res0 = sc_norm_from_dot(B_DOT(r, r));
  FN(apply_bc)(c, L, r);
  memcpy(rhat, r, lv->count * sizeof(SC));
  memset(nu, 0, lv->count * sizeof(SC));
  memset(p, 0, lv->count * sizeof(SC));
  if (res0 == 0.0) return 0;
  SC alpha = sc_from_real(1.0), beta, rho, rho_new = sc_from_real(1.0), omega = sc_from_real(1.0);
  int it = 0;
  for (; it < c->prob.cgs_max_iterations; ++it) {
    rho = rho_new;
    rho_new = B_DOT(rhat, r);
    beta = sc_mul(sc_div(rho_new, rho), sc_div(alpha, omega));
    /* p = r + beta * (p - omega * nu) ; boundary of p handled by applyBC_p@coarsest */
    FN(krylov_axpy)(c, L, p, sc_neg(omega), nu, p);
    FN(krylov_axpy)(c, L, p, beta, p, r);
    FN(apply_bc)(c, L, p);
    FN(krylov_apply)(c, L, st, p, NULL, nu, 0);
    alpha = sc_div(rho_new, B_DOT(rhat, nu));
    FN(krylov_axpy)(c, L, hh, alpha, p, x);
    FN(krylov_axpy)(c, L, s, sc_neg(alpha), nu, r);
    FN(apply_bc)(c, L, s);
    FN(krylov_apply)(c, L, st, s, NULL, t, 0);
    omega = sc_div(B_DOT(t, s), B_DOT(t, t));
    FN(krylov_axpy)(c, L, x, omega, s, hh);
    FN(apply_bc)(c, L, x);
    FN(krylov_axpy)(c, L, r, sc_neg(omega), t, s);
    const double res = sc_norm_from_dot(B_DOT(r, r));
    FN(apply_bc)(c, L, r);
    if (res < c->prob.cgs_target_res_reduction * res0) { ++it; break; }
    if (!(res == res)) { ++it; break; }  /* NaN guard: the DSL would spin to the iteration cap */
  }
  if (getenv("EVOMG_ORACLE_DEBUG")) fprintf(stderr, "[oracle] coarse BiCGStab: %d iterations (cap %d)\n", it, c->prob.cgs_max_iterations);
#undef B_DOT
  return it;
}

/* ---- op interpreter ----------------------------------------------------------------------------- */
static int FN(check_op)(FN(ctx_t)* c, const evomg_op_t* op) {
  const int lo = c->prob.min_level, hi = c->prob.max_level;
  if (op->level < lo || op->level > hi) return EVOMG_ERR_INVALID;
  if (op->dst < 0 || op->dst >= EVOMG_F_COUNT) return EVOMG_ERR_INVALID;
  switch (op->opcode) {
    case EVOMG_OP_SMOOTH: {
      if (op->src1 < 0 || op->src1 >= EVOMG_F_COUNT) return EVOMG_ERR_INVALID;
      int cells = 1;
      for (int d = 0; d < c->dim; ++d) { if (op->block[d] < 1) return EVOMG_ERR_INVALID; cells *= op->block[d]; }
      if (cells > EVOMG_MAX_BLOCK_CELLS) return EVOMG_ERR_INVALID;
      if (cells > 1 && op->partitioning != EVOMG_PART_SINGLE) return EVOMG_ERR_UNSUPPORTED;
      return EVOMG_OK;
    }
    case EVOMG_OP_RESIDUAL: case EVOMG_OP_RICHARDSON:
      if (op->src1 < 0 || op->src1 >= EVOMG_F_COUNT) return EVOMG_ERR_INVALID;
      if (op->opcode == EVOMG_OP_RESIDUAL && (op->src0 < 0 || op->src0 >= EVOMG_F_COUNT)) return EVOMG_ERR_INVALID;
      return EVOMG_OK;
    case EVOMG_OP_RESTRICT:
      if (op->level >= hi || op->src0 < 0 || op->src0 >= EVOMG_F_COUNT) return EVOMG_ERR_INVALID;
      return EVOMG_OK;
    case EVOMG_OP_PROLONG_ADD: case EVOMG_OP_PROLONG_SET:
      if (op->level <= lo || op->src0 < 0 || op->src0 >= EVOMG_F_COUNT) return EVOMG_ERR_INVALID;
      return EVOMG_OK;
    case EVOMG_OP_SET_ZERO: return EVOMG_OK;
    case EVOMG_OP_COPY:
      if (op->src0 < 0 || op->src0 >= EVOMG_F_COUNT) return EVOMG_ERR_INVALID;
      return EVOMG_OK;
    case EVOMG_OP_COARSE_SOLVE:
      if (op->level != lo) return EVOMG_ERR_INVALID;
      return EVOMG_OK;
    default: return EVOMG_ERR_INVALID;
  }
}

static int FN(exec_op)(FN(ctx_t)* c, const evomg_op_t* op) {
  const int L = op->level;
  switch (op->opcode) {
    case EVOMG_OP_SMOOTH: {
      const int cells = op->block[0] * op->block[1] * ((c->dim == 3) ? op->block[2] : 1);
      if (cells > 1) FN(smooth_block)(c, L, op->dst, op->src1, op->omega, op->block);
      else if (op->partitioning == EVOMG_PART_RED_BLACK) FN(smooth_rbgs)(c, L, op->dst, op->src1, op->omega);
      else FN(smooth_jacobi)(c, L, op->dst, op->src1, op->omega);
      return EVOMG_OK;
    }
    case EVOMG_OP_RESIDUAL:
      FN(residual_st)(c, L, c->lv[L].st, FN(field)(c, L, op->src0), FN(field)(c, L, op->src1), FN(field)(c, L, op->dst));
      return EVOMG_OK;
    case EVOMG_OP_RICHARDSON: {
      /* x += w * (f - A x): evaluated Jacobi-style from the old x */
      FN(level_t)* lv = &c->lv[L];
      SC* x = FN(field)(c, L, op->dst);
      SC* y = lv->alt;
      memcpy(y, x, lv->count * sizeof(SC));
      FN(residual_st)(c, L, lv->st, x, FN(field)(c, L, op->src1), y);   /* y(inner) = f - A x */
      FN(axpy_to)(c, L, y, x, sc_from_real(op->omega), y);
      FN(apply_bc)(c, L, y);
      lv->alt = x;
      FN(set_fieldptr)(c, L, op->dst, y);
      return EVOMG_OK;
    }
    case EVOMG_OP_RESTRICT:
      FN(restrict_fw)(c, L, FN(field)(c, L + 1, op->src0), FN(field)(c, L, op->dst));
      return EVOMG_OK;
    case EVOMG_OP_PROLONG_ADD:
      FN(prolong)(c, L, FN(field)(c, L - 1, op->src0), FN(field)(c, L, op->dst), 1, op->omega);
      return EVOMG_OK;
    case EVOMG_OP_PROLONG_SET:
      FN(prolong)(c, L, FN(field)(c, L - 1, op->src0), FN(field)(c, L, op->dst), 0, 1.0);
      return EVOMG_OK;
    case EVOMG_OP_SET_ZERO:
      FN(zero_inner)(c, L, FN(field)(c, L, op->dst));
      return EVOMG_OK;
    case EVOMG_OP_COPY: {
      SC* d = FN(field)(c, L, op->dst);
      const SC* s = FN(field)(c, L, op->src0);
      if (d != s) memcpy(d, s, c->lv[L].count * sizeof(SC));
      return EVOMG_OK;
    }
    case EVOMG_OP_COARSE_SOLVE:
      if (c->prob.cgs_kind == EVOMG_CGS_BICGSTAB) FN(coarse_bicgstab)(c);
      else if (FN(cg1_serves)(c, c->prob.min_level)) FN(coarse_cg1)(c);
      else FN(coarse_cg)(c);
      return EVOMG_OK;
  }
  return EVOMG_ERR_INVALID;
}

static int FN(run_ops)(FN(ctx_t)* c, const evomg_op_t* ops, int n_ops) {
  for (int i = 0; i < n_ops; ++i) {
    int rc = FN(check_op)(c, &ops[i]);
    if (rc != EVOMG_OK) { snprintf(c->err, sizeof c->err, "op %d (opcode %d) rejected: %d", i, ops[i].opcode, rc); return rc; }
  }
  for (int i = 0; i < n_ops; ++i) {
    int rc = FN(exec_op)(c, &ops[i]);
    if (rc != EVOMG_OK) return rc;
  }
  return EVOMG_OK;
}

static void FN(reset_fields)(FN(ctx_t)* c) {
  for (int L = c->prob.min_level; L <= c->prob.max_level; ++L) {
    FN(level_t)* lv = &c->lv[L];
    for (int f = 0; f < EVOMG_F_COUNT; ++f) {
      if (!lv->f[f]) continue;
      if (lv->init[f]) memcpy(lv->f[f], lv->init[f], lv->count * sizeof(SC));
      else memset(lv->f[f], 0, lv->count * sizeof(SC));
    }
  }
}

/* ---- outer loop A: "generate solver for u in solEq" [EXTERNAL: App. B.8]
 * residual -> initRes; repeat { cycle; residual; norm } until res <= target*initRes or maxNumIts. */
/* Residual norm of the solver loop on 2D levels with at least 33 nodes: the device fuses it into the residual sweep
 * (csrc/evomg_strip.h, STRIP_RESNORM / TAIL_NORM), which fixes the summation order: one partial sum per wave = (strip
 * of 120 stored columns, chunk of NORM_ROWS rows), lane l of the wave holds the column pair (c0, c0+1),
 * c0 = 120*strip - 4 + 2l (lanes 0, 1, 62 and 63 are halo lanes and contribute nothing) and adds r(c0,j)^2 then r(c0+1,j)^2 for the rows j of the
 * chunk in ascending order (unfused multiply + add); the 64 lane sums go through the wave tree; the wave partials
 * (wave = chunk * strips + strip) are then combined like the partials of FN(dot). */
#define NORM_STRIP_COLS 120
#define NORM_ROWS 16
static int FN(norm_is_fused)(FN(ctx_t)* c, int level) { return c->dim == 2 && c->lv[level].n >= 33; }
static double FN(norm_strip2d)(FN(ctx_t)* c, int level, const SC* r) {
  const int n = c->lv[level].n;
  const int strips = (n + NORM_STRIP_COLS - 1) / NORM_STRIP_COLS, chunks = (n + NORM_ROWS - 1) / NORM_ROWS;
  const int waves = strips * chunks;
  SC* partial = (SC*)malloc(sizeof(SC) * (size_t)waves);
#pragma omp parallel for schedule(static) if (OMP_WORTH(n))
  for (int wave = 0; wave < waves; ++wave) {
    const int strip = wave % strips, chunk = wave / strips;
    const int y0 = chunk * NORM_ROWS, y1 = (y0 + NORM_ROWS < n) ? y0 + NORM_ROWS : n;
    SC lane[64];
    for (int l = 0; l < 64; ++l) {
      SC acc = sc_zero();
      const int c0 = strip * NORM_STRIP_COLS - 4 + 2 * l;
      if (l >= 2 && l <= 61 && c0 >= 0 && c0 < n)
        for (int j = y0; j < y1; ++j) {
          if (j < 1 || j > n - 2) continue;
          for (int e = 0; e < 2; ++e) {
            const int i = c0 + e;
            if (i < 1 || i > n - 2) continue;
            const SC v = r[IDX2(n, i, j)];
            acc = sc_add(acc, sc_mul(v, v));
          }
        }
      lane[l] = acc;
    }
    partial[wave] = FN(tree64)(lane);
  }
  SC th[DOT_THREADS];
  for (int t = 0; t < DOT_THREADS; ++t) th[t] = sc_zero();
  for (int i = 0; i < waves; ++i) th[i % DOT_THREADS] = sc_add(th[i % DOT_THREADS], partial[i]);
  free(partial);
  return sc_norm_from_dot(FN(block_tree)(th));
}
/* Residual norm of the solver loop on 3D levels with at least 129 nodes per dimension: the device forms it in the
 * residual pass itself (csrc/evomg_tile3d.h, TILE_RESNORM). Order: one partial sum per (plane k, row j, strip s of P
 * stored columns, P even and the same for every strip: the n - 1 columns spread over ceil(n / 124) strips): lane l of the
 * wave holds the column pair (c0, c0+1), c0 = P s - 2 + 2 l (lanes 1 .. P/2 count, the others hold zero), adds r(c0)^2
 * then r(c0+1)^2 (inner columns only), the 64 lane values go through the wave tree; the partials,
 * indexed ((k-1) m + (j-1)) strips + s, are then summed by 1024 blocks of 256 threads striding over the array
 * (evomg_sum_stage1) and the block sums like the partials of FN(dot). */
#define TILE_STRIP_COLS 124
static int FN(norm_is_fused3d)(FN(ctx_t)* c, int level) {
  return c->dim == 3 && !SC_IS_COMPLEX && c->lv[level].n >= 129 && c->prob.bc_kind == EVOMG_BC_DIRICHLET;
}
static double FN(norm_tile3d)(FN(ctx_t)* c, int level, const SC* r) {
  const int n = c->lv[level].n, m = n - 2;
  const int strips = (n + TILE_STRIP_COLS - 1) / TILE_STRIP_COLS;
  /* the strips store equally many columns (even, at most TILE_STRIP_COLS): csrc/impl_launch3d.inl tile3d_xpitch */
  int xpitch = 2 * ((n - 1 + 2 * strips - 1) / (2 * strips));
  if (xpitch > TILE_STRIP_COLS) xpitch = TILE_STRIP_COLS;
  const long count = (long)m * m * strips;
  SC* part = (SC*)malloc(sizeof(SC) * (size_t)count);
#pragma omp parallel for schedule(static)
  for (int k = 1; k <= m; ++k)
    for (int j = 1; j <= m; ++j)
      for (int s = 0; s < strips; ++s) {
        SC lane[64];
        for (int l = 0; l < 64; ++l) {
          SC acc = sc_zero();
          const int c0 = s * xpitch - 2 + 2 * l;
          if (l >= 1 && 2 * l <= xpitch && c0 >= 0 && c0 < n)
            for (int e = 0; e < 2; ++e) {
              const int i = c0 + e;
              if (i < 1 || i > n - 2) continue;
              const SC v = r[IDX3(n, i, j, k)];
              acc = sc_add(acc, sc_mul(v, v));
            }
          lane[l] = acc;
        }
        part[((long)(k - 1) * m + (j - 1)) * strips + s] = FN(tree64)(lane);
      }
  SC block[DOT_BLOCKS];
#pragma omp parallel for schedule(static)
  for (int b = 0; b < DOT_BLOCKS; ++b) {
    SC th[DOT_THREADS];
    for (int t = 0; t < DOT_THREADS; ++t) {
      SC acc = sc_zero();
      for (long i = (long)b * DOT_THREADS + t; i < count; i += (long)DOT_BLOCKS * DOT_THREADS) acc = sc_add(acc, part[i]);
      th[t] = acc;
    }
    block[b] = FN(block_tree)(th);
  }
  free(part);
  SC th[DOT_THREADS];
  for (int t = 0; t < DOT_THREADS; ++t) th[t] = sc_zero();
  for (int i = 0; i < DOT_BLOCKS; ++i) th[i % DOT_THREADS] = sc_add(th[i % DOT_THREADS], block[i]);
  return sc_norm_from_dot(FN(block_tree)(th));
}
static double FN(solver_res_norm)(FN(ctx_t)* c, int level, const SC* r) {
  if (FN(norm_is_fused3d)(c, level)) return FN(norm_tile3d)(c, level, r);
  return FN(norm_is_fused)(c, level) ? FN(norm_strip2d)(c, level, r) : FN(norm)(c, level, r);
}

static void FN(solve_mg)(FN(ctx_t)* c, const evomg_op_t* ops, int n_ops, double wall_limit_ms,
                         evomg_result_t* out, double* hist, int cap) {
  const int L = c->prob.max_level;
  FN(level_t)* lv = &c->lv[L];
  int nh = 0;
  const double t0 = now_ms();
  FN(residual_st)(c, L, lv->st, FN(field)(c, L, EVOMG_F_SOL), FN(field)(c, L, EVOMG_F_RHS), FN(field)(c, L, EVOMG_F_RES));
  const double res0 = FN(solver_res_norm)(c, L, FN(field)(c, L, EVOMG_F_RES));
  if (hist && nh < cap) hist[nh] = res0;
  ++nh;
  double res = res0;
  int it = 0, status = EVOMG_SOLVE_MAXITER;
  if (res0 == 0.0) status = EVOMG_SOLVE_CONVERGED;
  else
    for (it = 1; it <= c->prob.max_iterations; ++it) {
      for (int i = 0; i < n_ops; ++i) FN(exec_op)(c, &ops[i]);
      FN(residual_st)(c, L, lv->st, FN(field)(c, L, EVOMG_F_SOL), FN(field)(c, L, EVOMG_F_RHS), FN(field)(c, L, EVOMG_F_RES));
      res = FN(solver_res_norm)(c, L, FN(field)(c, L, EVOMG_F_RES));
      if (hist && nh < cap) hist[nh] = res;
      ++nh;
      if (!(res == res) || res > 1.7e308) { status = EVOMG_SOLVE_DIVERGED; break; }
      if (res <= c->prob.target_res_reduction * res0) { status = EVOMG_SOLVE_CONVERGED; break; }
      if (wall_limit_ms > 0 && now_ms() - t0 > wall_limit_ms) { status = EVOMG_SOLVE_WALL_LIMIT; break; }
    }
  if (it > c->prob.max_iterations) it = c->prob.max_iterations;
  out->time_ms = now_ms() - t0;
  out->n_iterations = (double)it;
  out->init_res_norm = res0;
  out->final_res_norm = res;
  out->status = status;
  out->n_history = nh < cap || !hist ? nh : cap;
  out->conv_factor = (it > 0 && res0 > 0) ? pow(res / res0, 1.0 / (double)it) : 0.0;
}

/* ---- outer loop B: MG-preconditioned BiCGStab, literal restatement of
 * example_problems/Helmholtz/2D_FD_Helmholtz_fromL3.exa3:144-200. Field roles:
 * Solution -> tmp[0]@max, Residual -> RES@max, p/ap/s/t/h/resHat -> tmp[..]@max... the MG cycle
 * works on u (= SOL@max) and f (= RHS@max). RHS (the PDE right-hand side) is init[RHS]. */
static void FN(solve_pbicgstab)(FN(ctx_t)* c, const evomg_op_t* ops, int n_ops, double wall_limit_ms,
                                evomg_result_t* out, double* hist, int cap) {
  const int L = c->prob.max_level;
  FN(level_t)* lv = &c->lv[L];
  const SC* A = c->outer_st;
  const size_t bytes = lv->count * sizeof(SC);
  SC* Sol = FN(zalloc)(lv->count);
  SC* Res = FN(zalloc)(lv->count);
  SC* p = FN(zalloc)(lv->count);
  SC* ap = FN(zalloc)(lv->count);
  SC* s = FN(zalloc)(lv->count);
  SC* t = FN(zalloc)(lv->count);
  SC* hh = FN(zalloc)(lv->count);
  SC* rhat = FN(zalloc)(lv->count);
  SC* RHS = lv->init[EVOMG_F_RHS];
  int nh = 0;
  const double t0 = now_ms();
  if (lv->init[EVOMG_F_SOL]) memcpy(Sol, lv->init[EVOMG_F_SOL], bytes);
  FN(apply_bc)(c, L, Sol);
  /* Residual = RHS - A * Solution   (no BC declared on Residual@finest: exa3:26) */
  {
    const int n = lv->n;
#pragma omp parallel for schedule(static) if (OMP_WORTH(n))
    for (int j = 1; j < n - 1; ++j)
      for (int i = 1; i < n - 1; ++i)
        Res[IDX2(n, i, j)] = sc_sub(RHS[IDX2(n, i, j)], FN(apply_st2)(A, Sol, n, i, j));
  }
  const double res0 = FN(norm)(c, L, Res);
  if (hist && nh < cap) hist[nh] = res0;
  ++nh;
  double res = res0;
  int it = 0, status = EVOMG_SOLVE_MAXITER;
  SC alpha = sc_from_real(1.0), beta, rho, rho_new = sc_from_real(1.0), omega = sc_from_real(1.0);
  memcpy(rhat, Res, bytes);
  if (res0 == 0.0) status = EVOMG_SOLVE_CONVERGED;
  else
    for (it = 1; it <= c->prob.max_iterations; ++it) {
      rho = rho_new;
      rho_new = FN(dot)(c, L, rhat, Res);
      beta = sc_mul(sc_div(rho_new, rho), sc_div(alpha, omega));
      FN(axpy_to)(c, L, p, p, sc_neg(omega), ap);
      FN(axpy_to)(c, L, p, Res, beta, p);
      /* u = 0 ; f = p ; gen_mgCycle() */
      SC* u = FN(field)(c, L, EVOMG_F_SOL);
      memset(u, 0, bytes);
      memcpy(FN(field)(c, L, EVOMG_F_RHS), p, bytes);
      for (int i = 0; i < n_ops; ++i) FN(exec_op)(c, &ops[i]);
      u = FN(field)(c, L, EVOMG_F_SOL);
      FN(apply_op)(c, L, A, u, ap);
      alpha = sc_div(rho_new, FN(dot)(c, L, rhat, ap));
      FN(axpy_to)(c, L, hh, Sol, alpha, u);
      FN(axpy_to)(c, L, s, Res, sc_neg(alpha), ap);
      /* u = 0 ; f = s ; gen_mgCycle() */
      memset(u, 0, bytes);
      memcpy(FN(field)(c, L, EVOMG_F_RHS), s, bytes);
      for (int i = 0; i < n_ops; ++i) FN(exec_op)(c, &ops[i]);
      u = FN(field)(c, L, EVOMG_F_SOL);
      FN(apply_op)(c, L, A, u, t);
      omega = sc_div(FN(dot)(c, L, t, s), FN(dot)(c, L, t, t));
      FN(axpy_to)(c, L, Sol, hh, omega, u);
      FN(apply_bc)(c, L, Sol);
      FN(axpy_to)(c, L, Res, s, sc_neg(omega), t);
      res = FN(norm)(c, L, Res);
      if (hist && nh < cap) hist[nh] = res;
      ++nh;
      if (!(res == res) || res > 1.7e308) { status = EVOMG_SOLVE_DIVERGED; break; }
      if (res < c->prob.target_res_reduction * res0) { status = EVOMG_SOLVE_CONVERGED; break; }
      if (wall_limit_ms > 0 && now_ms() - t0 > wall_limit_ms) { status = EVOMG_SOLVE_WALL_LIMIT; break; }
    }
  if (it > c->prob.max_iterations) it = c->prob.max_iterations;
  out->time_ms = now_ms() - t0;
  /* keep the solution readable through CORR@max (otherwise unused on the finest level) */
  memcpy(lv->f[EVOMG_F_CORR], Sol, bytes);
  memcpy(lv->f[EVOMG_F_RES], Res, bytes);
  out->n_iterations = (double)it;
  out->init_res_norm = res0;
  out->final_res_norm = res;
  out->status = status;
  out->n_history = nh < cap || !hist ? nh : cap;
  /* the DSL prints ONE line: convergence factor = |res| / |initRes|  (exa3:193) */
  out->conv_factor = (res0 > 0) ? res / res0 : 0.0;
  free(Sol); free(Res); free(p); free(ap); free(s); free(t); free(hh); free(rhat);
}

static int FN(create)(const evomg_problem_t* pr, FN(ctx_t)** outc) {
  FN(ctx_t)* c = (FN(ctx_t)*)calloc(1, sizeof(FN(ctx_t)));
  if (!c) return EVOMG_ERR_DEVICE;
  c->prob = *pr;
  c->dim = pr->dimension;
  for (int e = 0; e < EVOMG_STENCIL_ENTRIES; ++e) c->outer_st[e] = sc_from_c64(pr->outer_stencil[e]);
  for (int L = pr->min_level; L <= pr->max_level; ++L) {
    FN(level_t)* lv = &c->lv[L];
    lv->L = L;
    lv->n = (1 << L) + 1;
    lv->count = (size_t)lv->n * (size_t)lv->n * (size_t)(c->dim == 3 ? lv->n : 1);
    for (int e = 0; e < EVOMG_STENCIL_ENTRIES; ++e) lv->st[e] = sc_from_c64(pr->mg_stencil[L][e]);
    lv->f[EVOMG_F_SOL] = FN(zalloc)(lv->count);
    lv->f[EVOMG_F_RHS] = FN(zalloc)(lv->count);
    lv->f[EVOMG_F_RES] = FN(zalloc)(lv->count);
    if (L == pr->max_level) lv->f[EVOMG_F_CORR] = FN(zalloc)(lv->count);
    if (L == pr->min_level) lv->f[EVOMG_F_CGS_RHS] = FN(zalloc)(lv->count);
    lv->alt = FN(zalloc)(lv->count);
  }
  *outc = c;
  return EVOMG_OK;
}

static void FN(destroy)(FN(ctx_t)* c) {
  for (int L = c->prob.min_level; L <= c->prob.max_level; ++L) {
    FN(level_t)* lv = &c->lv[L];
    for (int f = 0; f < EVOMG_F_COUNT; ++f) { free(lv->f[f]); free(lv->init[f]); }
    free(lv->alt);
    for (int t = 0; t < 6; ++t) free(lv->tmp[t]);
  }
  free(c);
}
