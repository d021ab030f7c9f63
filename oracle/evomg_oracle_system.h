/*
 * evomg_oracle_system.h — CPU ORACLE for systems of PDEs. TEST INFRASTRUCTURE ONLY, included by
 * evomg_oracle.c.
 *
 * Restates what the ExaStencils-generated executable computes for the statements the reference's lowering emits on a
 * problem with several unknowns per node (evostencils/code_generation/exastencils.py:769-873 with
 * evostencils/ir/transformations.py:51-145; problem: example_problems/LinearElasticity/2D_FD_LinearElasticity_fromL2.*):
 * 2D, real, Dirichlet, n_unknowns node fields coupled by 9-point stencils A[equation][unknown].
 *
 * PINNING: the statement text our lowering prints for these problems is pinned character for character against the
 * reference's own lowering (tests/golden/elasticity2d_*.exa3); the arithmetic below has no recorded reference output
 * to be pinned against (the reference records a run of the scalar Poisson problem only) and rests on the conventions
 * that run pinned for scalar problems (oracle/APPB_MATRIX.md) plus known-answer tests (tests/test_systems_oracle.py).
 *
 * Conventions specific to systems:
 *   * a `solve locally` block lists unknowns (u and / or v at the cells of a block); every stencil entry that hits a
 *     listed unknown goes into the local matrix, everything else to the right-hand side at its current value
 *     (SURVEY App. B.5); unknown order = field order, then cells with i0 outermost (the order the reference prints);
 *     dense solve = Gaussian elimination without pivoting in that order, skipping exactly-zero multipliers;
 *   * `color with` sweeps on a 9-point coupling are ORDER DEPENDENT in the reference (same-colour corner neighbours,
 *     updated in place in loop order, in parallel under OpenMP): [EXTERNAL], not reproducible. Here every colour pass
 *     is evaluated from the state before the pass (all nodes of the colour at once) — identical to the in-place form
 *     wherever that is well defined (decoupled sweeps, 5-point couplings);
 *   * `with jacobi` sweeps follow evomg_problem_t.jacobi_semantics exactly as for scalar problems;
 *   * coarse solver: CG on the stacked unknowns; residual norm of the solver loop: sqrt(sum over all unknowns).
 */

#define SYS_NF EVOMG_MAX_UNKNOWNS
#define SYS_SLOTS (EVOMG_F_COUNT * SYS_NF)
static const int sys_ox[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
static const int sys_oy[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};

typedef struct {
  int n;
  size_t count;
  double* f[SYS_SLOTS];
  double* init[SYS_SLOTS];
  double* alt[SYS_NF];
  double* tmp[2 * SYS_NF];      /* CG: p, ap per unknown */
  double st[SYS_NF][SYS_NF][9];
} sys_level_t;

typedef struct {
  evomg_problem_t prob;
  int nf;
  sys_level_t lv[EVOMG_MAX_LEVELS];
  char err[256];
} sys_ctx_t;

static int sys_canon(const sys_ctx_t* c, int level, int field) {
  if (field < 0 || field >= EVOMG_F_COUNT * c->nf) return -1;
  const int slot = field % EVOMG_F_COUNT, u = field / EVOMG_F_COUNT;
  /* gen_error_<f>@L and the solution of level L < max are the same array (exastencils.py:296-301) */
  if (slot == EVOMG_F_CORR && level < c->prob.max_level) return EVOMG_F_SOL + EVOMG_F_COUNT * u;
  return field;
}
static double* sys_field(sys_ctx_t* c, int level, int field) {
  const int k = sys_canon(c, level, field);
  return k < 0 ? NULL : c->lv[level].f[k];
}

static int sys_create(const evomg_problem_t* pr, sys_ctx_t** out) {
  sys_ctx_t* c = (sys_ctx_t*)calloc(1, sizeof *c);
  if (!c) return EVOMG_ERR_DEVICE;
  c->prob = *pr;
  c->nf = pr->n_unknowns;
  for (int L = pr->min_level; L <= pr->max_level; ++L) {
    sys_level_t* lv = &c->lv[L];
    lv->n = (1 << L) + 1;
    lv->count = (size_t)lv->n * lv->n;
    memcpy(lv->st, pr->sys_stencil[L], sizeof lv->st);
    for (int u = 0; u < c->nf; ++u) {
      const int b = EVOMG_F_COUNT * u;
      lv->f[b + EVOMG_F_SOL] = (double*)calloc(lv->count, sizeof(double));
      lv->f[b + EVOMG_F_RHS] = (double*)calloc(lv->count, sizeof(double));
      lv->f[b + EVOMG_F_RES] = (double*)calloc(lv->count, sizeof(double));
      if (L == pr->max_level) lv->f[b + EVOMG_F_CORR] = (double*)calloc(lv->count, sizeof(double));
      if (L == pr->min_level) lv->f[b + EVOMG_F_CGS_RHS] = (double*)calloc(lv->count, sizeof(double));
      lv->alt[u] = (double*)calloc(lv->count, sizeof(double));
    }
  }
  *out = c;
  return EVOMG_OK;
}
static void sys_destroy(sys_ctx_t* c) {
  for (int L = c->prob.min_level; L <= c->prob.max_level; ++L) {
    sys_level_t* lv = &c->lv[L];
    for (int k = 0; k < SYS_SLOTS; ++k) { free(lv->f[k]); free(lv->init[k]); }
    for (int u = 0; u < SYS_NF; ++u) free(lv->alt[u]);
    for (int t = 0; t < 2 * SYS_NF; ++t) free(lv->tmp[t]);
  }
  free(c);
}

/* ---- row i of the block operator at an inner node: sum_j sum_e A[i][j][e] * x_j(node + offset_e) ---- */
static inline double sys_apply_row(const sys_level_t* lv, int nf, int i, double* const* x, int a, int b) {
  const int n = lv->n;
  double s = 0.0;
  for (int j = 0; j < nf; ++j)
    for (int e = 0; e < 9; ++e)
      s = s + lv->st[i][j][e] * x[j][IDX2(n, a + sys_ox[e], b + sys_oy[e])];
  return s;
}

static void sys_residual(sys_ctx_t* c, int level, int i, double* const* x, const double* f, double* r) {
  const sys_level_t* lv = &c->lv[level];
  const int n = lv->n;
#pragma omp parallel for schedule(static) if (n >= 257)
  for (int b = 1; b < n - 1; ++b)
    for (int a = 1; a < n - 1; ++a) r[IDX2(n, a, b)] = f[IDX2(n, a, b)] - sys_apply_row(lv, c->nf, i, x, a, b);
}

/* ---- one `solve locally` block on the tile anchored at (a0, b0) with extents (wx, wy): unknown e = (unknown u of
 * the mask, cell (ci, cj)), field order, then cells with i0 outermost. x: current values; out[e]: relaxed values. ---- */
static void sys_local_solve(const sys_level_t* lv, int nf, int mask, double* const* x, double* const* f, int a0, int b0,
                            const int (*shape)[2], double om, double* out) {
  const int n = lv->n;
  int eu[8], ea[8], eb[8], m = 0;
  for (int u = 0; u < nf; ++u) {
    if (!((mask >> u) & 1)) continue;
    /* the cells of unknown u: its own block shape (exastencils.py:778-822 prints one equation per (unknown, cell of the
     * unknown's block), ir/transformations.py:51-121), cut at the far boundary */
    const int wx = (a0 + shape[u][0] <= n - 1) ? shape[u][0] : n - 1 - a0, wy = (b0 + shape[u][1] <= n - 1) ? shape[u][1] : n - 1 - b0;
    for (int ci = 0; ci < wx; ++ci)
      for (int cj = 0; cj < wy; ++cj) { eu[m] = u; ea[m] = a0 + ci; eb[m] = b0 + cj; ++m; }
  }
  double M[8][8], r[8];
  for (int p = 0; p < m; ++p) {
    for (int q = 0; q < m; ++q) M[p][q] = 0.0;
    const int i = eu[p];
    double s = f[i][IDX2(n, ea[p], eb[p])];
    for (int j = 0; j < nf; ++j)
      for (int e = 0; e < 9; ++e) {
        const int ta = ea[p] + sys_ox[e], tb = eb[p] + sys_oy[e];
        int hit = -1;
        if ((mask >> j) & 1)
          for (int q = 0; q < m; ++q)
            if (eu[q] == j && ea[q] == ta && eb[q] == tb) hit = q;
        if (hit >= 0) M[p][hit] = lv->st[i][j][e];
        else s = s - lv->st[i][j][e] * x[j][IDX2(n, ta, tb)];
      }
    r[p] = s;
  }
  for (int p = 0; p < m; ++p)
    for (int q = p + 1; q < m; ++q) {
      if (M[q][p] == 0.0) continue;
      const double fac = M[q][p] / M[p][p];
      for (int t = p + 1; t < m; ++t) M[q][t] = M[q][t] - fac * M[p][t];
      r[q] = r[q] - fac * r[p];
    }
  for (int p = m - 1; p >= 0; --p) {
    double s = r[p];
    for (int t = p + 1; t < m; ++t) s = s - M[p][t] * r[t];
    r[p] = s / M[p][p];
  }
  const double omm = 1.0 - om;
  for (int p = 0; p < m; ++p) out[p] = omm * x[eu[p]][IDX2(n, ea[p], eb[p])] + om * r[p];
}

/* block shape of the q-th LISTED unknown of a SMOOTH (include/evomg.h: block, block_other, block_third) */
static inline void sys_listed_shape(const evomg_op_t* op, int q, int* sx, int* sy) {
  *sx = op->block[0];
  *sy = op->block[1];
  if (q >= 1 && op->block_other[0] > 0) { *sx = op->block_other[0]; *sy = op->block_other[1]; }
  if (q >= 2 && op->block_third != 0) { *sx = op->block_third & 0xff; *sy = (op->block_third >> 8) & 0xff; }
}
/* the unknowns a SMOOTH lists; a scalar op program (mask 0) on a system of ONE unknown (EVOMG_STENCIL_COMPACT) means that one */
#define SYS_MASK_OF(c, op) ((op)->unknowns ? (op)->unknowns : ((c)->nf == 1 ? 1 : 0))
/* SMOOTH on a system. Single partitioning (`with jacobi`): non-overlapping tiles, all from the old state, next slot;
 * red-black: point blocks, one pass per colour, each from the state before the pass (header). */
static int sys_smooth(sys_ctx_t* c, const evomg_op_t* op) {
  sys_level_t* lv = &c->lv[op->level];
  const int n = lv->n, nf = c->nf, mask = SYS_MASK_OF(c, op), inner = n - 2;
  /* block shape per unknown of the mask: `block` for the first, `block_other` (where set) for the second; tiles repeat with
   * the bounding box of the statement's unknowns [EXTERNAL: the loop ExaStencils generates around a `solve locally`
   * is not visible from the reference; with equal shapes this is the IR's non-overlapping tiling, stencils/multiple.py:204-217] */
  int shape[SYS_NF][2];
  int bx = 1, by = 1, seen = 0;
  for (int u = 0; u < nf; ++u) {
    shape[u][0] = shape[u][1] = 0;
    if (!((mask >> u) & 1)) continue;
    sys_listed_shape(op, seen++, &shape[u][0], &shape[u][1]);
    if (shape[u][0] > bx) bx = shape[u][0];
    if (shape[u][1] > by) by = shape[u][1];
  }
  const int sol_slot = sys_canon(c, op->level, op->dst) % EVOMG_F_COUNT;
  const int rhs_slot = op->src1 % EVOMG_F_COUNT;
  double* x[SYS_NF];
  double* f[SYS_NF];
  for (int u = 0; u < nf; ++u) {
    x[u] = sys_field(c, op->level, EVOMG_FIELD(sol_slot, u));
    f[u] = sys_field(c, op->level, EVOMG_FIELD(rhs_slot, u));
  }
  const int passes = (op->partitioning == EVOMG_PART_RED_BLACK) ? 2 : 1;
  for (int pass = 0; pass < passes; ++pass) {
    for (int u = 0; u < nf; ++u)
      if ((mask >> u) & 1) memcpy(lv->alt[u], x[u], lv->count * sizeof(double));
    const int tx = (inner + bx - 1) / bx, ty = (inner + by - 1) / by;
#pragma omp parallel for schedule(static) if (n >= 257)
    for (int tj = 0; tj < ty; ++tj)
      for (int ti = 0; ti < tx; ++ti) {
        const int a0 = 1 + ti * bx, b0 = 1 + tj * by;
        if (passes == 2 && ((a0 + b0) & 1) != pass) continue;
        double out[8];
        sys_local_solve(lv, nf, mask, x, f, a0, b0, (const int (*)[2])shape, op->omega, out);
        int p = 0;
        for (int u = 0; u < nf; ++u) {
          if (!((mask >> u) & 1)) continue;
          const int wx = (a0 + shape[u][0] <= n - 1) ? shape[u][0] : n - 1 - a0, wy = (b0 + shape[u][1] <= n - 1) ? shape[u][1] : n - 1 - b0;
          for (int ci = 0; ci < wx; ++ci)
            for (int cj = 0; cj < wy; ++cj) lv->alt[u][IDX2(n, a0 + ci, b0 + cj)] = out[p++];
        }
      }
    /* a colour pass always takes effect (in place in the reference); a `with jacobi` sweep only if the slot advances */
    const int lost = passes == 1 && c->prob.jacobi_semantics == EVOMG_JACOBI_EXASTENCILS_V11;
    if (!lost)
      for (int u = 0; u < nf; ++u)
        if ((mask >> u) & 1) {
          const int k = sys_canon(c, op->level, EVOMG_FIELD(sol_slot, u));
          double* t = lv->f[k];
          lv->f[k] = lv->alt[u];
          lv->alt[u] = t;
          x[u] = lv->f[k];
        }
  }
  return EVOMG_OK;
}

static void sys_restrict(sys_ctx_t* c, int coarse_level, const double* fine, double* coarse) {
  const int nc = c->lv[coarse_level].n, nfine = c->lv[coarse_level + 1].n;
  for (int J = 1; J < nc - 1; ++J)
    for (int I = 1; I < nc - 1; ++I) {
      const int a = 2 * I, b = 2 * J;
      /* same operation order as the scalar oracle's FN(restrict_fw): centre, edges, corners */
      double e4 = (fine[IDX2(nfine, a - 1, b)] + fine[IDX2(nfine, a + 1, b)]) + (fine[IDX2(nfine, a, b - 1)] + fine[IDX2(nfine, a, b + 1)]);
      double q4 = (fine[IDX2(nfine, a - 1, b - 1)] + fine[IDX2(nfine, a + 1, b - 1)]) + (fine[IDX2(nfine, a - 1, b + 1)] + fine[IDX2(nfine, a + 1, b + 1)]);
      double v = 0.25 * fine[IDX2(nfine, a, b)];
      v = v + 0.125 * e4;
      v = v + 0.0625 * q4;
      coarse[IDX2(nc, I, J)] = v;
    }
}
static void sys_prolong(sys_ctx_t* c, int fine_level, const double* coarse, double* fine, int accumulate, double om) {
  const int nfine = c->lv[fine_level].n, nc = c->lv[fine_level - 1].n;
  for (int b = 1; b < nfine - 1; ++b)
    for (int a = 1; a < nfine - 1; ++a) {
      const int I = a >> 1, J = b >> 1;
      double v;
      if (!(a & 1) && !(b & 1)) v = coarse[IDX2(nc, I, J)];
      else if ((a & 1) && !(b & 1)) v = 0.5 * (coarse[IDX2(nc, I, J)] + coarse[IDX2(nc, I + 1, J)]);
      else if (!(a & 1) && (b & 1)) v = 0.5 * (coarse[IDX2(nc, I, J)] + coarse[IDX2(nc, I, J + 1)]);
      else v = 0.25 * ((coarse[IDX2(nc, I, J)] + coarse[IDX2(nc, I + 1, J)]) + (coarse[IDX2(nc, I, J + 1)] + coarse[IDX2(nc, I + 1, J + 1)]));
      if (accumulate) fine[IDX2(nfine, a, b)] = fine[IDX2(nfine, a, b)] + om * v;
      else fine[IDX2(nfine, a, b)] = v;
    }
}

/* stacked dot product in the order of the device's single-workgroup solver (csrc/evomg_system.h evomg_sys_cg): 1024
 * lanes stride over the stacked inner nodes (unknown-major, rows, columns), wave64 trees, then a tree over the 16 wave
 * sums */
static double sys_dot(sys_ctx_t* c, int level, double* const* a, double* const* b) {
  const int n = c->lv[level].n, m = n - 2;
  const long per = (long)m * m, total = per * c->nf;
  double lane[1024];
  for (int t = 0; t < 1024; ++t) {
    double s = 0.0;
    for (long g = t; g < total; g += 1024) {
      const int u = (int)(g / per);
      const long q = g % per;
      const size_t idx = IDX2(n, 1 + (int)(q % m), 1 + (int)(q / m));
      s = fma(a[u][idx], b[u][idx], s);
    }
    lane[t] = s;
  }
  double waves[64];
  for (int w = 0; w < 64; ++w) waves[w] = 0.0;
  for (int w = 0; w < 16; ++w) waves[w] = tree64_r(&lane[64 * w]);
  return tree64_r(waves);
}

static void sys_coarse_cg(sys_ctx_t* c) {
  const int L = c->prob.min_level, nf = c->nf;
  sys_level_t* lv = &c->lv[L];
  const int n = lv->n;
  double *x[SYS_NF], *b[SYS_NF], *r[SYS_NF], *p[SYS_NF], *ap[SYS_NF];
  for (int u = 0; u < nf; ++u) {
    x[u] = sys_field(c, L, EVOMG_FIELD(EVOMG_F_CORR, u));
    b[u] = sys_field(c, L, EVOMG_FIELD(EVOMG_F_CGS_RHS, u));
    r[u] = sys_field(c, L, EVOMG_FIELD(EVOMG_F_RES, u));
    for (int t = 0; t < 2; ++t)
      if (!lv->tmp[2 * u + t]) lv->tmp[2 * u + t] = (double*)calloc(lv->count, sizeof(double));
    p[u] = lv->tmp[2 * u];
    ap[u] = lv->tmp[2 * u + 1];
  }
  for (int u = 0; u < nf; ++u)
    for (int bb = 1; bb < n - 1; ++bb)
      for (int a = 1; a < n - 1; ++a) {
        const size_t i = IDX2(n, a, bb);
        r[u][i] = b[u][i] - sys_apply_row(lv, nf, u, x, a, bb);
        p[u][i] = r[u][i];
      }
  double rr = sys_dot(c, L, r, r);
  const double res0 = sqrt(rr);
  if (res0 == 0.0) return;
  for (int it = 0; it < c->prob.cgs_max_iterations; ++it) {
    for (int u = 0; u < nf; ++u)
      for (int bb = 1; bb < n - 1; ++bb)
        for (int a = 1; a < n - 1; ++a) ap[u][IDX2(n, a, bb)] = sys_apply_row(lv, nf, u, p, a, bb);
    const double alpha = rr / sys_dot(c, L, p, ap);
    for (int u = 0; u < nf; ++u)
      for (int bb = 1; bb < n - 1; ++bb)
        for (int a = 1; a < n - 1; ++a) {
          const size_t i = IDX2(n, a, bb);
          x[u][i] = fma(alpha, p[u][i], x[u][i]);
          r[u][i] = fma(-alpha, ap[u][i], r[u][i]);
        }
    const double rrn = sys_dot(c, L, r, r);
    if (!(sqrt(rrn) > c->prob.cgs_target_res_reduction * res0)) break;
    const double beta = rrn / rr;
    for (int u = 0; u < nf; ++u)
      for (int bb = 1; bb < n - 1; ++bb)
        for (int a = 1; a < n - 1; ++a) {
          const size_t i = IDX2(n, a, bb);
          p[u][i] = fma(beta, p[u][i], r[u][i]);
        }
    rr = rrn;
  }
}

static int sys_check_op(sys_ctx_t* c, const evomg_op_t* op) {
  const int lo = c->prob.min_level, hi = c->prob.max_level;
  if (op->level < lo || op->level > hi || !sys_field(c, op->level, op->dst)) return EVOMG_ERR_INVALID;
  switch (op->opcode) {
    case EVOMG_OP_SMOOTH: {
      if (!sys_field(c, op->level, op->src1)) return EVOMG_ERR_INVALID;
      const int mask = SYS_MASK_OF(c, op);
      if (mask <= 0 || mask >= (1 << c->nf)) return EVOMG_ERR_INVALID;
      if (op->block[0] < 1 || op->block[1] < 1) return EVOMG_ERR_INVALID;
      int k = 0;
      for (int u = 0; u < c->nf; ++u) k += (mask >> u) & 1;
      const int other = op->block_other[0] != 0 || op->block_other[1] != 0;
      if (other && (k < 2 || op->block_other[0] < 1 || op->block_other[1] < 1)) return EVOMG_ERR_INVALID;
      if (op->block_third != 0 && (k < 3 || !other || (op->block_third & 0xff) < 1 || ((op->block_third >> 8) & 0xff) < 1)) return EVOMG_ERR_INVALID;
      int cells = 0, differ = 0;
      for (int q = 0; q < k; ++q) {
        int sx, sy;
        sys_listed_shape(op, q, &sx, &sy);
        cells += sx * sy;
        differ = differ || sx != op->block[0] || sy != op->block[1];
      }
      if (cells > EVOMG_MAX_BLOCK_CELLS) return EVOMG_ERR_INVALID;
      if (op->partitioning == EVOMG_PART_RED_BLACK && cells != k) return EVOMG_ERR_UNSUPPORTED;
      /* differing per-unknown shapes: the tiling around such a `solve locally` is [EXTERNAL]; accepted only where the sweep's
       * result is never used (EVOMG_JACOBI_EXASTENCILS_V11), rejected where it would be (see csrc/impl_system.hip) */
      if (other && differ && c->prob.jacobi_semantics == EVOMG_JACOBI_TEXTBOOK) return EVOMG_ERR_UNSUPPORTED;
      if (op->partitioning != EVOMG_PART_SINGLE && op->partitioning != EVOMG_PART_RED_BLACK) return EVOMG_ERR_INVALID;
      /* dst / src1 name the slots of the first unknown of the mask */
      int first = 0;
      while (!((mask >> first) & 1)) ++first;
      if (op->dst / EVOMG_F_COUNT != first || op->src1 / EVOMG_F_COUNT != first) return EVOMG_ERR_INVALID;
      return EVOMG_OK;
    }
    case EVOMG_OP_RESIDUAL:
      if (!sys_field(c, op->level, op->src0) || !sys_field(c, op->level, op->src1) || op->src0 / EVOMG_F_COUNT != 0) return EVOMG_ERR_INVALID;
      return EVOMG_OK;
    case EVOMG_OP_RESTRICT: return (op->level < hi && sys_field(c, op->level + 1, op->src0)) ? EVOMG_OK : EVOMG_ERR_INVALID;
    case EVOMG_OP_PROLONG_ADD: case EVOMG_OP_PROLONG_SET:
      return (op->level > lo && sys_field(c, op->level - 1, op->src0)) ? EVOMG_OK : EVOMG_ERR_INVALID;
    case EVOMG_OP_SET_ZERO: return EVOMG_OK;
    case EVOMG_OP_COPY: return sys_field(c, op->level, op->src0) ? EVOMG_OK : EVOMG_ERR_INVALID;
    case EVOMG_OP_COARSE_SOLVE: return op->level == lo ? EVOMG_OK : EVOMG_ERR_INVALID;
    case EVOMG_OP_RICHARDSON: return EVOMG_ERR_UNSUPPORTED;
  }
  return EVOMG_ERR_INVALID;
}

static int sys_exec_op(sys_ctx_t* c, const evomg_op_t* op) {
  const int L = op->level, n = c->lv[L].n;
  switch (op->opcode) {
    case EVOMG_OP_SMOOTH: return sys_smooth(c, op);
    case EVOMG_OP_RESIDUAL: {
      double* x[SYS_NF];
      const int slot = sys_canon(c, L, op->src0) % EVOMG_F_COUNT;
      for (int u = 0; u < c->nf; ++u) x[u] = sys_field(c, L, EVOMG_FIELD(slot, u));
      sys_residual(c, L, op->dst / EVOMG_F_COUNT, x, sys_field(c, L, op->src1), sys_field(c, L, op->dst));
      return EVOMG_OK;
    }
    case EVOMG_OP_RESTRICT: sys_restrict(c, L, sys_field(c, L + 1, op->src0), sys_field(c, L, op->dst)); return EVOMG_OK;
    case EVOMG_OP_PROLONG_ADD: sys_prolong(c, L, sys_field(c, L - 1, op->src0), sys_field(c, L, op->dst), 1, op->omega); return EVOMG_OK;
    case EVOMG_OP_PROLONG_SET: sys_prolong(c, L, sys_field(c, L - 1, op->src0), sys_field(c, L, op->dst), 0, 1.0); return EVOMG_OK;
    case EVOMG_OP_SET_ZERO: {
      double* x = sys_field(c, L, op->dst);
      for (int b = 1; b < n - 1; ++b)
        for (int a = 1; a < n - 1; ++a) x[IDX2(n, a, b)] = 0.0;
      return EVOMG_OK;
    }
    case EVOMG_OP_COPY: {
      double* d = sys_field(c, L, op->dst);
      const double* s = sys_field(c, L, op->src0);
      if (d != s) memcpy(d, s, c->lv[L].count * sizeof(double));
      return EVOMG_OK;
    }
    case EVOMG_OP_COARSE_SOLVE: sys_coarse_cg(c); return EVOMG_OK;
  }
  return EVOMG_ERR_INVALID;
}

static int sys_run_ops(sys_ctx_t* c, const evomg_op_t* ops, int n_ops) {
  for (int i = 0; i < n_ops; ++i) {
    const int rc = sys_check_op(c, &ops[i]);
    if (rc != EVOMG_OK) { snprintf(c->err, sizeof c->err, "op %d (opcode %d) rejected: %d", i, ops[i].opcode, rc); return rc; }
  }
  for (int i = 0; i < n_ops; ++i) {
    const int rc = sys_exec_op(c, &ops[i]);
    if (rc != EVOMG_OK) return rc;
  }
  return EVOMG_OK;
}

/* residual of every equation + the combined norm sqrt(sum_u sum r_u^2), plain row order */
static double sys_residual_norm(sys_ctx_t* c) {
  const int L = c->prob.max_level, n = c->lv[L].n;
  double* x[SYS_NF];
  for (int u = 0; u < c->nf; ++u) x[u] = sys_field(c, L, EVOMG_FIELD(EVOMG_F_SOL, u));
  double s = 0.0;
  for (int u = 0; u < c->nf; ++u) {
    double* r = sys_field(c, L, EVOMG_FIELD(EVOMG_F_RES, u));
    sys_residual(c, L, u, x, sys_field(c, L, EVOMG_FIELD(EVOMG_F_RHS, u)), r);
    for (int b = 1; b < n - 1; ++b)
      for (int a = 1; a < n - 1; ++a) s = s + r[IDX2(n, a, b)] * r[IDX2(n, a, b)];
  }
  return sqrt(s);
}

static int sys_evaluate(sys_ctx_t* c, const evomg_op_t* ops, int n_ops, int samples, double wall_limit_ms,
                        evomg_result_t* out, double* hist, int cap) {
  for (int i = 0; i < n_ops; ++i) {
    const int rc = sys_check_op(c, &ops[i]);
    if (rc != EVOMG_OK) { snprintf(c->err, sizeof c->err, "op %d (opcode %d) rejected: %d", i, ops[i].opcode, rc); return rc; }
  }
  double tsum = 0.0;
  int done = 0;
  evomg_result_t r;
  memset(&r, 0, sizeof r);
  for (int smp = 0; smp < samples; ++smp) {
    for (int L = c->prob.min_level; L <= c->prob.max_level; ++L)
      for (int k = 0; k < SYS_SLOTS; ++k) {
        sys_level_t* lv = &c->lv[L];
        if (!lv->f[k]) continue;
        if (lv->init[k]) memcpy(lv->f[k], lv->init[k], lv->count * sizeof(double));
        else memset(lv->f[k], 0, lv->count * sizeof(double));
      }
    int nh = 0;
    const double t0 = now_ms();
    const double res0 = sys_residual_norm(c);
    if (hist && nh < cap) hist[nh] = res0;
    ++nh;
    double res = res0;
    int it = 0, status = EVOMG_SOLVE_MAXITER;
    if (res0 == 0.0) status = EVOMG_SOLVE_CONVERGED;
    else
      for (it = 1; it <= c->prob.max_iterations; ++it) {
        for (int i = 0; i < n_ops; ++i) sys_exec_op(c, &ops[i]);
        res = sys_residual_norm(c);
        if (hist && nh < cap) hist[nh] = res;
        ++nh;
        if (!(res == res) || res > 1.7e308) { status = EVOMG_SOLVE_DIVERGED; break; }
        if (res <= c->prob.target_res_reduction * res0) { status = EVOMG_SOLVE_CONVERGED; break; }
        if (wall_limit_ms > 0 && now_ms() - t0 > wall_limit_ms) { status = EVOMG_SOLVE_WALL_LIMIT; break; }
      }
    if (it > c->prob.max_iterations) it = c->prob.max_iterations;
    memset(&r, 0, sizeof r);
    r.time_ms = now_ms() - t0;
    r.n_iterations = (double)it;
    r.init_res_norm = res0;
    r.final_res_norm = res;
    r.status = status;
    r.n_history = (nh < cap || !hist) ? nh : cap;
    r.conv_factor = (it > 0 && res0 > 0) ? pow(res / res0, 1.0 / (double)it) : 0.0;
    tsum += r.time_ms;
    ++done;
    if (status != EVOMG_SOLVE_CONVERGED) break;
  }
  *out = r;
  out->time_ms = tsum / (double)done;
  return EVOMG_OK;
}
