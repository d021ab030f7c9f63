/*
 * evomg_oracle_fas.h — CPU ORACLE for the non-linear FAS backend. TEST INFRASTRUCTURE ONLY, included by
 * evomg_oracle.c.
 *
 * Restates what the ExaStencils-generated executable computes for the layer-4 statements the reference's second
 * backend emits (evostencils/code_generation/exastencils_FAS.py:99-319) on its non-linear example
 * (example_problems/FAS_2D_Basic/FAS_2D_Basic_template.exa4): -Lap u + gam u e^u = f, 2D, real, u = 0 on the boundary.
 * One function per statement kind; the formulas are the statement texts, operand by operand, left to right:
 *   A(X; S) := ( ( Laplace * X ) + ( gamSten * X ) ),  gamSten = gam * exp( S )          (template:17-29)
 *   smoother:  Solution += w * ( ( RHS - A(Solution; Solution) ) / ( Laplace:[0,0] [+ gam*(Solution + 1)*exp(Solution)] ) )
 *              n local steps at the node; Single: through the next slot + advance (a proper Jacobi sweep: the statement
 *              list restores the active slot after every node and advances explicitly, exastencils_FAS.py:243-262);
 *              RedBlack: `color with { ( i0 + i1 ) % 2, ...}` = colour 0, then colour 1, in place
 *   residual:  Residual = ( RHS - A(Solution; Solution) )
 *   FAS:       Approximation@c = R * Solution@f ; Solution@c = Approximation@c                      (:121-136)
 *              RHS@c = ( ( R * Residual@f ) + A(Approximation@c; Solution@c) )                      (:138-156)
 *              Solution@c -= Approximation@c                                                       (:182-193)
 *              Solution@f += ( w * ( P * Solution@c ) )                                             (:103-119)
 *   CGS:       cgs_max_iterations sweeps of `Smoother`: Newton-Jacobi, one local step, weight fas_cgs_omega (template:56-71)
 *   Solve:     UpResidual; repeat the cycle until res < target * res0 or max_iterations              (template:130-160)
 * [EXTERNAL] `RestrictionNode@f * Solution@c` as the reference prints it names the coarse-level field; read here as the
 * full-weighting restriction of the FINE solution, the only reading under which the statement is a FAS transfer.
 *
 * PARITY UNPINNED for the arithmetic: the reference records no output of a FAS run (its one recorded run is the linear
 * Poisson cycle, notebooks/tutorial.ipynb:3373). What IS pinned: the statement text, character for character, against
 * the imported reference (tests/golden/fas2d_*.exa4); and known answers (tests/test_fas_oracle.py: the manufactured
 * solution is recovered to O(h^2), the FAS cycle converges at a level-independent rate).
 */
#include <math.h>

#define FAS_SLOTS EVOMG_F_COUNT

typedef struct {
  int n;
  size_t count;
  double* f[FAS_SLOTS];
  double* init[FAS_SLOTS];
  double* next;                  /* Solution<next> */
  double st[5];
} fas_level_t;

typedef struct {
  evomg_problem_t prob;
  fas_level_t lv[EVOMG_MAX_LEVELS];
  char err[256];
} fas_ctx_t;

static double* fas_field(fas_ctx_t* c, int level, int field) {
  if (field < 0 || field >= FAS_SLOTS) return NULL;
  return c->lv[level].f[field];
}

static int fas_create(const evomg_problem_t* p, fas_ctx_t** out) {
  fas_ctx_t* c = (fas_ctx_t*)calloc(1, sizeof *c);
  c->prob = *p;
  for (int L = p->min_level; L <= p->max_level; ++L) {
    fas_level_t* lv = &c->lv[L];
    lv->n = (1 << L) + 1;
    lv->count = (size_t)lv->n * lv->n;
    for (int e = 0; e < 5; ++e) lv->st[e] = p->mg_stencil[L][e].re;
    for (int k = 0; k < 4; ++k) lv->f[k] = (double*)calloc(lv->count, sizeof(double));   /* SOL, RHS, RES, CORR = Approximation */
    lv->next = (double*)calloc(lv->count, sizeof(double));
  }
  *out = c;
  return EVOMG_OK;
}
static void fas_destroy(fas_ctx_t* c) {
  for (int L = c->prob.min_level; L <= c->prob.max_level; ++L) {
    for (int k = 0; k < FAS_SLOTS; ++k) { free(c->lv[L].f[k]); free(c->lv[L].init[k]); }
    free(c->lv[L].next);
  }
  free(c);
}

/* The non-linear stencil entry sigma(S) and the Newton term d/du [sigma(u) u] (include/evomg.h: EVOMG_NONLINEAR_*). Kind
 * GAM_U_EXP_U is the reference's example written out (FAS_2D_Basic_template.exa4:27-29,69); kind PROGRAM (ABI 6) runs the
 * two reverse-Polish programs of evomg_nl_program_t — whatever entry the template holds (code_generation/parser.py:146-169
 * admits any sympy term), compiled by evostencils_amd/code_generation/fas.py — one IEEE operation per opcode, in program
 * order: exactly what the device functor of csrc/evomg_fas.h executes. */
static inline double fas_nl_run(const evomg_nl_program_t* p, const uint8_t* code, int n, double u) {
  double stack[EVOMG_NL_MAX_STACK + 1];
  int top = 0;                                   /* number of values on the stack (the descriptor check bounds it) */
  for (int k = 0; k < n; ++k) {
    const int op = code[k];
    if (op == EVOMG_NL_PUSH_U) stack[top++] = u;
    else if (op == EVOMG_NL_DUP) { stack[top] = stack[top - 1]; ++top; }
    else if (op >= EVOMG_NL_PUSH_CONST && op < EVOMG_NL_PUSH_CONST + EVOMG_NL_MAX_CONSTS) stack[top++] = p->consts[op - EVOMG_NL_PUSH_CONST];
    else if (op >= EVOMG_NL_ADD && op <= EVOMG_NL_DIV) {
      const double a = stack[top - 2], b = stack[top - 1];
      stack[top - 2] = op == EVOMG_NL_ADD ? a + b : op == EVOMG_NL_SUB ? a - b : op == EVOMG_NL_MUL ? a * b : a / b;
      --top;
    } else {
      double* t = &stack[top - 1];
      switch (op) {
        case EVOMG_NL_NEG: *t = -*t; break;
        case EVOMG_NL_EXP: *t = exp(*t); break;
        case EVOMG_NL_SIN: *t = sin(*t); break;
        case EVOMG_NL_COS: *t = cos(*t); break;
        case EVOMG_NL_SQRT: *t = sqrt(*t); break;
        case EVOMG_NL_LOG: *t = log(*t); break;
        case EVOMG_NL_TANH: *t = tanh(*t); break;
        case EVOMG_NL_SINH: *t = sinh(*t); break;
        case EVOMG_NL_COSH: *t = cosh(*t); break;
        default: break;
      }
    }
  }
  return stack[0];
}
typedef struct { int kind; double gam; const evomg_nl_program_t* prog; } fas_nl_t;
static inline fas_nl_t fas_nl_of(const evomg_problem_t* p) {
  fas_nl_t v;
  v.kind = p->nonlinear_kind;
  v.gam = p->nonlinear_gamma;
  v.prog = &p->nonlinear_program;
  return v;
}
static inline double fas_nl_sigma(fas_nl_t nl, double s) {
  if (nl.kind == EVOMG_NONLINEAR_PROGRAM) return fas_nl_run(nl.prog, nl.prog->sigma, nl.prog->n_sigma, s);
  return nl.gam * exp(s);
}
static inline void fas_nl_both(fas_nl_t nl, double u, double* sigma, double* newton) {
  if (nl.kind == EVOMG_NONLINEAR_PROGRAM) {
    *sigma = fas_nl_run(nl.prog, nl.prog->sigma, nl.prog->n_sigma, u);
    *newton = fas_nl_run(nl.prog, nl.prog->jacobian, nl.prog->n_jacobian, u);
  } else {
    const double ex = exp(u);
    *sigma = nl.gam * ex;
    *newton = (nl.gam * (u + 1.0)) * ex;
  }
}
/* ( ( Laplace * X ) + ( gamSten * X ) ) at (i, j); `centre`: the value of X at the node (the smoother's local iterate);
 * `s`: the value of Solution at the node that gamSten reads */
static inline double fas_apply(const fas_level_t* lv, fas_nl_t gam, const double* x, double centre, double s, int i, int j) {
  const int n = lv->n;
  double a = lv->st[0] * centre;
  a = a + lv->st[1] * x[IDX2(n, i - 1, j)];
  a = a + lv->st[2] * x[IDX2(n, i + 1, j)];
  a = a + lv->st[3] * x[IDX2(n, i, j - 1)];
  a = a + lv->st[4] * x[IDX2(n, i, j + 1)];
  return a + fas_nl_sigma(gam, s) * centre;
}
/* the smoother's node update: n_steps local steps (at least one); newton: divisor diag + d/du [sigma(u) u] (the example:
 * gam*(u + 1)*exp(u)) */
static inline double fas_relax(const fas_level_t* lv, fas_nl_t gam, const double* x, const double* f, int i, int j, double om,
                               int newton, int n_steps) {
  const int n = lv->n;
  double u = x[IDX2(n, i, j)];
  for (int s = 0; s < (n_steps > 1 ? n_steps : 1); ++s) {
    double a = lv->st[0] * u;
    a = a + lv->st[1] * x[IDX2(n, i - 1, j)];
    a = a + lv->st[2] * x[IDX2(n, i + 1, j)];
    a = a + lv->st[3] * x[IDX2(n, i, j - 1)];
    a = a + lv->st[4] * x[IDX2(n, i, j + 1)];
    double sig, jac;
    fas_nl_both(gam, u, &sig, &jac);
    const double num = f[IDX2(n, i, j)] - (a + sig * u);
    const double den = newton ? lv->st[0] + jac : lv->st[0];
    u = u + om * (num / den);
  }
  return u;
}
static void fas_smooth(fas_ctx_t* c, int L, double om, int red_black, int n_newton) {
  fas_level_t* lv = &c->lv[L];
  const int n = lv->n;
  const fas_nl_t gam = fas_nl_of(&c->prob);
  double* x = lv->f[EVOMG_F_SOL];
  const double* f = lv->f[EVOMG_F_RHS];
  if (red_black) {
    for (int colour = 0; colour < 2; ++colour)
      for (int j = 1; j < n - 1; ++j)
        for (int i = 1; i < n - 1; ++i)
          if (((i + j) & 1) == colour) x[IDX2(n, i, j)] = fas_relax(lv, gam, x, f, i, j, om, n_newton > 0, n_newton);
    return;
  }
  double* y = lv->next;
  for (int j = 1; j < n - 1; ++j)
    for (int i = 1; i < n - 1; ++i) y[IDX2(n, i, j)] = fas_relax(lv, gam, x, f, i, j, om, n_newton > 0, n_newton);
  lv->f[EVOMG_F_SOL] = y;          /* advance */
  lv->next = x;
}
static void fas_residual(fas_ctx_t* c, int L) {
  fas_level_t* lv = &c->lv[L];
  const int n = lv->n;
  const fas_nl_t gam = fas_nl_of(&c->prob);
  const double *x = lv->f[EVOMG_F_SOL], *f = lv->f[EVOMG_F_RHS];
  double* r = lv->f[EVOMG_F_RES];
  for (int j = 1; j < n - 1; ++j)
    for (int i = 1; i < n - 1; ++i) r[IDX2(n, i, j)] = f[IDX2(n, i, j)] - fas_apply(lv, gam, x, x[IDX2(n, i, j)], x[IDX2(n, i, j)], i, j);
}
static inline double fas_fw(const double* fine, int nf, int I, int J) {     /* full weighting: centre, edges, corners */
  const int a = 2 * I, b = 2 * J;
  const double e4 = (fine[IDX2(nf, a - 1, b)] + fine[IDX2(nf, a + 1, b)]) + (fine[IDX2(nf, a, b - 1)] + fine[IDX2(nf, a, b + 1)]);
  const double q4 = (fine[IDX2(nf, a - 1, b - 1)] + fine[IDX2(nf, a + 1, b - 1)]) + (fine[IDX2(nf, a - 1, b + 1)] + fine[IDX2(nf, a + 1, b + 1)]);
  double v = 0.25 * fine[IDX2(nf, a, b)];
  v = v + 0.125 * e4;
  v = v + 0.0625 * q4;
  return v;
}
static void fas_restrict_solution(fas_ctx_t* c, int L) {
  fas_level_t *lc = &c->lv[L], *lf = &c->lv[L + 1];
  const int nc = lc->n;
  for (int J = 1; J < nc - 1; ++J)
    for (int I = 1; I < nc - 1; ++I) {
      const double v = fas_fw(lf->f[EVOMG_F_SOL], lf->n, I, J);
      lc->f[EVOMG_F_CORR][IDX2(nc, I, J)] = v;
      lc->f[EVOMG_F_SOL][IDX2(nc, I, J)] = v;
    }
}
static void fas_coarse_rhs(fas_ctx_t* c, int L) {
  fas_level_t *lc = &c->lv[L], *lf = &c->lv[L + 1];
  const int nc = lc->n;
  const fas_nl_t gam = fas_nl_of(&c->prob);
  const double *ap = lc->f[EVOMG_F_CORR], *sol = lc->f[EVOMG_F_SOL];
  for (int J = 1; J < nc - 1; ++J)
    for (int I = 1; I < nc - 1; ++I)
      lc->f[EVOMG_F_RHS][IDX2(nc, I, J)] = fas_fw(lf->f[EVOMG_F_RES], lf->n, I, J) + fas_apply(lc, gam, ap, ap[IDX2(nc, I, J)], sol[IDX2(nc, I, J)], I, J);
}
static void fas_subtract(fas_ctx_t* c, int L) {
  fas_level_t* lv = &c->lv[L];
  const int n = lv->n;
  for (int j = 1; j < n - 1; ++j)
    for (int i = 1; i < n - 1; ++i) lv->f[EVOMG_F_SOL][IDX2(n, i, j)] = lv->f[EVOMG_F_SOL][IDX2(n, i, j)] - lv->f[EVOMG_F_CORR][IDX2(n, i, j)];
}
static void fas_prolong_add(fas_ctx_t* c, int L, double om) {
  fas_level_t *lf = &c->lv[L], *lc = &c->lv[L - 1];
  const int nf = lf->n, nc = lc->n;
  const double* coarse = lc->f[EVOMG_F_SOL];
  double* fine = lf->f[EVOMG_F_SOL];
  for (int b = 1; b < nf - 1; ++b)
    for (int a = 1; a < nf - 1; ++a) {
      const int I = a >> 1, J = b >> 1;
      double v;
      if (!(a & 1) && !(b & 1)) v = coarse[IDX2(nc, I, J)];
      else if ((a & 1) && !(b & 1)) v = 0.5 * (coarse[IDX2(nc, I, J)] + coarse[IDX2(nc, I + 1, J)]);
      else if (!(a & 1) && (b & 1)) v = 0.5 * (coarse[IDX2(nc, I, J)] + coarse[IDX2(nc, I, J + 1)]);
      else v = 0.25 * ((coarse[IDX2(nc, I, J)] + coarse[IDX2(nc, I + 1, J)]) + (coarse[IDX2(nc, I, J + 1)] + coarse[IDX2(nc, I + 1, J + 1)]));
      fine[IDX2(nf, a, b)] = fine[IDX2(nf, a, b)] + om * v;
    }
}

static int fas_check_op(const fas_ctx_t* c, const evomg_op_t* op) {
  const int lo = c->prob.min_level, hi = c->prob.max_level, L = op->level;
  if (L < lo || L > hi) return EVOMG_ERR_INVALID;
  switch (op->opcode) {
    case EVOMG_OP_FAS_SMOOTH:
      return (op->block[0] >= 0 && op->block[0] <= 16 && (op->partitioning == EVOMG_PART_SINGLE || op->partitioning == EVOMG_PART_RED_BLACK)) ? EVOMG_OK : EVOMG_ERR_INVALID;
    case EVOMG_OP_FAS_RESIDUAL: case EVOMG_OP_FAS_SUBTRACT: case EVOMG_OP_FAS_CGS: return EVOMG_OK;
    case EVOMG_OP_FAS_RESTRICT_SOLUTION: case EVOMG_OP_FAS_COARSE_RHS: return L < hi ? EVOMG_OK : EVOMG_ERR_INVALID;
    case EVOMG_OP_PROLONG_ADD: return (L > lo && op->dst == EVOMG_F_SOL && op->src0 == EVOMG_F_SOL) ? EVOMG_OK : EVOMG_ERR_INVALID;
  }
  return EVOMG_ERR_UNSUPPORTED;
}
static int fas_exec(fas_ctx_t* c, const evomg_op_t* op) {
  const int L = op->level;
  switch (op->opcode) {
    case EVOMG_OP_FAS_SMOOTH: fas_smooth(c, L, op->omega, op->partitioning == EVOMG_PART_RED_BLACK, op->block[0]); return EVOMG_OK;
    case EVOMG_OP_FAS_RESIDUAL: fas_residual(c, L); return EVOMG_OK;
    case EVOMG_OP_FAS_RESTRICT_SOLUTION: fas_restrict_solution(c, L); return EVOMG_OK;
    case EVOMG_OP_FAS_COARSE_RHS: fas_coarse_rhs(c, L); return EVOMG_OK;
    case EVOMG_OP_FAS_SUBTRACT: fas_subtract(c, L); return EVOMG_OK;
    case EVOMG_OP_FAS_CGS:
      for (int s = 0; s < c->prob.cgs_max_iterations; ++s) fas_smooth(c, L, c->prob.fas_cgs_omega, 0, 1);
      return EVOMG_OK;
    case EVOMG_OP_PROLONG_ADD: fas_prolong_add(c, L, op->omega); return EVOMG_OK;
  }
  return EVOMG_ERR_INVALID;
}
static int fas_run_ops(fas_ctx_t* c, const evomg_op_t* ops, int n_ops) {
  for (int i = 0; i < n_ops; ++i) {
    const int rc = fas_check_op(c, &ops[i]);
    if (rc != EVOMG_OK) { snprintf(c->err, sizeof c->err, "op %d (opcode %d) rejected: %d", i, ops[i].opcode, rc); return rc; }
  }
  for (int i = 0; i < n_ops; ++i) fas_exec(c, &ops[i]);
  return EVOMG_OK;
}
static double fas_norm(const fas_ctx_t* c, int level, const double* p) {
  const int n = c->lv[level].n;
  double acc = 0.0;
  for (int b = 1; b < n - 1; ++b)
    for (int a = 1; a < n - 1; ++a) acc = acc + p[IDX2(n, a, b)] * p[IDX2(n, a, b)];
  return sqrt(acc);
}
static int fas_evaluate(fas_ctx_t* c, const evomg_op_t* ops, int n_ops, int samples, double wall_limit_ms, evomg_result_t* out,
                        double* hist, int cap) {
  for (int i = 0; i < n_ops; ++i) {
    const int rc = fas_check_op(c, &ops[i]);
    if (rc != EVOMG_OK) { snprintf(c->err, sizeof c->err, "op %d (opcode %d) rejected: %d", i, ops[i].opcode, rc); return rc; }
  }
  const int L = c->prob.max_level;
  double tsum = 0.0;
  evomg_result_t r;
  int done = 0;
  for (int smp = 0; smp < samples; ++smp) {
    memset(&r, 0, sizeof r);
    for (int l = c->prob.min_level; l <= c->prob.max_level; ++l) {
      fas_level_t* lv = &c->lv[l];
      for (int k = 0; k < 4; ++k) {
        if (lv->init[k]) memcpy(lv->f[k], lv->init[k], lv->count * sizeof(double));
        else memset(lv->f[k], 0, lv->count * sizeof(double));
      }
      memset(lv->next, 0, lv->count * sizeof(double));
    }
    const double t0 = now_ms();
    int nh = 0;
    fas_residual(c, L);
    const double res0 = fas_norm(c, L, c->lv[L].f[EVOMG_F_RES]);
    if (hist && nh < cap) hist[nh] = res0;
    ++nh;
    double res = res0;
    int it = 0, status = EVOMG_SOLVE_MAXITER;
    if (res0 == 0.0) status = EVOMG_SOLVE_CONVERGED;
    else
      for (it = 1; it <= c->prob.max_iterations; ++it) {
        for (int i = 0; i < n_ops; ++i) fas_exec(c, &ops[i]);
        fas_residual(c, L);
        res = fas_norm(c, L, c->lv[L].f[EVOMG_F_RES]);
        if (hist && nh < cap) hist[nh] = res;
        ++nh;
        if (!(res == res) || res > 1.7e308) { status = EVOMG_SOLVE_DIVERGED; break; }
        if (res < c->prob.target_res_reduction * res0) { status = EVOMG_SOLVE_CONVERGED; break; }
        if (wall_limit_ms > 0 && now_ms() - t0 > wall_limit_ms) { status = EVOMG_SOLVE_WALL_LIMIT; break; }
      }
    if (it > c->prob.max_iterations) it = c->prob.max_iterations;
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
