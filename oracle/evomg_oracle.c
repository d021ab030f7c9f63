/*
 * evomg_oracle.c — CPU ORACLE. TEST INFRASTRUCTURE ONLY: never imported, linked or called by the
 * product path (evostencils_amd/); used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker / reported baseline.
 *
 * What it restates: the numerics of one fitness evaluation of the reference, i.e. what the
 * ExaStencils-generated executable computes when run by
 *   evostencils/code_generation/exastencils.py:417-443 (evaluate) for the statement list emitted by
 *   evostencils/code_generation/exastencils.py:684-925 (generate_multigrid),
 * on the problems example_problems/Poisson/{2D,3D}_FD_Poisson_fromL2.* and
 * example_problems/Helmholtz/2D_FD_Helmholtz_fromL3.*.
 *
 * PARITY PINNED against the one numerical result the reference records for this path: notebooks/tutorial.ipynb:3373
 * ("convergence factor: 0.9212764170210773, number of iterations: 100.0" for the cycle at :3421-3470, 2D Poisson, levels
 * 5..9). With jacobi_semantics = EVOMG_JACOBI_EXASTENCILS_V11 this oracle's residual history, read the way the reference
 * reads its executable's stdout (6 significant digits per factor, three samples), gives exactly that value and count
 * (tests/test_reference_recorded_run.py); so does the independent numpy restatement oracle/numpy_mg.py, and no other
 * reading of SURVEY App. B does (oracle/appb_matrix.py -> oracle/APPB_MATRIX.md). ExaStencils itself (the JVM code generator
 * that emits the reference's arithmetic: github.com/lssfau/ExaStencils, tag v1.1, README.md:19-21) is not under
 * /root/reference and cannot run in this pipeline. What that one run cannot pin — Jacobi-type sweeps taking effect, 3D,
 * Helmholtz, systems — rests on known-answer tests (tests/test_oracle_known_answers.py, test_helmholtz_oracle.py,
 * test_systems_oracle.py) and on the algorithms spelled out in ExaSlang in
 * example_problems/Helmholtz/2D_FD_Helmholtz_fromL3.exa3 (BiCGStab, V-cycle, norms).
 *
 * It consumes the same op program (include/evomg.h) as the HIP backend and exposes the same entry points
 * with the prefix `oracle_` instead of `evomg_`.
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/evomg.h"

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

/* ===================== real instantiation ===================== */
#define SC double
#define FN(x) x##_r
#define SC_IS_COMPLEX 0
static inline double sc_add(double a, double b) { return a + b; }
static inline double sc_sub(double a, double b) { return a - b; }
static inline double sc_mul(double a, double b) { return a * b; }
static inline double sc_div(double a, double b) { return a / b; }
static inline double sc_rmul(double r, double a) { return r * a; }
static inline double sc_neg(double a) { return -a; }
static inline double sc_zero(void) { return 0.0; }
static inline double sc_from_real(double r) { return r; }
static inline double sc_from_c64(evomg_c64 z) { return z.re; }
static inline int sc_iszero(double a) { return a == 0.0; }
static inline double sc_norm_from_dot(double d) { return sqrt(d); }
static inline double sc_fma(double a, double b, double c) { return fma(a, b, c); }
#include "evomg_oracle_impl.h"
#undef SC
#undef FN
#undef SC_IS_COMPLEX
#define sc_add csc_add
#define sc_sub csc_sub
#define sc_mul csc_mul
#define sc_div csc_div
#define sc_rmul csc_rmul
#define sc_neg csc_neg
#define sc_zero csc_zero
#define sc_from_real csc_from_real
#define sc_from_c64 csc_from_c64
#define sc_iszero csc_iszero
#define sc_norm_from_dot csc_norm_from_dot
#define sc_fma csc_fma

/* ===================== complex instantiation ===================== */
typedef struct { double re, im; } cplx;
#define SC cplx
#define FN(x) x##_c
#define SC_IS_COMPLEX 1
static inline cplx csc_add(cplx a, cplx b) { cplx r = {a.re + b.re, a.im + b.im}; return r; }
static inline cplx csc_sub(cplx a, cplx b) { cplx r = {a.re - b.re, a.im - b.im}; return r; }
static inline cplx csc_mul(cplx a, cplx b) {
  cplx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
  return r;
}
/* textbook division (a * conj(b)) / |b|^2, fixed order */
static inline cplx csc_div(cplx a, cplx b) {
  const double d = b.re * b.re + b.im * b.im;
  cplx r = {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
  return r;
}
static inline cplx csc_rmul(double s, cplx a) { cplx r = {s * a.re, s * a.im}; return r; }
static inline cplx csc_neg(cplx a) { cplx r = {-a.re, -a.im}; return r; }
static inline cplx csc_zero(void) { cplx r = {0.0, 0.0}; return r; }
static inline cplx csc_from_real(double s) { cplx r = {s, 0.0}; return r; }
static inline cplx csc_from_c64(evomg_c64 z) { cplx r = {z.re, z.im}; return r; }
static inline int csc_iszero(cplx a) { return a.re == 0.0 && a.im == 0.0; }
/* |sqrt(z)| = sqrt(|z|): `fabs(curRes)` of a Complex<Real> sqrt(dot(r,r)) (Helmholtz exa3:86-88,191) */
/* a*b + c with one rounding per real product-accumulate, fixed order (device: evomg_scalar.h sc_fma) */
static inline cplx csc_fma(cplx a, cplx b, cplx c) {
  cplx r = {fma(-a.im, b.im, fma(a.re, b.re, c.re)), fma(a.im, b.re, fma(a.re, b.im, c.im))};
  return r;
}
static inline double csc_norm_from_dot(cplx d) { return sqrt(sqrt(d.re * d.re + d.im * d.im)); }
#include "evomg_oracle_impl.h"
#undef SC
#undef FN

/* ===================== systems of PDEs (real, 2D) ===================== */
#include "evomg_oracle_system.h"

/* ===================== non-linear FAS problems (real, 2D) ===================== */
#include "evomg_oracle_fas.h"

/* ===================== C entry points (same shapes as evomg_*) ===================== */
typedef struct {
  int is_fas;
  fas_ctx_t* fas;
  int is_system;
  sys_ctx_t* s;
  int is_complex;
  ctx_t_r* r;
  ctx_t_c* c;
} oracle_handle_t;

static char g_err[256];

const char* oracle_last_error(const oracle_handle_t* h) {
  if (!h) return g_err;
  if (h->is_fas) return h->fas->err;
  if (h->is_system) return h->s->err;
  return h->is_complex ? h->c->err : h->r->err;
}

static int check_problem(const evomg_problem_t* p) {
  if (!p || p->abi_version != EVOMG_ABI_VERSION) { snprintf(g_err, sizeof g_err, "abi mismatch"); return EVOMG_ERR_INVALID; }
  if (p->dimension != 2 && p->dimension != 3) { snprintf(g_err, sizeof g_err, "dimension"); return EVOMG_ERR_INVALID; }
  if (p->min_level < 1 || p->max_level >= EVOMG_MAX_LEVELS || p->min_level > p->max_level) {
    snprintf(g_err, sizeof g_err, "levels"); return EVOMG_ERR_INVALID;
  }
  if (p->scalar_kind == EVOMG_COMPLEX_F64 && p->dimension != 2) { snprintf(g_err, sizeof g_err, "complex 3D"); return EVOMG_ERR_UNSUPPORTED; }
  if (p->bc_kind == EVOMG_BC_HELMHOLTZ_ROBIN && p->scalar_kind != EVOMG_COMPLEX_F64) return EVOMG_ERR_INVALID;
  if (p->outer_kind == EVOMG_OUTER_PBICGSTAB && p->dimension != 2) return EVOMG_ERR_UNSUPPORTED;
  if (p->jacobi_semantics != EVOMG_JACOBI_TEXTBOOK && p->jacobi_semantics != EVOMG_JACOBI_EXASTENCILS_V11) {
    snprintf(g_err, sizeof g_err, "jacobi_semantics must be chosen"); return EVOMG_ERR_INVALID;
  }
  if (p->cgs_kind != EVOMG_CGS_CG && p->cgs_kind != EVOMG_CGS_BICGSTAB && p->cgs_kind != EVOMG_CGS_FAS_SMOOTHER && p->cgs_kind != EVOMG_CGS_CG_CLASSICAL) {
    snprintf(g_err, sizeof g_err, "bad coarse-grid solver kind"); return EVOMG_ERR_INVALID;
  }
  if (p->nonlinear_kind == EVOMG_NONLINEAR_PROGRAM) {
    /* both programs well-formed (the same check as csrc/evomg.hip: check_problem) */
    for (int which = 0; which < 2; ++which) {
      const int n = which ? p->nonlinear_program.n_jacobian : p->nonlinear_program.n_sigma;
      const uint8_t* code = which ? p->nonlinear_program.jacobian : p->nonlinear_program.sigma;
      int depth = 0, ok = n >= 1 && n <= EVOMG_NL_MAX_OPS;
      for (int k = 0; ok && k < n; ++k) {
        const int op = code[k];
        if (op == EVOMG_NL_PUSH_U || (op >= EVOMG_NL_PUSH_CONST && op < EVOMG_NL_PUSH_CONST + EVOMG_NL_MAX_CONSTS)) ++depth;
        else if (op == EVOMG_NL_DUP) { ok = depth >= 1; ++depth; }
        else if (op >= EVOMG_NL_ADD && op <= EVOMG_NL_DIV) { ok = depth >= 2; --depth; }
        else if (op >= EVOMG_NL_NEG && op <= EVOMG_NL_COSH) ok = depth >= 1;
        else ok = 0;
        ok = ok && depth <= EVOMG_NL_MAX_STACK;
      }
      if (!ok || depth != 1) { snprintf(g_err, sizeof g_err, "non-linear FAS: malformed program"); return EVOMG_ERR_INVALID; }
    }
  }
  if (p->nonlinear_kind != EVOMG_NONLINEAR_NONE) {
    if ((p->nonlinear_kind != EVOMG_NONLINEAR_GAM_U_EXP_U && p->nonlinear_kind != EVOMG_NONLINEAR_PROGRAM) || p->dimension != 2 || p->scalar_kind != EVOMG_REAL_F64 ||
        p->bc_kind != EVOMG_BC_DIRICHLET || p->outer_kind != EVOMG_OUTER_MG_ITERATION || p->cgs_kind != EVOMG_CGS_FAS_SMOOTHER ||
        p->n_unknowns > 1) {
      snprintf(g_err, sizeof g_err, "non-linear FAS: 2D real Dirichlet scalar problems, N(u) = gam u exp(u), smoother as coarse solver");
      return EVOMG_ERR_UNSUPPORTED;
    }
  } else if (p->cgs_kind == EVOMG_CGS_FAS_SMOOTHER) return EVOMG_ERR_INVALID;
  if (p->stencil_kind != EVOMG_STENCIL_STAR && p->stencil_kind != EVOMG_STENCIL_COMPACT) return EVOMG_ERR_INVALID;
  if (p->n_unknowns > 1 || p->stencil_kind == EVOMG_STENCIL_COMPACT) {
    if (p->n_unknowns > EVOMG_MAX_UNKNOWNS) { snprintf(g_err, sizeof g_err, "too many unknowns"); return EVOMG_ERR_UNSUPPORTED; }
    if (p->dimension != 2 || p->scalar_kind != EVOMG_REAL_F64 || p->bc_kind != EVOMG_BC_DIRICHLET ||
        p->outer_kind != EVOMG_OUTER_MG_ITERATION || (p->cgs_kind != EVOMG_CGS_CG && p->cgs_kind != EVOMG_CGS_CG_CLASSICAL)) {
      snprintf(g_err, sizeof g_err, "systems of PDEs: 2D real Dirichlet problems with the generated solver loop and CG only");
      return EVOMG_ERR_UNSUPPORTED;
    }
  }
  return EVOMG_OK;
}

int oracle_create(const evomg_problem_t* p, int device_id, oracle_handle_t** out) {
  (void)device_id;
  int rc = check_problem(p);
  if (rc != EVOMG_OK) return rc;
  oracle_handle_t* h = (oracle_handle_t*)calloc(1, sizeof *h);
  h->is_fas = p->nonlinear_kind != EVOMG_NONLINEAR_NONE;
  h->is_system = !h->is_fas && (p->n_unknowns > 1 || p->stencil_kind == EVOMG_STENCIL_COMPACT);
  h->is_complex = (p->scalar_kind == EVOMG_COMPLEX_F64);
  rc = h->is_fas ? fas_create(p, &h->fas) : h->is_system ? sys_create(p, &h->s) : h->is_complex ? create_c(p, &h->c) : create_r(p, &h->r);
  if (rc != EVOMG_OK) { free(h); return rc; }
  *out = h;
  return EVOMG_OK;
}

void oracle_destroy(oracle_handle_t* h) {
  if (!h) return;
  if (h->is_fas) fas_destroy(h->fas);
  else if (h->is_system) sys_destroy(h->s);
  else if (h->is_complex) destroy_c(h->c); else destroy_r(h->r);
  free(h);
}

int oracle_abi_version(void) { return EVOMG_ABI_VERSION; }

#define LEVEL_OK(P, L) ((L) >= (P).min_level && (L) <= (P).max_level)

int oracle_set_initial_field(oracle_handle_t* h, int level, int field, const double* host) {
  if (h && h->is_fas && host) {
    fas_ctx_t* c = h->fas;
    if (!LEVEL_OK(c->prob, level) || !fas_field(c, level, field)) return EVOMG_ERR_INVALID;
    fas_level_t* lv = &c->lv[level];
    if (!lv->init[field]) lv->init[field] = (double*)calloc(lv->count, sizeof(double));
    memcpy(lv->init[field], host, lv->count * sizeof(double));
    memcpy(lv->f[field], host, lv->count * sizeof(double));
    return EVOMG_OK;
  }
  if (h && h->is_system && host) {
    sys_ctx_t* c = h->s;
    if (!LEVEL_OK(c->prob, level)) return EVOMG_ERR_INVALID;
    const int k = sys_canon(c, level, field);
    sys_level_t* lv = &c->lv[level];
    if (k < 0 || !lv->f[k]) return EVOMG_ERR_INVALID;
    if (!lv->init[k]) lv->init[k] = (double*)calloc(lv->count, sizeof(double));
    memcpy(lv->init[k], host, lv->count * sizeof(double));
    memcpy(lv->f[k], host, lv->count * sizeof(double));
    return EVOMG_OK;
  }
  if (!h || !host || field < 0 || field >= EVOMG_F_COUNT) return EVOMG_ERR_INVALID;
  if (h->is_complex) {
    if (!LEVEL_OK(h->c->prob, level)) return EVOMG_ERR_INVALID;
    if (field == EVOMG_F_CORR && level < h->c->prob.max_level) field = EVOMG_F_SOL;
    level_t_c* lv = &h->c->lv[level];
    if (!lv->f[field]) return EVOMG_ERR_INVALID;
    if (!lv->init[field]) lv->init[field] = zalloc_c(lv->count);
    memcpy(lv->init[field], host, lv->count * sizeof(cplx));
    memcpy(lv->f[field], host, lv->count * sizeof(cplx));
  } else {
    if (!LEVEL_OK(h->r->prob, level)) return EVOMG_ERR_INVALID;
    if (field == EVOMG_F_CORR && level < h->r->prob.max_level) field = EVOMG_F_SOL;
    level_t_r* lv = &h->r->lv[level];
    if (!lv->f[field]) return EVOMG_ERR_INVALID;
    if (!lv->init[field]) lv->init[field] = zalloc_r(lv->count);
    memcpy(lv->init[field], host, lv->count * sizeof(double));
    memcpy(lv->f[field], host, lv->count * sizeof(double));
  }
  return EVOMG_OK;
}

int oracle_set_field(oracle_handle_t* h, int level, int field, const double* host) {
  if (h && h->is_fas && host) {
    if (!LEVEL_OK(h->fas->prob, level) || !fas_field(h->fas, level, field)) return EVOMG_ERR_INVALID;
    memcpy(fas_field(h->fas, level, field), host, h->fas->lv[level].count * sizeof(double));
    return EVOMG_OK;
  }
  if (h && h->is_system && host) {
    if (!LEVEL_OK(h->s->prob, level)) return EVOMG_ERR_INVALID;
    double* p = sys_field(h->s, level, field);
    if (!p) return EVOMG_ERR_INVALID;
    memcpy(p, host, h->s->lv[level].count * sizeof(double));
    return EVOMG_OK;
  }
  if (!h || !host || field < 0 || field >= EVOMG_F_COUNT) return EVOMG_ERR_INVALID;
  if (h->is_complex) {
    if (!LEVEL_OK(h->c->prob, level)) return EVOMG_ERR_INVALID;
    cplx* p = field_c(h->c, level, field);
    if (!p) return EVOMG_ERR_INVALID;
    memcpy(p, host, h->c->lv[level].count * sizeof(cplx));
  } else {
    if (!LEVEL_OK(h->r->prob, level)) return EVOMG_ERR_INVALID;
    double* p = field_r(h->r, level, field);
    if (!p) return EVOMG_ERR_INVALID;
    memcpy(p, host, h->r->lv[level].count * sizeof(double));
  }
  return EVOMG_OK;
}

int oracle_get_field(oracle_handle_t* h, int level, int field, double* host) {
  if (h && h->is_fas && host) {
    if (!LEVEL_OK(h->fas->prob, level) || !fas_field(h->fas, level, field)) return EVOMG_ERR_INVALID;
    memcpy(host, fas_field(h->fas, level, field), h->fas->lv[level].count * sizeof(double));
    return EVOMG_OK;
  }
  if (h && h->is_system && host) {
    if (!LEVEL_OK(h->s->prob, level)) return EVOMG_ERR_INVALID;
    const double* p = sys_field(h->s, level, field);
    if (!p) return EVOMG_ERR_INVALID;
    memcpy(host, p, h->s->lv[level].count * sizeof(double));
    return EVOMG_OK;
  }
  if (!h || !host || field < 0 || field >= EVOMG_F_COUNT) return EVOMG_ERR_INVALID;
  if (h->is_complex) {
    if (!LEVEL_OK(h->c->prob, level)) return EVOMG_ERR_INVALID;
    const cplx* p = field_c(h->c, level, field);
    if (!p) return EVOMG_ERR_INVALID;
    memcpy(host, p, h->c->lv[level].count * sizeof(cplx));
  } else {
    if (!LEVEL_OK(h->r->prob, level)) return EVOMG_ERR_INVALID;
    const double* p = field_r(h->r, level, field);
    if (!p) return EVOMG_ERR_INVALID;
    memcpy(host, p, h->r->lv[level].count * sizeof(double));
  }
  return EVOMG_OK;
}

int oracle_set_operators(oracle_handle_t* h, const evomg_problem_t* p) {
  if (!h || !p) return EVOMG_ERR_INVALID;
  if (h->is_system || h->is_fas) return EVOMG_ERR_UNSUPPORTED;
  if (h->is_complex) {
    ctx_t_c* c = h->c;
    if (p->min_level != c->prob.min_level || p->max_level != c->prob.max_level) return EVOMG_ERR_INVALID;
    for (int L = p->min_level; L <= p->max_level; ++L)
      for (int e = 0; e < EVOMG_STENCIL_ENTRIES; ++e) c->lv[L].st[e] = csc_from_c64(p->mg_stencil[L][e]);
    for (int e = 0; e < EVOMG_STENCIL_ENTRIES; ++e) c->outer_st[e] = csc_from_c64(p->outer_stencil[e]);
    memcpy(c->prob.mg_stencil, p->mg_stencil, sizeof p->mg_stencil);
    memcpy(c->prob.outer_stencil, p->outer_stencil, sizeof p->outer_stencil);
    c->prob.wavenumber = p->wavenumber;
  } else {
    ctx_t_r* c = h->r;
    if (p->min_level != c->prob.min_level || p->max_level != c->prob.max_level) return EVOMG_ERR_INVALID;
    for (int L = p->min_level; L <= p->max_level; ++L)
      for (int e = 0; e < EVOMG_STENCIL_ENTRIES; ++e) c->lv[L].st[e] = p->mg_stencil[L][e].re;
    memcpy(c->prob.mg_stencil, p->mg_stencil, sizeof p->mg_stencil);
  }
  return EVOMG_OK;
}

int oracle_run_ops(oracle_handle_t* h, const evomg_op_t* ops, int n_ops) {
  if (!h || (!ops && n_ops > 0) || n_ops < 0) return EVOMG_ERR_INVALID;
  if (h->is_fas) return fas_run_ops(h->fas, ops, n_ops);
  if (h->is_system) return sys_run_ops(h->s, ops, n_ops);
  return h->is_complex ? run_ops_c(h->c, ops, n_ops) : run_ops_r(h->r, ops, n_ops);
}

int oracle_norm(oracle_handle_t* h, int level, int field, double* out) {
  if (h && h->is_fas && out) {
    if (!LEVEL_OK(h->fas->prob, level) || !fas_field(h->fas, level, field)) return EVOMG_ERR_INVALID;
    *out = fas_norm(h->fas, level, fas_field(h->fas, level, field));
    return EVOMG_OK;
  }
  if (h && h->is_system && out) {
    if (!LEVEL_OK(h->s->prob, level)) return EVOMG_ERR_INVALID;
    const double* p = sys_field(h->s, level, field);
    if (!p) return EVOMG_ERR_INVALID;
    const int n = h->s->lv[level].n;
    double acc = 0.0;
    for (int b = 1; b < n - 1; ++b)
      for (int a = 1; a < n - 1; ++a) acc = acc + p[IDX2(n, a, b)] * p[IDX2(n, a, b)];
    *out = sqrt(acc);
    return EVOMG_OK;
  }
  if (!h || !out || field < 0 || field >= EVOMG_F_COUNT) return EVOMG_ERR_INVALID;
  if (h->is_complex) {
    if (!LEVEL_OK(h->c->prob, level)) return EVOMG_ERR_INVALID;
    const cplx* p = field_c(h->c, level, field);
    if (!p) return EVOMG_ERR_INVALID;
    *out = norm_c(h->c, level, p);
  } else {
    if (!LEVEL_OK(h->r->prob, level)) return EVOMG_ERR_INVALID;
    const double* p = field_r(h->r, level, field);
    if (!p) return EVOMG_ERR_INVALID;
    *out = norm_r(h->r, level, p);
  }
  return EVOMG_OK;
}

int oracle_evaluate(oracle_handle_t* h, const evomg_op_t* ops, int n_ops, int evaluation_samples,
                    double wall_limit_ms, evomg_result_t* out, double* res_history, int history_capacity) {
  if (!h || !out || (!ops && n_ops > 0) || n_ops < 0 || evaluation_samples < 1) return EVOMG_ERR_INVALID;
  memset(out, 0, sizeof *out);
  if (h->is_fas) return fas_evaluate(h->fas, ops, n_ops, evaluation_samples, wall_limit_ms, out, res_history, history_capacity);
  if (h->is_system) return sys_evaluate(h->s, ops, n_ops, evaluation_samples, wall_limit_ms, out, res_history, history_capacity);
  /* validate once */
  for (int i = 0; i < n_ops; ++i) {
    int rc = h->is_complex ? check_op_c(h->c, &ops[i]) : check_op_r(h->r, &ops[i]);
    if (rc != EVOMG_OK) return rc;
  }
  double tsum = 0.0;
  evomg_result_t r;
  for (int s = 0; s < evaluation_samples; ++s) {
    memset(&r, 0, sizeof r);
    if (h->is_complex) {
      reset_fields_c(h->c);
      if (h->c->prob.outer_kind == EVOMG_OUTER_PBICGSTAB) solve_pbicgstab_c(h->c, ops, n_ops, wall_limit_ms, &r, res_history, history_capacity);
      else solve_mg_c(h->c, ops, n_ops, wall_limit_ms, &r, res_history, history_capacity);
    } else {
      reset_fields_r(h->r);
      if (h->r->prob.outer_kind == EVOMG_OUTER_PBICGSTAB) return EVOMG_ERR_UNSUPPORTED;
      solve_mg_r(h->r, ops, n_ops, wall_limit_ms, &r, res_history, history_capacity);
    }
    tsum += r.time_ms;
    if (r.status != EVOMG_SOLVE_CONVERGED) { evaluation_samples = s + 1; break; }
  }
  *out = r;
  out->time_ms = tsum / (double)evaluation_samples;
  return EVOMG_OK;
}
