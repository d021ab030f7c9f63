/*
 * evomg.h — C-ABI of the MI355X-native multigrid fitness-evaluation backend.
 *
 * This header is the drop-in boundary for ONE path of jonas-schmitt/evostencils:
 * the generate -> compile -> run -> parse fitness evaluation of
 *   evostencils/code_generation/exastencils.py:485-537  ProgramGenerator.generate_and_evaluate
 * Instead of emitting ExaSlang text, starting two JVMs, running `make` and a
 * generated OpenMP executable, the Python side lowers the same IR DAG to a flat
 * *op program* (one evomg_op_t per ExaSlang statement the reference would emit,
 * exastencils.py:684-925) and hands it to evomg_evaluate(), which interprets it
 * with hand-written HIP kernels on one MI355X.
 *
 * Plain C, no torch types, no C++ in the signatures: bindable from ctypes/cffi.
 * Every function returns an int status (EVOMG_OK == 0) and never aborts; a bad
 * candidate is reported through evomg_result_t.status, mirroring the reference's
 * "sentinels, not exceptions" convention (exastencils.py:430-439,476-483).
 * A handle is not thread-safe; use one handle per GPU / per process.
 */
#ifndef EVOMG_H
#define EVOMG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVOMG_ABI_VERSION 6
#define EVOMG_MAX_LEVELS 24       /* levels are absolute: grid level L has 2^L+1 nodes per dimension */
#define EVOMG_MAX_DIM 3
#define EVOMG_STENCIL_ENTRIES 7   /* star stencil: centre, -x, +x, -y, +y, -z, +z */
#define EVOMG_MAX_BLOCK_CELLS 8   /* maximum_local_system_size default, optimization/program.py:107 */
#define EVOMG_MAX_UNKNOWNS 3      /* systems of PDEs: unknowns per node (LinearElasticity: u, v; ABI 6: three — ir/system.py:32-50 is general) */
#define EVOMG_SYS_STENCIL_ENTRIES 9 /* 2D compact stencil: centre, W, E, S, N, SW, SE, NW, NE */

/* status codes (function return values) */
enum {
  EVOMG_OK = 0,
  EVOMG_ERR_INVALID = 1,      /* bad argument / malformed op program */
  EVOMG_ERR_DEVICE = 2,       /* HIP runtime error (no device, OOM, launch failure) */
  EVOMG_ERR_UNSUPPORTED = 3   /* well-formed but not implemented (e.g. 3D complex) */
};

/* outcome of one candidate evaluation (evomg_result_t.status) */
enum {
  EVOMG_SOLVE_CONVERGED = 0,  /* res <= target_res_reduction * init_res */
  EVOMG_SOLVE_MAXITER = 1,    /* "maximum number of solver iterations" (exastencils.py:555-557) */
  EVOMG_SOLVE_DIVERGED = 2,   /* NaN/Inf residual norm */
  EVOMG_SOLVE_WALL_LIMIT = 3  /* wall_limit_ms exceeded: counterpart of the subprocess timeout (exastencils.py:430-431) */
};

/* scalar kind of all fields of a problem */
enum { EVOMG_REAL_F64 = 0, EVOMG_COMPLEX_F64 = 1 };

/* boundary handling applied after every field write */
enum {
  EVOMG_BC_DIRICHLET = 0,       /* boundary nodes are never written by field loops (Poisson 2D/3D) */
  EVOMG_BC_HELMHOLTZ_ROBIN = 1  /* Dirichlet 0 at y=0,1; x_b = x_in / (1 - i k h) at x=0,1
                                   (example_problems/Helmholtz/2D_FD_Helmholtz_fromL3.exa4:43-59) */
};

/* outer solver loop */
enum {
  EVOMG_OUTER_MG_ITERATION = 0, /* "generate solver for u in solEq": repeat cycle until reduction / maxNumIts
                                   (example_problems/Poisson/2D_FD_Poisson_fromL2.exa3:2-4) */
  EVOMG_OUTER_PBICGSTAB = 1     /* MG-preconditioned BiCGStab, 2 cycles per iteration
                                   (example_problems/Helmholtz/2D_FD_Helmholtz_fromL3.exa3:144-200) */
};

/* coarse-grid solver = gen_mgCycle@minLevel */
enum {
  EVOMG_CGS_CG = 0,       /* solver_cgs = "CG" (Poisson/2D_FD_Poisson_fromL2.exa3:12-14: at most cgs_max_iterations,
                             reduction cgs_target_res_reduction). ABI 5: where a single-reduction form of the recurrence
                             exists (Chronopoulos & Gear: both inner products of an iteration in ONE reduction, the same
                             iterates in exact arithmetic, the same stopping rule) the solve runs in that form — real 2D
                             grids up to 129^2 (csrc/evomg_cg1.h), real 3D grids of 33^3 (evomg_cg1_slab3d) — otherwise as
                             EVOMG_CGS_CG_CLASSICAL. The reference's one recorded number does not depend on the coarse
                             solver's arithmetic (oracle/APPB_MATRIX.md, row "coarse=exact") */
  EVOMG_CGS_BICGSTAB = 1, /* explicit BiCGStab text, Helmholtz/2D_FD_Helmholtz_fromL3.exa3:396-433 */
  EVOMG_CGS_FAS_SMOOTHER = 2, /* non-linear FAS problems: cgs_max_iterations Newton-Jacobi sweeps with weight fas_cgs_omega
                                 (`Function CGS@coarsest`: repeat 200 times Smoother, FAS_2D_Basic_template.exa4:56-61,63-71) */
  EVOMG_CGS_CG_CLASSICAL = 3  /* ABI 5: "CG" as the textbook two-reduction recurrence on every grid (p . A p, then r . r):
                                 what EVOMG_CGS_CG was up to ABI 4; kept as the second mode the single-reduction form is
                                 checked against */
};

/* Non-linear term N(u) of a FAS problem -Lap u + N(u) = f, SURVEY 8f N4 (exastencils_FAS.py; FAS_2D_Basic_template.exa4) */
enum {
  EVOMG_NONLINEAR_NONE = 0,
  EVOMG_NONLINEAR_GAM_U_EXP_U = 1, /* `Stencil gamSten { [0,0] => gam * exp(Solution) }` applied to Solution:
                                      N(u) = gam u e^u, N'(u) = gam (1 + u) e^u (template:27-29,69) */
  EVOMG_NONLINEAR_PROGRAM = 2      /* ABI 6: ANY stencil entry sigma(Solution) the reference's parser admits
                                      (code_generation/parser.py:146-169 takes an arbitrary sympy term): sigma and the
                                      Newton term d/du [ sigma(u) u ] — sympy's own derivative, as exastencils_FAS.py
                                      forms it — travel as two small straight-line programs, evomg_nl_program_t */
};
/* A straight-line program in reverse Polish notation over ONE variable (the value of Solution at the node), evaluated
 * with an operand stack of at most EVOMG_NL_MAX_STACK entries, left to right, one IEEE operation per opcode (no
 * contraction): the HIP kernels and the CPU oracle execute the same operations in the same order. */
#define EVOMG_NL_MAX_OPS 48
#define EVOMG_NL_MAX_CONSTS 8
#define EVOMG_NL_MAX_STACK 8
enum {
  EVOMG_NL_END = 0,
  EVOMG_NL_PUSH_U = 1,       /* the variable */
  EVOMG_NL_PUSH_CONST = 2,   /* 2 + k: consts[k], k < EVOMG_NL_MAX_CONSTS */
  EVOMG_NL_DUP = 10,
  EVOMG_NL_ADD = 16, EVOMG_NL_SUB = 17, EVOMG_NL_MUL = 18, EVOMG_NL_DIV = 19,   /* (a b op): a op b */
  EVOMG_NL_NEG = 20, EVOMG_NL_EXP = 21, EVOMG_NL_SIN = 22, EVOMG_NL_COS = 23, EVOMG_NL_SQRT = 24, EVOMG_NL_LOG = 25,
  EVOMG_NL_TANH = 26, EVOMG_NL_SINH = 27, EVOMG_NL_COSH = 28
};
typedef struct {
  int32_t n_sigma, n_jacobian;             /* opcodes in use (a program leaves exactly one value on the stack) */
  uint8_t sigma[EVOMG_NL_MAX_OPS];         /* the stencil entry, e.g. `gam * Solution * Solution` */
  uint8_t jacobian[EVOMG_NL_MAX_OPS];      /* d/du [ sigma(u) u ], the term the Newton-Jacobi denominator adds to the centre */
  double  consts[EVOMG_NL_MAX_CONSTS];     /* shared by both programs (`Val gam`, numbers) */
} evomg_nl_program_t;

/* What a `with jacobi` sweep — every smoother with Single partitioning, point or block
 * (exastencils.py:64-70,778-822) — does to the field it relaxes. The sweep itself is the same in both modes
 * (reads the active slot, writes the next slot); the modes differ in whether the next slot ever becomes active. */
enum {
  EVOMG_JACOBI_TEXTBOOK = 1,        /* ONE advance after the sweep: its result is the field from then on (the IR / LFA
                                       meaning, stencils/multiple.py:204-217; what exastencils.py:338-359 intends) */
  EVOMG_JACOBI_EXASTENCILS_V11 = 2  /* the next slot is never advanced: the sweep runs (and costs its time) but the field
                                       keeps its old values. This is what the reference's pipeline does with the
                                       ExaStencils version it pins (README.md:19-21, v1.1): the one numeric it records,
                                       notebooks/tutorial.ipynb:3373 (rho = 0.9212764170210773, 100 iterations), is
                                       reproduced to the last digit in this mode and by no other convention
                                       (oracle/APPB_MATRIX.md). */
};

/* Shape of a scalar problem's operator (ABI 5). The reference's parser takes stencils with arbitrary offsets
 * (code_generation/parser.py:8-22); none of its example problems uses a scalar one outside the star. */
enum {
  EVOMG_STENCIL_STAR = 0,     /* centre and +-1 per axis: mg_stencil (the tuned scalar path) */
  EVOMG_STENCIL_COMPACT = 1   /* 2D, any entry of the 3 x 3 neighbourhood: sys_stencil[level][0][0] with n_unknowns = 1 — the
                                 problem runs on the path of the systems of PDEs as a system of one unknown */
};

typedef struct { double re, im; } evomg_c64;

/*
 * Problem descriptor: replaces the .knowledge/.settings/.exa2/.exa3 file zoo the
 * reference parses (code_generation/parser.py:25-143). Pure data.
 */
typedef struct {
  int32_t abi_version;          /* must be EVOMG_ABI_VERSION */
  int32_t dimension;            /* 2 or 3 (knowledge: dimensionality) */
  int32_t min_level;            /* coarsest level (knowledge: minLevel) */
  int32_t max_level;            /* finest level   (knowledge: maxLevel) */
  int32_t scalar_kind;          /* EVOMG_REAL_F64 | EVOMG_COMPLEX_F64 */
  int32_t bc_kind;              /* EVOMG_BC_* */
  int32_t outer_kind;           /* EVOMG_OUTER_* */
  int32_t cgs_kind;             /* EVOMG_CGS_* */
  int32_t max_iterations;       /* solver_maxNumIts */
  int32_t cgs_max_iterations;   /* solver_cgs_maxNumIts */
  double  target_res_reduction;     /* solver_targetResReduction */
  double  cgs_target_res_reduction; /* solver_cgs_targetResReduction */
  /* System operator per absolute level (index = level), rediscretised per level
   * (SURVEY App. B.4). Star stencil, entry order: centre, -x, +x, -y, +y, -z, +z.
   * For real problems .im == 0. This is the operator the multigrid cycle acts on
   * (Laplace for Poisson, the shifted preconditioner M for Helmholtz). */
  evomg_c64 mg_stencil[EVOMG_MAX_LEVELS][EVOMG_STENCIL_ENTRIES];
  /* Operator of the outer Krylov solver on the finest level only (Helmholtz A);
   * ignored for EVOMG_OUTER_MG_ITERATION. */
  evomg_c64 outer_stencil[EVOMG_STENCIL_ENTRIES];
  /* Robin factor numerator data: wavenumber k (Helmholtz). */
  double wavenumber;
  int32_t jacobi_semantics;     /* EVOMG_JACOBI_*; 0 is rejected: the caller has to choose */
  /* Systems of PDEs (example_problems/LinearElasticity/2D_FD_LinearElasticity_fromL2.exa2:11-50; ir/system.py:32-135):
   * n_unknowns > 1 fields per node, coupled by the block operator sys_stencil[level][equation][unknown][entry]
   * (2D, real, Dirichlet; mg_stencil is ignored). n_unknowns <= 1: a scalar problem (with stencil_kind = EVOMG_STENCIL_COMPACT:
   * one whose nine-point stencil sits in sys_stencil[level][0][0], same restrictions as a system). */
  int32_t n_unknowns;
  double  sys_stencil[EVOMG_MAX_LEVELS][EVOMG_MAX_UNKNOWNS][EVOMG_MAX_UNKNOWNS][EVOMG_SYS_STENCIL_ENTRIES];
  /* Non-linear FAS problems (ABI 3; 2D, real, homogeneous Dirichlet; linear part = mg_stencil): nonlinear_kind != 0 selects
   * the FAS op vocabulary (EVOMG_OP_FAS_*) and the template's solver loop (repeat the cycle until the residual norm has
   * dropped by target_res_reduction or max_iterations, FAS_2D_Basic_template.exa4:130-160). */
  int32_t nonlinear_kind;       /* EVOMG_NONLINEAR_* */
  int32_t stencil_kind;         /* EVOMG_STENCIL_* (ABI 5; 0 in older descriptors = EVOMG_STENCIL_STAR) */
  double  nonlinear_gamma;      /* `Val gam : Real` */
  double  fas_cgs_omega;        /* weight of the coarse-grid solver's Newton-Jacobi sweeps (template:66: 0.8) */
  evomg_nl_program_t nonlinear_program;   /* ABI 6: nonlinear_kind == EVOMG_NONLINEAR_PROGRAM */
} evomg_problem_t;

/* field slots of one level (exastencils.py:14-26,295-316) */
enum {
  EVOMG_F_SOL = 0,      /* u@max on the finest level, gen_error_u@L below it (exastencils.py:296-301) */
  EVOMG_F_RHS = 1,      /* RHS_u@L */
  EVOMG_F_RES = 2,      /* gen_residual_u@L */
  EVOMG_F_CORR = 3,     /* gen_error_u@L; the same array as SOL below the finest level */
  EVOMG_F_CGS_RHS = 4,  /* gen_rhs_u@min (exastencils.py:890-892) */
  EVOMG_F_COUNT = 5
};
/* field id of slot `slot` of the `unknown`-th unknown of a system (unknown 0 = the plain slot ids) */
#define EVOMG_FIELD(slot, unknown) ((slot) + EVOMG_F_COUNT * (unknown))

/* opcodes: one per ExaSlang statement kind the reference emits */
enum {
  EVOMG_OP_SMOOTH = 1,       /* solve locally at dst@L [with jacobi] relax w {...}        exastencils.py:778-822 */
  EVOMG_OP_RESIDUAL = 2,     /* dst@L = src1@L - (A@L * src0@L)                           exastencils.py:837-853 */
  EVOMG_OP_RESTRICT = 3,     /* dst@L = R@(L+1) * src0@(L+1)                              exastencils.py:868-873 */
  EVOMG_OP_PROLONG_ADD = 4,  /* dst@L += w * (P@(L-1) * src0@(L-1))                       exastencils.py:742-743 */
  EVOMG_OP_PROLONG_SET = 5,  /* dst@L  = P@(L-1) * src0@(L-1)                             exastencils.py:870-873 */
  EVOMG_OP_SET_ZERO = 6,     /* dst@L = 0                                                 exastencils.py:709,756,836,892,895 */
  EVOMG_OP_COPY = 7,         /* dst@L = src0@L                                            exastencils.py:891,901 */
  EVOMG_OP_COARSE_SOLVE = 8, /* gen_mgCycle@min(): Krylov solve on gen_error/gen_rhs@min  exastencils.py:896 */
  EVOMG_OP_RICHARDSON = 9,   /* dst@L += w * (src1@L - A@L*dst@L)                         exastencils.py:710-726 */
  /* Non-linear FAS vocabulary (exastencils_FAS.py:99-319). Fields are fixed by the statement kind: Solution = SOL,
   * RHS, Residual = RES, Approximation = CORR (a separate array on EVERY level of a FAS problem). A(u) = L u + N(u). */
  EVOMG_OP_FAS_SMOOTH = 16,            /* point smoother: u += w (f - A(u)) / (diag L [+ N'(u)]), block[0] local Newton
                                          steps (0: Picard, divisor diag L only); PART_SINGLE: Jacobi through the next
                                          slot + advance; PART_RED_BLACK: two colours in place        :209-268 */
  EVOMG_OP_FAS_RESIDUAL = 17,          /* Residual@L = RHS@L - A(Solution@L)                          :158-180 */
  EVOMG_OP_FAS_RESTRICT_SOLUTION = 18, /* Approximation@L = R * Solution@(L+1); Solution@L = Approximation@L  :121-136 */
  EVOMG_OP_FAS_COARSE_RHS = 19,        /* RHS@L = R * Residual@(L+1) + (L@L Approximation@L + N-stencil(Solution@L) *
                                          Approximation@L)                                           :138-156, grammar 287-293 */
  EVOMG_OP_FAS_CGS = 20,               /* CGS@L ( ): cgs_max_iterations Newton-Jacobi sweeps           :195-207 */
  EVOMG_OP_FAS_SUBTRACT = 21           /* Solution@L -= Approximation@L                               :182-193 */
};

/* Pseudo-opcode for evomg_profile_begin only: the solver loop's own residual + norm pass on the finest level (it is not an
 * op of the cycle; SURVEY 3.1: `gen_solve` computes the residual norm once per iteration). */
#define EVOMG_PROF_LOOP_NORM 64

enum { EVOMG_PART_SINGLE = 0, EVOMG_PART_RED_BLACK = 1 };  /* ir/partitioning.py:18,34 */

typedef struct {
  int32_t opcode;        /* EVOMG_OP_* */
  int32_t level;         /* level of dst */
  int32_t dst;           /* EVOMG_F_* */
  int32_t src0;          /* EVOMG_F_* (operand field; for SMOOTH unused) */
  int32_t src1;          /* EVOMG_F_* (right-hand side for SMOOTH/RESIDUAL/RICHARDSON) */
  int32_t partitioning;  /* EVOMG_PART_*; RED_BLACK => "color with {((i0+i1[+i2])%2), ...}" and in place */
  int32_t block[EVOMG_MAX_DIM]; /* block shape of the local solve, (1,1,1) = point smoother */
  int32_t unknowns;      /* SMOOTH on a system of PDEs: bit mask of the unknowns whose equations are solved together
                            in one `solve locally` block (0b11 collective, 0b01 / 0b10 decoupled); 0 on scalar problems.
                            On systems dst/src1 name the slots of the FIRST unknown of the mask, RESIDUAL's src0 the
                            solution slot of unknown 0 (the row reads all unknowns) */
  int32_t block_other[EVOMG_MAX_DIM]; /* ABI 4, SMOOTH on a system whose mask names two unknowns: block shape of the SECOND
                            unknown's cells when it differs from the first's (`block`); all zero = the same shape. The
                            grammar registers one BlockShape terminal per combination of per-unknown shapes
                            (grammar/multigrid.py:388-407), so most collective block smoothers of a system look like this.
                            Tiles repeat with the componentwise maximum of the two shapes — an [EXTERNAL] guess: the loop
                            ExaStencils generates around such a `solve locally` is not visible from the reference. Accepted
                            only under EVOMG_JACOBI_EXASTENCILS_V11, where the sweep is launched for its time and its
                            result never becomes the field (rho / iterations exact); EVOMG_ERR_UNSUPPORTED under
                            EVOMG_JACOBI_TEXTBOOK, where the unverified tiling would decide the candidate's fitness */
  int32_t block_third;   /* ABI 6 (the field was reserved before), SMOOTH on a system whose mask names THREE unknowns with
                            differing shapes: the third listed unknown's block shape, packed bx | by << 8 | bz << 16; 0 = the
                            same as the second's. Same acceptance rule as block_other */
  double  omega;         /* relaxation factor */
} evomg_op_t;
#define EVOMG_PACK_BLOCK(bx, by, bz) ((int32_t)((bx) | ((by) << 8) | ((bz) << 16)))

typedef struct {
  double  time_ms;         /* mean wall time of the solve loop over evaluation_samples runs */
  double  conv_factor;     /* (final/init)^(1/n): geometric mean of the per-iteration residual ratios */
  double  n_iterations;    /* iterations performed */
  double  init_res_norm;
  double  final_res_norm;
  int32_t status;          /* EVOMG_SOLVE_* */
  int32_t n_history;       /* number of residual norms written to res_history (init + one per iteration) */
} evomg_result_t;

typedef struct {
  double ms;               /* mean duration of one launch (HIP events on the handle's stream) */
  double bytes;            /* algorithmic bytes of one launch (SURVEY 8d table x points of the level) */
  double gbps;             /* bytes / ms */
  int32_t launches;
  int32_t reserved;
} evomg_bench_t;

typedef struct evomg_handle evomg_handle_t;

/* -- lifetime ------------------------------------------------------------ */
/* Allocates every level's fields on device `device_id` once; fields are reused across candidates. */
int  evomg_create(const evomg_problem_t* problem, int device_id, evomg_handle_t** out);
void evomg_destroy(evomg_handle_t* h);
const char* evomg_last_error(const evomg_handle_t* h);  /* h may be NULL: error of the last failed create */
int  evomg_abi_version(void);
/* Number of GPUs this process can use (0: none, or the runtime failed to start). One process per GPU picks its device
 * as local_rank % evomg_device_count() (the reference runs one MPI rank per worker, README.md: `mpiexec -n N`). */
int  evomg_device_count(void);

/* -- problem data (K10): dense host arrays, x fastest, (2^L+1)^d scalars; complex = interleaved re,im -- */
/* Sets the *initial* content of a field (restored at the start of every solve). Only SOL and RHS of the
 * finest level are meaningful inputs (u with its boundary values, RHS_u).
 * RHS of the finest level (EVOMG_OUTER_MG_ITERATION): handles of one process that upload the same data share ONE device copy
 * of it — with several candidates in flight it is read from the Infinity Cache instead of once per handle from HBM. Invisible
 * to the caller: a handle that writes the field (evomg_set_field, an op program with it as a destination, another upload) is
 * given an array of its own first. EVOMG_NO_SHARED_RHS=1 switches it off. */
int  evomg_set_initial_field(evomg_handle_t* h, int level, int field, const double* host);
/* Reads back the current content of a field after an evaluate/run_ops call. */
int  evomg_get_field(evomg_handle_t* h, int level, int field, double* host);
/* Overwrites the current content of a field (test hook for per-kernel parity). */
int  evomg_set_field(evomg_handle_t* h, int level, int field, const double* host);
/* Re-targets the operators, e.g. Helmholtz wavenumber sweep k, 2k, 4k (exastencils.py:518-532). */
int  evomg_set_operators(evomg_handle_t* h, const evomg_problem_t* problem);

/* -- the call -------------------------------------------------------------- */
/* One candidate: validates the op program, runs the solver loop `evaluation_samples` times.
 * res_history may be NULL; otherwise receives the residual norms of the last sample. */
int  evomg_evaluate(evomg_handle_t* h, const evomg_op_t* ops, int n_ops,
                    int evaluation_samples, double wall_limit_ms,
                    evomg_result_t* out, double* res_history, int history_capacity);
/* Executes the op program once on the current field contents (no solver loop). Test hook. */
int  evomg_run_ops(evomg_handle_t* h, const evomg_op_t* ops, int n_ops);
/* L2 norm of the current content of a field over the inner nodes (the solver loop's norm). */
int  evomg_norm(evomg_handle_t* h, int level, int field, double* out);

/* Time objective without contention (optimization/program.py:449-451 uses time / iterations as an objective): runs exactly
 * n_iterations solver-loop iterations of the candidate from the initial state — no convergence test — and reports the mean
 * wall time of one iteration. With several candidates in flight on one GPU a solve's own wall time includes its neighbours;
 * calling this while nothing else runs on the device and multiplying by the iteration count of the real evaluation gives
 * the time the candidate needs alone (population.evaluate_population_in_flight(..., time_fn=...)). Stops early if the
 * residual becomes NaN / Inf; n_done receives the iterations actually run (may be NULL). */
int  evomg_time_iterations(evomg_handle_t* h, const evomg_op_t* ops, int n_ops, int n_iterations,
                           double* ms_per_iteration, int* n_done);
/* Krylov iterations of ALL coarse-grid solves of the last evomg_evaluate (its last sample) — what varies along a solve and
 * between candidates while every other launch of a cycle costs the same each iteration. 0 for problems whose coarse solver
 * does not iterate to a tolerance. */
int  evomg_coarse_iterations(evomg_handle_t* h, long long* total);
/* Three-point fit of the candidate's time on an otherwise idle device:
 *     time(its, K) ~ ms_fixed + ms_per_iteration * its + ms_per_coarse_iteration * K
 * (its: solver-loop iterations, K: coarse-solver iterations, evomg_coarse_iterations). Runs n_short and n_long (> n_short)
 * iterations with the coarse solver capped at ONE iteration, and n_long iterations as they are. Together with `its` and K of
 * the real evaluation this is the candidate's time-to-solution ALONE on the GPU, whatever ran beside that evaluation. */
int  evomg_time_model(evomg_handle_t* h, const evomg_op_t* ops, int n_ops, int n_short, int n_long,
                      double* ms_fixed, double* ms_per_iteration, double* ms_per_coarse_iteration);
/* The same fit with ms_per_coarse_iteration KNOWN: what one iteration of the coarse-grid solver costs depends on the problem
 * and its coarsest level, not on the candidate, so a population measures it on its first few candidates (evomg_time_model)
 * and every later candidate pays for the two cheap probes only (the third one, n_long iterations with the coarse solver
 * uncapped, is four fifths of a model's cost on the headline sizes). */
int  evomg_time_model_known_coarse(evomg_handle_t* h, const evomg_op_t* ops, int n_ops, int n_short, int n_long,
                                   double ms_per_coarse_iteration, double* ms_fixed, double* ms_per_iteration);

/* -- measurement ------------------------------------------------------------ */
/* Times `reps` back-to-back launches of one op on the handle's stream with HIP events. */
int  evomg_bench_op(evomg_handle_t* h, const evomg_op_t* op, int warmup, int reps, evomg_bench_t* out);
/* Live measurement inside a solve: between profile_begin and profile_end every launch of an op with this opcode
 * on this level is bracketed by HIP events on the handle's stream; profile_end reports the mean duration per
 * launch, the launch count and the algorithmic bytes of one launch. */
int  evomg_profile_begin(evomg_handle_t* h, int opcode, int level);
int  evomg_profile_end(evomg_handle_t* h, evomg_bench_t* out);
/* Name of the device kernel an op dispatches to (for matching rocprofv3 kernel-trace rows). */
int  evomg_kernel_name(evomg_handle_t* h, const evomg_op_t* op, char* buf, int buf_len);

/* -- population exchange: one collective per generation ----------------------------------------------
 * Candidates shard embarrassingly across the GPUs of a node, one process per GPU; the ONLY exchange is the fitness table of
 * a generation — the RCCL-over-xGMI counterpart of the reference's mpi4py `allgather` (optimization/program.py:285-310,
 * :534-536 after the round-robin re-evaluation, :582 per generation). Rows are EVOMG_FITNESS_COLUMNS doubles:
 * (candidate index, convergence factor, time-to-solution ms, iterations). librccl is opened on first use. */
#define EVOMG_FITNESS_COLUMNS 4
typedef struct evomg_comm evomg_comm_t;
/* One communicator per process: rank of `world`, bound to GPU `device_id`. The RCCL unique id travels through the file
 * `id_path` (unique per job, on a file system all ranks see: /dev/shm or /tmp on one node): rank 0 writes it atomically, the
 * others wait at most timeout_ms (< 0: forever) for it. Collective: every rank must call it. */
int  evomg_comm_create(int rank, int world, int device_id, const char* id_path, int timeout_ms, evomg_comm_t** out);
void evomg_comm_destroy(evomg_comm_t* c);
const char* evomg_comm_last_error(const evomg_comm_t* c);   /* c may be NULL: error of the last failed create */
/* Every rank contributes n_local rows (the same n_local everywhere: pad); `all` receives world * n_local rows in rank
 * order on EVERY rank (ncclAllGather) — program.py:536,582. Host pointers; staged through device buffers of the handle. */
int  evomg_allgather_fitness(evomg_comm_t* c, const double* local, int n_local, double* all);
/* The same table on rank `root` only (`all` may be NULL elsewhere) — north_star's "tiny gather back to rank 0". */
int  evomg_gather_fitness(evomg_comm_t* c, const double* local, int n_local, double* all, int root);
/* RCCL's version code (e.g. 22707), 0 when librccl cannot be opened. Initialises nothing on the GPU. */
int  evomg_rccl_version(void);

#ifdef __cplusplus
}
#endif
#endif /* EVOMG_H */
