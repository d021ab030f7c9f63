// evomg_fas.h — non-linear FAS problems: -Lap u + sigma(u) u = f (the reference's example: sigma = gam e^u),
// 2D, real, u = 0 on the boundary; the
// statement kinds the reference's second backend emits (evostencils/code_generation/exastencils_FAS.py:99-319) on its
// example problem (example_problems/FAS_2D_Basic/FAS_2D_Basic_template.exa4).
//
// First implementation, as for the systems of PDEs: correct and complete for the op vocabulary, plain kernels (thread per
// node, no fusion) — the reference runs this problem on levels 6..10. Operation order = the layer-4 statement texts, left
// to right = oracle/evomg_oracle_fas.h. exp() is the device library's: results agree with the CPU oracle to rounding of
// the exponential, not bit for bit (tests: 1e-12 relative per statement, 1e-10 on residual histories).
//   A(X; S) := ( ( Laplace * X ) + ( gamSten * X ) ),  gamSten = gam * exp( S )
#pragma once
#include "evomg_kernels.h"
#include "evomg_krylov.h"   // the cross-workgroup exchange primitives (xchg_store / xchg_load / grid_publish)

// ---- the non-linear stencil entry sigma(S) and the Newton term d/du [sigma(u) u] -----------------------------------------
// kind EVOMG_NONLINEAR_GAM_U_EXP_U: the reference's example, sigma = gam exp(S), written out (one exponential serves both);
// kind EVOMG_NONLINEAR_PROGRAM (ABI 6): whatever entry the problem's template holds, as two straight-line programs in
// reverse Polish notation (include/evomg.h: evomg_nl_program_t; compiled from sympy by code_generation/fas.py, the derivative
// formed as the reference forms it, exastencils_FAS.py: simplify(diff(sigma u, u))). The operand stack is EIGHT named registers
// moved by static shifts — a stack indexed at run time would live in scratch memory —; the program is a kernel argument, so
// the opcode dispatch is a scalar branch. One IEEE operation per opcode, in program order: the oracle executes the same list.
struct FasNl {
  int kind;
  double gam;
  evomg_nl_program_t prog;
};
__device__ __forceinline__ double fas_nl_run(const evomg_nl_program_t& p, const uint8_t (&code)[EVOMG_NL_MAX_OPS], int n, double u) {
  static_assert(EVOMG_NL_MAX_STACK == 8, "the operand stack is eight named registers");
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;      // s0 = top of the stack
  for (int k = 0; k < n; ++k) {
    const int op = code[k];
    if (op == EVOMG_NL_PUSH_U || op == EVOMG_NL_DUP || (op >= EVOMG_NL_PUSH_CONST && op < EVOMG_NL_PUSH_CONST + EVOMG_NL_MAX_CONSTS)) {
      const double v = op == EVOMG_NL_PUSH_U ? u : op == EVOMG_NL_DUP ? s0 : p.consts[op - EVOMG_NL_PUSH_CONST];
      s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v;
    } else if (op >= EVOMG_NL_ADD && op <= EVOMG_NL_DIV) {
      const double a = s1, b = s0;
      const double r = op == EVOMG_NL_ADD ? a + b : op == EVOMG_NL_SUB ? a - b : op == EVOMG_NL_MUL ? a * b : a / b;
      s0 = r; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7;
    } else {
      switch (op) {
        case EVOMG_NL_NEG: s0 = -s0; break;
        case EVOMG_NL_EXP: s0 = exp(s0); break;
        case EVOMG_NL_SIN: s0 = sin(s0); break;
        case EVOMG_NL_COS: s0 = cos(s0); break;
        case EVOMG_NL_SQRT: s0 = sqrt(s0); break;
        case EVOMG_NL_LOG: s0 = log(s0); break;
        case EVOMG_NL_TANH: s0 = tanh(s0); break;
        case EVOMG_NL_SINH: s0 = sinh(s0); break;
        case EVOMG_NL_COSH: s0 = cosh(s0); break;
        default: break;
      }
    }
  }
  return s0;
}
// the stencil entry at Solution = s
__device__ __forceinline__ double fas_nl_sigma(const FasNl& nl, double s) {
  if (nl.kind == EVOMG_NONLINEAR_PROGRAM) return fas_nl_run(nl.prog, nl.prog.sigma, nl.prog.n_sigma, s);
  return nl.gam * exp(s);
}
// the entry at u and the Newton term at u (what the smoother's denominator adds to the centre coefficient)
__device__ __forceinline__ void fas_nl_both(const FasNl& nl, double u, double* sigma, double* newton) {
  if (nl.kind == EVOMG_NONLINEAR_PROGRAM) {
    *sigma = fas_nl_run(nl.prog, nl.prog.sigma, nl.prog.n_sigma, u);
    *newton = fas_nl_run(nl.prog, nl.prog.jacobian, nl.prog.n_jacobian, u);
  } else {
    const double ex = exp(u);
    *sigma = nl.gam * ex;
    *newton = (nl.gam * (u + 1.0)) * ex;
  }
}
// `centre`: the value of X at the node (the smoother's local iterate); s: Solution at the node (what gamSten reads)
__device__ __forceinline__ double fas_apply(const Stencil<double>& st, const FasNl& nl, const double* x, const Geo& g, long c, double centre, double s) {
  double a = st.c[0] * centre;
  a = a + st.c[1] * x[c - 1];
  a = a + st.c[2] * x[c + 1];
  a = a + st.c[3] * x[c - g.pitch];
  a = a + st.c[4] * x[c + g.pitch];
  return a + fas_nl_sigma(nl, s) * centre;
}
// Solution += w * ( ( RHS - A(Solution) ) / ( Laplace:[0,0] [+ d/du (sigma(u) u): gam*(Solution + 1)*exp(Solution)] ) ), n local steps
__device__ __forceinline__ double fas_relax(const Stencil<double>& st, const FasNl& nl, const double* x, const double* f, const Geo& g, long c,
                                            double om, int newton, int n_steps) {
  double u = x[c];
  for (int s = 0; s < (n_steps > 1 ? n_steps : 1); ++s) {
    double a = st.c[0] * u;
    a = a + st.c[1] * x[c - 1];
    a = a + st.c[2] * x[c + 1];
    a = a + st.c[3] * x[c - g.pitch];
    a = a + st.c[4] * x[c + g.pitch];
    double sig, jac;
    fas_nl_both(nl, u, &sig, &jac);
    const double num = f[c] - (a + sig * u);
    const double den = newton ? st.c[0] + jac : st.c[0];
    u = u + om * (num / den);
  }
  return u;
}
// colour < 0: Jacobi sweep x -> y (inner nodes; y keeps its zero boundary); colour 0 | 1: one colour in place (y == x)
__global__ void __launch_bounds__(256) evomg_fas_smooth(Geo g, Stencil<double> st, FasNl nl, const double* x, const double* f, double* y,
                                                        double om, int newton, int n_steps, int colour) {
  int i, j, k;
  if (!node_of_thread<2>(g, i, j, k) || !is_inner<2>(g, i, j, 0)) return;
  if (colour >= 0 && ((i + j) & 1) != colour) return;
  const long c = gidx<2>(g, i, j, 0);
  y[c] = fas_relax(st, nl, x, f, g, c, om, newton, n_steps);
}
// Residual = ( RHS - A(Solution) )
__global__ void __launch_bounds__(256) evomg_fas_residual(Geo g, Stencil<double> st, FasNl nl, const double* __restrict__ x,
                                                          const double* __restrict__ f, double* __restrict__ r) {
  int i, j, k;
  if (!node_of_thread<2>(g, i, j, k) || !is_inner<2>(g, i, j, 0)) return;
  const long c = gidx<2>(g, i, j, 0);
  r[c] = f[c] - fas_apply(st, nl, x, g, c, x[c], x[c]);
}
__device__ __forceinline__ double fas_fw(const Geo& gf, const double* __restrict__ fine, int I, int J) {   // = evomg_restrict<double, 2>
  const long c = gidx<2>(gf, 2 * I, 2 * J, 0);
  const long p = gf.pitch;
  const double e = (fine[c - 1] + fine[c + 1]) + (fine[c - p] + fine[c + p]);
  const double q = (fine[c - p - 1] + fine[c - p + 1]) + (fine[c + p - 1] + fine[c + p + 1]);
  double v = 0.25 * fine[c];
  v = v + 0.125 * e;
  v = v + 0.0625 * q;
  return v;
}
// Approximation@c = R * Solution@f ; Solution@c = Approximation@c
__global__ void __launch_bounds__(256) evomg_fas_restrict_solution(Geo gc, Geo gf, const double* __restrict__ fine_sol,
                                                                   double* __restrict__ approx, double* __restrict__ sol) {
  int I, J, K;
  if (!node_of_thread<2>(gc, I, J, K) || !is_inner<2>(gc, I, J, 0)) return;
  const double v = fas_fw(gf, fine_sol, I, J);
  const long c = gidx<2>(gc, I, J, 0);
  approx[c] = v;
  sol[c] = v;
}
// RHS@c = ( ( R * Residual@f ) + A(Approximation@c; Solution@c) )
__global__ void __launch_bounds__(256) evomg_fas_coarse_rhs(Geo gc, Geo gf, Stencil<double> st, FasNl nl, const double* __restrict__ fine_res,
                                                            const double* __restrict__ approx, const double* __restrict__ sol,
                                                            double* __restrict__ rhs) {
  int I, J, K;
  if (!node_of_thread<2>(gc, I, J, K) || !is_inner<2>(gc, I, J, 0)) return;
  const long c = gidx<2>(gc, I, J, 0);
  rhs[c] = fas_fw(gf, fine_res, I, J) + fas_apply(st, nl, approx, gc, c, approx[c], sol[c]);
}
// Solution -= Approximation
__global__ void __launch_bounds__(256) evomg_fas_subtract(Geo g, double* __restrict__ sol, const double* __restrict__ approx) {
  int i, j, k;
  if (!node_of_thread<2>(g, i, j, k) || !is_inner<2>(g, i, j, 0)) return;
  const long c = gidx<2>(g, i, j, 0);
  sol[c] = sol[c] - approx[c];
}

// CGS@coarsest: `sweeps` Newton-Jacobi sweeps (one local step, weight om) in ONE launch of ONE 1024-thread workgroup when
// the level has at most 65^2 nodes: a sweep there is a few thousand nodes, 200 launches of it are 200 x 5 us of launch
// floor for 200 x ~1 us of work. Thread t owns the inner nodes t, t + 1024, ...; a barrier between sweeps; the two slots
// alternate. Same per-node formula as evomg_fas_smooth (fas_relax): bit-identical to the launch-per-sweep form.
// After an even number of sweeps the result is in `a`, after an odd number in `b` (the host swaps accordingly).
__global__ void __launch_bounds__(1024) evomg_fas_cgs(Geo g, Stencil<double> st, FasNl nl, double* a, double* b, const double* f,
                                                      double om, int sweeps) {
  const int m = g.n - 2;
  const int total = m * m;
  double* x = a;
  double* y = b;
  for (int s = 0; s < sweeps; ++s) {
    for (int q = threadIdx.x; q < total; q += 1024) {
      const int i = 1 + q % m, j = 1 + q / m;
      const long c = gidx<2>(g, i, j, 0);
      y[c] = fas_relax(st, nl, x, f, g, c, om, 1, 1);
    }
    __syncthreads();
    double* t = x; x = y; y = t;
  }
}

// The same sweeps with the iterate in LDS: both slots of the 65^2 field (2 x 33 KB) stay on the compute unit
// for the whole call (each with the boundary values its array holds), the right-hand side of a thread's (up to four) nodes in registers; a sweep reads its five values from
// one LDS image and writes the other — no global-memory round trip between the sweeps, what is left is the arithmetic
// (one exponential, one division per node). Per-node formula and operation order = fas_relax(.., newton = 1, n_steps = 1):
// bit-identical to evomg_fas_cgs and to the launch-per-sweep form. a / b: as evomg_fas_cgs (the result of the last sweep
// and of the one before it are written back to the slots where the global-memory form would have left them).
// dynamic LDS: 2 * n * n doubles.
__global__ void __launch_bounds__(1024) evomg_fas_cgs_lds(Geo g, Stencil<double> st, FasNl nl, double* a, double* b, const double* f,
                                                          double om, int sweeps) {
  extern __shared__ __align__(16) double fas_lds[];
  const int n = g.n, m = n - 2, total = m * m, nn = n * n;
  double* cur = fas_lds;
  double* nxt = fas_lds + nn;
  for (int q = threadIdx.x; q < nn; q += 1024) {
    const int i = q % n, j = q / n;
    const bool inner = i >= 1 && i <= m && j >= 1 && j <= m;
    cur[q] = a[gidx<2>(g, i, j, 0)];
    nxt[q] = inner ? 0.0 : b[gidx<2>(g, i, j, 0)];        // each slot keeps its own boundary values, as in global memory
  }
  constexpr int MAXN = 4;                                  // 63^2 inner nodes / 1024 threads
  double fv[MAXN];
  int lq[MAXN];
#pragma unroll
  for (int e = 0; e < MAXN; ++e) {
    const int q = threadIdx.x + 1024 * e;
    const bool ok = q < total;
    const int i = 1 + (ok ? q % m : 0), j = 1 + (ok ? q / m : 0);
    lq[e] = ok ? j * n + i : -1;
    fv[e] = ok ? f[gidx<2>(g, i, j, 0)] : 0.0;
  }
  __syncthreads();
  for (int s = 0; s < sweeps; ++s) {
#pragma unroll
    for (int e = 0; e < MAXN; ++e) {
      if (lq[e] < 0) continue;
      const int c = lq[e];
      const double u = cur[c];
      double ap = st.c[0] * u;
      ap = ap + st.c[1] * cur[c - 1];
      ap = ap + st.c[2] * cur[c + 1];
      ap = ap + st.c[3] * cur[c - n];
      ap = ap + st.c[4] * cur[c + n];
      double sig, jac;
      fas_nl_both(nl, u, &sig, &jac);
      ap = ap + sig * u;
      const double num = fv[e] - ap;
      const double den = st.c[0] + jac;
      nxt[c] = u + om * (num / den);
    }
    __syncthreads();
    double* t = cur; cur = nxt; nxt = t;
  }
  // cur = result of the last sweep, nxt = of the one before; the global-memory form leaves sweep k's output in b for odd k
  double* out_last = (sweeps & 1) ? b : a;
  double* out_prev = (sweeps & 1) ? a : b;
#pragma unroll
  for (int e = 0; e < MAXN; ++e) {
    if (lq[e] < 0) continue;
    const int q = threadIdx.x + 1024 * e;
    const long c = gidx<2>(g, 1 + q % m, 1 + q / m, 0);
    out_last[c] = cur[lq[e]];
    if (sweeps > 0) out_prev[c] = nxt[lq[e]];
  }
}

// ---- CGS@coarsest across several compute units: overlapping tiles, a rendezvous every nine sweeps ------------------------
// The 200 Newton-Jacobi sweeps on 65^2 are VALU bound on ONE compute unit (an exponential and a division per node:
// ~70 FP64 operations x 4 nodes per thread, 2.25 us per sweep, 75 % of a FAS solve at levels 6..10). A rendezvous per
// sweep would cost more than the sweep (1.0-1.8 us, profiles/r03_rendezvous.md), so the grid is cut into 16 x 16 tiles with
// a halo of EIGHT nodes: a workgroup holds the 32 x 32 extended tile (one node per thread) plus one read-only ring in LDS
// and runs NINE sweeps on a region that shrinks by one node per sweep — sweep s of a round updates the nodes at least s
// away from the extended tile's edge, which read only values that sweep s - 1 computed from true data — after which
// exactly the owned tile is still exact. Then the owned values are published (agent-scope stores, one sequence flag per
// workgroup, as evomg_cg_multi_wg3d), every workgroup waits for all flags (bounded: a peer that never arrives sets
// info[0] = 1 instead of hanging) and reloads its extended tile. The redundant halo work is 2.6x the nodes, spread over
// 16 compute units; the per-node formula is fas_relax(.., newton = 1, n_steps = 1) = evomg_fas_cgs_lds: bit-identical.
// Two LDS images alternate by the parity of the GLOBAL sweep number; each carries the boundary values of its global
// slot (a: even, b: odd), as the global-memory form does. a / b afterwards: as evomg_fas_cgs_lds.
// xchg: [2][pitch * n] exchange slots (alternating by round); flags: [groups], zeroed by the host.
#define EVOMG_FAS_TILE 16
#define EVOMG_FAS_HALO 8
__host__ __device__ inline int fas_tiles_per_side(int n) { return (n - 2 + EVOMG_FAS_TILE - 1) / EVOMG_FAS_TILE; }
__device__ __forceinline__ bool grid_wait(const unsigned* flags, int groups, unsigned seq, int* lds_flag) {
  if (threadIdx.x == 0) *lds_flag = 0;
  __syncthreads();
  if ((int)threadIdx.x < groups) {
    int budget = 1 << 22;
    while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) {
      if (--budget == 0) { *lds_flag = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return *lds_flag == 0;
}
__global__ void __launch_bounds__(1024) evomg_fas_cgs_tiles(Geo g, Stencil<double> st, FasNl nl, double* a, double* b, const double* f,
                                                            double om, int sweeps, double* xchg, unsigned* flags, int* info) {
  constexpr int T = EVOMG_FAS_TILE, H = EVOMG_FAS_HALO, W = T + 2 * H, IW = W + 2;
  static_assert(W * W == 1024, "one node of the extended tile per thread");
  __shared__ double img[2][IW * IW];
  __shared__ int lds_flag;
  const int n = g.n, m = n - 2;
  const int tiles = fas_tiles_per_side(n), groups = tiles * tiles;          // gridDim.x == groups (host)
  const int w = (int)blockIdx.x, tx = w % tiles, ty = w / tiles;
  const int tid = (int)threadIdx.x, lx = tid % W, ly = tid / W;
  const int gi = tx * T + 1 - H + lx, gj = ty * T + 1 - H + ly;             // this thread's node
  const bool inner = gi >= 1 && gi <= m && gj >= 1 && gj <= m;
  const bool owned = inner && lx >= H && lx < H + T && ly >= H && ly < H + T;
  const long c = inner ? gidx<2>(g, gi, gj, 0) : 0;
  const double fv = inner ? f[c] : 0.0;
  const int me = (ly + 1) * IW + lx + 1;
  const size_t slot = (size_t)g.pitch * n;
  unsigned seq = 0;
  bool alive = true;
  double prev = 0.0, last = 0.0;
  const double* src = a;             // where the field after `done` sweeps lies (inner nodes)
  bool src_plain = true;             // ... written before this launch (plain loads) or by peers during it
  int done = 0;
  while (done < sweeps && alive) {
    const int todo = (sweeps - done) < H + 1 ? (sweeps - done) : H + 1;
    const int p0 = done & 1;
    for (int q = tid; q < IW * IW; q += 1024) {
      const int ei = tx * T - H + q % IW, ej = ty * T - H + q / IW;         // the node behind image position q
      double v0 = 0.0, v1 = 0.0;                                           // image p0 (the field now) and the other one
      if (ei >= 0 && ei < n && ej >= 0 && ej < n) {
        const long cq = gidx<2>(g, ei, ej, 0);
        if (ei >= 1 && ei <= m && ej >= 1 && ej <= m) v0 = src_plain ? src[cq] : xchg_load(src + cq);
        else {                                                              // each slot keeps its own boundary values
          v0 = (p0 ? b : a)[cq];
          v1 = (p0 ? a : b)[cq];
        }
      }
      img[p0][q] = v0;
      img[p0 ^ 1][q] = v1;
    }
    __syncthreads();
    for (int s = 0; s < todo; ++s) {
      const int cur = (done + s) & 1;
      if (inner && lx >= s && lx < W - s && ly >= s && ly < W - s) {
        const double* im = img[cur];
        const double u = im[me];
        double ap = st.c[0] * u;
        ap = ap + st.c[1] * im[me - 1];
        ap = ap + st.c[2] * im[me + 1];
        ap = ap + st.c[3] * im[me - IW];
        ap = ap + st.c[4] * im[me + IW];
        double sig, jac;
        fas_nl_both(nl, u, &sig, &jac);
        ap = ap + sig * u;
        const double num = fv - ap;
        const double den = st.c[0] + jac;
        const double un = u + om * (num / den);
        img[cur ^ 1][me] = un;
        prev = u;
        last = un;
      }
      __syncthreads();
    }
    done += todo;
    double* dst = xchg + (size_t)(seq & 1) * slot;
    if (owned) xchg_store(dst + c, last);
    grid_publish(flags, w, ++seq);
    alive = grid_wait(flags, groups, seq, &lds_flag);
    src = dst;
    src_plain = false;
  }
  if (alive && sweeps > 0 && owned) {
    ((sweeps & 1) ? b : a)[c] = last;
    ((sweeps & 1) ? a : b)[c] = prev;
  }
  if (!alive && tid == 0) info[0] = 1;
}
