// evomg_scalar.h — scalar primitives shared by all device kernels.
//
// The formulas are, operation for operation, those of the CPU oracle (oracle/evomg_oracle.c: sc_* / csc_*);
// both sides are compiled with -ffp-contract=off, so every pointwise kernel reproduces the oracle bit for bit.
#pragma once
#include <hip/hip_runtime.h>

struct cplx {
  double re, im;
};

#define EVOMG_HD __host__ __device__ __forceinline__

EVOMG_HD double sc_add(double a, double b) { return a + b; }
EVOMG_HD double sc_sub(double a, double b) { return a - b; }
EVOMG_HD double sc_mul(double a, double b) { return a * b; }
EVOMG_HD double sc_div(double a, double b) { return a / b; }
EVOMG_HD double sc_rmul(double r, double a) { return r * a; }
EVOMG_HD double sc_neg(double a) { return -a; }
EVOMG_HD bool sc_iszero(double a) { return a == 0.0; }
EVOMG_HD double sc_norm_from_dot(double d) { return sqrt(d); }

// fused multiply-add a*b + c (one rounding per real product-accumulate) — used inside the Krylov solvers only;
// same formulas as the oracle's sc_fma / csc_fma
EVOMG_HD double sc_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

EVOMG_HD cplx sc_add(cplx a, cplx b) { return cplx{a.re + b.re, a.im + b.im}; }
EVOMG_HD cplx sc_sub(cplx a, cplx b) { return cplx{a.re - b.re, a.im - b.im}; }
EVOMG_HD cplx sc_mul(cplx a, cplx b) { return cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
EVOMG_HD cplx sc_div(cplx a, cplx b) {
  const double d = b.re * b.re + b.im * b.im;
  return cplx{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
EVOMG_HD cplx sc_fma(cplx a, cplx b, cplx c) {
  return cplx{__builtin_fma(-a.im, b.im, __builtin_fma(a.re, b.re, c.re)),
              __builtin_fma(a.im, b.re, __builtin_fma(a.re, b.im, c.im))};
}
EVOMG_HD cplx sc_rmul(double s, cplx a) { return cplx{s * a.re, s * a.im}; }
EVOMG_HD cplx sc_neg(cplx a) { return cplx{-a.re, -a.im}; }
EVOMG_HD bool sc_iszero(cplx a) { return a.re == 0.0 && a.im == 0.0; }
EVOMG_HD double sc_norm_from_dot(cplx d) { return sqrt(sqrt(d.re * d.re + d.im * d.im)); }

// ---- quotients by a divisor that is constant over a kernel (stencil centre, block pivots) --------------------------------
// The f64 division the compiler emits for a / b is
//   s0 = div_scale(b), s1 = div_scale(a);  y = rcp(s0) refined by two Newton steps (four fma);
//   m = s1 * y;  e = fma(-s0, m, s1);  q = div_fmas(e, y, m);  div_fixup(q, b, a)
// (llvm AMDGPU LowerFDIV64). y depends on b only: const_div_make forms it once per kernel with those very operations, and
// a quotient costs the remaining three — m, e, fma(e, y, m): the same roundings in the same order, so the same bits as
// a / b whenever div_scale leaves its operands alone, i.e. for normal b and a with exponent(a) > -970 and
// |exponent(a) - exponent(b)| < 768 (fields of a multigrid solve: yes; checked bit for bit by tools/check_const_div.hip).
// Outside that range (and for a = -0, which comes out as +0) the result may differ from a / b in the last place, and a
// non-finite numerator gives NaN where a / b gives inf. Neither is reachable with a result that matters: the solver loops
// stop at the first residual norm that is inf or NaN (status DIVERGED, same test on both sides), the norm — a sum of
// squares — is inf from |r| ~ 1e154 on, and the divisors here are 4/h^2 ... 6/h^2 and block pivots of that size, so a
// numerator would have to grow from below 1e154 past 2^768 * b ~ 1e237 within ONE cycle to be rescaled; a field that has
// already overflowed makes the norm non-finite either way (inf on the CPU, NaN here) and ends the solve at the same
// iteration with the same status. tests/test_parity_gpu.py::test_diverging_candidate_3d_tiles pins that.
// Complex: plain sc_div.
template <typename T> struct ConstDiv { T b, y; };
__device__ __forceinline__ ConstDiv<double> const_div_make(double b) {
  double y = __builtin_amdgcn_rcp(b);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  return ConstDiv<double>{b, y};
}
__device__ __forceinline__ double const_div(double a, const ConstDiv<double>& d) {
  const double m = a * d.y;
  return __builtin_fma(__builtin_fma(-d.b, m, a), d.y, m);
}
__device__ __forceinline__ ConstDiv<cplx> const_div_make(cplx b) { return ConstDiv<cplx>{b, b}; }
__device__ __forceinline__ cplx const_div(cplx a, const ConstDiv<cplx>& d) { return sc_div(a, d.b); }

template <typename T> struct scalar_traits;
template <> struct scalar_traits<double> {
  static EVOMG_HD double zero() { return 0.0; }
  static EVOMG_HD double from_real(double r) { return r; }
  static EVOMG_HD double from_c64(double re, double) { return re; }
  static constexpr bool is_complex = false;
};
template <> struct scalar_traits<cplx> {
  static EVOMG_HD cplx zero() { return cplx{0.0, 0.0}; }
  static EVOMG_HD cplx from_real(double r) { return cplx{r, 0.0}; }
  static EVOMG_HD cplx from_c64(double re, double im) { return cplx{re, im}; }
  static constexpr bool is_complex = true;
};

// star stencil: centre, -x, +x, -y, +y, -z, +z (include/evomg.h)
template <typename T> struct Stencil {
  T c[7];
};
