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
