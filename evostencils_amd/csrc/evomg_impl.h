// evomg_impl.h — the scalar (real / complex) implementation class: device field storage, the op-program interpreter with
// its peephole fusions, the solver loops. The class body is spread over impl_*.inl by concern; impl_real.hip and
// impl_complex.hip instantiate it for double and cplx.
#pragma once
#include "evomg_runtime.h"
#include "evomg_kernels.h"
#include "evomg_krylov.h"
#include "evomg_cg1.h"
#include "evomg_strip.h"
#include "evomg_strip_cross.h"
#include "evomg_small2d.h"
#include "evomg_small3d.h"
#include "evomg_tile3d.h"
#include "evomg_batch.h"
#include "evomg_strip_block.h"
#ifndef EVOMG_FUSED_PF
#define EVOMG_FUSED_PF 2   // prefetch depth of the fused sweeps (fewer waves per SIMD than the plain ones)
#endif

namespace evomg_detail {

template <typename T> struct type_name;
template <> struct type_name<double> { static const char* get() { return "double"; } };
template <> struct type_name<cplx> { static const char* get() { return "cplx"; } };

template <typename T> struct Level {
  Geo g{};
  size_t elems = 0;             // allocated elements (pitch * rows)
  T* f[EVOMG_F_COUNT] = {};
  T* init[EVOMG_F_COUNT] = {};  // device copies of the initial contents (nullptr = zero)
  T* alt = nullptr;             // ping-pong twin for out-of-place smoothers
  const T* alt_mirror = nullptr;   // array whose boundary values `alt` is known to share (nullptr: unknown)
  T* tmp[kTmpVectors] = {};     // Krylov work vectors, allocated on demand
  Stencil<T> st{};
};

template <typename T> class Impl : public ImplBase {
 public:
#include "impl_api.inl"

 private:
#include "impl_storage.inl"
#include "impl_peephole.inl"
#include "impl_interp.inl"
#include "impl_launch2d.inl"
#include "impl_launch3d.inl"
#include "impl_coarse.inl"
#include "impl_solvers.inl"
#include "impl_members.inl"
};

}  // namespace evomg_detail
