// impl_complex.hip — Impl<cplx>: complex scalar problems (2D Helmholtz with Robin columns).
#include "evomg_impl.h"

namespace evomg_detail {
ImplBase* make_impl_complex(const evomg_problem_t& p, int device) { return new Impl<cplx>(p, device); }
}  // namespace evomg_detail
