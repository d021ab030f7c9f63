// impl_real.hip — Impl<double>: real scalar problems (2D / 3D Poisson and everything Dirichlet).
#include "evomg_impl.h"

namespace evomg_detail {
ImplBase* make_impl_real(const evomg_problem_t& p, int device) { return new Impl<double>(p, device); }
}  // namespace evomg_detail
