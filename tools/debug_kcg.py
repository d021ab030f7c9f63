"""Developer tool: the launch-per-iteration coarse CG (kcg_*) against the oracle for growing iteration caps."""
import os
os.environ.setdefault("OMP_NUM_THREADS", "4")
import sys
sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np
from evostencils_amd import abi, native, problems
from oracle import oracle
import dataclasses
for dim, lo, maxit in ((2, 8, 1), (2, 8, 2), (2, 8, 5), (2, 8, 1000)):
    pd = dataclasses.replace(problems.poisson_2d(lo, lo + 1), cgs_max_iterations=maxit)
    rng = np.random.default_rng(lo * 10 + dim)
    shape = ((1 << lo) + 1,) * dim
    rhs = rng.standard_normal(shape)
    x0 = rng.standard_normal(shape) * 1e-3
    x0[0,:]=x0[-1,:]=x0[:,0]=x0[:,-1]=0
    h = native.create(pd); ho = oracle.create(pd)
    for hh in (h, ho):
        hh.set_field(lo, abi.F_CGS_RHS, rhs); hh.set_field(lo, abi.F_CORR, x0); hh.run_ops([abi.make_op(abi.OP_COARSE_SOLVE, lo)])
    a, b = h.get_field(lo, abi.F_CORR), ho.get_field(lo, abi.F_CORR)
    ra, rb = h.get_field(lo, abi.F_RES), ho.get_field(lo, abi.F_RES)
    print("maxit", maxit, "x equal", np.array_equal(a, b), "max |dx|", np.abs(a-b).max(), "rel", np.abs(a-b).max()/np.abs(b).max(), "r equal", np.array_equal(ra[1:-1,1:-1], rb[1:-1,1:-1]), np.abs(ra-rb)[1:-1,1:-1].max())
