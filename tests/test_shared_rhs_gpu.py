"""The finest level's right-hand side is the same array for every candidate of a problem and read-only to a multigrid solve: the
handles of a process read ONE device copy of it (csrc/evomg_runtime.h: shared_array_acquire; round 6 — sixteen C2 solves in
flight 421 -> 439 solves/s). What must hold: sharing changes no result; a handle that writes RHS@max (set_field, an op program
that names it as a destination, a second upload) gets an array of its own first and leaves the others' untouched; the shared
copy outlives the handle that uploaded it."""
import dataclasses

import numpy as np
import pytest

from evostencils_amd import abi, candidates, problems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()
    return nat


LO, HI = 3, 7


def _ops():
    return candidates.v_cycle_ops(LO, HI, 1, 1, omega=1.0)


def test_sharing_changes_no_result_and_survives_the_uploader(native, monkeypatch):
    pd = problems.poisson_2d(LO, HI)
    monkeypatch.setenv("EVOMG_NO_SHARED_RHS", "1")
    h0 = native.create(pd)
    want = h0.evaluate(_ops(), 1, 0, 64)
    u0, f0 = h0.get_field(HI, abi.F_SOL), h0.get_field(HI, abi.F_RHS)
    h0.close()
    monkeypatch.delenv("EVOMG_NO_SHARED_RHS")
    a, b = native.create(pd), native.create(pd)
    ra = a.evaluate(_ops(), 1, 0, 64)
    a.close()                                            # the handle whose upload became the shared copy goes first
    rb = b.evaluate(_ops(), 1, 0, 64)
    assert np.array_equal(ra.history, want.history) and np.array_equal(rb.history, want.history)
    assert np.array_equal(b.get_field(HI, abi.F_SOL), u0) and np.array_equal(b.get_field(HI, abi.F_RHS), f0)
    c = native.create(pd)                                # ... and a third one still finds (or re-creates) it
    assert np.array_equal(c.evaluate(_ops(), 1, 0, 64).history, want.history)
    b.close()
    c.close()


def test_a_writer_gets_its_own_array(native, monkeypatch):
    pd = problems.poisson_2d(LO, HI)
    a, b, c = native.create(pd), native.create(pd), native.create(pd)
    f = b.get_field(HI, abi.F_RHS)
    want = b.evaluate(_ops(), 1, 0, 64).history
    # set_field on the shared array
    a.set_field(HI, abi.F_RHS, 2.0 * f)
    assert np.array_equal(a.get_field(HI, abi.F_RHS), 2.0 * f)
    assert np.array_equal(b.get_field(HI, abi.F_RHS), f) and np.array_equal(c.get_field(HI, abi.F_RHS), f)
    # an op program that names RHS@max as its destination
    c.run_ops([abi.make_op(abi.OP_SET_ZERO, HI, abi.F_RHS)])
    assert not c.get_field(HI, abi.F_RHS)[1:-1, 1:-1].any()              # (the op zeroes the inner nodes)
    assert np.array_equal(b.get_field(HI, abi.F_RHS), f)
    assert np.array_equal(b.evaluate(_ops(), 1, 0, 64).history, want)
    # evaluate() restores the handle's initial fields: c solves the shared problem again, from its own copy
    assert np.array_equal(c.evaluate(_ops(), 1, 0, 64).history, want)
    assert np.array_equal(c.get_field(HI, abi.F_RHS), f)
    # a second upload to one handle leaves the others alone
    a.set_initial_field(HI, abi.F_RHS, 3.0 * f)
    assert np.array_equal(a.get_field(HI, abi.F_RHS), 3.0 * f) and np.array_equal(b.get_field(HI, abi.F_RHS), f)
    ha = a.evaluate(_ops(), 1, 0, 64).history
    assert np.array_equal(b.evaluate(_ops(), 1, 0, 64).history, want)
    for h in (a, b, c):
        h.close()
    monkeypatch.setenv("EVOMG_NO_SHARED_RHS", "1")      # the same second upload on a handle that never shared anything
    d = native.create(pd)
    d.set_initial_field(HI, abi.F_RHS, 3.0 * f)
    assert np.array_equal(d.evaluate(_ops(), 1, 0, 64).history, ha) and not np.array_equal(ha, want)
    d.close()


def test_the_preconditioned_outer_solver_keeps_private_arrays(native):
    """RHS@max is a work array of the cycle there (the preconditioner's input): never shared."""
    pd = dataclasses.replace(problems.helmholtz_2d(3, 6, k=0.625 * 64), max_iterations=4)
    ops = candidates.v_cycle_ops(3, 6, 2, 1, omega=0.6)
    a, b = native.create(pd), native.create(pd)
    ra, rb = a.evaluate(ops, 1, 0, 64), b.evaluate(ops, 1, 0, 64)
    assert np.array_equal(ra.history, rb.history)
    assert np.array_equal(a.get_field(6, abi.F_CORR), b.get_field(6, abi.F_CORR))
    a.close()
    assert np.array_equal(b.evaluate(ops, 1, 0, 64).history, ra.history)
    b.close()
