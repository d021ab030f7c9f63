"""The drop-in class on the GPU: same calls the optimizer makes, checked against the oracle."""
import os

import numpy as np
import pytest

from evostencils_amd import abi, problems
from evostencils_amd.code_generation.hip import ProgramGenerator
from helpers import build_case, context_for
import cycles

pytestmark = pytest.mark.gpu
INF = 1e100


@pytest.fixture(scope="module")
def native():
    from evostencils_amd import native as nat
    nat.library()
    return nat


def test_reference_style_construction_and_attributes(native, tmp_path):
    base = tmp_path
    (base / "Poisson").mkdir()
    (base / "Poisson" / "p.knowledge").write_text("dimensionality = 2\n\nminLevel = 4\nmaxLevel = 7\n")
    (base / "Poisson" / "p.settings").write_text('user = "x"\nconfigName = "2D_FD_Poisson_fromL2"\nbasePathPrefix = "./Poisson/"\n')
    gen = ProgramGenerator("/nonexistent/Compiler.jar", str(base), "Poisson/p.settings", "Poisson/p.knowledge",
                           "lib/linux.platform", mpi_rank=0, cycle_name="gen_mgCycle", solver_iteration_limit=500)
    assert (gen.dimension, gen.min_level, gen.max_level) == (2, 4, 7)
    assert gen.problem_name == "2D_FD_Poisson_fromL2" and gen.uses_FAS is False and gen.mpi_rank == 0
    assert gen.finest_grid[0].size == (128, 128) and gen.finest_grid[0].level == 7
    assert gen.coarsening_factor == [(2, 2)] and [f.name for f in gen.fields] == ["u"]
    assert {e.level for e in gen.equations} == {4, 5, 6, 7}
    storages = gen.generate_storage(4, 7, gen.finest_grid)
    gen.initialize_code_generation(4, 7)
    ctx = context_for(gen, 4, 7)
    expression, _ = cycles.CASES["poisson2d_L7_v21_rbgs"][3](ctx)
    t, rho, its = gen.generate_and_evaluate(expression, storages, 4, 7, "", infinity=INF, evaluation_samples=3)
    assert 0 < t < 1e4 and 0 < rho < 0.2 and its == int(its) and 5 <= its <= 20
    gen._counter, gen._average_generation_time = 0, 0        # the optimizer resets these (program.py:850-851)


@pytest.mark.parametrize("name", ["poisson2d_L8_v11_jacobi", "poisson2d_L7_blocks", "poisson3d_L4_mixed"])
def test_generate_and_evaluate_matches_oracle(native, name):
    from oracle import oracle
    gen, expression, lo, hi = build_case(name, library=native.library())
    gen.initialize_code_generation(lo, hi)
    t, rho, its = gen.generate_and_evaluate(expression, None, lo, hi, "", infinity=INF, evaluation_samples=2)
    ops = gen.generate_op_program(expression, lo, hi)
    ref = oracle.create(problems.BY_NAME[cycles.CASES[name][0]](lo, hi)).evaluate(ops, 1, 0, 128)
    rt, rrho, rits = ProgramGenerator.interpret_outcome(ref, INF)
    assert its == rits
    assert rho == pytest.approx(rrho, rel=1e-10)


def test_bad_candidates_return_sentinels_not_exceptions(native):
    gen, expression, lo, hi = build_case("poisson2d_L8_v11_jacobi", library=native.library())
    gen.initialize_code_generation(lo, hi)
    # divergent smoother: Jacobi with w = 1.9 everywhere (relaxation index 36)
    ctx = context_for(gen, lo, hi)
    bad, _ = cycles.v_cycle(ctx, 2, [[("jacobi", 36)] * 3] * 2, [[("jacobi", 36)] * 3] * 2)
    t, rho, its = gen.generate_and_evaluate(bad, None, lo, hi, "", infinity=INF, evaluation_samples=3)
    assert rho > 1                                    # what discards the candidate (exastencils.py:436-437)
    # not lowerable -> (inf, inf, inf) like a failed code generation (exastencils.py:499-510)
    from evostencils_amd.ir import base
    assert gen.generate_and_evaluate(base.Addition(expression, expression), None, lo, hi, "") == (INF, INF, INF)
    # a solver_program this generator did not produce cannot be spliced: a failed candidate, not an exception
    assert gen.generate_and_evaluate(expression, None, lo, hi, "Function gen_mgCycle@7 {\n}\n\n") == (INF, INF, INF)


def test_level_windows_and_multi_run_solver_programs(native):
    """SURVEY 8 row a7 on the GPU: a window over the top levels calls the template's generated cycle underneath; a
    second run's candidate is evaluated underneath the first run's function handed back as `solver_program`
    (optimization/program.py:811-898). Both against the oracle on the composed op program."""
    from oracle import oracle
    from helpers import window_context, generator_for
    gen = generator_for("2D_FD_Poisson_fromL2", 4, 8, library=native.library())
    gen.initialize_code_generation(4, 8)
    rb = [("rbgs", 21)]
    top, _ = cycles.v_cycle(window_context(gen, 6, 8), 2, [rb, [("jacobi", 12)]], [[("block", 14, (2, 1))], rb], cgc_index=17)
    t, rho, its = gen.generate_and_evaluate(top, None, 6, 8, "", infinity=INF, evaluation_samples=1)
    ref = oracle.create(gen.problem).evaluate(gen.generate_op_program(top, 6, 8), 1, 0, 128)
    _, want_rho, want_its = ProgramGenerator.interpret_outcome(ref, INF)
    assert its == want_its and rho == pytest.approx(want_rho, rel=1e-10) and rho < 0.5
    solver_program = gen.generate_cycle_function(top, None, 6, 8, 8)
    low, _ = cycles.v_cycle(window_context(gen, 4, 6), 2, [rb, rb], [[("rbgs", 20)], rb], cgc_index=19)
    t, rho, its = gen.generate_and_evaluate(low, None, 4, 6, solver_program, infinity=INF, evaluation_samples=1)
    ops = gen.generate_op_program(low, 4, 6, solver_program)
    assert [o.level for o in ops if o.opcode == abi.OP_COARSE_SOLVE] == [4]
    ref = oracle.create(gen.problem).evaluate(ops, 1, 0, 128)
    _, want_rho, want_its = ProgramGenerator.interpret_outcome(ref, INF)
    assert its == want_its and rho == pytest.approx(want_rho, rel=1e-10) and rho < 0.5
    gen.close()


def test_reinitialize_bumps_the_problem_size(native):
    gen, expression, lo, hi = build_case("poisson2d_L8_v11_jacobi", library=native.library())
    gen.reinitialize(lo + 1, hi + 1, {})
    assert (gen.min_level, gen.max_level) == (lo + 1, hi + 1) and gen.finest_grid[0].level == hi + 1
    gen.initialize_code_generation(lo + 1, hi + 1)
    ctx = context_for(gen, lo + 1, hi + 1)
    bigger, _ = cycles.v_cycle(ctx, 2, [[("jacobi", 14)]] * 2, [[("jacobi", 14)]] * 2)
    t, rho, its = gen.generate_and_evaluate(bigger, None, lo + 1, hi + 1, "", evaluation_samples=1)
    assert 0.3 < rho < 0.4 and its < 40


def test_helmholtz_wavenumber_sweep(native):
    """'k' in global_variable_values -> evaluated at k, 2k, 4k and averaged (exastencils.py:518-535)."""
    from test_helmholtz_oracle import reference_vcycle_ops
    from oracle import oracle
    lo, hi = 3, 6
    pd = problems.helmholtz_2d(lo, hi, k=5.0)
    gen = ProgramGenerator(problem=pd, library=native.library(), solver_iteration_limit=10000)
    gen.initialize_code_generation(lo, hi)
    ops = reference_vcycle_ops(lo, hi)
    acc = np.zeros(3)
    for k in (5.0, 10.0, 20.0):
        out = oracle.create(pd, {"k": k}).evaluate(ops, 1, 0, 10002)
        acc += np.array(ProgramGenerator.interpret_outcome(out, INF, 10000, krylov_outer=True))
    want = acc / 3
    total = np.zeros(3)
    mapping = {"k": 5.0}
    for _ in range(3):
        total += np.array(gen.compile_and_run(ops, lo, hi, mapping=mapping, evaluation_samples=1))
        mapping["k"] *= 2
    got = total / 3
    assert got[2] == want[2]
    assert got[1] == pytest.approx(want[1], rel=2e-6)     # one unit of the sixth printed digit


def test_concurrent_3d_candidates_with_multi_workgroup_coarse_solver(native):
    """Throughput mode on 3D problems: each candidate's coarse solve on 33^3 is EIGHT co-resident workgroups that
    rendezvous through global memory (evomg_cg_multi_wg3d). Six candidates in flight must neither dead-lock nor
    change any result (every handle has its own exchange area)."""
    from evostencils_amd import population
    lo, hi = 5, 6
    pd = problems.poisson_3d(lo, hi)

    def make():
        g = ProgramGenerator(problem=pd, library=native.library(), jacobi_semantics="textbook")
        g.initialize_code_generation(lo, hi)
        return g

    specs = [[("jacobi", 12)], [("rbgs", 18)], [("jacobi", 14), ("jacobi", 14)], [("rbgs", 21)], [("block", 12, (2, 1, 1))],
             [("rbgs", 16), ("jacobi", 12)]] * 2

    def evaluate(gen, spec):
        ctx = context_for(gen, lo, hi)
        expression, _ = cycles.v_cycle(ctx, hi - lo, [spec] * (hi - lo), [spec] * (hi - lo))
        return gen.generate_and_evaluate(expression, None, lo, hi, "", infinity=INF, evaluation_samples=1)

    solo = make()
    want = [evaluate(solo, s) for s in specs]
    solo.close()
    workers = population.ConcurrentEvaluator(make, 6)
    got = workers.map(evaluate, specs)
    workers.close()
    for (t0, rho0, its0), (t1, rho1, its1) in zip(want, got):
        assert its0 == its1 and rho0 == rho1 and its0 < INF
