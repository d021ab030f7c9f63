"""The problems' own reference solvers (the V-cycle each `.exa3` file configures) as op programs, and the two
command-line counterparts of the reference's scripts/evaluate_reference_solver.py / evaluate_evolved_solver.py."""
import os
import runpy
import sys

import pytest

from evostencils_amd import abi, problem_files, problems, reference_solver

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = os.path.join(REPO, "tests", "golden", "problems")


def test_reference_solver_program_follows_the_hints():
    ops = reference_solver.reference_solver_ops(problems.poisson_2d(5, 9))     # exa3: 2 pre, 1 post, RB-GS 1.15
    top = [o for o in ops if o.level == 9 and o.opcode == abi.OP_SMOOTH]
    assert len(top) == 3 and all(o.partitioning == abi.PART_RED_BLACK and o.omega == 1.15 for o in top)
    assert sum(o.opcode == abi.OP_COARSE_SOLVE for o in ops) == 1
    assert reference_solver.reference_solver_ops(problems.poisson_3d(2, 4))[0].omega == 1.25
    p = problem_files.load_problem(FIXTURES, "Diffusion/2D_FD_AnisotropicDiffusion.settings",
                                   "Diffusion/2D_FD_AnisotropicDiffusion.knowledge")
    ops = reference_solver.reference_solver_ops(p)                               # the fixture asks for V(2,2), w = 1.0
    top = [o for o in ops if o.level == p.max_level and o.opcode == abi.OP_SMOOTH]
    assert len(top) == 4 and top[0].omega == 1.0
    jac = reference_solver.reference_solver_ops(p, {"solver_smoother_jacobiType": "true", "solver_smoother_damping": 0.8})
    assert all(o.partitioning == abi.PART_SINGLE for o in jac if o.opcode == abi.OP_SMOOTH)


def test_reference_solvers_converge_on_the_oracle():
    from oracle import oracle
    for pd, bound in ((problems.poisson_2d(3, 7), 0.1), (problems.poisson_3d(2, 5), 0.2)):
        out = oracle.create(pd).evaluate(reference_solver.reference_solver_ops(pd), 1, 0, 128)
        assert out.status == abi.SOLVE_CONVERGED and out.conv_factor < bound


@pytest.mark.gpu
def test_evaluation_scripts_run_against_the_backend(tmp_path, capsys, monkeypatch):
    from test_individual_strings import TUTORIAL_INDIVIDUAL
    individual = tmp_path / "individual_0.txt"
    individual.write_text(TUTORIAL_INDIVIDUAL)
    monkeypatch.setattr(sys, "argv", ["evaluate_evolved_solver.py", str(individual), "--levels", "5", "9", "--samples", "3"])
    runpy.run_path(os.path.join(REPO, "scripts", "evaluate_evolved_solver.py"), run_name="__main__")
    line = capsys.readouterr().out.strip().splitlines()[-1]
    # the stored individual is the reference's recorded best individual: same line as notebooks/tutorial.ipynb:3373
    assert line.startswith("Solving Time: ") and "Convergence factor: 0.9212764170210773" in line and "Number of Iterations: 100.0" in line
    monkeypatch.setattr(sys, "argv", ["evaluate_reference_solver.py", FIXTURES, "Harmonic/3D_FD_Harmonic.settings",
                                      "Harmonic/3D_FD_Harmonic.knowledge", "--nruns", "3"])
    runpy.run_path(os.path.join(REPO, "scripts", "evaluate_reference_solver.py"), run_name="__main__")
    lines = capsys.readouterr().out.strip().splitlines()
    assert lines[-2].startswith("Average solving time: ") and lines[-1].startswith("Average number of iterations: ")
    assert 3 <= float(lines[-1].split(": ")[1]) <= 40
