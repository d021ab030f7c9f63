"""Counterpart of the reference's scripts/evaluate_evolved_solver.py: reads a stored individual (the `str(individual)`
prefix string the optimizer writes to hof_*/individual_*.txt), rebuilds the IR with the grammar's productions and
evaluates it (20 samples, as Optimizer.generate_and_evaluate_program_from_grammar_representation,
optimization/program.py:904-933).

usage: python scripts/evaluate_evolved_solver.py INDIVIDUAL.txt [base_path settings_path knowledge_path]
       [--problem NAME --levels LO HI]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("individual")
    ap.add_argument("paths", nargs="*")
    ap.add_argument("--problem", default="2D_FD_Poisson_fromL2")
    ap.add_argument("--levels", type=int, nargs=2, metavar=("MIN", "MAX"))
    ap.add_argument("--samples", type=int, default=20)
    args = ap.parse_args()
    from evostencils_amd import problems
    from evostencils_amd.code_generation.hip import ProgramGenerator
    from evostencils_amd.grammar import productions
    from evostencils_amd.ir import base, system
    if len(args.paths) == 3:
        gen = ProgramGenerator("", *args.paths)
        if args.levels:
            gen.reinitialize(args.levels[0], args.levels[1], {})
    else:
        make = problems.BY_NAME[args.problem]
        gen = ProgramGenerator(problem=make(*args.levels) if args.levels else make())
    lo, hi = gen.min_level, gen.max_level
    approximation = system.Approximation("x", [base.Approximation(f.name, g) for f, g in zip(gen.fields, gen.finest_grid)])
    top = [eq for eq in gen.equations if eq.level == hi]
    rhs = system.RightHandSide("b", [base.RightHandSide(eq.rhs_name, g) for eq, g in zip(top, gen.finest_grid)])
    ctx = productions.generate_context(approximation, rhs, gen.dimension, gen.coarsening_factor, hi, gen.equations,
                                       gen.operators, gen.fields, depth=hi - lo)
    with open(args.individual, "r") as fh:
        grammar_string = fh.read()
    expression, _ = productions.compile_individual(grammar_string, ctx)
    storages = gen.generate_storage(lo, hi, gen.finest_grid)
    gen.initialize_code_generation(lo, hi)
    solving_time, convergence_factor, number_of_iterations = gen.generate_and_evaluate(
        expression, storages, lo, hi, "", infinity=1e100, evaluation_samples=args.samples)
    print(f"Solving Time: {solving_time}, Convergence factor: {convergence_factor}, "
          f"Number of Iterations: {number_of_iterations}", flush=True)
    gen.close()


if __name__ == "__main__":
    main()
