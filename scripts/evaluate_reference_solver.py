"""Counterpart of the reference's scripts/evaluate_reference_solver.py: runs the problem's own reference solver
(the V-cycle its `.exa3` file configures) `nruns` times and prints the two averages that script prints. No JVM, no
make: the solver is an op program executed by the HIP backend.

usage: python scripts/evaluate_reference_solver.py [base_path settings_path knowledge_path] [--problem NAME --levels LO HI]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="*", help="base_path settings_path knowledge_path (as the reference's ProgramGenerator)")
    ap.add_argument("--problem", default="2D_FD_Poisson_fromL2", help="built-in descriptor when no paths are given")
    ap.add_argument("--levels", type=int, nargs=2, metavar=("MIN", "MAX"))
    ap.add_argument("--nruns", type=int, default=20)
    args = ap.parse_args()
    from evostencils_amd import native, problem_files, problems, reference_solver
    if len(args.paths) == 3:
        pd = problem_files.load_problem(*args.paths)
        if args.levels:
            pd = pd.with_levels(*args.levels)
    else:
        make = problems.BY_NAME[args.problem]
        pd = make(*args.levels) if args.levels else make()
    ops = reference_solver.reference_solver_ops(pd)
    handle = native.create(pd)
    out = handle.evaluate(ops, args.nruns, 0, 0)
    iterations = out.n_iterations if out.status == 0 else pd.max_iterations
    print(f"Average solving time: {out.time_ms}", flush=True)
    print(f"Average number of iterations: {iterations}", flush=True)
    handle.close()


if __name__ == "__main__":
    main()
