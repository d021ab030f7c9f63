"""Developer tool: A/B the marching-strip kernel settings in interleaved rounds inside ONE process on ONE device
(run-to-run / box-to-box spread on the pool is about +-5 %, larger than the effects being measured)."""
import os, sys, itertools, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
from evostencils_amd import abi, native, problems

level = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
one = (1, 1, 1)
ops = {"rbgs": abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_RED_BLACK, one, 1.0),
       "jacobi": abi.make_op(abi.OP_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, one, 0.8),
       "residual": abi.make_op(abi.OP_RESIDUAL, level, abi.F_RES, abi.F_SOL, abi.F_RHS)}
pd = problems.poisson_2d(level - 1, level)
configs = [("generic", None, None)] + [("strip", r, nt) for r, nt in itertools.product((16, 32, 64, 128), (0, 1))]
handles = []
for kind, rows, nt in configs:
    if kind == "generic":
        os.environ["EVOMG_NO_STRIP"] = "1"
    else:
        os.environ.pop("EVOMG_NO_STRIP", None)
        os.environ["EVOMG_STRIP_ROWS"] = str(rows)
        os.environ["EVOMG_STRIP_NT"] = str(nt)
    handles.append(native.create(pd))
times = {(i, name): [] for i in range(len(configs)) for name in ops}
for r in range(rounds):
    for i, h in enumerate(handles):
        for name, op in ops.items():
            times[(i, name)].append(h.bench_op(op, 3, 30).ms)
bytes_per = 24.0 * ((1 << level) + 1) ** 2
for i, (kind, rows, nt) in enumerate(configs):
    line = f"{kind:8s} rows={rows} nt={nt}:"
    for name in ops:
        t = times[(i, name)]
        med, best = statistics.median(t), min(t)
        line += f"  {name} med {med*1e3:6.1f} us min {best*1e3:6.1f} us ({bytes_per/med/1e6/80:.1f}%)"
    print(line, flush=True)
