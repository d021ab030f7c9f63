"""evostencils_amd — MI355X-native multigrid fitness-evaluation backend.

Drop-in for ONE path of jonas-schmitt/evostencils: the generate -> compile -> run -> parse
fitness evaluation of ``evostencils.code_generation.exastencils.ProgramGenerator``
(reference: evostencils/code_generation/exastencils.py:39-537). The candidate multigrid cycle
(an IR DAG) is lowered to a flat op program and executed by hand-written HIP kernels for gfx950
behind the C-ABI declared in ``include/evomg.h``.
"""
__version__ = "0.1.0"

import os as _os

# Candidates in flight on one GPU (population.ConcurrentEvaluator) each drive their own HIP stream. The HIP runtime maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue run IN ORDER behind each
# other — a candidate's 1 ms single-workgroup coarse solve then blocks another candidate's sweeps. Measured on MI355X
# (profiles/r03_throughput_queues.md): 6 candidates in flight 182 solves/s on 4 queues, 12 in flight on 16 queues 255;
# round 5, with the coarse solve at two thirds of its former time (profiles/r05_throughput.md): 12 on 16 queues 332, 16 on 24
# queues 356, 20 on 24 314 — the population of BASELINE config C5 47.5 -> 53.8 solves/s.
# The variable is read when the runtime initialises, i.e. it has to be in the environment before the first HIP call of
# the process: importing this package first is enough; a value the user set wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
