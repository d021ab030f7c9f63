"""evostencils_amd — MI355X-native multigrid fitness-evaluation backend.

Drop-in for ONE path of jonas-schmitt/evostencils: the generate -> compile -> run -> parse
fitness evaluation of ``evostencils.code_generation.exastencils.ProgramGenerator``
(reference: evostencils/code_generation/exastencils.py:39-537). The candidate multigrid cycle
(an IR DAG) is lowered to a flat op program and executed by hand-written HIP kernels for gfx950
behind the C-ABI declared in ``include/evomg.h``.
"""
__version__ = "0.1.0"
