"""Expression-graph IR consumed by the backend (the input format of the hot path).

Own, minimal restatement of the node types the reference's lowering dispatches on
(reference: evostencils/ir/{base,system,smoother,partitioning}.py). The lowering in
``evostencils_amd.code_generation`` is duck-typed on class *names* and attributes, so DAGs built
from the reference's own classes are accepted as well.
"""
