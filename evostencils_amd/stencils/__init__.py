"""Constant-coefficient and periodic stencil containers used by the IR and the lowering
(reference: evostencils/stencils/{constant,multiple}.py). Only what the hot path needs."""
