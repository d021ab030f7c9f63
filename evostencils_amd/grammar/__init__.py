"""Problem-description data classes shared by the optimizer side and the backend
(reference: evostencils/grammar/multigrid.py:15-71,122-173). The G3P search space itself stays in
the reference's Python optimizer and is out of scope here."""
