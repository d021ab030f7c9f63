"""The backend: counterpart of evostencils/code_generation/ (exastencils.py + parser.py)."""
