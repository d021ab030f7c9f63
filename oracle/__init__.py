"""CPU oracle — TEST INFRASTRUCTURE ONLY. See oracle/evomg_oracle.c for what it restates and why
its parity with ExaStencils is unpinned. Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package."""
