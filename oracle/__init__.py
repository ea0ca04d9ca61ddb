"""CPU oracle for the GPE eigenvalue-residual path.  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the product."""
