"""CPU oracle for the kraken2 classify path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package
(as the checker).  PARITY UNPINNED vs kraken2: see oracle/k2_oracle.h.
"""
