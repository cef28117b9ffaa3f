"""CPU oracle for the SimT hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(simt_amd/) never does.  See oracle/simt_oracle.py for the restatement and oracle/gen_golden.py for how it is
pinned against the reference.
"""
