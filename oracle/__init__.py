"""CPU parity oracle for the DFU3D pseudo-box path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  See oracle/penet_oracle.py for the pinning
statement.
"""
