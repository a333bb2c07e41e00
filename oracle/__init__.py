"""CPU restatements of the reference semantics (test infrastructure only; PARITY UNPINNED, see
nm_oracle.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package."""
