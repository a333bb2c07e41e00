"""CPU restatements of the reference semantics (test infrastructure only; parity PINNED to the reference's own code executed on
a test-side TensorFlow stand-in, see nm_oracle.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package."""
