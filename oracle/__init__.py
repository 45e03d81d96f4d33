"""oracle/ -- CPU checkers (TEST INFRASTRUCTURE).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; the product never does."""
