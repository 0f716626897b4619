"""CPU oracle of the sampling and training paths: TEST INFRASTRUCTURE (see the module docstrings).  Only tests/,
__graft_entry__.smoke() and bench.py cpu_baseline import from here."""
