"""ORACLE -- test infrastructure only.

CPU restatement of the reference's algorithm for the hot path, used
as the checker by tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg.
Nothing under `wsovod_amd/` (the product path) imports this package.
"""
