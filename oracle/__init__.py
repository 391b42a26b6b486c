"""CPU oracles for the artdeco_amd hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything from this package, and only as the checker.  The product path
(artdeco_amd/) never imports it and has no CPU fallback.
"""
