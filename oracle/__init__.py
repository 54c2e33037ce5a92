"""CPU oracle of the hot path -- TEST INFRASTRUCTURE ONLY (see oracle.cpp header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
