"""CPU oracle for the MPPI hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this package; the product (benchnav_amd/) never does.
"""
