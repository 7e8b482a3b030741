"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's hot path, used as the parity checker by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under the product package
(self-diagnosing-gan_amd/) may import this package.
"""
