"""oracle/ — TEST INFRASTRUCTURE: the CPU parity checkers for the wavefront-alignment hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under pywfa_amd/ does (the product path fails loudly without the HIP library).
"""
