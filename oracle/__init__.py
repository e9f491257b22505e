"""CPU oracle for the Bit-Swap hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under bitswap_amd/ does.
"""
from .oracle import *  # noqa: F401,F403
