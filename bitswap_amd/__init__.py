"""bitswap_amd -- MI355X-native Bit-Swap / BB-ANS compression hot path.

Hand-written HIP kernels (csrc/) behind a C ABI (include/bitswap_hip.h) for the discretized
logistic CDF tables and the rANS push/pop steps, plus the host-side mirror of the reference's
`ANS` / `Model` / bins interfaces and the batched chain driver.
"""
__version__ = "0.1.0"
