"""MI355X-native rollout + update engine behind the TCE/BBRL agent API.

The hot path runs as hand-written HIP kernels for gfx950 in libtce_hip.so
(C ABI: include/tce_hip.h), reached through ``tce_rl_amd.ops``.
"""
__version__ = "0.1.0"
