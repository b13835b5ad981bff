"""m3p2i_aip_amd -- MI355X-native MPPI / M3P2I command() hot path.

Host side stays Python (mirrors the reference's M3P2I / Objective / IsaacGymWrapper plugin
API); the rollout, cost and update arithmetic runs in hand-written HIP kernels behind the
C-ABI of ``include/m3p2i_hip.h`` (``m3p2i_aip_amd/lib/libm3p2i_hip.so``).
"""
__version__ = "0.1.0"
