"""rl4co_amd — MI355X-native autoregressive rollout engine for the RL4CO hot path.

The package mirrors the reference's Python surfaces for that path only
(``RL4COEnvBase.reset/step/get_reward`` and ``AutoregressivePolicy.forward``) on top of
hand-written HIP kernels for gfx950 reached through a C-ABI (``include/rl4co_amd.h``).
See DESIGN.md for the path, the boundary and what is out of scope.
"""
__version__ = "0.1.0"
