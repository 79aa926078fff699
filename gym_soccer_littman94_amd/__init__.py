"""MI355X-native batched Littman-94 grid soccer (step/reset hot path of mimoralea/gym-soccer-littman94).

Host side: Python mirroring the reference's SoccerSimultaneousEnv surface; device side: hand-written
HIP kernels for gfx950 behind a C ABI (libsoccer_hip.so, include/soccer_hip.h).
"""
from . import planners, policies  # noqa: F401
from .compat import install_as_gym_soccer  # noqa: F401
from .core import DeviceArray, SoccerBatch  # noqa: F401
from .envs import SoccerSimultaneousEnv, VectorSoccerEnv  # noqa: F401
from .registration import make, register_all  # noqa: F401

register_all()

__all__ = ["SoccerBatch", "DeviceArray", "SoccerSimultaneousEnv", "VectorSoccerEnv", "make", "register_all",
           "planners", "policies", "install_as_gym_soccer"]
