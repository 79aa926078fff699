from .soccer_env import SoccerSimultaneousEnv
from .vector_env import VectorSoccerEnv

__all__ = ["SoccerSimultaneousEnv", "VectorSoccerEnv"]
