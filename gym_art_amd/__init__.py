"""gym_art_amd -- MI355X-native batched quadrotor simulator with the Gym surface of
amolchanov86/gym_art's `QuadrotorEnv` (include/gaq.h)."""
from .quadrotor import QuadrotorEnv  # noqa: F401
from .quadrotor_multi import QuadrotorEnvMulti  # noqa: F401

__all__ = ["QuadrotorEnv", "QuadrotorEnvMulti"]
