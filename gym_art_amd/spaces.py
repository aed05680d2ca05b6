"""Observation / action space containers.

Uses gym's (or gymnasium's) `spaces.Box` and `EnvSpec` when one of them is importable, so that
Garage-style code doing isinstance checks keeps working; otherwise small stand-ins with the same
attributes (low, high, shape, dtype, sample, contains) are used -- neither package is installed in
the build image.
"""
import numpy as np


class _Box(object):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.low = np.asarray(low, dtype=self.dtype)
        self.high = np.asarray(high, dtype=self.dtype)
        self.shape = self.low.shape

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return "Box(%s, %s)" % (self.shape, self.dtype)


class _EnvSpec(object):
    def __init__(self, id, max_episode_steps=None, **kw):
        self.id = id
        self.max_episode_steps = max_episode_steps


def _pick():
    for mod in ("gym", "gymnasium"):
        try:
            m = __import__(mod)
            box = m.spaces.Box
            try:
                reg = __import__(mod + ".envs.registration", fromlist=["EnvSpec"])
                return box, reg.EnvSpec
            except Exception:
                return box, _EnvSpec
        except Exception:
            continue
    return _Box, _EnvSpec


Box, EnvSpec = _pick()


def _env_base():
    """`gym.Env` when the classic gym is importable (the reference subclasses it, quadrotor.py:647, and wrappers such as
    Garage's check isinstance); plain `object` otherwise.  gymnasium's Env is NOT used: its reset/step signatures differ
    from the reference's, which this class keeps."""
    try:
        import gym
        return gym.Env
    except Exception:
        return object


EnvBase = _env_base()
