"""Import shims for running the *unmodified* reference (gym_art) in the build
container, where gym / transforms3d / noise / pyglet / tensorflow are absent.

Container-only tooling: used by make_golden.py to generate the fixtures in this
directory.  None of the stubbed packages does arithmetic on the hot path
(SURVEY.md §8c): gym supplies base classes/spaces, transforms3d is only used by
an un-called random-state helper, noise/pyglet only at render() time and
tensorflow only for tf_control=True.

Nothing here is imported by the product (gym_art_amd) or by the GPU-box tests.
"""
import sys
import types

import numpy as np


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install():
    if "gym" in sys.modules and getattr(sys.modules["gym"], "_gaq_stub", False):
        return
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference

    # ---- gym ------------------------------------------------------------
    gym = _mod("gym")
    gym._gaq_stub = True

    class Env(object):
        metadata = {}
        spec = None

    class EzPickle(object):
        def __init__(self, *a, **kw):
            self._ezpickle_args = a
            self._ezpickle_kwargs = kw

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)

    class EnvSpec(object):
        def __init__(self, id, max_episode_steps=None, **kw):
            self.id = id
            self.max_episode_steps = max_episode_steps

    def np_random(seed=None):
        # gym seeds from OS entropy when seed is None (QuadrotorEnv.__init__ does that, quadrotor.py:823); a fixed
        # default keeps make_golden.py reproducible run to run (the constructor's own reset() otherwise consumes a
        # varying number of global numpy draws in its yaw rejection loop)
        rs = np.random.RandomState(20240 if seed is None else seed)
        return rs, seed

    gym.Env = Env
    utils = _mod("gym.utils")
    utils.EzPickle = EzPickle
    seeding = _mod("gym.utils.seeding")
    seeding.np_random = np_random
    utils.seeding = seeding
    spaces = _mod("gym.spaces")
    spaces.Box = Box
    envs = _mod("gym.envs")
    reg = _mod("gym.envs.registration")
    reg.EnvSpec = EnvSpec
    envs.registration = reg
    error = _mod("gym.error")

    class Error(Exception):
        pass

    error.Error = Error
    gym.utils, gym.spaces, gym.envs, gym.error = utils, spaces, envs, error

    # ---- transforms3d / noise / tensorflow --------------------------------
    t3d = _mod("transforms3d")
    t3d.euler = _mod("transforms3d.euler")
    _mod("noise")
    _mod("tensorflow")

    # ---- pyglet -----------------------------------------------------------
    pyglet = _mod("pyglet")
    pyglet.options = {}
    gl = _mod("pyglet.gl")
    gl.__all__ = []
    graphics = _mod("pyglet.graphics")

    class Group(object):
        def __init__(self, parent=None):
            self.parent = parent

    class Batch(object):
        pass

    graphics.Group = Group
    graphics.OrderedGroup = Group
    graphics.Batch = Batch
    pyglet.gl, pyglet.graphics = gl, graphics

    import matplotlib
    matplotlib.use("Agg")
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
