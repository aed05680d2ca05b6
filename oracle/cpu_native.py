"""ctypes driver of oracle/cpu_native.cpp: the compiled, OpenMP-threaded CPU port of the hot path (bench.py's
`cpu_baseline` leg, "cpu-c++" of BASELINE.md 4.3).  TEST / MEASUREMENT INFRASTRUCTURE ONLY (see the .cpp header).

    python -m oracle.cpu_native --envs 262144 --seconds 4 --threads 16
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from tests import hh  # noqa: E402  (ctypes mirrors of StepCfg / the model struct of quad_core.hpp)

SRC = os.path.join(HERE, "cpu_native.cpp")
OUT = os.path.join(HERE, "_native", "libcpu_native.so")
_lib = None


def build():
    """g++ -O3 -fopenmp; AVX2+FMA (every host this runs on has them; -march=native would tie the .so to the build
    container's CPU, and the file travels to the GPU box)."""
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    newest = max(os.path.getmtime(SRC), os.path.getmtime(hh.CORE))
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
        subprocess.check_call(["g++", "-std=c++17", "-O3", "-mavx2", "-mfma", "-fopenmp", "-fPIC", "-shared",
                               "-Wno-unknown-pragmas", "-o", OUT, SRC])
    return OUT


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        assert L.cn_sizeof_cfg() == C.sizeof(hh.StepCfg) and L.cn_sizeof_model() == C.sizeof(hh.HHModel)
        L.cn_create.restype = C.c_void_p
        L.cn_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.cn_destroy.argtypes = [C.c_void_p]
        L.cn_set_state.argtypes = [C.c_void_p, C.c_void_p]
        L.cn_reset.argtypes = [C.c_void_p, C.c_int]
        L.cn_step.argtypes = [C.c_void_p] * 5 + [C.c_int]
        L.cn_run.restype = C.c_int64
        L.cn_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double)]
        _lib = L
    return _lib


class Batch(object):
    def __init__(self, n, const, dt=0.005, sim_steps=2, ep_len=500, noise=0, auto_reset=0, seed=0, **cfg_kw):
        self.h = None
        self.L = lib()
        self.n = n
        self.model = hh.make_model(const)
        self.cfg = hh.make_cfg(dt, sim_steps, ep_len, self.model, noise=noise, auto_reset=auto_reset, **cfg_kw)
        self.cfg.seed = seed
        self.h = self.L.cn_create(C.byref(self.cfg), C.byref(self.model), n)
        self.D = self.cfg.obs_dim

    def close(self):
        if self.h:
            self.L.cn_destroy(self.h)
            self.h = None

    __del__ = close

    def set_state(self, planes):
        p = np.ascontiguousarray(planes[:39], dtype=np.float64)
        assert p.shape == (39, self.n)
        self.L.cn_set_state(self.h, p.ctypes.data)

    def reset(self, threads=1):
        self.L.cn_reset(self.h, threads)

    def step(self, actions, threads=1):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n, 4)
        obs = np.empty((self.n, self.D), np.float32)
        rew = np.empty(self.n, np.float32)
        done = np.empty(self.n, np.uint8)
        self.L.cn_step(self.h, a.ctypes.data, obs.ctypes.data, rew.ctypes.data, done.ctypes.data, threads)
        return obs, rew, done.astype(bool)

    def run(self, seconds, threads, ring=4, seed=0):
        rng = np.random.RandomState(seed)
        acts = rng.uniform(-1, 1, size=(ring, self.n, 4)).astype(np.float32)
        el = C.c_double(0)
        steps = self.L.cn_run(self.h, acts.ctypes.data, ring, float(seconds), int(threads), C.byref(el))
        return int(steps), el.value


# Hummingbird ("DefaultQuad") derived constants in the golden fixtures' `const_*` naming (values: SURVEY 8a2, checked
# against the reference in tests/test_quad_params.py)
HUMMINGBIRD = dict(mass=0.816, inertia=(3.746575e-3, 3.746575e-3, 6.149342e-3), thrust_max=(5.603472,) * 4,
                   torque_max=(0.2801736,) * 4,
                   prop_pos=((0.12, -0.12, 7.174e-3), (-0.12, -0.12, 7.174e-3), (-0.12, 0.12, 7.174e-3), (0.12, 0.12, 7.174e-3)),
                   damp_time_up=0.0, damp_time_down=0.0, motor_linearity=1.0, arm=0.169706, thrust_noise_sigma=0.01,
                   vel_damp=0.0, damp_omega_quadratic=0.0, C_rot_drag=0.0, C_rot_roll=0.0)


def usable_cores():
    """(cpus the host reports, cpus in this process's affinity mask, cgroup cpu quota or None)"""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except Exception:
        pass
    return os.cpu_count() or 1, aff, quota


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1 << 18)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    threads = args.threads or usable_cores()[1]
    b = Batch(args.envs, HUMMINGBIRD, noise=1, auto_reset=1)
    b.reset(threads)
    b.run(0.3, threads)                         # first touch / thread pool warm-up
    steps, el = b.run(args.seconds, threads)
    print(json.dumps({"envs": args.envs, "steps": steps, "seconds": el, "threads": threads,
                      "env_steps_per_s": args.envs * steps / el}))
