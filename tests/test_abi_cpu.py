"""CPU-side checks of the boundary: libgaq.so loads, exports every symbol include/gaq.h declares, struct
layouts agree between header, library and binding, and the product fails loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

from gym_art_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "gaq.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gaq_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(n for n, _, _ in _lib.SYMBOLS) == declared        # binding covers the whole header
    assert lib.gaq_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_the_header():
    import subprocess
    import tempfile
    code = ('#include <stdio.h>\n#include <stddef.h>\n#include "gaq.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(gaq_config),'
            ' sizeof(gaq_model), sizeof(gaq_rew_coeff), offsetof(gaq_config, rew), offsetof(gaq_config, model), sizeof(gaq_quad_params),'
            ' sizeof(gaq_randomizer), offsetof(gaq_randomizer, base), offsetof(gaq_config, swarm));return 0;}')
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "t.c"), os.path.join(td, "t")
        open(src, "w").write(code)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, src])     # header is plain C
        out = subprocess.check_output([exe]).decode().split()
    assert [int(x) for x in out] == [C.sizeof(_lib.GaqConfig), C.sizeof(_lib.GaqModel), C.sizeof(_lib.GaqRewCoeff),
                                     _lib.GaqConfig.rew.offset, _lib.GaqConfig.model.offset, C.sizeof(_lib.GaqQuadParams),
                                     C.sizeof(_lib.GaqRandomizer), _lib.GaqRandomizer.base.offset, _lib.GaqConfig.swarm.offset]
    assert C.sizeof(_lib.GaqQuadParams) == 8 * 40
    assert C.sizeof(_lib.GaqModel) == 8 * _lib.MODEL_DOUBLES == 8 * 33


def test_plain_c_client_builds_against_the_header_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/c_abi_demo.c: gcc, include/gaq.h, -lgaq -- nothing else.  On a GPU-less host it must refuse (exit 2)."""
    import subprocess
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + os.path.join(ROOT, "gym_art_amd"), "-lgaq",
                           "-Wl,-rpath," + os.path.join(ROOT, "gym_art_amd"), "-lm"])
    if _no_gpu():
        r = subprocess.run([exe, "8", "2"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 2 and "no CPU path" in r.stderr


def _no_gpu():
    return _lib.load().gaq_num_devices() == 0


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a GPU-less host")
def test_fails_loudly_without_a_gpu():
    from gym_art_amd import QuadrotorEnv
    with pytest.raises(_lib.GaqError, match="no CPU fallback"):
        QuadrotorEnv(num_envs=4)


def test_argument_validation_needs_no_gpu():
    lib = _lib.load()
    cfg = _lib.GaqConfig()
    h = C.c_void_p()
    assert lib.gaq_create(C.byref(cfg), C.byref(h)) == -1            # struct_size / version not filled in
    assert b"mismatch" in lib.gaq_last_error()
    cfg.struct_size, cfg.abi_version, cfg.num_envs = C.sizeof(cfg), _lib.ABI_VERSION, 0
    assert lib.gaq_create(C.byref(cfg), C.byref(h)) == -1 and b"num_envs" in lib.gaq_last_error()
    cfg.num_envs, cfg.sim_freq, cfg.sim_steps, cfg.ep_len = 8, 200.0, 2, 70000
    assert lib.gaq_create(C.byref(cfg), C.byref(h)) == -1 and b"ep_len" in lib.gaq_last_error()
    cfg.ep_len, cfg.control = 500, 7
    assert lib.gaq_create(C.byref(cfg), C.byref(h)) == -1 and b"control" in lib.gaq_last_error()
    with pytest.raises(ValueError):
        _lib.check(-1)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under gym_art_amd/ may import it or the host harness."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gym_art_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "host_harness/" not in txt.replace(
                    "tests/host_harness", ""), f


def test_host_side_option_validation_needs_no_gpu():
    """Constructor options are checked on the host before anything touches the device."""
    from gym_art_amd import QuadrotorEnv, QuadrotorEnvMulti
    with pytest.raises(ValueError, match="precision"):
        QuadrotorEnv(precision="fp16")
    with pytest.raises(ValueError, match="power of two"):
        QuadrotorEnvMulti(num_agents=6, num_worlds=2)
    with pytest.raises(TypeError, match="unknown swarm option"):
        QuadrotorEnv(num_envs=8, swarm={"agents": 8, "radius": 1.0})
    with pytest.raises(ValueError, match="multiples of the number of agents"):
        QuadrotorEnv(num_envs=12, swarm={"agents": 8})
    with pytest.raises(TypeError):
        QuadrotorEnvMulti(num_agents=8, num_worlds=2, num_envs=16)
    prm = QuadrotorEnv._parse_swarm({"agents": 4, "prox_dist": 2.0}, 16, 8)
    assert prm["agents"] == 4 and prm["prox_dist"] == 2.0 and prm["collision_dist"] is None and prm["w_collision"] == 1.0
    with pytest.raises(AttributeError):
        QuadrotorEnv(obs_repr="xyz_vxyz_euler_omega")           # broken beyond repair in the reference, absent here
    with pytest.raises(NotImplementedError):
        QuadrotorEnv(tf_control=True)
    with pytest.raises(NotImplementedError, match="no CPU backend"):
        QuadrotorEnv(backend="cpu")                              # one product path: the HIP library
    with pytest.raises(ValueError, match="at least one device"):
        QuadrotorEnv(num_envs=128, device_ids=[])


def test_fork_class_has_the_forks_constructor():
    """gym_art_amd.quadrotor_multi.QuadrotorEnv mirrors the fork's constructor (gym_art/quadrotor_multi/quadrotor_multi.py:665-670):
    the same argument names in the same order with the same defaults (then this build's extensions as keywords)."""
    import inspect
    from gym_art_amd.quadrotor_multi import QuadrotorEnv as ForkEnv
    from gym_art_amd.quadrotor import GRAV
    want = [("dynamics_params", "defaultquad"), ("dynamics_change", None), ("dynamics_randomize_every", None), ("dyn_sampler_1", None),
            ("dyn_sampler_2", None), ("raw_control", True), ("raw_control_zero_middle", True), ("dim_mode", '3D'), ("tf_control", False),
            ("sim_freq", 200.), ("sim_steps", 2), ("obs_repr", "xyz_vxyz_R_omega"), ("ep_time", 4), ("obstacles_num", 0), ("room_size", 10),
            ("init_random_state", False), ("rew_coeff", None), ("sense_noise", None), ("verbose", False), ("gravity", GRAV),
            ("resample_goal", False), ("t2w_std", 0.005), ("t2t_std", 0.0005), ("excite", False), ("dynamics_simplification", False)]
    ps = list(inspect.signature(ForkEnv.__init__).parameters.values())[1:]
    assert [(p.name, p.default) for p in ps[:len(want)]] == want
    assert ps[len(want)].kind is inspect.Parameter.VAR_KEYWORD
