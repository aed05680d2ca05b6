"""CPU: the (configuration -> feature mask -> kernel instantiation) map of libgaq, enumerated through the library's own
selection logic (`gaq_plan`: pure host code, no device) -- every reachable combination of the options that pick a kernel must land on
an instantiation that exists, so that a hole in the table of gaq_kernels.hpp (GAQ_STEP_ALL / GAQ_ROLL_ALL) is a test failure here and
not a runtime "internal: no kernel instantiation" on a GPU box."""
import ctypes as C
import itertools
import os
import re

import pytest

from gym_art_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HUMMINGBIRD = dict(mass=0.816, inertia=(3.746575e-3, 3.746575e-3, 6.149342e-3), thrust_max=(5.603472,) * 4, torque_max=(0.2801736,) * 4,
                   prop_pos=(0.12, -0.12, 7.174e-3, -0.12, -0.12, 7.174e-3, -0.12, 0.12, 7.174e-3, 0.12, 0.12, 7.174e-3),
                   damp_time_up=0.0, damp_time_down=0.0, linearity=1.0, arm=0.169706, ou_sigma=0.01)


def base_cfg(n=4096, **over):
    cfg = _lib.GaqConfig()
    cfg.struct_size, cfg.abi_version = C.sizeof(cfg), _lib.ABI_VERSION
    cfg.num_envs, cfg.sim_freq, cfg.sim_steps, cfg.ep_len = n, 200.0, 2, 500
    cfg.room_size, cfg.gravity = 10.0, 9.81
    cfg.rew.pos, cfg.rew.effort, cfg.rew.crash, cfg.rew.orient, cfg.rew.spin = 1.0, 0.05, 1.0, 1.0, 0.1
    for k, v in HUMMINGBIRD.items():
        if isinstance(v, tuple):
            getattr(cfg.model, k)[:] = v
        else:
            setattr(cfg.model, k, v)
    for k, v in over.items():
        obj = cfg
        *path, leaf = k.split(".")
        for part in path:
            obj = getattr(obj, part)
        setattr(obj, leaf, v)
    return cfg


def plan(cfg, lag=-1, drag=-1, every=0, cus=256):
    out = _lib.GaqPlanInfo()
    _lib.check(_lib.load().gaq_plan(C.byref(cfg), lag, drag, every, cus, C.byref(out)))
    return out


def table(name):
    src = open(os.path.join(ROOT, "gym_art_amd", "csrc", "gaq_kernels.hpp")).read()
    vals = set()
    for m in re.finditer(r"#define %s_PART\d\(X\)(.*)" % name, src):
        vals.update(int(v) for v in re.findall(r"X\((\d+)u\)", m.group(1)))
    return vals


def test_every_reachable_configuration_has_its_kernel():
    """The cross product of every option that takes part in the kernel choice (uniform and per-env models x motor lag x rotor drag x
    controller x thrust-noise source x the seven observation flags x sensor-noise models x the extras that force a generic tier x the
    three layouts x fp32 x per-episode re-randomisation x batch sizes on both sides of the small-batch rule x two device sizes; more
    than two sub-steps switch the pre-drawn noise off)."""
    step_all, roll_all = table("GAQ_STEP"), table("GAQ_ROLL")
    assert len(step_all) >= 60 and len(roll_all) == 16
    obs_sets = [0, 1, 2, 3, 4, 8, 12, 14, 15, 16, 18, 32, 64, 96, 33, 97]
    seen_step, seen_roll, checked = set(), set(), 0
    for per_env, lag, drag in itertools.product((0, 1), (0, 1), (0, 1)):
        for control, noise in itertools.product((0, 1, 2), (0, 1, 2)):
            for obs_flags in obs_sets:
                for sense, extra in itertools.product((0, 1, 2), ("", "aux", "sense_input", "resample_goal", "excite", "swarm", "action_change")):
                    if extra == "swarm" and (obs_flags & 16):
                        continue          # refused by gaq_create (checked below)
                    for alias, fp32 in ((0, 0), (1, 0), (2, 0), (1, 1)):
                        for n, every, cus, sim_steps in ((1, 0, 256, 2), (65536, 0, 256, 2), (131072, 1, 256, 2), (131072, 0, 256, 1),
                                                         (1 << 20, 0, 256, 2), (1 << 20, 0, 256, 4), (65536, 0, 64, 2)):
                            if every and not per_env:
                                every = 0
                            kw = {"per_env_params": per_env, "control": control, "noise": noise, "obs_flags": obs_flags, "obs_state_alias": alias,
                                  "fp32_state": fp32, "auto_reset": 1, "sim_steps": sim_steps}
                            if not per_env:
                                kw["model.damp_time_up"] = 0.15 if lag else 0.0
                                kw["model.damp_time_down"] = 0.15 if lag else 0.0
                                kw["model.c_drag"] = 0.1 if drag else 0.0
                            if sense:
                                kw.update({"sense.enabled": 1, "sense.pos_norm_std": 0.005, "sense.gyro_noise_density": 0.000175,
                                           "sense.gyro_norm_std": 0.0 if sense == 1 else 0.01, "sense.gyro_bias_correlation_time": 1000.0})
                            if extra == "aux":
                                kw["aux_outputs"] = 1
                            elif extra == "sense_input":
                                kw["sense_input"] = 1
                            elif extra in ("resample_goal", "excite"):
                                kw[extra] = 1
                            elif extra == "swarm":
                                kw.update({"swarm.agents": 8, "swarm.goal_radius": 0.5, "swarm.collision_dist": 0.3, "swarm.prox_dist": 1.2})
                                if n % 8:
                                    continue
                            elif extra == "action_change":
                                kw["rew.action_change"] = 0.1
                            cfg = base_cfg(n, **kw)
                            p = plan(cfg, lag if per_env else -1, drag if per_env else -1, every, cus)
                            checked += 1
                            where = (kw, n, every, cus, sim_steps, p.step_variant)
                            if fp32 and p.state_layout == 0:
                                assert not p.launchable, where         # fp32_state is refused, never dropped silently
                                continue
                            assert p.step_instantiated == 1 and p.step_variant in step_all, where
                            if p.state_layout != 0 and (p.step_variant & 8):
                                assert per_env and drag and not p.launchable, where    # rotor drag arriving on a split-state handle: refused loudly
                            else:
                                assert p.launchable == 1, where
                            seen_step.add(p.step_variant)
                            for twin, bit in ((p.rows_variant, 4096), (p.ctr_variant, 8192)):      # what a launch runs with packed rows / in graph-safe mode
                                if twin >= 0:
                                    assert twin == (p.step_variant | bit) and twin in step_all, where
                                    seen_step.add(twin)
                            if p.rollout_variant >= 0:
                                assert p.rollout_instantiated == 1 and p.rollout_variant in roll_all, where
                                seen_roll.add(p.rollout_variant)
                            assert p.lds_per_wave * 4 <= 160 * 1024, where
    assert checked > 50000
    # ... and the other way round: no instantiation is dead weight (compile time and library size)
    assert seen_step == step_all, sorted(step_all - seen_step)
    assert seen_roll == roll_all, sorted(roll_all - seen_roll)


def test_small_batch_rule_follows_the_device_size():
    """Non-temporal streaming up to two waves per SIMD, pre-drawn noise from two waves per SIMD up -- counted on the device's OWN
    compute units (hipDeviceProp_t::multiProcessorCount), not on a hard-coded 256 x 4: a partition with a quarter of the
    CUs switches at a quarter of the batch."""
    F_PREDRAW, F_NT = 128, 256
    cfg = lambda n: base_cfg(n, noise=1, obs_state_alias=1, auto_reset=1)
    for cus in (256, 64, 32):
        simds = cus * 4
        for tiles, want in ((simds // 2, F_NT), (simds, F_NT), (simds + 1, F_NT | F_PREDRAW), (2 * simds, F_NT | F_PREDRAW),
                            (2 * simds + 1, F_PREDRAW), (16 * simds, F_PREDRAW)):
            p = plan(cfg(tiles * 64), cus=cus)
            assert p.step_variant == (20 | want), (cus, tiles, p.step_variant)


def test_plan_validates_like_create():
    lib = _lib.load()
    out = _lib.GaqPlanInfo()
    cfg = base_cfg(4096)
    cfg.struct_size = 1
    assert lib.gaq_plan(C.byref(cfg), -1, -1, 0, 256, C.byref(out)) == -1 and b"mismatch" in lib.gaq_last_error()
    cfg = base_cfg(4096, **{"obs_flags": 16, "swarm.agents": 8, "swarm.prox_dist": 1.0})
    assert lib.gaq_plan(C.byref(cfg), -1, -1, 0, 256, C.byref(out)) == -1 and b"quaternion" in lib.gaq_last_error()
    cfg = base_cfg(4096, sim_steps=65)
    assert lib.gaq_plan(C.byref(cfg), -1, -1, 0, 256, C.byref(out)) == -1


def test_product_library_is_not_a_measurement_build():
    """GAQ_ABLATE's timing-only ablations exist in -DGAQ_DIAG_BUILD libraries only: the in-tree library must not be one."""
    assert _lib.load().gaq_is_diag_build() == 0


def test_feature_bit_names_of_the_tools_and_tests_match_the_header():
    """tools/kernel_coverage.py names the bits of a feature mask, tests/test_gpu_kernel_coverage.py turns masks back into constructor
    arguments by them: both tables have to be quad_core.hpp's enum, bit for bit."""
    import importlib.util
    src = open(os.path.join(ROOT, "gym_art_amd", "csrc", "quad_core.hpp")).read()
    header = {m.group(1): int(m.group(2)) for m in re.finditer(r"\bF_([A-Z0-9_]+)\s*=\s*(\d+)", src)}
    assert len(header) == 19 and sorted(header.values()) == [1 << k for k in range(19)], header
    tool = open(os.path.join(ROOT, "tools", "kernel_coverage.py")).read()
    bits = dict((name, int(val)) for val, name in re.findall(r'\((\d+), "([A-Z0-9_]+)"\)', tool))
    assert bits == header, (bits, header)
    spec = importlib.util.spec_from_file_location("cov", os.path.join(ROOT, "tests", "test_gpu_kernel_coverage.py"))
    cov = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cov)
    for name, val in header.items():
        assert getattr(cov, name) == val, name
    # ... and every instantiated mask has a recipe (or is one of the two reference kernels), built from known bits only
    for mask in cov.instantiated("GAQ_STEP"):
        assert mask < (1 << 19)
        rc = cov.recipe(mask)
        assert (rc is None) == (mask in (8, 9)), mask


def test_step_counter_arithmetic_for_every_launch_size():
    """a CPU model of the self-advancing graph-safe step counter (gaq_kernels.hpp step_counter_checkin) against the
    library's own ctr_waves / ctr_shift / ctr_inc0 (gaq_plan; gaq_create uses the same function) over every launch size there is:
    every tile count up to 4096 exhaustively, then the sizes around every power of two up to the 2^27-env limit of a handle.
    Model: the counter is a sum of words holding step << shift; wave w of the launch adds inc0 (w == 0) or 1 AFTER its own read, in any
    order; a wave may read at any moment.  What must hold: (i) one launch adds exactly 2^shift; (ii) whenever a wave reads, what has
    landed of THIS launch is < 2^shift -- the shift drops it, every wave sees the same index however late it is scheduled;
    (iii) waves covers every tile in whole workgroups of four; (iv) 2^(shift-1) < waves <= 2^shift (the tightest shift: 64 - shift bits
    are left for the step index, > 2^37 steps at the largest handle)."""
    import random
    lib = _lib.load()
    out = _lib.GaqPlanInfo()
    sizes = set(range(1, 4096 * 64 + 1, 64)) | set(range(1, 300))
    for p in range(8, 28):
        for d in (-129, -65, -64, -63, -1, 0, 1, 63, 64, 65, 255, 256, 257):
            n = (1 << p) + d
            if 0 < n <= (1 << 27):
                sizes.add(n)
    rng = random.Random(4)
    cfg = base_cfg(64, noise=1, obs_state_alias=1, auto_reset=1)
    for n in sorted(sizes):
        cfg.num_envs = n
        _lib.check(lib.gaq_plan(C.byref(cfg), -1, -1, 0, 256, C.byref(out)))
        waves, sh, inc0 = out.ctr_waves, out.ctr_shift, out.ctr_inc0
        tiles = (n + 63) // 64
        assert waves % 4 == 0 and waves >= tiles and waves - tiles < 4, n                       # (iii)
        assert (1 << sh) >= waves and (sh == 0 or (1 << (sh - 1)) < waves), n                   # (iv)
        assert inc0 >= 1 and inc0 + (waves - 1) == (1 << sh), n                                 # (i)
        # (ii): the most that can have landed when wave w reads is everything but its own check-in
        assert (1 << sh) - inc0 < (1 << sh) and (1 << sh) - 1 < (1 << sh)
        worst_for_wave0, worst_for_others = waves - 1, (1 << sh) - 1
        assert worst_for_wave0 < (1 << sh) and worst_for_others < (1 << sh), n
    # ... and the model run as a process for a few sizes: random interleavings of reads and check-ins over several launches, the
    # check-ins spread over 64 words like on the device
    for n in (1, 64 * 5, 64 * 1023 + 7, 65536, 2 ** 21 + 64 * 3, 100 * 64):
        cfg.num_envs = n
        _lib.check(lib.gaq_plan(C.byref(cfg), -1, -1, 0, 256, C.byref(out)))
        waves, sh, inc0 = out.ctr_waves, out.ctr_shift, out.ctr_inc0
        words = [0] * 64
        words[0] = 12345 << sh                                # the host wrote the step index into the first word
        for launch in range(3):
            order = list(range(waves))
            rng.shuffle(order)                                # the order in which the waves get to run
            pending = []                                      # check-ins issued (after the wave's read) but not landed yet
            for w in order:
                while pending and rng.random() < 0.7:         # some earlier check-ins land before this wave reads
                    k, inc = pending.pop(rng.randrange(len(pending)))
                    words[k] += inc
                seen = sum(words) >> sh
                assert seen == 12345 + launch, (n, launch, w)
                pending.append((w % 64, inc0 if w == 0 else 1))
            for k, inc in pending:                            # the kernel boundary: everything has landed
                words[k] += inc
            assert sum(words) == (12345 + launch + 1) << sh


def test_shard_ranges_of_the_one_process_multi_device_handle():
    """gaq_shard_range (pure host arithmetic behind gaq_create_sharded and QuadrotorEnv(device_ids=...)): contiguous, covering, whole
    64-env tiles and whole swarm worlds per shard, balanced to one unit, tail shards empty when there are fewer tiles than devices --
    and BASELINE config 4's split: 2^20 envs over 8 devices = 131 072 each."""
    lib = _lib.load()
    f, c = C.c_int64(0), C.c_int64(0)

    def split(n, K, align=1):
        out = []
        for k in range(K):
            _lib.check(lib.gaq_shard_range(n, K, k, align, C.byref(f), C.byref(c)))
            out.append((f.value, c.value))
        return out

    assert split(1 << 20, 8) == [(k * 131072, 131072) for k in range(8)]
    for n, K, align in ((1 << 20, 8, 1), (1000, 4, 1), (64, 8, 1), (65, 2, 1), (1, 3, 1), (8 * 131072, 8, 8), (96, 8, 8), (16 * 77, 3, 16),
                        (2 ** 20 + 17, 7, 1), (12345, 5, 1)):
        spans = split(n, K, align)
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1
        unit = 64 * align // __import__("math").gcd(64, align)
        live = [s for s in spans if s[1] > 0]
        assert all(fk % unit == 0 for fk, _ in live) and all(ck % unit == 0 for _, ck in live[:-1])
        assert all(ck % align == 0 for _, ck in live) or n % align != 0
        full = [ck for _, ck in live[:-1]]
        assert not full or max(full) - min(full) <= unit
        assert all(ck == 0 for _, ck in spans[len(live):])          # empty shards only at the tail
    assert lib.gaq_shard_range(0, 2, 0, 1, C.byref(f), C.byref(c)) == -1 and lib.gaq_shard_range(10, 2, 2, 1, C.byref(f), C.byref(c)) == -1
