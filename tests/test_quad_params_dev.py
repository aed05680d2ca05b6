"""The device-side parameter pipeline (gym_art_amd/csrc/quad_params_dev.hpp: QuadLink + update_model + limits +
RelativeSampler per env), compiled for the host, against the reference's own numbers (fixtures G4, G4b) and against the
host pipeline's distribution.  The same header runs in libgaq's rerandomize kernel."""
import ctypes as C

import numpy as np

from gym_art_amd import quad_models, quad_params as qp, quadrotor_randomization as qr
from tests import golden_util as gu
from tests import hh


class Derived(C.Structure):
    _fields_ = [("mass", C.c_double), ("inertia", C.c_double * 3), ("thrust_max", C.c_double * 4), ("torque_max", C.c_double * 4),
                ("prop_pos", C.c_double * 12), ("damp_time_up", C.c_double), ("damp_time_down", C.c_double),
                ("linearity", C.c_double), ("arm", C.c_double), ("ou_sigma", C.c_double), ("vel_damp", C.c_double),
                ("damp_omega_quadratic", C.c_double), ("c_drag", C.c_double), ("c_roll", C.c_double), ("com", C.c_double * 3),
                ("t2t", C.c_double), ("motor_x", C.c_double), ("motor_y", C.c_double)]


def derive(row, clip=0, by_density=0):
    L = hh.lib()
    assert L.hh_sizeof_derived() == C.sizeof(Derived)
    out = Derived()
    row = np.ascontiguousarray(row, dtype=np.float64)
    L.hh_derive_tree(row.ctypes.data_as(C.POINTER(C.c_double)), clip, by_density, C.byref(out))
    return out


def tree_from_flat_params(prm):
    """golden `param_*` block (flattened 'geom.body.l' keys) -> nested tree"""
    tree = {}
    for k, v in prm.items():
        node = tree
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = np.asarray(v, dtype=np.float64)
    return tree


def check_against_const(d, const):
    for key, ref in (("mass", const["mass"]), ("inertia", const["inertia"]), ("thrust_max", const["thrust_max"]),
                     ("torque_max", const["torque_max"]), ("prop_pos", np.asarray(const["prop_pos"]).reshape(12)),
                     ("arm", const["arm"]), ("ou_sigma", const["thrust_noise_sigma"]), ("com", const["com"]),
                     ("damp_time_up", const["damp_time_up"]), ("linearity", const["motor_linearity"])):
        got = np.array(getattr(d, key)) if hasattr(getattr(d, key), "__len__") else getattr(d, key)
        assert gu.rel_err(got, ref) <= 1e-12, (key, got, ref)
    # the compact parameter path's construction: bit-exact by the same operations
    for j, (sx, sy) in enumerate(((1, -1), (-1, -1), (-1, 1), (1, 1))):
        assert d.prop_pos[3 * j] == sx * d.motor_x - d.com[0] and d.prop_pos[3 * j + 1] == sy * d.motor_y - d.com[1]
        assert d.torque_max[j] == d.t2t * d.thrust_max[j]


def test_tree_layout_round_trip():
    tree = qp.batch_tree([quad_models.model_params(m) for m in ("crazyflie", "defaultquad", "mediumquad")])
    assert qp.tree_is_flat_compatible(tree) and qp.TREE_DOUBLES == 40
    rows = qp.flatten_tree(tree)
    back = qp.unflatten_tree(rows)
    assert np.array_equal(qp.flatten_tree(back), rows)
    rq = qr.RandomQuad().sample(5, rng=np.random.RandomState(0))
    assert qp.tree_is_flat_compatible(rq) and qp.tree_links_by_density(rq) and not qp.tree_links_by_density(tree)
    rows = qp.flatten_tree(rq)
    assert np.all(rows[:, 8] == 0) and np.array_equal(rows[:, 3], rq["geom"]["body"]["density"])
    assert np.array_equal(qp.flatten_tree(qp.unflatten_tree(rows, by_density=True)), rows)


def test_derivation_matches_the_reference_constants():
    """QuadLink + update_model per env: the three shipped models (G4b) and 32 RelativeSampler-perturbed CrazyFlies (G4)."""
    d = gu.load("g4b_models")
    for name in ("DefaultQuad", "Crazyflie", "MediumQuad"):
        tree = qp.batch_tree([tree_from_flat_params(gu.sub(d, name + "_param_"))])
        check_against_const(derive(qp.flatten_tree(tree)[0]), gu.sub(d, name + "_const_"))
    g4 = gu.load("g4_randomized")
    for blk in gu.env_blocks(g4):
        tree = qp.batch_tree([tree_from_flat_params(gu.sub(blk, "param_"))])
        check_against_const(derive(qp.flatten_tree(tree)[0]), gu.sub(blk, "const_"))


def test_random_quad_derivation_and_sampler():
    """RandomQuad on the device path: the density-based tree (mass = density x volume, arms.l from the motor position)
    derived per env against the reference's constants for its own 16 RandomQuad draws (G4b), and random_quad_tree (the
    sampler, Philox) against the host's randomquad_parameters distribution, leaf by leaf and on derived quantities."""
    from scipy import stats
    d = gu.load("g4b_models")
    for i in range(int(d["n_random"])):
        tree = qp.batch_tree([tree_from_flat_params(gu.sub(d, "rq%d_param_" % i))])
        assert qp.tree_links_by_density(tree)
        check_against_const(derive(qp.flatten_tree(tree)[0], by_density=1), gu.sub(d, "rq%d_const_" % i))
    L = hh.lib()
    n = 6000
    out = np.zeros((n, 40))
    for i in range(n):
        L.hh_random_quad_tree(C.c_uint64(5), C.c_uint64(i), C.c_uint64(i % 4), out[i].ctypes.data_as(C.POINTER(C.c_double)))
    host = qp.flatten_tree(qr.RandomQuad().sample(n, rng=np.random.RandomState(8)))
    for k in range(40):
        a, b = out[:, k], host[:, k]
        if b.std() == 0:
            assert np.all(a == b[0]), k
        else:
            assert stats.ks_2samp(a, b).pvalue > 1e-4, (k, a.mean(), b.mean())
    md, _ = qp.derive_models(qp.unflatten_tree(out, by_density=True))
    mh, _ = qp.derive_models(qp.unflatten_tree(host, by_density=True))
    for key in ("mass", "arm"):
        assert stats.ks_2samp(md[key], mh[key]).pvalue > 1e-4, key
    assert stats.ks_2samp(md["inertia"][:, 2], mh["inertia"][:, 2]).pvalue > 1e-4
    dm = derive(out[0], by_density=1)
    assert gu.rel_err(dm.mass, md["mass"][0]) <= 1e-12 and gu.rel_err(np.array(dm.inertia), md["inertia"][0]) <= 1e-12


def test_limits_match_the_host_pipeline():
    rng = np.random.RandomState(3)
    base = qp.broadcast_tree(quad_models.model_params("crazyflie"), 64)
    rows = qp.flatten_tree(base) * rng.uniform(-0.5, 3.0, size=(64, 40))       # wild values: every limit is hit somewhere
    host = qr.check_quad_param_limits(qp.unflatten_tree(rows))
    models, _ = qp.derive_models(host)
    for i in range(64):
        dm = derive(rows[i], clip=1)
        if not np.isfinite(models["inertia"][i]).all():
            continue
        assert gu.rel_err(dm.mass, models["mass"][i]) <= 1e-12
        assert gu.rel_err(np.array(dm.inertia), models["inertia"][i]) <= 1e-9
        assert gu.rel_err(np.array(dm.thrust_max), models["thrust_max"][i]) <= 1e-12


def test_sampler_distribution_matches_relative_sampler():
    """perturb_tree (Philox streams) against RelativeSampler(noise_ratio=0.2, 'normal') of the host pipeline (numpy): same
    per-leaf mean / std, same limits, same propeller-radius rule; 'uniform' likewise."""
    from scipy import stats
    L = hh.lib()
    base = qp.flatten_tree(qp.broadcast_tree(quad_models.model_params("crazyflie"), 1))[0]
    n = 6000
    for sampler, name in ((0, "normal"), (1, "uniform")):
        ratio = np.full(40, 0.2)
        out = np.zeros((n, 40))
        for i in range(n):
            L.hh_perturb_tree(base.ctypes.data_as(C.POINTER(C.c_double)), ratio.ctypes.data_as(C.POINTER(C.c_double)), sampler,
                              C.c_uint64(77), C.c_uint64(i), C.c_uint64(i % 3), out[i].ctypes.data_as(C.POINTER(C.c_double)))
        btree = qp.broadcast_tree(quad_models.model_params("crazyflie"), n)
        host = qp.flatten_tree(qr.RelativeSampler(btree, noise_ratio=0.2, sampler=name).sample(btree, rng=np.random.RandomState(5)))
        for k in range(40):
            a, b = out[:, k], host[:, k]
            scale = max(abs(base[k]), 1e-12)
            assert abs(a.mean() - b.mean()) <= 0.01 * scale + 1e-15, (name, k, a.mean(), b.mean())
            assert abs(a.std() - b.std()) <= 0.012 * scale + 1e-15, (name, k, a.std(), b.std())
            if a.std() > 0:
                assert stats.ks_2samp(a, b).pvalue > 1e-4, (name, k)
            if name == "uniform" and k != 16:        # (leaf 16, the propeller radius, is a function of leaf 29)
                lo, hi = sorted((base[k] * 0.8, base[k] * 1.2))
                assert a.min() >= lo - 1e-15 and a.max() <= hi + 1e-15
        # propeller radius follows the sampled thrust-to-weight ratio exactly (:41-44)
        assert np.allclose(out[:, 16], base[16] * np.sqrt(base[26 + 3] / out[:, 29]), rtol=1e-14)
        assert len(np.unique(out[:, 29])) > 0.99 * n         # distinct streams per env / resample count
