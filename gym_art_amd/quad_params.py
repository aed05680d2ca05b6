"""Host-side model parameters -> derived dynamics constants, vectorised over N parameter sets.

Restates, for a whole batch at once, what the reference does per env with Python objects
(~2 ms each):
  * QuadLink (gym_art/quadrotor/inertia.py:182-310): composite-body mass, centre of mass and
    diagonal inertia of body + payload + 4 arms + 4 motors + 4 propellers;
  * QuadrotorDynamics.update_model (gym_art/quadrotor/quadrotor.py:142-208): thrust_max,
    torque_max, propeller positions, arm length, OU sigma.

A *parameter tree* is the reference's nested dict (geom/damp/noise/motor); in a *batched tree*
every numeric leaf is an array with a leading env axis: [N] for scalars, [N,k] for the short
vectors (motor_pos.xyz [N,3], payload_pos.xy [N,2], motor.assymetry [N,4]).

This runs on the host in fp64 (it is per-episode set-up, not the per-step path) and its output
feeds gaq_set_params / gaq_config.model.  Checked against the reference's own numbers by
tests/test_quad_params.py (fixtures g4_randomized, g4b_models).
"""
import copy

import numpy as np

GRAV = 9.81


def batch_tree(trees):
    """List of N parameter trees -> one batched tree."""
    def rec(nodes):
        first = nodes[0]
        if isinstance(first, dict):
            return {k: rec([n[k] for n in nodes]) for k in first}
        return np.array([np.asarray(n, dtype=np.float64) for n in nodes], dtype=np.float64)
    return rec(list(trees))


def broadcast_tree(tree, n):
    """One parameter tree -> batched tree of n identical rows."""
    def rec(node):
        if isinstance(node, dict):
            return {k: rec(v) for k, v in node.items()}
        a = np.asarray(node, dtype=np.float64)
        return np.broadcast_to(a, (n,) + a.shape).copy()
    return rec(tree)


def tree_size(btree):
    return int(np.asarray(btree["motor"]["thrust_to_weight"]).shape[0])


def unbatch_tree(btree, i):
    def rec(node):
        if isinstance(node, dict):
            return {k: rec(v) for k, v in node.items()}
        v = np.asarray(node)[i]
        return v.tolist() if v.ndim else float(v)
    return rec(btree)


def _box_mass(p):
    """BoxLink.compute_m (inertia.py:96-97) when a density is given instead of a mass."""
    if "m" in p:
        return np.asarray(p["m"], dtype=np.float64)
    return p["density"] * p["l"] * p["w"] * p["h"]


def _cyl_mass(p):
    """CylinderLink.compute_m (inertia.py:155-156)."""
    if "m" in p:
        return np.asarray(p["m"], dtype=np.float64)
    return p["density"] * np.pi * p["h"] * p["r"] ** 2


def _box_inertia(m, l, w, h):
    """BoxLink.I_com diagonal (inertia.py:88-94)."""
    return np.stack([m * (h ** 2 + w ** 2) / 12.0, m * (l ** 2 + h ** 2) / 12.0, m * (w ** 2 + l ** 2) / 12.0], axis=-1)


def _cyl_inertia(m, h, r):
    """CylinderLink.I_com diagonal (inertia.py:147-154)."""
    a = m * (3 * r ** 2 + h ** 2) / 12.0
    return np.stack([a, a, 0.5 * m * r ** 2], axis=-1)


def quadlink(geom):
    """Vectorised QuadLink (inertia.py:182-310): returns mass [N], com [N,3], inertia diag [N,3],
    prop_pos [N,4,3] (motor positions relative to the COM, :307), motor_xyz [N,3]."""
    g = geom
    body, payload, arms, motors, props = g["body"], g["payload"], g["arms"], g["motors"], g["propellers"]
    n = int(np.asarray(g["motor_pos"]["xyz"]).shape[0])
    arm_angle = np.asarray(g["arms_pos"]["angle"], dtype=np.float64) / 180.0 * np.pi        # deg2rad :37
    arm_angle = np.where(arm_angle == 0.0, 0.01, arm_angle)                                  # :218-219
    motor_xyz = np.asarray(g["motor_pos"]["xyz"], dtype=np.float64)                          # [N,3]
    delta_y = motor_xyz[:, 1] - body["w"] / 2.0                                              # :221
    arms = dict(arms)
    if "l" not in arms:
        arms["l"] = delta_y / np.sin(arm_angle)                                              # :223-224
    arm_xyz = np.stack([motor_xyz[:, 0] - delta_y / (2 * np.tan(arm_angle)),
                        motor_xyz[:, 1] - delta_y / 2.0,
                        np.asarray(g["arms_pos"]["z"], dtype=np.float64) * np.ones(n)], axis=1)   # :230-232
    x_sign = np.array([1.0, -1.0, -1.0, 1.0])                                                # :238-240
    y_sign = np.array([-1.0, -1.0, 1.0, 1.0])
    sign = np.stack([x_sign, y_sign, np.ones(4)], axis=1)                                    # [4,3]
    motors_coord = sign[None] * motor_xyz[:, None, :]                                        # [N,4,3]
    props_coord = motors_coord.copy()
    props_coord[:, :, 2] += (motors["h"] / 2.0 + props["h"])[:, None]                        # :243
    arm_angles = arm_angle[:, None] * np.array([-1.0, 1.0, -1.0, 1.0])[None]                 # :244-248
    arms_coord = sign[None] * arm_xyz[:, None, :]

    m_body, m_payload = _box_mass(body), _box_mass(payload)
    m_arm, m_motor, m_prop = _box_mass(arms), _cyl_mass(motors), _cyl_mass(props)
    I_body = _box_inertia(m_body, body["l"], body["w"], body["h"])
    I_payload = _box_inertia(m_payload, payload["l"], payload["w"], payload["h"])
    I_arm = _box_inertia(m_arm, arms["l"], arms["w"], arms["h"])
    I_motor = _cyl_inertia(m_motor, motors["h"], motors["r"])
    I_prop = _cyl_inertia(m_prop, props["h"], props["r"])

    pay_xy = np.asarray(g["payload_pos"]["xy"], dtype=np.float64)
    payload_xyz = np.concatenate([pay_xy, (np.sign(g["payload_pos"]["z_sign"]) * (body["h"] + payload["h"]) / 2.0)[:, None]],
                                 axis=1)                                                     # :268
    mass = m_body + m_payload + 4 * m_arm + 4 * m_motor + 4 * m_prop                         # :309-310
    com = (m_payload[:, None] * payload_xyz + m_arm[:, None] * arms_coord.sum(1) + m_motor[:, None] * motors_coord.sum(1)
           + m_prop[:, None] * props_coord.sum(1)) / mass[:, None]                           # :280-281
    def shifted(xyz):
        return xyz - (com[:, None, :] if xyz.ndim == 3 else com)

    def translate_diag(I, m, xyz):
        """diagonal of translate_I (inertia.py:22-35) for a tensor whose own diagonal is I"""
        x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
        return I + np.stack([m * (y ** 2 + z ** 2), m * (x ** 2 + z ** 2), m * (x ** 2 + y ** 2)], axis=-1)

    inertia = translate_diag(I_body, m_body, shifted(np.zeros((n, 3))))
    inertia = inertia + translate_diag(I_payload, m_payload, shifted(payload_xyz))
    # arms are rotated about z by +-arm_angle (LinkPose alpha, :166-177): diag(R I R^T)
    c2, s2 = np.cos(arm_angles) ** 2, np.sin(arm_angles) ** 2                                # [N,4]
    I_arm_rot = np.stack([c2 * I_arm[:, None, 0] + s2 * I_arm[:, None, 1],
                          s2 * I_arm[:, None, 0] + c2 * I_arm[:, None, 1],
                          np.broadcast_to(I_arm[:, None, 2], c2.shape)], axis=-1)            # [N,4,3]
    inertia = inertia + translate_diag(I_arm_rot, m_arm[:, None], shifted(arms_coord)).sum(1)
    inertia = inertia + translate_diag(I_motor[:, None, :], m_motor[:, None], shifted(motors_coord)).sum(1)
    inertia = inertia + translate_diag(I_prop[:, None, :], m_prop[:, None], shifted(props_coord)).sum(1)
    prop_pos = shifted(motors_coord)                                                         # :307
    return dict(mass=mass, com=com, inertia=inertia, prop_pos=prop_pos, motor_xyz=motor_xyz)


def quadlink_simplified(geom):
    """Vectorised QuadLinkSimplified (inertia.py:312-440): two perpendicular rods carrying the whole mass, massless
    propellers.  Same return dict as quadlink().  Quirks kept: only motor_pos.x enters the arm length (:346-349);
    the mass is body + payload + 4 arms + 4 motors, propellers dropped (:352-359), with the arms priced as boxes,
    so arms without "l" (RandomQuad) raise TypeError like BoxLink(**arms) does; the motor positions are NOT moved to
    the COM frame (they are not in `poses`, :400-414)."""
    g = geom
    body, payload, arms, motors = g["body"], g["payload"], g["arms"], g["motors"]
    n = int(np.asarray(g["motor_pos"]["xyz"]).shape[0])
    mx = np.asarray(g["motor_pos"]["xyz"], dtype=np.float64)[:, 0]
    arm_length = np.sqrt(mx ** 2 * 2) * 2                                                   # :346
    mp = arm_length * np.sqrt(2) / 4                                                        # :348
    motor_xyz = np.stack([mp, mp, np.zeros(n)], axis=1)                                     # :349
    if "mass" in g:
        mass = np.asarray(g["mass"], dtype=np.float64) * np.ones(n)
    else:
        if "l" not in arms:
            raise TypeError("BoxLink.__init__() missing 1 required positional argument: 'l'")
        mass = _box_mass(body) + _box_mass(payload) + 4 * _box_mass(arms) + 4 * _cyl_mass(motors) + 0.0   # :352-359
        mass = mass * np.ones(n)
    m_rod = mass / 2.0                                                                      # :361
    arm_angle = np.asarray(g["arms_pos"]["angle"], dtype=np.float64) / 180.0 * np.pi * np.ones(n)
    arm_angle = np.where(arm_angle == 0.0, 0.01, arm_angle)                                 # :371-373
    arm_xyz = np.stack([np.zeros(n), np.zeros(n), np.asarray(g["arms_pos"]["z"], dtype=np.float64) * np.ones(n)], axis=1)
    sign = np.stack([np.array([1.0, -1.0, -1.0, 1.0]), np.array([-1.0, -1.0, 1.0, 1.0]), np.ones(4)], axis=1)
    motors_coord = sign[None] * motor_xyz[:, None, :]                                       # :384
    com = (m_rod[:, None] * arm_xyz + m_rod[:, None] * arm_xyz) / mass[:, None]             # :408-409 (propellers weigh 0)
    rel = arm_xyz - com
    I0 = m_rod * arm_length ** 2 / 12.0                                                     # RodLink.I_com = diag(I0, 0, I0) (:119-124)
    x, y, z = rel[:, 0], rel[:, 1], rel[:, 2]
    shift = np.stack([m_rod * (y ** 2 + z ** 2), m_rod * (x ** 2 + z ** 2), m_rod * (x ** 2 + y ** 2)], axis=-1)
    inertia = np.zeros((n, 3))
    for sgn in (-1.0, 1.0):                                                                 # rods at -+arm_angle (:387-390)
        c2, s2 = np.cos(sgn * arm_angle) ** 2, np.sin(sgn * arm_angle) ** 2
        inertia = inertia + np.stack([c2 * I0, s2 * I0, I0], axis=-1) + shift               # diag(R I R^T) + translate_I
    return dict(mass=mass, com=com, inertia=inertia, prop_pos=motors_coord, motor_xyz=motor_xyz)


def slice_tree(btree, lo, hi):
    """Rows [lo, hi) of a batched tree (views)."""
    def rec(node):
        return {k: rec(v) for k, v in node.items()} if isinstance(node, dict) else np.asarray(node)[lo:hi]
    return rec(btree)


def derive_models(btree, dynamics_simplification=False, chunk=1 << 16):
    """Batched tree -> dict of gaq_model fields ([N] / [N,k] float64), i.e. QuadrotorDynamics.update_model
    (quadrotor.py:142-208); `dynamics_simplification` selects QuadLinkSimplified (:143-146).  Large batches are
    processed `chunk` rows at a time: the ~100 temporaries of the inertia composition then stay in cache (2^20
    parameter sets: 6 s instead of 12 s)."""
    n_all = tree_size(btree)
    if n_all > chunk:
        parts = [_derive_models(slice_tree(btree, lo, min(lo + chunk, n_all)), dynamics_simplification)
                 for lo in range(0, n_all, chunk)]
        return ({k: np.concatenate([p[0][k] for p in parts], axis=0) for k in parts[0][0]},
                {k: np.concatenate([p[1][k] for p in parts], axis=0) for k in parts[0][1]})
    return _derive_models(btree, dynamics_simplification)


def _derive_models(btree, dynamics_simplification=False):
    q = quadlink_simplified(btree["geom"]) if dynamics_simplification else quadlink(btree["geom"])
    motor = btree["motor"]
    n = q["mass"].shape[0]
    asym = np.asarray(motor["assymetry"], dtype=np.float64).reshape(n, 4)
    asym = asym * 4.0 / np.sum(asym, axis=1, keepdims=True)                                  # :174
    t2w = np.asarray(motor["thrust_to_weight"], dtype=np.float64)
    thrust_max = GRAV * q["mass"][:, None] * t2w[:, None] * asym / 4.0                       # :175
    torque_max = np.asarray(motor["torque_to_thrust"], dtype=np.float64)[:, None] * thrust_max   # :176
    out = dict(
        mass=q["mass"], inertia=q["inertia"], thrust_max=thrust_max, torque_max=torque_max,
        prop_pos=q["prop_pos"].reshape(n, 12),
        damp_time_up=np.asarray(motor["damp_time_up"], dtype=np.float64),
        damp_time_down=np.asarray(motor["damp_time_down"], dtype=np.float64),
        linearity=np.asarray(motor["linearity"], dtype=np.float64),
        arm=np.linalg.norm(q["motor_xyz"][:, :2], axis=1),                                   # :200
        ou_sigma=0.2 * np.asarray(btree["noise"]["thrust_noise_ratio"], dtype=np.float64),   # :198
        vel_damp=np.asarray(btree["damp"]["vel"], dtype=np.float64),
        damp_omega_quadratic=np.asarray(btree["damp"]["omega_quadratic"], dtype=np.float64),
        c_drag=np.asarray(motor["C_drag"], dtype=np.float64),
        c_roll=np.asarray(motor["C_roll"], dtype=np.float64),
    )
    # torque_to_inertia (:202-205): row sums of G_omega @ [[0,0,0],[0,1,1],[1,1,0],[1,0,1]], with G_omega =
    # (1/I)[:,None] * (thrust_max * (prop_pos x z).T + torque_max * ccw on the z row) (:182-190)
    pp = q["prop_pos"]                                                                       # [N,4,3]
    cross = np.stack([pp[:, :, 1], -pp[:, :, 0], np.zeros_like(pp[:, :, 0])], axis=1)        # (prop_pos x z).T -> [N,3,4]
    G = thrust_max[:, None, :] * cross
    G[:, 2, :] = G[:, 2, :] + torque_max * np.array([-1.0, 1.0, -1.0, 1.0])
    G = (1.0 / q["inertia"])[:, :, None] * G
    t2i = np.sum(G @ np.array([[0., 0., 0.], [0., 1., 1.], [1., 1., 0.], [1., 0., 1.]]), axis=2)
    extra = dict(com=q["com"], motor_assymetry=asym, thrust_to_weight=t2w,
                 torque_to_thrust=np.asarray(motor["torque_to_thrust"], dtype=np.float64), torque_to_inertia=t2i)
    return {k: np.array(np.broadcast_to(v, (n,) + np.shape(v)[1:]), dtype=np.float64, order="C") for k, v in out.items()}, \
        {k: np.array(v, dtype=np.float64) for k, v in extra.items()}


def update_tree(tree, change):
    """dict_update_existing (quad_utils.py:171-176): overwrite existing leaves only (KeyError otherwise)."""
    for key in change.keys():
        if isinstance(tree[key], dict):
            update_tree(tree[key], change[key])
        else:
            tree[key] = copy.deepcopy(change[key])
    return tree


# ---- flat form of a parameter tree: gaq_quad_params (include/gaq.h), the device-side pipeline's input -------------
TREE_LEAVES = (   # (path, width) in the order of gaq::TreeLeaf (csrc/quad_params_dev.hpp)
    (("geom", "body", "l"), 1), (("geom", "body", "w"), 1), (("geom", "body", "h"), 1), (("geom", "body", "m"), 1),
    (("geom", "payload", "l"), 1), (("geom", "payload", "w"), 1), (("geom", "payload", "h"), 1), (("geom", "payload", "m"), 1),
    (("geom", "arms", "l"), 1), (("geom", "arms", "w"), 1), (("geom", "arms", "h"), 1), (("geom", "arms", "m"), 1),
    (("geom", "motors", "h"), 1), (("geom", "motors", "r"), 1), (("geom", "motors", "m"), 1),
    (("geom", "propellers", "h"), 1), (("geom", "propellers", "r"), 1), (("geom", "propellers", "m"), 1),
    (("geom", "motor_pos", "xyz"), 3), (("geom", "arms_pos", "angle"), 1), (("geom", "arms_pos", "z"), 1),
    (("geom", "payload_pos", "xy"), 2), (("geom", "payload_pos", "z_sign"), 1),
    (("damp", "vel"), 1), (("damp", "omega_quadratic"), 1), (("noise", "thrust_noise_ratio"), 1),
    (("motor", "thrust_to_weight"), 1), (("motor", "assymetry"), 4), (("motor", "torque_to_thrust"), 1),
    (("motor", "linearity"), 1), (("motor", "C_drag"), 1), (("motor", "C_roll"), 1), (("motor", "damp_time_up"), 1),
    (("motor", "damp_time_down"), 1))
TREE_DOUBLES = sum(w for _, w in TREE_LEAVES)      # 40


def tree_is_flat_compatible(tree):
    """True when the tree has exactly the shipped models' leaves (links with masses `m`, arms with a length `l`): what
    the device-side pipeline handles.  RandomQuad trees (densities, no arm length) are not."""
    def leaves(node, path=()):
        for k, v in node.items():
            if isinstance(v, dict):
                yield from leaves(v, path + (k,))
            else:
                yield path + (k,)
    want = {p for p, _ in TREE_LEAVES}
    if tree_links_by_density(tree):
        want = {(p[:-1] + ("density",)) if (p[0] == "geom" and p[-1] == "m") else p for p in want} - {("geom", "arms", "l")}
    return set(leaves(tree)) == want


def tree_links_by_density(tree):
    """True for RandomQuad-shaped trees: links carry a `density` instead of a mass `m`, the arms no length `l`."""
    return "density" in tree["geom"]["body"]


def flatten_tree(btree):
    """Batched tree -> [N, 40] float64 rows of gaq_quad_params.  RandomQuad-shaped trees (tree_links_by_density) put the
    densities into the five `m` slots and 0 into arms.l (derived on the other side, inertia.py:223-224)."""
    n = tree_size(btree)
    by_density = tree_links_by_density(btree)
    out = np.zeros((n, TREE_DOUBLES))
    col = 0
    for path, w in TREE_LEAVES:
        node = btree
        if by_density and path[-1] == "m" and path[0] == "geom":
            path = path[:-1] + ("density",)
        if by_density and path == ("geom", "arms", "l"):
            col += w
            continue
        for k in path:
            node = node[k]
        out[:, col:col + w] = np.broadcast_to(np.asarray(node, dtype=np.float64).reshape(-1, w) if np.ndim(node) else np.asarray(node, dtype=np.float64), (n, w))
        col += w
    return out


def unflatten_tree(rows, by_density=False):
    """[N, 40] rows -> batched tree (the inverse of flatten_tree)."""
    rows = np.asarray(rows, dtype=np.float64)
    tree, col = {}, 0
    for path, w in TREE_LEAVES:
        if by_density and path == ("geom", "arms", "l"):
            col += w
            continue
        if by_density and path[-1] == "m" and path[0] == "geom":
            path = path[:-1] + ("density",)
        node = tree
        for k in path[:-1]:
            node = node.setdefault(k, {})
        node[path[-1]] = rows[:, col].copy() if w == 1 else rows[:, col:col + w].copy()
        col += w
    return tree


def ratio_rows(base_btree, noise_ratio, custom=None):
    """Per-leaf noise ratios of a RelativeSampler (get_dyn_randomization_params, quadrotor_randomization.py:48-68) as
    [N, 40] rows: `noise_ratio` everywhere, overridden leaf by leaf from the nested dict `custom`."""
    def rec(node):
        return {k: rec(v) for k, v in node.items()} if isinstance(node, dict) else np.full(np.shape(node), float(noise_ratio))
    tree = rec(base_btree)
    if custom is not None:
        n = tree_size(base_btree)
        update_tree(tree, broadcast_tree(custom, n))
    return flatten_tree(tree)
