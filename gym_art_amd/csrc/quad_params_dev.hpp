// quad_params_dev.hpp -- a quadrotor's parameter tree -> derived dynamics constants, and the per-episode parameter
// sampler, as per-env scalar code that runs on the device (the rerandomize kernel of gaq.hip) and, compiled by g++, in
// the host harness (tests/host_harness: pinned to the reference's numbers, fixtures G4 / G4b, <= 1e-12).
//
// What it restates (reference = amolchanov86/gym_art, gym_art/quadrotor/):
//   derive_tree()   inertia.py:182-310 QuadLink (composite-body mass, centre of mass, diagonal inertia of body + payload +
//                   4 arms + 4 motors + 4 propellers) and quadrotor.py:142-208 QuadrotorDynamics.update_model
//   clip_tree()     quadrotor_randomization.py:16-46 check_quad_param_limits
//   perturb_tree()  quadrotor_randomization.py:70-104 perturb_dyn_parameters (RelativeSampler :345-358)
// The reference does this with Python objects per env (~2 ms each) on the host at every
// `dynamics_randomize_every`-th reset (quadrotor.py:1063-1066); gym_art_amd/quad_params.py is the vectorised host version.
// Here it runs inside the reset path on the GPU so that per-episode re-randomisation of 2^20 envs costs microseconds.
//
//   random_quad_tree()  quadrotor_randomization.py:142-243 randomquad_parameters (the RandomQuad sampler)
// Supported trees: the shipped models' shape (quad_models.py: every link has a mass `m`, the arms a length `l`) and RandomQuad's
// (`by_density`: the five `m` slots hold DENSITIES -- the mass is density x volume, inertia.py:96-97 / :155-156 -- and the arm
// length is derived from the motor position, :223-224).  QuadLinkSimplified stays on the host path (gaq_set_params).
#pragma once

#include "quad_core.hpp"

namespace gaq {

// flat layout of the parameter tree = gaq_quad_params (include/gaq.h), in the reference's dict order
enum TreeLeaf {
  TL_BODY = 0,      // l, w, h, m
  TL_PAYLOAD = 4,   // l, w, h, m
  TL_ARMS = 8,      // l, w, h, m
  TL_MOTORS = 12,   // h, r, m
  TL_PROPS = 15,    // h, r, m
  TL_MOTOR_POS = 18,  // xyz
  TL_ARMS_ANGLE = 21, TL_ARMS_Z = 22,
  TL_PAYLOAD_XY = 23, TL_PAYLOAD_ZSIGN = 25,
  TL_DAMP_VEL = 26, TL_DAMP_OMEGA_Q = 27,
  TL_NOISE_RATIO = 28,
  TL_T2W = 29, TL_ASYM = 30 /* 4 */, TL_T2T = 34, TL_LINEARITY = 35, TL_C_DRAG = 36, TL_C_ROLL = 37,
  TL_DAMP_UP = 38, TL_DAMP_DOWN = 39,
  TL_COUNT = 40
};

struct ParamTree { double v[TL_COUNT]; };

// what update_model derives, in gaq_model's terms (+ the construction hints of the compact parameter path)
struct DerivedModel {
  double mass, inertia[3], thrust_max[4], torque_max[4], prop_pos[12];
  double damp_time_up, damp_time_down, linearity, arm, ou_sigma, vel_damp, damp_omega_quadratic, c_drag, c_roll;
  double com[3], t2t, motor_x, motor_y;
};

GAQ_HD double clip_lo(double x, double lo) { return x < lo ? lo : x; }          // np.clip(x, lo, None): NaN stays NaN
GAQ_HD double clip_both(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// check_quad_param_limits (quadrotor_randomization.py:16-46).  `init` != nullptr: the propeller radius follows the
// thrust-to-weight ratio, r = r0 * (t2w_init / t2w_new) ** 0.5 (:41-44; the arithmetic the reference executes).
GAQ_HD void clip_tree(ParamTree& t, const ParamTree* init) {
  for (int k = TL_BODY; k < TL_MOTOR_POS; ++k) t.v[k] = clip_lo(t.v[k], 0.0);                       // :19-22
  t.v[TL_MOTOR_POS] = clip_lo(t.v[TL_MOTOR_POS], 0.005);                                              // :23
  t.v[TL_MOTOR_POS + 1] = clip_lo(t.v[TL_MOTOR_POS + 1], 0.005);
  const double bw4 = t.v[TL_BODY + 1] / 4.0;                                                          // :24-25
  t.v[TL_PAYLOAD_XY] = clip_both(t.v[TL_PAYLOAD_XY], -bw4, bw4);
  t.v[TL_PAYLOAD_XY + 1] = clip_both(t.v[TL_PAYLOAD_XY + 1], -bw4, bw4);
  t.v[TL_ARMS_ANGLE] = clip_both(t.v[TL_ARMS_ANGLE], 0.0, 90.0);                                      // :26
  t.v[TL_DAMP_VEL] = clip_both(t.v[TL_DAMP_VEL], 0.0, 1.0);                                           // :29-30
  t.v[TL_DAMP_OMEGA_Q] = clip_both(t.v[TL_DAMP_OMEGA_Q], 0.0, 1.0);
  t.v[TL_T2W] = clip_lo(t.v[TL_T2W], 1.2);                                                            // :33
  t.v[TL_T2T] = clip_both(t.v[TL_T2T], 0.001, 1.0);                                                   // :34
  t.v[TL_LINEARITY] = clip_both(t.v[TL_LINEARITY], 0.0, 1.0);                                         // :35
  for (int j = 0; j < 4; ++j) t.v[TL_ASYM + j] = clip_both(t.v[TL_ASYM + j], 0.9, 1.1);               // :36
  t.v[TL_C_DRAG] = clip_lo(t.v[TL_C_DRAG], 0.0); t.v[TL_C_ROLL] = clip_lo(t.v[TL_C_ROLL], 0.0);       // :37-40
  t.v[TL_DAMP_UP] = clip_lo(t.v[TL_DAMP_UP], 0.0); t.v[TL_DAMP_DOWN] = clip_lo(t.v[TL_DAMP_DOWN], 0.0);
  if (init) t.v[TL_PROPS + 1] = init->v[TL_PROPS + 1] * sqrt(init->v[TL_T2W] / t.v[TL_T2W]);          // :41-44
}

// QuadLink (inertia.py:182-310) + update_model (quadrotor.py:142-208) for one tree
GAQ_HD void derive_tree(const ParamTree& t, DerivedModel& m, bool by_density = false) {
  const double* body = t.v + TL_BODY;
  const double* payload = t.v + TL_PAYLOAD;
  const double mot_h = t.v[TL_MOTORS], mot_r = t.v[TL_MOTORS + 1];
  const double prp_h = t.v[TL_PROPS], prp_r = t.v[TL_PROPS + 1];
  const double mx = t.v[TL_MOTOR_POS], my = t.v[TL_MOTOR_POS + 1], mz = t.v[TL_MOTOR_POS + 2];
  double ang = t.v[TL_ARMS_ANGLE] / 180.0 * 3.141592653589793;                                        // deg2rad (inertia.py:37)
  if (ang == 0.0) ang = 0.01;                                                                          // :218-219
  const double delta_y = my - body[1] / 2.0;                                                           // :221
  double arms[4] = {t.v[TL_ARMS], t.v[TL_ARMS + 1], t.v[TL_ARMS + 2], t.v[TL_ARMS + 3]};
  double m_body = body[3], m_payload = payload[3], m_arm = arms[3], m_motor = t.v[TL_MOTORS + 2], m_prop = t.v[TL_PROPS + 2];
  if (by_density) {
    arms[0] = delta_y / sin(ang);                                                                      // arms without "l" (:223-224)
    m_body = body[3] * body[0] * body[1] * body[2];                                                    // BoxLink.compute_m (:96-97)
    m_payload = payload[3] * payload[0] * payload[1] * payload[2];
    m_arm = arms[3] * arms[0] * arms[1] * arms[2];
    m_motor = t.v[TL_MOTORS + 2] * 3.141592653589793 * mot_h * (mot_r * mot_r);                        // CylinderLink.compute_m (:155-156)
    m_prop = t.v[TL_PROPS + 2] * 3.141592653589793 * prp_h * (prp_r * prp_r);
  }
  const double ax = mx - delta_y / (2.0 * tan(ang)), ay = my - delta_y / 2.0, az = t.v[TL_ARMS_Z];     // :230-232
  const double sx[4] = {1.0, -1.0, -1.0, 1.0}, sy[4] = {-1.0, -1.0, 1.0, 1.0};                        // :238-240
  const double prop_dz = mot_h / 2.0 + prp_h;                                                          // :243
  // link inertias about their own centres (BoxLink :88-94, CylinderLink :147-154)
  auto box = [](double mm, double l, double w, double h, double I[3]) {
    I[0] = mm * (h * h + w * w) / 12.0; I[1] = mm * (l * l + h * h) / 12.0; I[2] = mm * (w * w + l * l) / 12.0;
  };
  auto cyl = [](double mm, double h, double r, double I[3]) {
    const double a = mm * (3.0 * (r * r) + h * h) / 12.0;
    I[0] = a; I[1] = a; I[2] = 0.5 * mm * (r * r);
  };
  double I_body[3], I_payload[3], I_arm[3], I_motor[3], I_prop[3];
  box(m_body, body[0], body[1], body[2], I_body);
  box(m_payload, payload[0], payload[1], payload[2], I_payload);
  box(m_arm, arms[0], arms[1], arms[2], I_arm);
  cyl(m_motor, mot_h, mot_r, I_motor);
  cyl(m_prop, prp_h, prp_r, I_prop);
  const double zs = t.v[TL_PAYLOAD_ZSIGN];
  const double sgn = zs > 0.0 ? 1.0 : (zs < 0.0 ? -1.0 : 0.0);
  const double pay[3] = {t.v[TL_PAYLOAD_XY], t.v[TL_PAYLOAD_XY + 1], sgn * (body[2] + payload[2]) / 2.0};   // :268
  const double mass = m_body + m_payload + 4.0 * m_arm + 4.0 * m_motor + 4.0 * m_prop;                // :309-310
  // centre of mass (:280-281): sums over the four copies in index order, like the reference's array sums
  double s_arm[3] = {0, 0, 0}, s_mot[3] = {0, 0, 0}, s_prp[3] = {0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    s_arm[0] += sx[i] * ax; s_arm[1] += sy[i] * ay; s_arm[2] += az;
    s_mot[0] += sx[i] * mx; s_mot[1] += sy[i] * my; s_mot[2] += mz;
    s_prp[0] += sx[i] * mx; s_prp[1] += sy[i] * my; s_prp[2] += mz + prop_dz;
  }
  double com[3];
  for (int k = 0; k < 3; ++k) com[k] = (m_payload * pay[k] + m_arm * s_arm[k] + m_motor * s_mot[k] + m_prop * s_prp[k]) / mass;
  // inertia about the centre of mass: translate_I (:22-35), diagonal only (quadrotor.py:152)
  auto add = [&](const double I[3], double mm, double x, double y, double z, double acc[3]) {
    acc[0] += I[0] + mm * (y * y + z * z); acc[1] += I[1] + mm * (x * x + z * z); acc[2] += I[2] + mm * (x * x + y * y);
  };
  double inertia[3] = {0, 0, 0};
  add(I_body, m_body, 0.0 - com[0], 0.0 - com[1], 0.0 - com[2], inertia);
  add(I_payload, m_payload, pay[0] - com[0], pay[1] - com[1], pay[2] - com[2], inertia);
  double acc_arm[3] = {0, 0, 0}, acc_mot[3] = {0, 0, 0}, acc_prp[3] = {0, 0, 0};
  // arms are rotated about z by +-arm_angle (LinkPose alpha, :166-177): diag(R I R^T).  cos(-a) = cos(a) and sin(-a)^2 = sin(a)^2
  // bit for bit, so the four arms share one rotated tensor (two fp64 trig calls instead of eight)
  const double ca = cos(ang), sn = sin(ang);
  const double c2 = ca * ca, s2 = sn * sn;
  const double I_rot[3] = {c2 * I_arm[0] + s2 * I_arm[1], s2 * I_arm[0] + c2 * I_arm[1], I_arm[2]};
  for (int i = 0; i < 4; ++i) {
    add(I_rot, m_arm, sx[i] * ax - com[0], sy[i] * ay - com[1], az - com[2], acc_arm);
    add(I_motor, m_motor, sx[i] * mx - com[0], sy[i] * my - com[1], mz - com[2], acc_mot);
    add(I_prop, m_prop, sx[i] * mx - com[0], sy[i] * my - com[1], (mz + prop_dz) - com[2], acc_prp);
  }
  for (int k = 0; k < 3; ++k) inertia[k] = ((inertia[k] + acc_arm[k]) + acc_mot[k]) + acc_prp[k];
  m.mass = mass;
  for (int k = 0; k < 3; ++k) { m.inertia[k] = inertia[k]; m.com[k] = com[k]; }
  for (int i = 0; i < 4; ++i) {                                                                        // prop_pos (:307)
    m.prop_pos[3 * i] = sx[i] * mx - com[0]; m.prop_pos[3 * i + 1] = sy[i] * my - com[1]; m.prop_pos[3 * i + 2] = mz - com[2];
  }
  // update_model (quadrotor.py:172-176, :198-200)
  double asum = 0.0;
  for (int j = 0; j < 4; ++j) asum += t.v[TL_ASYM + j];
  for (int j = 0; j < 4; ++j) {
    const double as = t.v[TL_ASYM + j] * 4.0 / asum;
    m.thrust_max[j] = 9.81 * mass * t.v[TL_T2W] * as / 4.0;
    m.torque_max[j] = t.v[TL_T2T] * m.thrust_max[j];
  }
  m.t2t = t.v[TL_T2T]; m.motor_x = mx; m.motor_y = my;
  m.damp_time_up = t.v[TL_DAMP_UP]; m.damp_time_down = t.v[TL_DAMP_DOWN]; m.linearity = t.v[TL_LINEARITY];
  m.arm = sqrt(mx * mx + my * my);
  m.ou_sigma = 0.2 * t.v[TL_NOISE_RATIO];
  m.vel_damp = t.v[TL_DAMP_VEL]; m.damp_omega_quadratic = t.v[TL_DAMP_OMEGA_Q];
  m.c_drag = t.v[TL_C_DRAG]; m.c_roll = t.v[TL_C_ROLL];
}

// perturb_dyn_parameters (quadrotor_randomization.py:70-104): every numeric leaf v is redrawn around its nominal value,
// normal(loc = v, scale = |ratio/2 * v|) or uniform(v - v ratio, v + v ratio), then the limits are re-applied against the
// nominal tree.  Draws: Philox streams keyed by (seed, global env index, resample count) -- the reference draws from
// numpy's global MT19937, so the DISTRIBUTION is reproduced, not the stream.
enum { RNG_PARAM0 = 200 };
GAQ_HD void perturb_tree(const ParamTree& base, const double ratio[TL_COUNT], int sampler, uint64_t seed, uint64_t env_global,
                         uint64_t resample_count, ParamTree& out) {
  for (int b = 0; b < TL_COUNT / 4; ++b) {
    const Philox r(seed, env_global, resample_count, RNG_PARAM0 + (uint32_t)b);
    float nrm[4];
    normals4(r, nrm);
    for (int k = 0; k < 4; ++k) {
      const int leaf = 4 * b + k;
      const double v = base.v[leaf], rt = ratio[leaf];
      if (sampler == 0) {
        out.v[leaf] = v + fabs((rt / 2.0) * v) * (double)nrm[k];
      } else {
        const double lo = v - v * rt, hi = v + v * rt;
        const double a = lo < hi ? lo : hi, c = lo < hi ? hi : lo;
        out.v[leaf] = a + (c - a) * r.u01(k);
      }
    }
  }
  clip_tree(out, &base);
}

// randomquad_parameters (quadrotor_randomization.py:142-243): a random quadrotor -- overall size, body / payload / arm / motor
// proportions, link densities, thrust-to-weight, motor constants -- in the reference's order of construction; the tree comes out in
// the `by_density` form.  Draws: 8 Philox blocks of uniforms + 3 of normals, keyed like perturb_tree's.
GAQ_HD void random_quad_tree(uint64_t seed, uint64_t env_global, uint64_t resample_count, ParamTree& t) {
  double u[32];
  float nr[12];
  for (int b = 0; b < 8; ++b) {
    const Philox r(seed, env_global, resample_count, RNG_PARAM0 + 16u + (uint32_t)b);
    for (int k = 0; k < 4; ++k) u[4 * b + k] = r.u01(k);
  }
  for (int b = 0; b < 3; ++b) {
    const Philox r(seed, env_global, resample_count, RNG_PARAM0 + 24u + (uint32_t)b);
    normals4(r, nr + 4 * b);
  }
  auto U = [&](int i, double lo, double hi) { return lo + (hi - lo) * u[i]; };
  auto N = [&](int i, double loc, double scale) { return loc + scale * (double)nr[i]; };
  const double dlo[5] = {500., 200., 500., 500., 200.}, dhi[5] = {2000., 2000., 2000., 4500., 300.};       // :154-156
  const double dens[5] = {U(0, dlo[0], dhi[0]), U(1, dlo[1], dhi[1]), U(2, dlo[2], dhi[2]), U(3, dlo[3], dhi[3]), U(4, dlo[4], dhi[4])};
  const double total_w = U(5, 0.05, 0.2);                                                                  // :167
  const double total_l = clip_lo(N(0, 1., 0.1), 1.0) * total_w;                                            // :168
  const double motor_z = N(1, 0., total_w / 8.);                                                           // :169
  const double mot_r = total_w * N(2, 0.1, 0.01), mot_h = mot_r * N(3, 1.0, 0.05);                         // :171-172
  const double w_coeff = U(6, 0.25, 0.5);                                                                  // :175-176
  const double body_w = w_coeff * total_w;
  const double l_scale = 1. - (w_coeff - 0.25) / (0.5 - 0.25);                                             // :179
  const double body_l = clip_lo(N(4, 1., l_scale), 1.0) * body_w, body_h = U(7, 0.1, 1.5) * body_w;       // :180-181
  const double pay_w = U(8, 0.25, 1.0) * body_w, pay_l = U(9, 0.25, 1.0) * body_l, pay_h = U(10, 0.25, 1.0) * body_h;   // :184-187
  const double pay_x = N(5, 0., body_w / 10.), pay_y = N(6, 0., body_w / 10.);                             // :189
  const double zs = U(11, -1., 1.);
  const double z_sign = zs > 0 ? 1.0 : (zs < 0 ? -1.0 : 0.0);                                              // :190
  const double arm_w = total_w * N(7, 0.05, 0.005), arm_h = total_w * N(8, 0.05, 0.005);                   // :194-195
  const double angle = N(9, 45., 10.);                                                                     // :196
  const double t2w = U(12, 1.5, 3.5);                                                                      // :199
  const double damp_up = U(14, 0.15, 0.2);                                                                 // :220
  t.v[TL_BODY] = body_l; t.v[TL_BODY + 1] = body_w; t.v[TL_BODY + 2] = body_h; t.v[TL_BODY + 3] = dens[0];
  t.v[TL_PAYLOAD] = pay_l; t.v[TL_PAYLOAD + 1] = pay_w; t.v[TL_PAYLOAD + 2] = pay_h; t.v[TL_PAYLOAD + 3] = dens[1];
  t.v[TL_ARMS] = 0.0; t.v[TL_ARMS + 1] = arm_w; t.v[TL_ARMS + 2] = arm_h; t.v[TL_ARMS + 3] = dens[2];    // arms.l: derived
  t.v[TL_MOTORS] = mot_h; t.v[TL_MOTORS + 1] = mot_r; t.v[TL_MOTORS + 2] = dens[3];
  t.v[TL_PROPS] = 0.01; t.v[TL_PROPS + 1] = 0.3 * total_w * sqrt(t2w / 2.0); t.v[TL_PROPS + 2] = dens[4];   // :201-202
  t.v[TL_MOTOR_POS] = total_w / 2.; t.v[TL_MOTOR_POS + 1] = total_l / 2.; t.v[TL_MOTOR_POS + 2] = motor_z;  // :170
  t.v[TL_ARMS_ANGLE] = angle; t.v[TL_ARMS_Z] = motor_z - mot_h / 2.;                                       // :196
  t.v[TL_PAYLOAD_XY] = pay_x; t.v[TL_PAYLOAD_XY + 1] = pay_y; t.v[TL_PAYLOAD_ZSIGN] = z_sign;
  t.v[TL_DAMP_VEL] = 0.0; t.v[TL_DAMP_OMEGA_Q] = 0.0;                                                      // :211-213
  t.v[TL_NOISE_RATIO] = U(13, 0.01, 0.05);                                                                 // :217
  t.v[TL_T2W] = t2w;
  for (int j = 0; j < 4; ++j) t.v[TL_ASYM + j] = U(16 + j, 0.9, 1.1);                                     // :224
  t.v[TL_T2T] = U(15, 0.005, 0.025);                                                                       // :223
  t.v[TL_LINEARITY] = 1.0; t.v[TL_C_DRAG] = 0.0; t.v[TL_C_ROLL] = 0.0;
  t.v[TL_DAMP_UP] = damp_up; t.v[TL_DAMP_DOWN] = 1.0 * damp_up;                                            // :221, :229
  clip_tree(t, nullptr);                                                                                   // :241
}

}  // namespace gaq
