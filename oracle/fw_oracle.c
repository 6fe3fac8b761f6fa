/*
 * fw_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * PARITY UNPINNED vs PyBullet: the reference's arithmetic lives in un-vendored,
 * un-pinned PyFlyt + pybullet (absent from /root/reference and from this
 * image); the reference has no tests, fixtures or golden vectors.  This file
 * is a plain-C, double-precision, scalar restatement of
 *   (a) the reference's own step loop / reward / observation code, cited
 *       file:line below (paths relative to the reference repo root), and
 *   (b) the published algorithms of the absent dependencies (PyFlyt's
 *       Khan&Nahon-2015 flat-plate lifting surface, first-order motor,
 *       WaypointHandler; Bullet's semi-implicit Euler + exponential-map
 *       quaternion update, getEulerFromQuaternion / getQuaternionFromEuler /
 *       getMatrixFromQuaternion), restated from the SURVEY.md appendix A spec,
 *       with every constant that the reference does not contain exposed in
 *       fw_config (include/fwsim.h).
 * It is pinned only by the known-answer tests derivable from the reference
 * text (tests/test_oracle_known_answers.py; SURVEY.md section 8c list).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (pyflyt-drone_amd/) never does.
 *
 * Exported API mirrors include/fwsim.h one-to-one with the prefix fwo_ and
 * HOST pointers everywhere.
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/fwsim.h"

#define FWO_PI 3.14159265358979323846

/* ------------------------------------------------------------------------- */
/* Counter-based RNG: Philox4x32-10 (Salmon et al., SC'11).                  */
/* key = (seed lo, seed hi); counter = (block, episode, global env id, stream)*/
/* ------------------------------------------------------------------------- */
enum { FWO_STREAM_SCENARIO = 0, FWO_STREAM_NOISE = 1 };

static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 64 random bits number j of (stream, episode, env) */
static uint64_t rng_u64(uint64_t seed, uint32_t env, uint32_t episode, uint32_t stream, uint32_t j) {
  uint32_t ctr[4] = { j >> 1, episode, env, stream };
  uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
  uint32_t o[4];
  philox4x32_10(ctr, key, o);
  uint32_t lo = o[2 * (j & 1)], hi = o[2 * (j & 1) + 1];
  return ((uint64_t)hi << 32) | lo;
}
static double rng_uniform01(uint64_t seed, uint32_t env, uint32_t ep, uint32_t stream, uint32_t j) {
  return (double)(rng_u64(seed, env, ep, stream, j) >> 11) * (1.0 / 9007199254740992.0);
}
static double rng_uniform(uint64_t seed, uint32_t env, uint32_t ep, uint32_t j, double lo, double hi) {
  /* numpy Generator.uniform: low + (high-low)*random() */
  return lo + (hi - lo) * rng_uniform01(seed, env, ep, FWO_STREAM_SCENARIO, j);
}
/* two standard normals for Aviary step `astep` (Box-Muller) */
static void rng_normal2(uint64_t seed, uint32_t env, uint32_t ep, uint32_t astep, double z[2]) {
  uint64_t a = rng_u64(seed, env, ep, FWO_STREAM_NOISE, 2 * astep);
  uint64_t b = rng_u64(seed, env, ep, FWO_STREAM_NOISE, 2 * astep + 1);
  double u1 = (double)((a >> 11) + 1) * (1.0 / 9007199254740992.0); /* (0,1] */
  double u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);       /* [0,1) */
  double r = sqrt(-2.0 * log(u1));
  z[0] = r * cos(2.0 * FWO_PI * u2);
  z[1] = r * sin(2.0 * FWO_PI * u2);
}

/* scenario draw indices (stream FWO_STREAM_SCENARIO) */
enum {
  J_WIND_BASE = 0, J_WIND_AMP = 3, J_WIND_PHASE = 6,
  J_THETA = 8, J_PHI = 16, J_DIST = 24,
  J_DUCK_X = 32, J_DUCK_Y = 33, J_DUCK_YAW = 34,
  J_OBST = 40 /* + 3*attempt + {h,x,y} */
};

/* ------------------------------------------------------------------------- */
/* small vector helpers                                                      */
/* ------------------------------------------------------------------------- */
static double dot3(const double a[3], const double b[3]) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1]*b[2] - a[2]*b[1]; o[1] = a[2]*b[0] - a[0]*b[2]; o[2] = a[0]*b[1] - a[1]*b[0];
}
static double norm3(const double a[3]) { return sqrt(dot3(a, a)); }

/* Bullet btMatrix3x3::setRotation == pybullet.getMatrixFromQuaternion
 * (call sites envs/fixedwing_objlock_env.py:275, fixedwing_waypoint_objlock_env.py:239) */
static void mat_from_quat(const double q[4], double m[9]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double d = x*x + y*y + z*z + w*w;
  double s = 2.0 / d;
  double xs = x*s, ys = y*s, zs = z*s;
  double wx = w*xs, wy = w*ys, wz = w*zs;
  double xx = x*xs, xy = x*ys, xz = x*zs;
  double yy = y*ys, yz = y*zs, zz = z*zs;
  m[0] = 1.0 - (yy + zz); m[1] = xy - wz;         m[2] = xz + wy;
  m[3] = xy + wz;         m[4] = 1.0 - (xx + zz); m[5] = yz - wx;
  m[6] = xz - wy;         m[7] = yz + wx;         m[8] = 1.0 - (xx + yy);
}
static void mat_vec(const double m[9], const double v[3], double o[3]) {   /* o = M v */
  o[0] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  o[1] = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  o[2] = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
}
static void matT_vec(const double m[9], const double v[3], double o[3]) {  /* o = M^T v */
  o[0] = m[0]*v[0] + m[3]*v[1] + m[6]*v[2];
  o[1] = m[1]*v[0] + m[4]*v[1] + m[7]*v[2];
  o[2] = m[2]*v[0] + m[5]*v[1] + m[8]*v[2];
}

/* pybullet.getEulerFromQuaternion (Bullet getEulerZYX with the gimbal guard) */
static void euler_from_quat(const double q[4], double e[3]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double sqx = x*x, sqy = y*y, sqz = z*z, squ = w*w;
  double sarg = -2.0 * (x*z - w*y);
  if (sarg <= -0.99999) {
    e[0] = 0.0; e[1] = -0.5 * FWO_PI; e[2] = 2.0 * atan2(x, -y);
  } else if (sarg >= 0.99999) {
    e[0] = 0.0; e[1] = 0.5 * FWO_PI; e[2] = 2.0 * atan2(-x, y);
  } else {
    e[0] = atan2(2.0 * (y*z + w*x), squ - sqx - sqy + sqz);
    e[1] = asin(sarg);
    e[2] = atan2(2.0 * (x*y + w*z), squ + sqx - sqy - sqz);
  }
}
/* pybullet.getQuaternionFromEuler (Bullet setEulerZYX(yaw,pitch,roll));
 * call site envs/fixedwing_envs/fixedwing_base_env.py:288 */
static void quat_from_euler(const double e[3], double q[4]) {
  double hr = 0.5 * e[0], hp = 0.5 * e[1], hy = 0.5 * e[2];
  double cr = cos(hr), sr = sin(hr), cp = cos(hp), sp = sin(hp), cy = cos(hy), sy = sin(hy);
  q[0] = sr*cp*cy - cr*sp*sy;
  q[1] = cr*sp*cy + sr*cp*sy;
  q[2] = cr*cp*sy - sr*sp*cy;
  q[3] = cr*cp*cy + sr*sp*sy;
}

/* ------------------------------------------------------------------------- */
/* env struct                                                                */
/* ------------------------------------------------------------------------- */
typedef struct {
  double pos[3], quat[4], vel[3], omega[3];      /* world-frame velocities (Bullet base) */
  double act[FW_NUM_ACTUATORS];                  /* 5 surface actuations + throttle */
  double action[4];                              /* raw action (fixedwing_base_env.py:328) */
  double setpoint[4];                            /* roll,pitch,yaw,thrust01 (:330-331) */
  int64_t step_count, tick_count, episode;
  int termination, truncation, collision, oob, env_complete;
  int contact;                                   /* Aviary.contact_array any() */
  int num_reached;
  int obs_target_index;                          /* num_reached as seen by the last compute_state() */
  double new_distance, old_distance;
  double wind_base[3], wind_amp[3], wind_phase;
  double ep_return;
  int64_t ep_len;
  double targets[FW_MAX_TARGETS][3];
  double reward;
  /* observation as last computed by compute_state() */
  double attitude[23];
  double target_deltas[FW_MAX_TARGETS + 1][3];
  int n_deltas;
  /* objlock tail (kept in canonical-record order, see FW_ST_*) */
  double task[FW_STATE_DIM - FW_S_TASK];
  double obj_target_vector[3], obj_duck_vision[FW_VISION_HIST * FW_VISION_FEATS + 4];
  int duck_strike;
} oenv;

typedef struct {
  double area, aspect, Cl_alpha_3D, alpha_0_base, alpha_stall_P_base, alpha_stall_N_base;
  double theta_f, tau_f, torque_unit[3];
} osurf_derived;

struct fw_env {            /* the opaque handle type of fwsim.h, oracle flavour */
  fw_config cfg;
  int n;
  uint64_t seed;
  int64_t env_offset;
  oenv* e;
  osurf_derived sd[FW_NUM_SURFACES];
  double max_rpm;
  double inertia[9], inertia_inv[9];
  int obs_dim, att_dim;
  /* analytic camera (objlock tasks): body-frame axes, focal length [px], image size, duck radius */
  double cam_f[3], cam_r[3], cam_d[3], cam_focal, duck_radius;
  int cam_w, cam_h, camera_ratio_ticks;
  int64_t max_steps;
  int env_step_ratio, ticks_per_aviary;
  char err[256];
};
static char g_err[256];

/* ------------------------------------------------------------------------- */
/* config validation (fixedwing_base_env.py:48-57,65-72,113-115,128-133)     */
/* ------------------------------------------------------------------------- */
static int validate(const fw_config* c, char* msg, int n) {
#define FAIL(...) do { if (msg && n > 0) snprintf(msg, (size_t)n, __VA_ARGS__); return FW_EINVAL; } while (0)
  if (c->abi_version != FW_ABI_VERSION) { if (msg && n > 0) snprintf(msg, (size_t)n, "abi_version %d != %d", c->abi_version, FW_ABI_VERSION); return FW_EVERSION; }
  if (c->agent_hz <= 0 || 120 % c->agent_hz != 0) {
    int lowest = c->agent_hz > 0 ? (int)(120 / ((int)(120 / c->agent_hz) + 1)) : 1;
    int highest = (c->agent_hz > 0 && c->agent_hz <= 120) ? (int)(120 / (int)(120 / c->agent_hz)) : 120;
    FAIL("`agent_hz` must be round denominator of 120, try %d or %d.", lowest, highest);
  }
  if (c->angle_representation != 0 && c->angle_representation != 1)
    FAIL("angle_representation must be either `euler` or `quaternion`, not %d", c->angle_representation);
  if (c->wind_mode < FW_WIND_OFF || c->wind_mode > FW_WIND_GUST_SINE) FAIL("Unsupported wind mode: %d", c->wind_mode);
  if (c->wind_mode != FW_WIND_OFF && c->wind_randomize_on_reset) {
    for (int i = 0; i < 3; ++i) {
      if (!(c->wind_enu_mps_range[i][0] <= c->wind_enu_mps_range[i][1])) FAIL("Invalid wind_enu_mps_range");
      if (!(c->gust_amp_enu_mps_range[i][0] <= c->gust_amp_enu_mps_range[i][1])) FAIL("Invalid gust_amp_enu_mps_range");
    }
  }
  if (c->task < FW_TASK_WAYPOINTS || c->task > FW_TASK_WAYPOINT_OBJLOCK) FAIL("unknown task %d", c->task);
  if (c->dtype != FW_F64 && c->dtype != FW_F32) FAIL("unknown dtype %d", c->dtype);
  if (c->num_targets < 0 || c->num_targets > FW_MAX_TARGETS) FAIL("num_targets must be in [0,%d]", FW_MAX_TARGETS);
  if (c->task != FW_TASK_OBJLOCK && (c->context_length < 0 || c->context_length > FW_MAX_TARGETS + 1)) FAIL("bad context_length");
  if (c->n_collision_pts < 0 || c->n_collision_pts > FW_MAX_COLLISION_PTS) FAIL("bad n_collision_pts");
  if (c->num_obstacles < 0 || c->num_obstacles > FW_MAX_OBSTACLES) FAIL("bad num_obstacles");
  if (c->physics_hz <= 0 || c->control_hz <= 0 || c->physics_hz % c->control_hz != 0) FAIL("physics_hz must be a multiple of control_hz");
  if (!(c->mass > 0.0)) FAIL("mass must be > 0");
  if (c->wind_coupling < FW_WIND_COUPLE_NONE || c->wind_coupling > FW_WIND_COUPLE_AIRSPEED) FAIL("bad wind_coupling");
  return FW_OK;
#undef FAIL
}

static int obs_dim_of(const fw_config* c) {
  int att = (c->angle_representation == 0 ? 12 : 13) + 4 + 6;
  /* duck_vision_use_deltas=False (envs/fixedwing_objlock_env.py:163-165, 440-441): the history alone */
  if (c->task == FW_TASK_OBJLOCK) return att + 3 + FW_VISION_FEATS * FW_VISION_HIST + (c->duck_vision_no_deltas ? 0 : 4);
  return att + 3 * c->context_length;
}

int32_t fwo_sizeof_config(void) { return (int32_t)sizeof(fw_config); }
int32_t fwo_abi_version(void) { return FW_ABI_VERSION; }
int32_t fwo_state_dim(void) { return FW_STATE_DIM; }
int32_t fwo_obs_dim(const fw_config* c) { return c ? obs_dim_of(c) : FW_EINVAL; }
int32_t fwo_validate_config(const fw_config* c, char* msg, int32_t n) { return validate(c, msg, n); }

/* ------------------------------------------------------------------------- */
/* Lifting surface (PyFlyt LiftingSurface, Khan & Nahon 2015) */
/* ------------------------------------------------------------------------- */
static double deg2rad(double d) { return d * (FWO_PI / 180.0); }

static void derive_surface(const fw_surface_params* p, osurf_derived* d) {
  d->area = p->chord * p->span;
  d->aspect = p->span / p->chord;
  d->Cl_alpha_3D = p->Cl_alpha_2D * (d->aspect / (d->aspect + ((2.0 * (d->aspect + 4.0)) / (d->aspect + 2.0))));
  d->alpha_0_base = deg2rad(p->alpha_0_base_deg);
  d->alpha_stall_P_base = deg2rad(p->alpha_stall_P_base_deg);
  d->alpha_stall_N_base = deg2rad(p->alpha_stall_N_base_deg);
  d->theta_f = acos(2.0 * p->flap_to_chord - 1.0);
  d->tau_f = 1.0 - ((d->theta_f - sin(d->theta_f)) / FWO_PI);
  cross3(p->lift_unit, p->forward_unit, d->torque_unit);
}

/* numpy.interp(x, [x0,x1], [y0,y1]) incl. its end clamping */
static double interp2(double x, double x0, double x1, double y0, double y1) {
  if (x <= x0) return y0;
  if (x >= x1) return y1;
  return y0 + (y1 - y0) * ((x - x0) / (x1 - x0));
}

/* returns Cl, Cd, CM for angle of attack alpha [rad], flap deflection [rad] */
static void aero_coeffs(const fw_surface_params* p, const osurf_derived* d, double alpha, double defl,
                        double* Cl_o, double* Cd_o, double* CM_o) {
  double Cl3 = d->Cl_alpha_3D, AR = d->aspect;
  double delta_Cl = Cl3 * d->tau_f * p->eta * defl;
  double delta_Cl_max = p->flap_to_chord * delta_Cl;
  double Cl_max_P = Cl3 * (d->alpha_stall_P_base - d->alpha_0_base) + delta_Cl_max;
  double Cl_max_N = Cl3 * (d->alpha_stall_N_base - d->alpha_0_base) + delta_Cl_max;
  double alpha_0 = d->alpha_0_base - (delta_Cl / Cl3);
  double alpha_stall_P = alpha_0 + (Cl_max_P / Cl3);
  double alpha_stall_N = alpha_0 + (Cl_max_N / Cl3);
  double Cl, Cd, CM;

  if (alpha_stall_N < alpha && alpha < alpha_stall_P) {          /* no stall */
    Cl = Cl3 * (alpha - alpha_0);
    double alpha_i = Cl / (FWO_PI * AR);
    double alpha_eff = alpha - alpha_0 - alpha_i;
    double CT = p->Cd_0 * cos(alpha_eff);
    double CN = (Cl + (CT * sin(alpha_eff))) / cos(alpha_eff);
    Cd = (CN * sin(alpha_eff)) + (CT * cos(alpha_eff));
    CM = -CN * (0.25 - (0.175 * (1.0 - ((2.0 * alpha_eff) / FWO_PI))));
  } else {
    double alpha_i;
    if (alpha > 0.0) {                                            /* positive stall */
      double Cl_stall = Cl3 * (alpha_stall_P - alpha_0);
      double alpha_i_at_stall = Cl_stall / (FWO_PI * AR);
      alpha_i = interp2(alpha, alpha_stall_P, FWO_PI / 2.0, alpha_i_at_stall, 0.0);
    } else {                                                      /* negative stall */
      double Cl_stall = Cl3 * (alpha_stall_N - alpha_0);
      double alpha_i_at_stall = Cl_stall / (FWO_PI * AR);
      alpha_i = interp2(alpha, -FWO_PI / 2.0, alpha_stall_N, 0.0, alpha_i_at_stall);
    }
    double alpha_eff = alpha - alpha_0 - alpha_i;
    double Cd_90 = (-4.26e-2 * (defl * defl)) + (2.1e-1 * defl) + 1.98;   /* defl in rad (Khan&Nahon) */
    double CN = Cd_90 * sin(alpha_eff) *
                (1.0 / (0.56 + 0.44 * fabs(sin(alpha_eff))) - 0.41 * (1.0 - exp(-17.0 / AR)));
    double CT = 0.5 * p->Cd_0 * cos(alpha_eff);
    Cl = (CN * cos(alpha_eff)) - (CT * sin(alpha_eff));
    Cd = (CN * sin(alpha_eff)) + (CT * cos(alpha_eff));
    CM = -CN * (0.25 - (0.175 * (1.0 - ((2.0 * fabs(alpha_eff)) / FWO_PI))));
  }
  *Cl_o = Cl; *Cd_o = Cd; *CM_o = CM;
}

/* force/torque of one surface in the body(link) frame given its local air velocity */
static void surface_force(const fw_config* c, const fw_surface_params* p, const osurf_derived* d,
                          double actuation, const double v_local[3], double f[3], double tq[3]) {
  double lifting_airspeed = dot3(v_local, p->lift_unit);
  double forward_airspeed = dot3(v_local, p->forward_unit);
  double freestream_speed = sqrt(forward_airspeed * forward_airspeed + lifting_airspeed * lifting_airspeed);
  double alpha = atan2(-lifting_airspeed, forward_airspeed);
  double defl = deg2rad(p->deflection_limit_deg * actuation);
  double Cl, Cd, CM;
  aero_coeffs(p, d, alpha, defl, &Cl, &Cd, &CM);
  double Q = 0.5 * c->air_density * (freestream_speed * freestream_speed);
  double Q_area = Q * d->area;
  double lift = Cl * Q_area, drag = Cd * Q_area;
  double force_normal = (lift * cos(alpha)) + (drag * sin(alpha));
  double force_parallel = (lift * sin(alpha)) - (drag * cos(alpha));
  for (int k = 0; k < 3; ++k) {
    f[k] = p->lift_unit[k] * force_normal + p->forward_unit[k] * force_parallel;
    tq[k] = Q_area * CM * p->chord * d->torque_unit[k];
  }
}

/* ------------------------------------------------------------------------- */
/* wind (fixedwing_base_env.py:145-171)                                      */
/* ------------------------------------------------------------------------- */
static void wind_at(const struct fw_env* h, const oenv* e, double t, double w[3]) {
  const fw_config* c = &h->cfg;
  if (c->wind_mode == FW_WIND_OFF) { w[0] = w[1] = w[2] = 0.0; return; }
  if (c->wind_mode == FW_WIND_CONSTANT) { for (int k = 0; k < 3; ++k) w[k] = e->wind_base[k]; return; }
  double s = sin(2.0 * FWO_PI * c->gust_freq_hz * t + e->wind_phase);
  for (int k = 0; k < 3; ++k) w[k] = e->wind_base[k] + e->wind_amp[k] * s;
}

/* ------------------------------------------------------------------------- */
/* one physics tick: actuators -> wrench -> Bullet-style integration         */
/* ------------------------------------------------------------------------- */
static void physics_tick(struct fw_env* h, oenv* e, uint32_t genv, int tick_in_aviary, const double z2[2]) {
  const fw_config* c = &h->cfg;
  const double dt = 1.0 / (double)c->physics_hz;
  (void)genv;

  /* update_control (mode 0 mixer; cmd is refreshed on control ticks only, but the
   * setpoint is constant within an agent step so every tick sees the same cmd) */
  double cmd[FW_NUM_ACTUATORS];
  for (int a = 0; a < FW_NUM_ACTUATORS; ++a) {
    cmd[a] = 0.0;
    for (int k = 0; k < 4; ++k) cmd[a] += c->mixer[a][k] * e->setpoint[k];
  }

  /* update_physics: actuator lags */
  for (int s = 0; s < FW_NUM_SURFACES; ++s)
    e->act[s] += (dt / c->surfaces[s].tau) * (cmd[s] - e->act[s]);
  double* thr = &e->act[FW_NUM_SURFACES];
  *thr += (dt / c->motor.tau) * (cmd[FW_NUM_SURFACES] - *thr);
  *thr += z2[tick_in_aviary] * (*thr) * c->motor.noise_ratio;

  double R[9];
  mat_from_quat(e->quat, R);
  double v_b[3], w_b[3];
  matT_vec(R, e->vel, v_b);
  matT_vec(R, e->omega, w_b);

  double wind[3], wind_b[3] = {0, 0, 0};
  wind_at(h, e, (double)e->tick_count / (double)c->physics_hz, wind);
  if (c->wind_coupling == FW_WIND_COUPLE_AIRSPEED) matT_vec(R, wind, wind_b);

  double F_b[3] = {0, 0, 0}, T_b[3] = {0, 0, 0};
  for (int s = 0; s < FW_NUM_SURFACES; ++s) {
    const fw_surface_params* p = &c->surfaces[s];
    double wxr[3], v_local[3], f[3], tq[3], rxf[3];
    cross3(w_b, p->pos, wxr);
    for (int k = 0; k < 3; ++k) v_local[k] = v_b[k] + wxr[k] - wind_b[k];
    surface_force(c, p, &h->sd[s], e->act[s], v_local, f, tq);
    cross3(p->pos, f, rxf);
    for (int k = 0; k < 3; ++k) { F_b[k] += f[k]; T_b[k] += rxf[k] + tq[k]; }
  }
  { /* motor (PyFlyt Motors): rpm = throttle*max_rpm; thrust = rpm^2 k_T; torque = rpm^2 k_Q */
    double rpm = (*thr) * h->max_rpm;
    double f[3], tq[3], rxf[3];
    for (int k = 0; k < 3; ++k) {
      f[k] = (rpm * rpm) * c->motor.thrust_coef * c->motor.thrust_unit[k];
      tq[k] = (rpm * rpm) * c->motor.torque_coef * c->motor.thrust_unit[k];
    }
    cross3(c->motor.pos, f, rxf);
    for (int k = 0; k < 3; ++k) { F_b[k] += f[k]; T_b[k] += rxf[k] + tq[k]; }
  }

  /* stepSimulation: single rigid body (all URDF joints fixed), semi-implicit Euler */
  double F_w[3], acc[3];
  mat_vec(R, F_b, F_w);
  if (c->wind_coupling == FW_WIND_COUPLE_FORCE)
    for (int k = 0; k < 3; ++k) F_w[k] += c->wind_force_coef * wind[k];
  for (int k = 0; k < 3; ++k) acc[k] = F_w[k] / c->mass;
  acc[2] -= c->gravity;

  double Iw[3], gyro[3] = {0, 0, 0}, rhs[3], alpha_b[3], alpha_w[3];
  mat_vec(h->inertia, w_b, Iw);
  if (c->gyroscopic) cross3(w_b, Iw, gyro);
  for (int k = 0; k < 3; ++k) rhs[k] = T_b[k] - gyro[k];
  mat_vec(h->inertia_inv, rhs, alpha_b);
  mat_vec(R, alpha_b, alpha_w);

  for (int k = 0; k < 3; ++k) { e->vel[k] += acc[k] * dt; e->omega[k] += alpha_w[k] * dt; }
  for (int k = 0; k < 3; ++k) e->pos[k] += e->vel[k] * dt;

  /* quaternion exponential map (Bullet pQuatUpdate): q <- dq(omega*dt) * q, normalised */
  {
    double wv[3] = { e->omega[0], e->omega[1], e->omega[2] };
    double ang = norm3(wv);
    if (ang * dt > 0.5 * FWO_PI * 0.5) ang = (0.5 * FWO_PI * 0.5) / dt;   /* ANGULAR_MOTION_THRESHOLD */
    double ax[3], cw;
    if (ang < 0.001) {
      double k = 0.5 * dt - (dt * dt * dt) * 0.020833333333 * ang * ang;
      for (int i = 0; i < 3; ++i) ax[i] = wv[i] * k;
    } else {
      double k = sin(0.5 * ang * dt) / ang;
      for (int i = 0; i < 3; ++i) ax[i] = wv[i] * k;
    }
    cw = cos(ang * dt * 0.5);
    double x = e->quat[0], y = e->quat[1], z = e->quat[2], w = e->quat[3];
    /* dq * q, dq = (ax, cw) */
    double nx = cw * x + ax[0] * w + ax[1] * z - ax[2] * y;
    double ny = cw * y + ax[1] * w + ax[2] * x - ax[0] * z;
    double nz = cw * z + ax[2] * w + ax[0] * y - ax[1] * x;
    double nw = cw * w - ax[0] * x - ax[1] * y - ax[2] * z;
    double inv = 1.0 / sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
    e->quat[0] = nx * inv; e->quat[1] = ny * inv; e->quat[2] = nz * inv; e->quat[3] = nw * inv;
  }
  e->tick_count += 1;

  /* contacts: analytic ground plane z=0 against body-fixed points */
  mat_from_quat(e->quat, R);
  for (int i = 0; i < c->n_collision_pts; ++i) {
    double zc = e->pos[2] + R[6] * c->collision_pts[i][0] + R[7] * c->collision_pts[i][1] + R[8] * c->collision_pts[i][2];
    if (zc <= 0.0) e->contact = 1;
  }
}

static void object_contacts(const struct fw_env* h, oenv* e);
static void camera_capture(const struct fw_env* h, oenv* e);
/* Aviary.step(): contact_array &= False; updates_per_step ticks */
static void aviary_step(struct fw_env* h, oenv* e, uint32_t genv) {
  e->contact = 0;
  double z2[2] = {0.0, 0.0};
  if (h->cfg.motor.noise_ratio != 0.0)
    rng_normal2(h->seed, genv, (uint32_t)e->episode, (uint32_t)(e->tick_count / h->ticks_per_aviary), z2);
  for (int t = 0; t < h->ticks_per_aviary; ++t) {
    physics_tick(h, e, genv, t & 1, z2);
    if (h->cfg.task != FW_TASK_WAYPOINTS) object_contacts(h, e);
  }
  /* update_last(): the camera captures every physics_camera_ratio ticks (:631-641) */
  if (h->cfg.task != FW_TASK_WAYPOINTS && h->camera_ratio_ticks > 0 && (e->tick_count % h->camera_ratio_ticks) == 0)
    camera_capture(h, e);
}

/* ------------------------------------------------------------------------- */
/* compute_attitude + task state                                             */
/* ------------------------------------------------------------------------- */
static int n_targets_left(const struct fw_env* h, const oenv* e) { return h->cfg.num_targets - e->num_reached; }

/* fixedwing_base_env.py:267-290 + PyFlyt Fixedwing.update_state (body-frame velocities) */
static void compute_attitude(const struct fw_env* h, oenv* e, double ang_pos[3], double quat_rt[4]) {
  double R[9], ang_vel[3], lin_vel[3];
  mat_from_quat(e->quat, R);
  matT_vec(R, e->omega, ang_vel);
  matT_vec(R, e->vel, lin_vel);
  euler_from_quat(e->quat, ang_pos);
  quat_from_euler(ang_pos, quat_rt);          /* p.getQuaternionFromEuler(ang_pos) :288 */
  int o = 0;
  for (int k = 0; k < 3; ++k) e->attitude[o++] = ang_vel[k];
  if (h->cfg.angle_representation == 0) for (int k = 0; k < 3; ++k) e->attitude[o++] = ang_pos[k];
  else for (int k = 0; k < 4; ++k) e->attitude[o++] = quat_rt[k];
  for (int k = 0; k < 3; ++k) e->attitude[o++] = lin_vel[k];
  for (int k = 0; k < 3; ++k) e->attitude[o++] = e->pos[k];
  for (int k = 0; k < 4; ++k) e->attitude[o++] = e->action[k];
  for (int k = 0; k < FW_NUM_ACTUATORS; ++k) e->attitude[o++] = e->act[k];
}

/* FixedwingWaypointsEnv.compute_state (upstream; mirrored at
 * envs/fixedwing_waypoint_objlock_env.py:197-227) + WaypointHandler.distance_to_targets */
static void compute_state_waypoints(struct fw_env* h, oenv* e) {
  double ang_pos[3], q[4], R[9];
  compute_attitude(h, e, ang_pos, q);
  mat_from_quat(q, R);
  int nleft = n_targets_left(h, e);
  e->n_deltas = nleft;
  e->obs_target_index = e->num_reached;
  for (int i = 0; i < nleft; ++i) {
    double d[3];
    for (int k = 0; k < 3; ++k) d[k] = e->targets[e->num_reached + i][k] - e->pos[k];
    matT_vec(R, d, e->target_deltas[i]);      /* np.matmul(targets - lin_pos, rotation) */
  }
  if (nleft > 0) {
    e->old_distance = e->new_distance;
    e->new_distance = norm3(e->target_deltas[0]);
  }
}

/* fixedwing_base_env.py:296-312 */
static void compute_base_term_trunc_reward(struct fw_env* h, oenv* e) {
  if (e->step_count > h->max_steps) e->truncation |= 1;
  if (e->contact) { e->reward = -100.0; e->collision = 1; e->termination |= 1; }
  if (norm3(e->pos) > h->cfg.flight_dome_size) { e->reward = -100.0; e->oob = 1; e->termination |= 1; }
}

/* upstream FixedwingWaypointsEnv.compute_term_trunc_reward (mirrored at
 * envs/fixedwing_waypoint_objlock_env.py:286-294) */
static void compute_term_trunc_reward_waypoints(struct fw_env* h, oenv* e) {
  compute_base_term_trunc_reward(h, e);
  if (n_targets_left(h, e) <= 0) return;
  if (!h->cfg.sparse_reward) {
    double progress = (e->old_distance != 0.0) ? (e->old_distance - e->new_distance) : 0.0;
    e->reward += fmax(3.0 * progress, 0.0);
    e->reward += 1.0 / e->new_distance;
  }
  if (e->new_distance < h->cfg.goal_reach_distance) {      /* target_reached */
    e->reward = 100.0;
    e->num_reached += 1;                                   /* advance_targets */
    int all = (n_targets_left(h, e) == 0);
    e->truncation |= all;
    e->env_complete = all;
  }
}


/* ========================================================================= */
/* ObjLock task (envs/fixedwing_objlock_env.py).                              */
/* The reference derives its vision features from PyBullet's rendered          */
/* segmentation / depth images (:643-761).  Rendering is replaced by an        */
/* ANALYTIC camera (build-owned): the duck is a sphere,   */
/* obstacles are vertical cylinders, the ground is the plane z = 0; a "frame"   */
/* is the 8 numbers the reference extracts from the images.  Everything         */
/* downstream of the frame follows the reference line by line.                  */
/* ========================================================================= */
#define TK(e, k) ((e)->task[(k)])
static double f32r(double x) { return (double)(float)x; }

/* _reset_duck_state :409-419 */
static void reset_duck_state(oenv* e) {
  TK(e, FW_ST_LOCK_STEPS) = 0.0; TK(e, FW_ST_PREV_EST) = -1.0;
  TK(e, FW_ST_LAST_CX) = 0.5; TK(e, FW_ST_LAST_CY) = 0.5; TK(e, FW_ST_LAST_AREA) = 0.0; TK(e, FW_ST_LAST_DEPTH) = 0.0;
  TK(e, FW_ST_SINCE_SEEN) = 60.0; TK(e, FW_ST_HIST_FILLED) = 0.0; TK(e, FW_ST_FRAME_HAS) = 0.0;
  for (int k = 0; k < 8; ++k) TK(e, FW_ST_FRAME + k) = 0.0;
  for (int k = 0; k < FW_VISION_HIST * FW_VISION_FEATS; ++k) TK(e, FW_ST_HIST + k) = 0.0;
  TK(e, FW_ST_DUCK_PHASE) = 0.0; TK(e, FW_ST_SEEN_CONSEC) = 0.0;
}

/* _spawn_duck :461-491 and _spawn_obstacles :507-565 (draw order: x, y, yaw; then per attempt h, x, y) */
static void objlock_spawn(struct fw_env* h, oenv* e, uint32_t genv, uint32_t ep) {
  const fw_config* c = &h->cfg;
  const double r = c->flight_dome_size / 2.0;
  TK(e, FW_ST_DUCK_POS + 0) = rng_uniform(h->seed, genv, ep, J_DUCK_X, -r, r);
  TK(e, FW_ST_DUCK_POS + 1) = rng_uniform(h->seed, genv, ep, J_DUCK_Y, -r, r);
  TK(e, FW_ST_DUCK_POS + 2) = 0.05;
  (void)rng_uniform(h->seed, genv, ep, J_DUCK_YAW, -FWO_PI, FWO_PI);      /* yaw: drawn, irrelevant for a sphere */
  int n = 0;
  for (int i = 0; i < c->num_obstacles; ++i) {
    double hh = rng_uniform(h->seed, genv, ep, J_OBST + 3 * i + 0, c->obstacle_height_range[0], c->obstacle_height_range[1]);
    double x = rng_uniform(h->seed, genv, ep, J_OBST + 3 * i + 1, -r, r);
    double y = rng_uniform(h->seed, genv, ep, J_OBST + 3 * i + 2, -r, r);
    double dx = x - TK(e, FW_ST_DUCK_POS), dy = y - TK(e, FW_ST_DUCK_POS + 1);
    if (sqrt(dx * dx + dy * dy) < 10.0) continue;                          /* :536-539 */
    if (x * x + y * y < 100.0) continue;                                   /* :542-543 */
    TK(e, FW_ST_OBST + 3 * n + 0) = x; TK(e, FW_ST_OBST + 3 * n + 1) = y; TK(e, FW_ST_OBST + 3 * n + 2) = hh;
    ++n;
  }
  for (int i = n; i < FW_MAX_OBSTACLES; ++i) for (int k = 0; k < 3; ++k) TK(e, FW_ST_OBST + 3 * i + k) = 0.0;
  TK(e, FW_ST_NUM_OBST) = (double)n;
}

/* contacts with the duck sphere and the obstacle cylinders (Aviary.contact_array) */
static void object_contacts(const struct fw_env* h, oenv* e) {
  const fw_config* c = &h->cfg;
  double R[9];
  mat_from_quat(e->quat, R);
  const double cx = TK(e, FW_ST_DUCK_POS), cy = TK(e, FW_ST_DUCK_POS + 1), cz = TK(e, FW_ST_DUCK_POS + 2) + h->duck_radius;
  const int nob = (int)TK(e, FW_ST_NUM_OBST);
  for (int i = 0; i < c->n_collision_pts; ++i) {
    double pb[3] = { c->collision_pts[i][0], c->collision_pts[i][1], c->collision_pts[i][2] }, pw[3];
    mat_vec(R, pb, pw);
    for (int k = 0; k < 3; ++k) pw[k] += e->pos[k];
    double dx = pw[0] - cx, dy = pw[1] - cy, dz = pw[2] - cz;
    if (dx * dx + dy * dy + dz * dz <= h->duck_radius * h->duck_radius) e->contact = 1;
    for (int o = 0; o < nob; ++o) {
      double ox = pw[0] - TK(e, FW_ST_OBST + 3 * o), oy = pw[1] - TK(e, FW_ST_OBST + 3 * o + 1);
      if (ox * ox + oy * oy <= c->obstacle_radius * c->obstacle_radius && pw[2] <= TK(e, FW_ST_OBST + 3 * o + 2)) e->contact = 1;
    }
  }
}

/* depth along the view axis of the first thing a camera ray hits (ground or cylinder), clamped to [near, far] */
static double ray_depth(const struct fw_env* h, const oenv* e, const double cam[3], const double dw[3]) {
  const fw_config* c = &h->cfg;
  double best = c->camera_far;
  if (dw[2] < 0.0) { double t = -cam[2] / dw[2]; if (t > 0.0 && t < best) best = t; }
  const int nob = (int)TK(e, FW_ST_NUM_OBST);
  for (int o = 0; o < nob; ++o) {
    double ox = cam[0] - TK(e, FW_ST_OBST + 3 * o), oy = cam[1] - TK(e, FW_ST_OBST + 3 * o + 1), hh = TK(e, FW_ST_OBST + 3 * o + 2);
    double a = dw[0] * dw[0] + dw[1] * dw[1], b = 2.0 * (ox * dw[0] + oy * dw[1]), cc = ox * ox + oy * oy - c->obstacle_radius * c->obstacle_radius;
    if (a <= 0.0) continue;
    double disc = b * b - 4.0 * a * cc;
    if (disc < 0.0) continue;
    double t = (-b - sqrt(disc)) / (2.0 * a);
    if (t <= 0.0) continue;
    double z = cam[2] + t * dw[2];
    if (z < 0.0 || z > hh) continue;
    if (t < best) best = t;
  }
  return best < c->camera_near ? c->camera_near : best;
}

/* does the segment cam -> P pass through an obstacle cylinder? (occlusion of the duck) */
static int occluded(const struct fw_env* h, const oenv* e, const double cam[3], const double P[3]) {
  const fw_config* c = &h->cfg;
  double dw[3] = { P[0] - cam[0], P[1] - cam[1], P[2] - cam[2] };
  const int nob = (int)TK(e, FW_ST_NUM_OBST);
  for (int o = 0; o < nob; ++o) {
    double ox = cam[0] - TK(e, FW_ST_OBST + 3 * o), oy = cam[1] - TK(e, FW_ST_OBST + 3 * o + 1), hh = TK(e, FW_ST_OBST + 3 * o + 2);
    double a = dw[0] * dw[0] + dw[1] * dw[1], b = 2.0 * (ox * dw[0] + oy * dw[1]), cc = ox * ox + oy * oy - c->obstacle_radius * c->obstacle_radius;
    if (a <= 0.0) continue;
    double disc = b * b - 4.0 * a * cc;
    if (disc < 0.0) continue;
    double t = (-b - sqrt(disc)) / (2.0 * a);
    if (t <= 0.0 || t >= 1.0) continue;
    double z = cam[2] + t * dw[2];
    if (z >= 0.0 && z <= hh) return 1;
  }
  return 0;
}

/* Depth-buffer value of a fragment at view-axis depth t: the inverse of _depth_buffer_to_meters (:691-696),
 * z = far*near / (far - (far-near)*d)  <=>  d = far*(t-near) / (t*(far-near)).  The analytic depth image is float64 (PyBullet's
 * depthImg is float32: a quantisation of <= 3e-8 of the buffer value that belongs to the renderer, not to the functionals). */
static double depth_buffer_of(double t, double near, double far) {
  if (t < near) t = near;
  if (t > far) t = far;
  return far * (t - near) / (t * (far - near));
}
static double depth_buffer_to_meters(double d) {           /* :691-696 */
  const double near = 0.1, far = 255.0;
  double denom = far - (far - near) * d;
  if (fabs(denom) < 1e-9) return far;
  return (far * near) / denom;
}

/* Camera.capture_image() replaced by an analytic render of the scene (duck = sphere resting on the ground, obstacles =
 * vertical cylinders, ground = plane z = 0, sky = depth buffer 1.0), pixel centres at (x, y), x to the right, y down.
 * The frame holds the 8 numbers the reference computes FROM THE IMAGES, by the same functionals:
 *   duck mask (seg == duck id)            = the pixels whose ray hits the sphere between the clip planes, unless the line
 *                                           of sight to the sphere's centre is blocked by a cylinder (then no duck pixel)
 *   cx, cy   = mean(xs)/(w-1), mean(ys)/(h-1) over the mask                                           :670-674
 *   area     = count_nonzero(mask)/(h*w)                                                              :675
 *   depth_m  = metres(min of the depth buffer over the mask)                                          :731-743
 *   d_left / d_center / d_right = metres(mean of the DEPTH-BUFFER values of the non-duck pixels of the thirds
 *              [0, w//3), [w//3, 2w//3), [2w//3, w) of row h//2), 0.0 for an empty third or a zero mean  :698-729 */
static void camera_capture(const struct fw_env* h, oenv* e) {
  const fw_config* c = &h->cfg;
  double R[9], cam[3], off_w[3];
  mat_from_quat(e->quat, R);
  mat_vec(R, c->camera_offset, off_w);
  for (int k = 0; k < 3; ++k) cam[k] = e->pos[k] + off_w[k];
  const int Wi = h->cam_w, Hi = h->cam_h;
  const double W = (double)Wi, H = (double)Hi, F = h->cam_focal, near = c->camera_near, far = c->camera_far;
  const double u0 = 0.5 * (W - 1.0), v0 = 0.5 * (H - 1.0), Rd = h->duck_radius;
  double* fr = &TK(e, FW_ST_FRAME);
  /* --- duck mask statistics: literal loop over the pixels of a conservative bounding box --- */
  double C[3] = { TK(e, FW_ST_DUCK_POS), TK(e, FW_ST_DUCK_POS + 1), TK(e, FW_ST_DUCK_POS + 2) + Rd };
  double relw[3] = { C[0] - cam[0], C[1] - cam[1], C[2] - cam[2] }, relb[3];
  matT_vec(R, relw, relb);
  const double zc = dot3(relb, h->cam_f), xc = dot3(relb, h->cam_r), yc = dot3(relb, h->cam_d);
  const double k2 = zc * zc + xc * xc + yc * yc - Rd * Rd;
  double visible = 0.0, cx = 0.0, cy = 0.0, area = 0.0, depth = 0.0;
  int duck_in_frame = 0;
  if (zc - Rd > near && zc - Rd < far && !occluded(h, e, cam, C)) {
    /* |a - xc/zc| <= R sqrt(zc^2 + xc^2) / ((zc - R) zc) for every point of the sphere (a = P_r / P_f) */
    const double hw = F * Rd * sqrt(zc * zc + xc * xc) / ((zc - Rd) * zc) + 2.0, hh = F * Rd * sqrt(zc * zc + yc * yc) / ((zc - Rd) * zc) + 2.0;
    const double uc = u0 + F * xc / zc, vc = v0 + F * yc / zc;
    int xa = (int)floor(fmax(uc - hw, 0.0)), xb = (int)ceil(fmin(uc + hw, W - 1.0));
    int ya = (int)floor(fmax(vc - hh, 0.0)), yb = (int)ceil(fmin(vc + hh, H - 1.0));
    double count = 0.0, sx = 0.0, sy = 0.0, dmin = 2.0;
    for (int y = ya; y <= yb; ++y)
      for (int x = xa; x <= xb; ++x) {
        const double a = ((double)x - u0) / F, b = ((double)y - v0) / F;
        const double q = 1.0 + a * a + b * b, p = zc + a * xc + b * yc, disc = p * p - q * k2;
        if (disc < 0.0 || p <= 0.0) continue;
        const double t = (p - sqrt(disc)) / q;               /* view-axis depth of the first hit */
        if (!(t > near && t < far)) continue;
        count += 1.0; sx += (double)x; sy += (double)y;
        const double d = depth_buffer_of(t, near, far);
        if (d < dmin) dmin = d;
      }
    if (count > 0.0) {
      duck_in_frame = 1;
      visible = 1.0;
      cx = (sx / count) / fmax(1.0, W - 1.0);               /* mean(xs)/(w-1)  :673 */
      cy = (sy / count) / fmax(1.0, H - 1.0);
      area = count / fmax(1.0, H * W);                      /* count/(h*w)     :675 */
      depth = depth_buffer_to_meters(dmin);                 /* :731-743 */
    }
  }
  fr[0] = visible; fr[1] = cx; fr[2] = cy; fr[3] = area; fr[4] = depth;
  /* --- obstacle zones: mean depth-buffer value of the non-duck pixels of each third of row h//2  :698-729 --- */
  const int y_mid = Hi / 2, x_1 = Wi / 3, x_2 = (2 * Wi) / 3;
  const double bm = ((double)y_mid - v0) / F;
  double zsum[3] = {0.0, 0.0, 0.0}, zcnt[3] = {0.0, 0.0, 0.0};
  for (int x = 0; x < Wi; ++x) {
    const double a = ((double)x - u0) / F;
    if (duck_in_frame) {                                    /* mask = seg != duck_id */
      const double q = 1.0 + a * a + bm * bm, p = zc + a * xc + bm * yc, disc = p * p - q * k2;
      if (disc >= 0.0 && p > 0.0) { const double t = (p - sqrt(disc)) / q; if (t > near && t < far) continue; }
    }
    double db[3], dw[3];
    for (int k = 0; k < 3; ++k) db[k] = h->cam_f[k] + a * h->cam_r[k] + bm * h->cam_d[k];
    mat_vec(R, db, dw);
    const int z = x < x_1 ? 0 : (x < x_2 ? 1 : 2);
    zsum[z] += depth_buffer_of(ray_depth(h, e, cam, dw), near, far);    /* sky: ray_depth = far -> buffer 1.0 */
    zcnt[z] += 1.0;
  }
  for (int z = 0; z < 3; ++z) {
    const double mean = zcnt[z] > 0.0 ? zsum[z] / zcnt[z] : 0.0;                       /* float(np.mean(vals)) :718 */
    /* `d_buf > 0.0` (:725-727) with a guard band: a third whose every fragment sits on the near plane has mean buffer exactly 0
     * here and 1e-16 in an implementation that sums in another order; below 1e-12 (t within 1e-14 m of the near plane) is "0" */
    fr[5 + z] = mean > 1e-12 ? depth_buffer_to_meters(mean) : 0.0;
  }
  TK(e, FW_ST_FRAME_HAS) = 1.0;
}

/* FPV image of the SAME analytic scene camera_capture() applies the reference's functionals to, rendered pixel by pixel at
 * `res` x `res` (focal length scaled with the width: same FOV, same body-fixed camera; res == camera_resolution is the image
 * a14's numbers are functionals of).  What Camera.capture_image() hands the env in the reference (:603-622): a segmentation
 * image and a depth-buffer image -- here channel 0 = duck mask (1.0 where seg == duck id), channel 1 = depth-buffer value in
 * [0, 1] of the nearest fragment (duck pixels: the sphere; others: ground / cylinder; sky 1.0).  A network that consumes it
 * (the reference's CNN path replaces segImg by a network's mask, envs/fixedwing_envs/objlock_yolo_env.py:646-716) sees the
 * scene of the env's CURRENT pose.  out: float32 [2][res][res] of env e. */
static void render_env(const struct fw_env* h, const oenv* e, int res, float* out) {
  const fw_config* c = &h->cfg;
  double R[9], cam[3], off_w[3];
  mat_from_quat(e->quat, R);
  mat_vec(R, c->camera_offset, off_w);
  for (int k = 0; k < 3; ++k) cam[k] = e->pos[k] + off_w[k];
  const double W = (double)res, F = 0.5 * W / tan(0.5 * c->camera_fov_deg * (FWO_PI / 180.0));
  const double near = c->camera_near, far = c->camera_far, u0 = 0.5 * (W - 1.0), v0 = u0, Rd = h->duck_radius;
  double C[3] = { TK(e, FW_ST_DUCK_POS), TK(e, FW_ST_DUCK_POS + 1), TK(e, FW_ST_DUCK_POS + 2) + Rd };
  double relw[3] = { C[0] - cam[0], C[1] - cam[1], C[2] - cam[2] }, relb[3];
  matT_vec(R, relw, relb);
  const double zc = dot3(relb, h->cam_f), xc = dot3(relb, h->cam_r), yc = dot3(relb, h->cam_d);
  const double k2 = zc * zc + xc * xc + yc * yc - Rd * Rd;
  const int duck_possible = (zc - Rd > near && zc - Rd < far && !occluded(h, e, cam, C));
  for (int y = 0; y < res; ++y)
    for (int x = 0; x < res; ++x) {
      const double a = ((double)x - u0) / F, b = ((double)y - v0) / F;
      double mask = 0.0, d;
      int is_duck = 0;
      double t_duck = 0.0;
      if (duck_possible) {
        const double q = 1.0 + a * a + b * b, p = zc + a * xc + b * yc, disc = p * p - q * k2;
        if (disc >= 0.0 && p > 0.0) { t_duck = (p - sqrt(disc)) / q; if (t_duck > near && t_duck < far) is_duck = 1; }
      }
      if (is_duck) { mask = 1.0; d = depth_buffer_of(t_duck, near, far); }
      else {
        double db[3], dw[3];
        for (int k = 0; k < 3; ++k) db[k] = h->cam_f[k] + a * h->cam_r[k] + b * h->cam_d[k];
        mat_vec(R, db, dw);
        d = depth_buffer_of(ray_depth(h, e, cam, dw), near, far);
      }
      out[(size_t)y * res + x] = (float)mask;
      out[(size_t)res * res + (size_t)y * res + x] = (float)d;
    }
}

/* _compute_vision_features :643-689 on the latest frame; returns the 9 float32 features */
static void vision_features(oenv* e, double out[FW_VISION_FEATS]) {
  double visible = 0.0, dl = 0.0, dc = 0.0, dr = 0.0;
  if (TK(e, FW_ST_FRAME_HAS) != 0.0) {
    const double* fr = &TK(e, FW_ST_FRAME);
    dl = fr[5]; dc = fr[6]; dr = fr[7];
    if (fr[0] == 0.0) {
      TK(e, FW_ST_SINCE_SEEN) = fmin(TK(e, FW_ST_SINCE_SEEN) + 1.0, 60.0);            /* :664 */
    } else {
      TK(e, FW_ST_LAST_CX) = fr[1]; TK(e, FW_ST_LAST_CY) = fr[2]; TK(e, FW_ST_LAST_AREA) = fr[3];
      TK(e, FW_ST_LAST_DEPTH) = fr[4]; TK(e, FW_ST_SINCE_SEEN) = 0.0; visible = 1.0;   /* :673-681 */
    }
  }
  /* _build_vision_vector :745-761 (np.float32) */
  out[0] = visible; out[1] = f32r(TK(e, FW_ST_LAST_CX)); out[2] = f32r(TK(e, FW_ST_LAST_CY));
  out[3] = f32r(TK(e, FW_ST_LAST_AREA)); out[4] = f32r(TK(e, FW_ST_LAST_DEPTH));
  out[5] = f32r(TK(e, FW_ST_SINCE_SEEN) / 60.0); out[6] = f32r(dl); out[7] = f32r(dc); out[8] = f32r(dr);
}

/* the 31-float duck_vision observation as a pure function of the stored history:
 * after the shift of :437-442, base = hist[0] and prev = hist[1] */
static void duck_vision_from_hist(const oenv* e, double out[FW_VISION_HIST * FW_VISION_FEATS + 4]) {
  const double* hist = &e->task[FW_ST_HIST];
  for (int k = 0; k < FW_VISION_HIST * FW_VISION_FEATS; ++k) out[k] = hist[k];
  double* dl = out + FW_VISION_HIST * FW_VISION_FEATS;
  dl[0] = dl[1] = dl[2] = dl[3] = 0.0;
  if (e->task[FW_ST_HIST_FILLED] >= 2.0 && hist[0] > 0.5 && hist[FW_VISION_FEATS] > 0.5)
    for (int k = 0; k < 4; ++k) dl[k] = (double)((float)hist[1 + k] - (float)hist[FW_VISION_FEATS + 1 + k]);   /* float32 arithmetic :454-457 */
}

/* per-env scratch of the objlock observation */
typedef struct { double target_vector[3]; double duck_vision[FW_VISION_HIST * FW_VISION_FEATS + 4]; } oobj;

/* compute_state :253-287 (+ _build_duck_vision_observation :421-459) */
static void compute_state_objlock(struct fw_env* h, oenv* e, oobj* ob) {
  double ang_pos[3], q[4], R[9], diff[3];
  compute_attitude(h, e, ang_pos, q);
  mat_from_quat(q, R);
  for (int k = 0; k < 3; ++k) diff[k] = TK(e, FW_ST_DUCK_POS + k) - e->pos[k];
  matT_vec(R, diff, ob->target_vector);                                   /* rot.T @ diff :277 */
  double base[FW_VISION_FEATS];
  vision_features(e, base);
  double* hist = &TK(e, FW_ST_HIST);
  for (int r = FW_VISION_HIST - 1; r >= 1; --r) for (int k = 0; k < FW_VISION_FEATS; ++k) hist[r * FW_VISION_FEATS + k] = hist[(r - 1) * FW_VISION_FEATS + k];
  for (int k = 0; k < FW_VISION_FEATS; ++k) hist[k] = base[k];
  TK(e, FW_ST_HIST_FILLED) = fmin(TK(e, FW_ST_HIST_FILLED) + 1.0, (double)FW_VISION_HIST);
  duck_vision_from_hist(e, ob->duck_vision);
  e->n_deltas = 0;
  e->obs_target_index = 0;
}

/* _apply_obstacle_avoidance_reward :376-407 */
static void obstacle_penalty(const struct fw_env* h, oenv* e, const double vis[FW_VISION_FEATS], double scale_mult) {
  double d_obs = INFINITY; int any = 0;
  for (int k = 6; k < 9; ++k) if (vis[k] > 0.0 && isfinite(vis[k])) { if (vis[k] < d_obs) d_obs = vis[k]; any = 1; }
  if (!any) return;
  double d_safe = h->cfg.obstacle_safe_distance_m;
  if (d_safe <= 0.0 || d_obs >= d_safe) return;
  double penalty = h->cfg.obstacle_avoid_reward_scale * scale_mult * (d_safe - d_obs) / d_safe;
  if (penalty > h->cfg.obstacle_avoid_max_penalty) penalty = h->cfg.obstacle_avoid_max_penalty;
  e->reward -= penalty;
}

/* compute_term_trunc_reward :289-372 */
static void compute_term_trunc_reward_objlock(struct fw_env* h, oenv* e, const oobj* ob, int* duck_strike) {
  const fw_config* c = &h->cfg;
  compute_base_term_trunc_reward(h, e);
  if (e->collision || e->oob) return;                                      /* :293-294 */
  const double* vis = ob->duck_vision;                                     /* newest frame = first 9 entries */
  obstacle_penalty(h, e, vis, 0.5);                                        /* :403 */
  const double dist_to_duck = norm3(ob->target_vector);
  if (!c->sparse_reward) {
    e->reward += c->duck_distance_reward_scale / fmax(dist_to_duck, 2.0);  /* :303 */
    if (vis[0] > 0.5) {
      double cx = vis[1], cy = vis[2], area = vis[3], est_dist = vis[4];
      e->reward += c->duck_visible_step_reward;
      e->reward += c->duck_area_reward_scale * fmax(0.0, area);
      double dist_to_center = sqrt((cx - 0.5) * (cx - 0.5) + (cy - 0.5) * (cy - 0.5));
      double r_lock = fmax(c->duck_lock_center_radius, 1e-6);
      double center_score = fmax(0.0, (r_lock - dist_to_center) / r_lock);
      e->reward += c->duck_centering_reward_scale * center_score;
      if (dist_to_center < r_lock) {
        TK(e, FW_ST_LOCK_STEPS) = fmin(TK(e, FW_ST_LOCK_STEPS) + 1.0, (double)c->duck_lock_hold_steps);
        e->reward += c->duck_lock_step_reward;
      } else {
        TK(e, FW_ST_LOCK_STEPS) = fmax(TK(e, FW_ST_LOCK_STEPS) - (double)c->duck_lock_decay_steps, 0.0);
      }
      if (TK(e, FW_ST_PREV_EST) >= 0.0 && est_dist > 0.0 && isfinite(est_dist)) {
        double diff = TK(e, FW_ST_PREV_EST) - est_dist, clip_m = c->duck_approach_reward_clip_m;
        if (clip_m > 0.0) diff = fmax(-clip_m, fmin(diff, clip_m));
        e->reward += diff * c->duck_approach_reward_scale;
      }
      TK(e, FW_ST_PREV_EST) = (est_dist > 0.0 && isfinite(est_dist)) ? est_dist : -1.0;
    } else {
      if (TK(e, FW_ST_LOCK_STEPS) > 0.0) e->reward -= c->duck_lock_lost_penalty;
      TK(e, FW_ST_LOCK_STEPS) = fmax(TK(e, FW_ST_LOCK_STEPS) - (double)c->duck_lock_decay_steps, 0.0);
      TK(e, FW_ST_PREV_EST) = -1.0;
    }
  }
  if (TK(e, FW_ST_LOCK_STEPS) >= (double)c->duck_lock_hold_steps && dist_to_duck <= c->duck_strike_distance_m) {   /* :366-372 */
    e->termination = 1;
    e->reward += c->duck_strike_reward;
    e->env_complete = 1;
    *duck_strike = 1;
  }
}


/* ========================================================================= */
/* Combined task (envs/fixedwing_waypoint_objlock_env.py): N waypoints, then   */
/* the duck.  Same analytic camera; no vision history in the observation.       */
/* ========================================================================= */
/* _spawn_duck :394-436 (at the last waypoint's x,y, on the ground) and _spawn_obstacles :452-503 */
static void combined_spawn(struct fw_env* h, oenv* e, uint32_t genv, uint32_t ep) {
  const fw_config* c = &h->cfg;
  if (c->num_targets > 0) {
    TK(e, FW_ST_DUCK_POS + 0) = e->targets[c->num_targets - 1][0];
    TK(e, FW_ST_DUCK_POS + 1) = e->targets[c->num_targets - 1][1];
  } else { TK(e, FW_ST_DUCK_POS + 0) = 10.0; TK(e, FW_ST_DUCK_POS + 1) = 0.0; }       /* :417 */
  TK(e, FW_ST_DUCK_POS + 2) = 0.05;
  (void)rng_uniform(h->seed, genv, ep, J_DUCK_YAW, -FWO_PI, FWO_PI);
  const double r = c->flight_dome_size / 2.0;
  int n = 0;
  for (int i = 0; i < c->num_obstacles; ++i) {
    double hh = rng_uniform(h->seed, genv, ep, J_OBST + 3 * i + 0, c->obstacle_height_range[0], c->obstacle_height_range[1]);
    double x = rng_uniform(h->seed, genv, ep, J_OBST + 3 * i + 1, -r, r);
    double y = rng_uniform(h->seed, genv, ep, J_OBST + 3 * i + 2, -r, r);
    if (x * x + y * y < 100.0) continue;                                   /* :480-481 (no duck-distance rejection here) */
    TK(e, FW_ST_OBST + 3 * n + 0) = x; TK(e, FW_ST_OBST + 3 * n + 1) = y; TK(e, FW_ST_OBST + 3 * n + 2) = hh;
    ++n;
  }
  for (int i = n; i < FW_MAX_OBSTACLES; ++i) for (int k = 0; k < 3; ++k) TK(e, FW_ST_OBST + 3 * i + k) = 0.0;
  TK(e, FW_ST_NUM_OBST) = (double)n;
}

/* compute_state :197-276 */
static void compute_state_combined(struct fw_env* h, oenv* e) {
  double ang_pos[3], q[4], R[9];
  compute_attitude(h, e, ang_pos, q);
  mat_from_quat(q, R);
  const int nleft = n_targets_left(h, e);
  e->obs_target_index = e->num_reached;
  for (int i = 0; i < nleft; ++i) {
    double d[3];
    for (int k = 0; k < 3; ++k) d[k] = e->targets[e->num_reached + i][k] - e->pos[k];
    matT_vec(R, d, e->target_deltas[i]);
  }
  if (nleft > 0) { e->old_distance = e->new_distance; e->new_distance = norm3(e->target_deltas[0]); }
  {                                                                        /* duck delta appended as the last row :234-246 */
    double d[3];
    for (int k = 0; k < 3; ++k) d[k] = TK(e, FW_ST_DUCK_POS + k) - e->pos[k];
    matT_vec(R, d, e->target_deltas[nleft]);
  }
  e->n_deltas = nleft + 1;
  double feature[FW_VISION_FEATS];
  vision_features(e, feature);
  for (int k = 0; k < FW_VISION_FEATS; ++k) e->obj_duck_vision[k] = feature[k];
  int phase = (int)TK(e, FW_ST_DUCK_PHASE);
  if (nleft == 0) {                                                        /* all_targets_reached :255-270 */
    phase |= 2;
    if (!(phase & 1)) {
      int visible = feature[0] > 0.5 && TK(e, FW_ST_LAST_AREA) >= h->cfg.duck_switch_min_area;
      TK(e, FW_ST_SEEN_CONSEC) = visible ? TK(e, FW_ST_SEEN_CONSEC) + 1.0 : 0.0;
      if (TK(e, FW_ST_SEEN_CONSEC) >= (double)h->cfg.duck_switch_min_consecutive_seen) phase |= 1;
    }
  } else {
    phase = 0;                                                             /* :271-273 */
  }
  TK(e, FW_ST_DUCK_PHASE) = (double)phase;
}

/* compute_term_trunc_reward :278-343 */
static void compute_term_trunc_reward_combined(struct fw_env* h, oenv* e) {
  const fw_config* c = &h->cfg;
  compute_base_term_trunc_reward(h, e);
  if (e->collision || e->oob) return;                                      /* :282-283 */
  if (n_targets_left(h, e) > 0) {                                          /* waypoint phase :286-302 */
    if (!c->sparse_reward) {
      double progress = (e->old_distance != 0.0) ? (e->old_distance - e->new_distance) : 0.0;
      e->reward += fmax(3.0 * progress, 0.0);
      e->reward += 1.0 / e->new_distance;
    }
    if (e->new_distance < c->goal_reach_distance) {
      e->reward = 100.0;
      e->num_reached += 1;
      if (n_targets_left(h, e) == 0) { e->termination = 0; e->truncation = 0; }   /* :297-300 */
    }
    obstacle_penalty(h, e, e->obj_duck_vision, 1.0);
  } else {                                                                 /* duck phase :305-343 */
    e->termination = 0;
    obstacle_penalty(h, e, e->obj_duck_vision, 0.5);
    if (((int)TK(e, FW_ST_DUCK_PHASE)) & 1) {
      const double last_depth = TK(e, FW_ST_LAST_DEPTH), cx = TK(e, FW_ST_LAST_CX), cy = TK(e, FW_ST_LAST_CY);
      if (!c->sparse_reward && last_depth > 0.0) e->reward += 1.0 / fmax(last_depth, 2.0);
      if (cx > 0.0) {
        double dc = sqrt((cx - 0.5) * (cx - 0.5) + (cy - 0.5) * (cy - 0.5));
        if (dc < 0.35) { TK(e, FW_ST_LOCK_STEPS) += 1.0; e->reward += c->duck_lock_step_reward; }
        else TK(e, FW_ST_LOCK_STEPS) = 0.0;
      } else {
        TK(e, FW_ST_LOCK_STEPS) = 0.0;
      }
      const double est = last_depth;
      if (TK(e, FW_ST_PREV_EST) >= 0.0 && est > 0.0) {
        double diff = TK(e, FW_ST_PREV_EST) - est;
        if (diff > 0.0) e->reward += diff * c->duck_approach_reward_scale;
      }
      TK(e, FW_ST_PREV_EST) = est;
      if (TK(e, FW_ST_LOCK_STEPS) >= (double)c->duck_lock_hold_steps && est > 0.0 && est <= c->duck_strike_distance_m) {
        e->termination = 1;
        e->reward += c->duck_strike_reward;
        e->env_complete = 1;
        e->duck_strike = 1;
      }
    }
  }
}

static void compute_state(struct fw_env* h, oenv* e) {
  switch (h->cfg.task) {
    case FW_TASK_OBJLOCK: {
      oobj ob;
      compute_state_objlock(h, e, &ob);
      memcpy(e->obj_target_vector, ob.target_vector, sizeof ob.target_vector);
      memcpy(e->obj_duck_vision, ob.duck_vision, sizeof ob.duck_vision);
      break;
    }
    case FW_TASK_WAYPOINT_OBJLOCK: compute_state_combined(h, e); break;
    case FW_TASK_WAYPOINTS: default: compute_state_waypoints(h, e); break;
  }
}
static void compute_term_trunc_reward(struct fw_env* h, oenv* e) {
  switch (h->cfg.task) {
    case FW_TASK_OBJLOCK: {
      oobj ob;
      memcpy(ob.target_vector, e->obj_target_vector, sizeof ob.target_vector);
      memcpy(ob.duck_vision, e->obj_duck_vision, sizeof ob.duck_vision);
      compute_term_trunc_reward_objlock(h, e, &ob, &e->duck_strike);
      break;
    }
    case FW_TASK_WAYPOINT_OBJLOCK: compute_term_trunc_reward_combined(h, e); break;
    case FW_TASK_WAYPOINTS: default: compute_term_trunc_reward_waypoints(h, e); break;
  }
}

/* flatten: envs/flatten_waypoint_env.py:52-72 */
static void flatten_obs(const struct fw_env* h, const oenv* e, double* out) {
  int att = h->att_dim, o = 0;
  if (h->cfg.task == FW_TASK_OBJLOCK) {          /* envs/flatten_objlock_env.py:41-46: concat(...).astype(np.float32) */
    for (int k = 0; k < att; ++k) out[o++] = (double)(float)e->attitude[k];
    for (int k = 0; k < 3; ++k) out[o++] = (double)(float)e->obj_target_vector[k];
    const int nv = FW_VISION_HIST * FW_VISION_FEATS + (h->cfg.duck_vision_no_deltas ? 0 : 4);      /* :440-441 */
    for (int k = 0; k < nv; ++k) out[o++] = e->obj_duck_vision[k];
    return;
  }
  for (int k = 0; k < att; ++k) out[o++] = e->attitude[k];
  int ctx = h->cfg.context_length;
  for (int i = 0; i < ctx; ++i)
    for (int k = 0; k < 3; ++k) out[o++] = (i < e->n_deltas) ? e->target_deltas[i][k] : 0.0;
}

/* ------------------------------------------------------------------------- */
/* reset: begin_reset / scenario / end_reset (fixedwing_base_env.py:193-257)  */
/* ------------------------------------------------------------------------- */
static void reset_duck_state(oenv* e);
static void objlock_spawn(struct fw_env* h, oenv* e, uint32_t genv, uint32_t ep);
static void combined_spawn(struct fw_env* h, oenv* e, uint32_t genv, uint32_t ep);
/* `ov` / `li`: caller-supplied scenario of the new episode (fw_scenario, may be NULL) and the env's local index in it */
static void env_reset(struct fw_env* h, oenv* e, uint32_t genv, const fw_scenario* ov, int li) {
  const fw_config* c = &h->cfg;
  e->episode += 1;                       /* index of the episode that starts now */
  uint32_t ep = (uint32_t)e->episode;
  e->step_count = 0; e->termination = 0; e->truncation = 0;
  e->collision = 0; e->oob = 0; e->env_complete = 0; e->contact = 0;
  e->reward = 0.0; e->ep_return = 0.0; e->ep_len = 0;
  for (int k = 0; k < 4; ++k) { e->action[k] = 0.0; e->setpoint[k] = 0.0; }
  /* Aviary(...): start pose, PyFlyt starting_velocity, zero actuators */
  for (int k = 0; k < 3; ++k) { e->pos[k] = c->start_pos[k]; e->vel[k] = c->start_vel[k]; e->omega[k] = 0.0; }
  quat_from_euler(c->start_orn, e->quat);
  for (int k = 0; k < FW_NUM_ACTUATORS; ++k) e->act[k] = 0.0;
  e->tick_count = 0;
  /* wind sampling order: base(3), gust amp(3), phase (fixedwing_base_env.py:139-165) */
  for (int k = 0; k < 3; ++k) {
    e->wind_base[k] = c->wind_enu_mps[k];
    e->wind_amp[k] = c->gust_amp_enu_mps[k];
  }
  e->wind_phase = c->gust_phase_rad;
  if (c->wind_mode != FW_WIND_OFF && c->wind_randomize_on_reset) {
    for (int k = 0; k < 3; ++k)
      e->wind_base[k] = rng_uniform(h->seed, genv, ep, J_WIND_BASE + k, c->wind_enu_mps_range[k][0], c->wind_enu_mps_range[k][1]);
    if (c->wind_mode == FW_WIND_GUST_SINE) {
      for (int k = 0; k < 3; ++k)
        e->wind_amp[k] = rng_uniform(h->seed, genv, ep, J_WIND_AMP + k, c->gust_amp_enu_mps_range[k][0], c->gust_amp_enu_mps_range[k][1]);
      if (c->wind_randomize_phase) e->wind_phase = rng_uniform(h->seed, genv, ep, J_WIND_PHASE, 0.0, 2.0 * FWO_PI);
    }
  }
  if (ov && c->wind_mode != FW_WIND_OFF) {
    if (ov->wind_base) for (int k = 0; k < 3; ++k) e->wind_base[k] = ov->wind_base[3 * li + k];
    if (ov->gust_amp) for (int k = 0; k < 3; ++k) e->wind_amp[k] = ov->gust_amp[3 * li + k];
    if (ov->gust_phase) e->wind_phase = ov->gust_phase[li];
  }
  /* WaypointHandler.reset: polar sampling */
  e->num_reached = 0; e->new_distance = 0.0; e->old_distance = 0.0;
  memset(e->targets, 0, sizeof(e->targets));
  if (c->task != FW_TASK_OBJLOCK) {
    for (int i = 0; i < c->num_targets; ++i) {
      double theta = rng_uniform(h->seed, genv, ep, J_THETA + i, 0.0, 2.0 * FWO_PI);
      double phi = rng_uniform(h->seed, genv, ep, J_PHI + i, 0.0, 2.0 * FWO_PI);
      double dist = rng_uniform(h->seed, genv, ep, J_DIST + i, 1.0, c->waypoint_spawn_size * 0.9);
      double x = dist * sin(phi) * cos(theta);
      double y = dist * sin(phi) * sin(theta);
      double z = fabs(dist * cos(phi));
      e->targets[i][0] = x; e->targets[i][1] = y;
      e->targets[i][2] = z > c->waypoint_min_height ? z : c->waypoint_min_height;
    }
    if (ov && ov->targets)
      for (int t = 0; t < c->num_targets; ++t) for (int k = 0; k < 3; ++k) e->targets[t][k] = ov->targets[((size_t)li * FW_MAX_TARGETS + t) * 3 + k];
  }
  memset(e->task, 0, sizeof(e->task));
  e->duck_strike = 0;
  if (c->task == FW_TASK_OBJLOCK) { reset_duck_state(e); objlock_spawn(h, e, genv, ep); }   /* :240-245 */
  if (c->task == FW_TASK_WAYPOINT_OBJLOCK) { reset_duck_state(e); combined_spawn(h, e, genv, ep); }   /* duck under the (possibly supplied) last waypoint */
  if (ov && c->task != FW_TASK_WAYPOINTS) {
    if (ov->duck_pos) for (int k = 0; k < 3; ++k) TK(e, FW_ST_DUCK_POS + k) = ov->duck_pos[3 * li + k];
    if (ov->obstacles && ov->num_obstacles) {
      int n = ov->num_obstacles[li];
      n = n < 0 ? 0 : (n > c->num_obstacles ? c->num_obstacles : n);                   /* never more than the config allows */
      for (int o = 0; o < FW_MAX_OBSTACLES; ++o)
        for (int k = 0; k < 3; ++k) TK(e, FW_ST_OBST + 3 * o + k) = o < n ? ov->obstacles[((size_t)li * FW_MAX_OBSTACLES + o) * 3 + k] : 0.0;
      TK(e, FW_ST_NUM_OBST) = (double)n;
    }
  }
  /* end_reset: 10 warm-up Aviary steps with a zero setpoint, then compute_state */
  for (int i = 0; i < c->warmup_aviary_steps; ++i) aviary_step(h, e, genv);
  compute_state(h, e);
}

/* ------------------------------------------------------------------------- */
/* env.step  (fixedwing_base_env.py:314-348)                                  */
/* ------------------------------------------------------------------------- */
static void env_step(struct fw_env* h, oenv* e, uint32_t genv, const double action[4]) {
  e->reward = -0.1;                                             /* :325 */
  for (int k = 0; k < 4; ++k) { e->action[k] = action[k]; e->setpoint[k] = action[k]; }
  e->setpoint[3] = (e->setpoint[3] / 2.0) + 0.5;                /* :330 */
  for (int i = 0; i < h->env_step_ratio; ++i) {                 /* :334 */
    if (e->termination || e->truncation) break;                 /* :336 */
    aviary_step(h, e, genv);                                    /* :339 */
    compute_state(h, e);                                        /* :342 */
    compute_term_trunc_reward(h, e);                            /* :343 */
  }
  e->step_count += 1;                                           /* :346 */
}

/* ------------------------------------------------------------------------- */
/* exported API                                                               */
/* ------------------------------------------------------------------------- */
static int invert3(const double m[9], double inv[9]) {
  double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  if (det == 0.0) return -1;
  double id = 1.0 / det;
  inv[0] = (m[4] * m[8] - m[5] * m[7]) * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  inv[3] = (m[5] * m[6] - m[3] * m[8]) * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  inv[6] = (m[3] * m[7] - m[4] * m[6]) * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return 0;
}

int32_t fwo_create(const fw_config* cfg, int32_t num_envs, int32_t device, uint64_t seed,
                   int64_t global_env_offset, fw_handle* out) {
  (void)device;
  if (!cfg || !out || num_envs <= 0) { snprintf(g_err, sizeof g_err, "bad arguments"); return FW_EINVAL; }
  int rc = validate(cfg, g_err, (int)sizeof g_err);
  if (rc != FW_OK) return rc;
  struct fw_env* h = (struct fw_env*)calloc(1, sizeof *h);
  if (!h) return FW_ENOMEM;
  h->cfg = *cfg; h->n = num_envs; h->seed = seed; h->env_offset = global_env_offset;
  h->e = (oenv*)calloc((size_t)num_envs, sizeof(oenv));
  if (!h->e) { free(h); return FW_ENOMEM; }
  for (int s = 0; s < FW_NUM_SURFACES; ++s) derive_surface(&cfg->surfaces[s], &h->sd[s]);
  h->max_rpm = sqrt(cfg->motor.total_thrust / cfg->motor.thrust_coef);
  const double* I = cfg->inertia;
  double m[9] = { I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2] };
  memcpy(h->inertia, m, sizeof m);
  if (invert3(m, h->inertia_inv) != 0) { snprintf(g_err, sizeof g_err, "singular inertia"); free(h->e); free(h); return FW_EINVAL; }
  h->att_dim = (cfg->angle_representation == 0 ? 12 : 13) + 4 + 6;
  h->obs_dim = obs_dim_of(cfg);
  h->max_steps = (int64_t)(cfg->agent_hz * cfg->max_duration_seconds);   /* int(agent_hz*max_duration) :101 */
  h->env_step_ratio = (int)(120 / cfg->agent_hz);                        /* :102 */
  h->ticks_per_aviary = cfg->physics_hz / cfg->control_hz;
  {
    double th = cfg->camera_angle_deg * (FWO_PI / 180.0);
    h->cam_f[0] = cos(th); h->cam_f[1] = 0.0; h->cam_f[2] = sin(th);      /* tilt about body y; negative = looking down */
    h->cam_r[0] = 0.0; h->cam_r[1] = -1.0; h->cam_r[2] = 0.0;             /* body y is LEFT => image x grows to the right */
    cross3(h->cam_f, h->cam_r, h->cam_d);                                 /* image y grows downwards */
    h->cam_w = h->cam_h = cfg->camera_resolution > 0 ? cfg->camera_resolution : 128;
    h->cam_focal = 0.5 * h->cam_w / tan(0.5 * cfg->camera_fov_deg * (FWO_PI / 180.0));
    h->duck_radius = cfg->duck_radius_per_scale * cfg->duck_global_scaling;
    h->camera_ratio_ticks = h->ticks_per_aviary * cfg->duck_camera_capture_interval_steps;
  }
  /* envs start un-reset: a terminated shell so that step() before reset() is inert */
  for (int i = 0; i < num_envs; ++i) { h->e[i].termination = 1; h->e[i].quat[3] = 1.0; h->e[i].episode = -1; }
  *out = h;
  return FW_OK;
}

static void store_T(const struct fw_env* h, void* base, size_t idx, double v) {
  if (h->cfg.dtype == FW_F64) ((double*)base)[idx] = v; else ((float*)base)[idx] = (float)v;
}
static double load_T(const struct fw_env* h, const void* base, size_t idx) {
  return h->cfg.dtype == FW_F64 ? ((const double*)base)[idx] : (double)((const float*)base)[idx];
}
static void write_obs(const struct fw_env* h, const oenv* e, void* obs, size_t row) {
  double tmp[64];
  flatten_obs(h, e, tmp);
  for (int k = 0; k < h->obs_dim; ++k) store_T(h, obs, row * (size_t)h->obs_dim + (size_t)k, tmp[k]);
}

int32_t fwo_reset(fw_handle h, const uint8_t* mask, const fw_scenario* scenario, void* obs_out, void* stream) {
  (void)stream;
  if (!h) return FW_EINVAL;
  if (scenario && scenario->obstacles && !scenario->num_obstacles) { snprintf(h->err, sizeof h->err, "fw_scenario: obstacles without num_obstacles"); return FW_EINVAL; }
  for (int i = 0; i < h->n; ++i) {
    if (!mask || mask[i]) env_reset(h, &h->e[i], (uint32_t)(h->env_offset + i), scenario, i);
    if (obs_out) write_obs(h, &h->e[i], obs_out, (size_t)i);
  }
  return FW_OK;
}

int32_t fwo_step(fw_handle h, const void* actions, void* obs, void* reward, uint8_t* terminated,
                 uint8_t* truncated, void* terminal_obs, int32_t* info, void* stream) {
  (void)stream;
  if (!h || !actions) return FW_EINVAL;
  /* envs are independent: the multi-core CPU baseline just splits them over threads */
#pragma omp parallel for schedule(static) if (h->n >= 64)
  for (int i = 0; i < h->n; ++i) {
    oenv* e = &h->e[i];
    uint32_t genv = (uint32_t)(h->env_offset + i);
    double a[4];
    for (int k = 0; k < 4; ++k) a[k] = load_T(h, actions, (size_t)i * 4 + (size_t)k);
    env_step(h, e, genv, a);
    e->ep_return += e->reward; e->ep_len += 1;
    int done = e->termination || e->truncation;
    if (reward) store_T(h, reward, (size_t)i, e->reward);
    if (terminated) terminated[i] = (uint8_t)e->termination;
    if (truncated) truncated[i] = (uint8_t)e->truncation;
    if (info) {
      int32_t* r = info + (size_t)i * FW_INFO_DIM;
      r[FW_INFO_NUM_TARGETS_REACHED] = e->num_reached;
      r[FW_INFO_COLLISION] = e->collision;
      r[FW_INFO_OUT_OF_BOUNDS] = e->oob;
      r[FW_INFO_ENV_COMPLETE] = e->env_complete;
      r[FW_INFO_DUCK_STRIKE] = e->duck_strike;
      r[FW_INFO_IS_SUCCESS] = (h->cfg.task == FW_TASK_OBJLOCK) ? e->duck_strike : 0;
      r[FW_INFO_EP_LEN] = (int32_t)e->ep_len;
      r[FW_INFO_RESERVED] = 0;
    }
    if (done && h->cfg.auto_reset) {
      if (terminal_obs) write_obs(h, e, terminal_obs, (size_t)i);
      env_reset(h, e, genv, NULL, 0);
    }
    if (obs) write_obs(h, e, obs, (size_t)i);
  }
  return FW_OK;
}

int32_t fwo_observe(fw_handle h, void* obs_out, void* stream) {
  (void)stream;
  if (!h || !obs_out) return FW_EINVAL;
  for (int i = 0; i < h->n; ++i) {
    oenv tmp = h->e[i];
    tmp.num_reached = tmp.obs_target_index;
    if (h->cfg.task == FW_TASK_WAYPOINTS)
      compute_state(h, &tmp);        /* on a copy: no new/old-distance side effect */
    write_obs(h, &tmp, obs_out, (size_t)i);
  }
  return FW_OK;
}

int32_t fwo_seed(fw_handle h, uint64_t seed) {
  if (!h) return FW_EINVAL;
  h->seed = seed;
  for (int i = 0; i < h->n; ++i) h->e[i].episode = -1;
  return FW_OK;
}

int32_t fwo_get_state(fw_handle h, double* s) {
  if (!h || !s) return FW_EINVAL;
  for (int i = 0; i < h->n; ++i) {
    const oenv* e = &h->e[i];
    double* r = s + (size_t)i * FW_STATE_DIM;
    memset(r, 0, sizeof(double) * FW_STATE_DIM);
    for (int k = 0; k < 3; ++k) { r[FW_S_POS + k] = e->pos[k]; r[FW_S_VEL + k] = e->vel[k]; r[FW_S_OMEGA + k] = e->omega[k]; }
    /* FW_S_ACTION is the action visible in the env's current observation: a done env in
     * bare-Gymnasium mode keeps its stale self.state although self.action is overwritten (:328) */
    const int aoff = (h->cfg.angle_representation == 0 ? 12 : 13);
    for (int k = 0; k < 4; ++k) { r[FW_S_QUAT + k] = e->quat[k]; r[FW_S_ACTION + k] = e->attitude[aoff + k]; }
    for (int k = 0; k < FW_NUM_ACTUATORS; ++k) r[FW_S_ACT + k] = e->act[k];
    r[FW_S_STEP_COUNT] = (double)e->step_count; r[FW_S_TICK_COUNT] = (double)e->tick_count;
    r[FW_S_EPISODE] = (double)e->episode;
    r[FW_S_FLAGS] = (double)(e->termination | (e->truncation << 1) | (e->collision << 2) | (e->oob << 3) | (e->env_complete << 4) | (e->obs_target_index << 8));
    r[FW_S_NUM_REACHED] = (double)e->num_reached; r[FW_S_NEW_DIST] = e->new_distance;
    for (int k = 0; k < 3; ++k) { r[FW_S_WIND + k] = e->wind_base[k]; r[FW_S_WIND + 3 + k] = e->wind_amp[k]; }
    r[FW_S_WIND + 6] = e->wind_phase;
    r[FW_S_EP_RETURN] = e->ep_return;
    for (int t = 0; t < FW_MAX_TARGETS; ++t) for (int k = 0; k < 3; ++k) r[FW_S_TARGETS + 3 * t + k] = e->targets[t][k];
    memcpy(r + FW_S_TASK, e->task, sizeof(e->task));
  }
  return FW_OK;
}

int32_t fwo_set_state(fw_handle h, const double* s) {
  if (!h || !s) return FW_EINVAL;
  for (int i = 0; i < h->n; ++i) {
    oenv* e = &h->e[i];
    const double* r = s + (size_t)i * FW_STATE_DIM;
    for (int k = 0; k < 3; ++k) { e->pos[k] = r[FW_S_POS + k]; e->vel[k] = r[FW_S_VEL + k]; e->omega[k] = r[FW_S_OMEGA + k]; }
    for (int k = 0; k < 4; ++k) { e->quat[k] = r[FW_S_QUAT + k]; e->action[k] = r[FW_S_ACTION + k]; }
    for (int k = 0; k < FW_NUM_ACTUATORS; ++k) e->act[k] = r[FW_S_ACT + k];
    e->step_count = (int64_t)r[FW_S_STEP_COUNT]; e->tick_count = (int64_t)r[FW_S_TICK_COUNT];
    e->episode = (int64_t)r[FW_S_EPISODE];
    int fl = (int)r[FW_S_FLAGS];
    e->termination = fl & 1; e->truncation = (fl >> 1) & 1; e->collision = (fl >> 2) & 1; e->oob = (fl >> 3) & 1; e->env_complete = (fl >> 4) & 1;
    e->obs_target_index = (fl >> 8) & 15;
    e->num_reached = (int)r[FW_S_NUM_REACHED]; e->new_distance = r[FW_S_NEW_DIST];
    for (int k = 0; k < 3; ++k) { e->wind_base[k] = r[FW_S_WIND + k]; e->wind_amp[k] = r[FW_S_WIND + 3 + k]; }
    e->wind_phase = r[FW_S_WIND + 6];
    e->ep_return = r[FW_S_EP_RETURN];
    e->ep_len = e->step_count;
    for (int t = 0; t < FW_MAX_TARGETS; ++t) for (int k = 0; k < 3; ++k) e->targets[t][k] = r[FW_S_TARGETS + 3 * t + k];
    memcpy(e->task, r + FW_S_TASK, sizeof(e->task));
    oenv tmp = *e;                    /* refresh the cached observation without side effects */
    tmp.num_reached = e->obs_target_index;
    if (h->cfg.task == FW_TASK_OBJLOCK) {   /* vision history is state: rebuild obs from it without shifting */
      double ap[3], q[4], R[9], diff[3];
      compute_attitude(h, &tmp, ap, q);
      mat_from_quat(q, R);
      for (int k = 0; k < 3; ++k) diff[k] = tmp.task[FW_ST_DUCK_POS + k] - tmp.pos[k];
      matT_vec(R, diff, tmp.obj_target_vector);
      duck_vision_from_hist(&tmp, tmp.obj_duck_vision);
      memcpy(e->obj_target_vector, tmp.obj_target_vector, sizeof tmp.obj_target_vector);
      memcpy(e->obj_duck_vision, tmp.obj_duck_vision, sizeof tmp.obj_duck_vision);
    } else if (h->cfg.task == FW_TASK_WAYPOINT_OBJLOCK) {
      double ap[3], q[4], R[9];
      compute_attitude(h, &tmp, ap, q);
      mat_from_quat(q, R);
      const int nl = h->cfg.num_targets - tmp.num_reached;
      for (int t = 0; t <= nl; ++t) {
        double d[3];
        for (int k = 0; k < 3; ++k) d[k] = (t < nl ? tmp.targets[tmp.num_reached + t][k] : tmp.task[FW_ST_DUCK_POS + k]) - tmp.pos[k];
        matT_vec(R, d, tmp.target_deltas[t]);
      }
      tmp.n_deltas = nl + 1;
    } else
    compute_state(h, &tmp);
    memcpy(e->attitude, tmp.attitude, sizeof e->attitude);
    memcpy(e->target_deltas, tmp.target_deltas, sizeof e->target_deltas);
    e->n_deltas = tmp.n_deltas;
  }
  return FW_OK;
}

/* fw_render twin: HOST float32 [N][2][res][res] */
int32_t fwo_render(fw_handle h, int32_t res, float* out, void* stream) {
  (void)stream;
  if (!h || !out) return FW_EINVAL;
  if (h->cfg.task == FW_TASK_WAYPOINTS) { snprintf(h->err, sizeof h->err, "fw_render: the waypoints task has no camera"); return FW_EUNSUPPORTED; }
  if (res < 1 || res > 1024) { snprintf(h->err, sizeof h->err, "fw_render: res must be in [1, 1024]"); return FW_EINVAL; }
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int i = 0; i < h->n; ++i) render_env(h, &h->e[i], res, out + (size_t)i * 2 * res * res);
  return FW_OK;
}

int32_t fwo_set_threads(int32_t n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n; return 1;
#endif
}
int32_t fwo_num_envs(fw_handle h) { return h ? h->n : FW_EINVAL; }
const char* fwo_last_error(fw_handle h) { return h ? h->err : g_err; }
int32_t fwo_destroy(fw_handle h) {
  if (!h) return FW_EINVAL;
  free(h->e); free(h);
  return FW_OK;
}

/* ---- unit-level entry points used by the known-answer tests ---- */
void fwo_aero_coeffs(const fw_surface_params* p, double alpha, double defl, double out[3]) {
  osurf_derived d; derive_surface(p, &d);
  aero_coeffs(p, &d, alpha, defl, &out[0], &out[1], &out[2]);
}
/* derived constants: area, aspect, Cl_alpha_3D, theta_f, tau_f */
void fwo_surface_constants(const fw_surface_params* p, double out[5]) {
  osurf_derived d; derive_surface(p, &d);
  out[0] = d.area; out[1] = d.aspect; out[2] = d.Cl_alpha_3D; out[3] = d.theta_f; out[4] = d.tau_f;
}
void fwo_surface_force(const fw_config* c, int s, double actuation, const double v_local[3], double f[3], double tq[3]) {
  osurf_derived d; derive_surface(&c->surfaces[s], &d);
  surface_force(c, &c->surfaces[s], &d, actuation, v_local, f, tq);
}
void fwo_euler_from_quat(const double q[4], double e[3]) { euler_from_quat(q, e); }
void fwo_quat_from_euler(const double e[3], double q[4]) { quat_from_euler(e, q); }
void fwo_mat_from_quat(const double q[4], double m[9]) { mat_from_quat(q, m); }
void fwo_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }
double fwo_rng_uniform01(uint64_t seed, uint32_t env, uint32_t ep, uint32_t stream, uint32_t j) { return rng_uniform01(seed, env, ep, stream, j); }
void fwo_rng_normal2(uint64_t seed, uint32_t env, uint32_t ep, uint32_t astep, double z[2]) { rng_normal2(seed, env, ep, astep, z); }
/* wind field value at time t for explicit (base, amp, phase) */
void fwo_wind_at(const fw_config* c, const double base[3], const double amp[3], double phase, double t, double w[3]) {
  struct fw_env h; oenv e;
  memset(&h, 0, sizeof h); memset(&e, 0, sizeof e);
  h.cfg = *c;
  for (int k = 0; k < 3; ++k) { e.wind_base[k] = base[k]; e.wind_amp[k] = amp[k]; }
  e.wind_phase = phase;
  wind_at(&h, &e, t, w);
}
/* camera depth-buffer -> metres (envs/fixedwing_objlock_env.py:691-696) */
double fwo_depth_buffer_to_meters(double depth_buffer) { return depth_buffer_to_meters(depth_buffer); }
