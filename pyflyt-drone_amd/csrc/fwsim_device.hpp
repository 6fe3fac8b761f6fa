// fwsim_device.hpp -- gfx950 device code of the vectorised fixed-wing env step.
//
// Two lane mappings of the same physics (template parameter G = lanes per env):
//   G = 1  one wavefront lane per env, the 5 lifting surfaces in a rolled loop whose
//          constants arrive by scalar loads.  Throughput mapping for large N.
//   G = 8  eight lanes per env: lanes 0-4 each evaluate ONE lifting surface with their
//          constants resident in VGPRs, the wrench is summed with 3 DPP steps and the
//          (cheap) rigid-body update is replicated in all 8 lanes.  Latency mapping for
//          small N: 4096 envs become 512 waves instead of 64, and the dependent-issue
//          chain per tick is ~5x shorter (one surface instead of five in series).
// State is SoA in HBM ([field][Npad], coalesced per field); LDS holds the padded
// observation tile so the row-major obs[N,D] the policy GEMM wants is written with
// coalesced stores.  All 8 physics ticks, the 4 reward/termination evaluations, the
// observation and the SB3-style auto-reset of one agent step run in registers inside
// a single launch.  No MFMA: this is element-wise physics.
//
// Algorithm provenance (reference paths relative to the reference repo root):
//   step loop / reward / termination : envs/fixedwing_envs/fixedwing_base_env.py:296-348
//   observation layout               : envs/fixedwing_objlock_env.py:260-267, envs/flatten_waypoint_env.py:52-72
//   wind                             : envs/fixedwing_envs/fixedwing_base_env.py:108-173
//   aero / motor coefficients        : my_models/fixedwing/fixewing.yaml:1-71
//   un-vendored PyFlyt/Bullet parts  : SURVEY.md appendix A (spec), DESIGN.md section 2
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fwsim.h"

namespace fwsim {

constexpr int kWave = 64;
constexpr int kMaxObs = 64;

// ---- SoA field indices (real-valued fields) ----
enum RF : int {
  RF_POS = 0, RF_QUAT = 3, RF_VEL = 7, RF_OMEGA = 10, RF_ACT = 13, RF_ACTION = 19,
  RF_NEW_DIST = 23, RF_WIND = 24, RF_EP_RETURN = 31, RF_TARGETS = 32,
  RF_TASK = 32 + 3 * FW_MAX_TARGETS,              // 56
  RF_COUNT = RF_TASK + (FW_STATE_DIM - FW_S_TASK) // 56 + 115 = 171
};
// ---- SoA integer fields ----
enum IF : int { IF_STEP = 0, IF_TICK = 1, IF_EPISODE = 2, IF_FLAGS = 3, IF_NUM_REACHED = 4, IF_COUNT = 5 };

enum Flags : int { FL_TERM = 1, FL_TRUNC = 2, FL_COLLISION = 4, FL_OOB = 8, FL_COMPLETE = 16, FL_MASK = 0xFF, FL_TGT_SHIFT = 8 };

template <typename T>
struct SurfC {
  T dt_tau;                 // physics_period / tau
  T lift[3], fwd[3], pos[3], tq[3];
  T hra;                    // 0.5 * rho * area
  T chord;
  T Cl3, inv_Cl3, inv_piAR;
  T a0b;                    // alpha_0_base [rad]
  T k_dCl;                  // Cl3 * tau_f * eta * deg2rad(deflection_limit): delta_Cl = k_dCl * actuation
  T ClmaxPb, ClmaxNb;       // Cl3 * (alpha_stall_{P,N}_base - alpha_0_base)
  T ftc;                    // flap_to_chord
  T Cd0;
  T defl_scale;             // deg2rad(deflection_limit)
  T k_exp;                  // 0.41 * (1 - exp(-17/AR))
};

template <typename T>
struct Params {
  SurfC<T> s[FW_NUM_SURFACES];
  T motor_dt_tau, noise_ratio;
  T m_force[3], m_torque[3], m_pos[3];   // thrust/torque at throttle=1 (max_rpm^2 * coef * unit)
  T m_wrench_t[3];                       // m_pos x m_force + m_torque (motor torque about the COM at throttle=1)
  T mixer[FW_NUM_ACTUATORS][4];
  T inv_mass, gravity;
  T I[9], Iinv[9];
  T coll[FW_MAX_COLLISION_PTS][3];
  T dt, inv_physics_hz, physics_hz_T;
  T dome, reach, min_height, spawn_hi;
  T start_pos[3], start_quat[4], start_vel[3];
  T wind_base[3], wind_amp[3], wind_phase, gust_omega, wind_force_coef;
  T gust_sd, gust_cd;                     // sin / cos of the gust phase advance per physics tick
  double wind_base_range[3][2], wind_amp_range[3][2];
  T warm[19];                             // cached post-warm-up rigid(13)+act(6) state
  T warm_obs[24], warm_R[9];              // ... its observation's attitude block (zero action) and the rotation its target deltas use
  int32_t warm_valid, warm_ticks;
  int32_t n_coll, gyroscopic;
  int32_t task, angle_repr, att_dim, obs_dim, ctx;
  int32_t num_targets, sparse, auto_reset;
  int32_t max_steps, step_ratio, ticks_per_aviary, warmup_aviary_steps;
  int32_t wind_mode, wind_randomize, wind_randomize_phase, wind_coupling;
  int32_t has_noise;
  uint32_t seed_lo, seed_hi;
  int64_t env_offset;
};

template <typename T>
struct DevState {
  T* r;          // [RF_COUNT][npad]
  int32_t* i;    // [IF_COUNT][npad]
  int32_t n, npad;
  // "shadow" = the pre-simulated start of each env's NEXT episode (see shadow_* in fwsim.hip)
  T* rs;                     // [RF_COUNT][npad]  same layout as r; written ONLY by shadow workers
  int32_t* is;               // [npad]            physics ticks of the shadow state
  T* sobs;                   // [npad][obs_dim]   the observation the finished shadow's episode starts with (zero action)
  unsigned long long* sreq;  // [npad]  live -> worker:  (episode wanted << 32) | launch index of the request
  unsigned long long* sdone; // [npad]  worker -> live:  (episode built << 32) | (launch index & 0xFFFFFF) << 8 | progress
  uint32_t epoch;            // launch index of this fw_step: set IN the kernel from `lctr` (launch_index below)
  int32_t shadow_on;
  int32_t mbox_off;          // camera tasks, two-wave step workgroups: byte offset in dynamic LDS of the capture mailbox (fwsim_objlock.hpp)
  int32_t stash_off;         // byte offset in dynamic LDS of the step waves' output stash ([envs per wave][4] doubles, behind the tile / camera map)
  uint32_t* lctr;            // [fw_step grid]  per-workgroup launch counters (launch_index below)
  unsigned long long* stats;  // [FW_CTR_DIM]  hand-off counters (fw_get_counters), bumped on resets only
#ifdef FW_PROFILE
  long long* prof;           // dev-only per-wave cycle accounting (fwsim.hip)
#endif
};

// State layout: tiles of TILE = envs-per-wave consecutive envs; inside a tile field f of env e sits at f * TILE + e % TILE
// (AoSoA).  One wave works on exactly one tile, so inside a kernel every field is `base + f * TILE * sizeof(T)` from ONE
// per-lane address with compile-time offsets -- the plain SoA form (stride = the runtime env count) needed a separate
// 64-bit address per field, ~100 of them live at once, and hipcc spilled them to scratch (each reload a memory round trip).
__host__ __device__ inline size_t tile_index(int tile, int fields, int f, int env) {
  return (size_t)(env / tile) * fields * tile + (size_t)f * tile + (size_t)(env % tile);
}
// View of tile `blk` in which the usual `field * npad + env` indexing (npad := TILE, env = GLOBAL env id) lands in the tile.
template <typename T, int TILE>
__device__ __forceinline__ DevState<T> tile_view(const DevState<T>& G, int blk) {
  DevState<T> D = G;
  const size_t t = (size_t)blk * TILE;
  D.r = G.r + t * (RF_COUNT - 1);
  D.i = G.i + t * (IF_COUNT - 1);
  if (G.rs) D.rs = G.rs + t * (RF_COUNT - 1);
  D.npad = TILE;
  return D;
}

constexpr double kPi = 3.14159265358979323846;

// ------------------------------------------------------------------------
// Launch index kept in DEVICE memory.  The shadow / scenario hand-off compares launch indices; a host counter passed
// by value is frozen into a captured hipGraph node and repeats on every replay.  Instead every workgroup of the
// fw_step grid owns one private word `lctr[blockIdx.x]`: it reads it when it starts and stores it + 1 when it is
// finished.  The grid of a handle never changes and every launch runs every workgroup exactly once, so all the words
// advance in lockstep -- the step workgroup of a tile and the worker workgroup of the same tile (the only two parties
// that ever compare tags) always agree on the index, without atomics or any cross-workgroup traffic, eager launches
// and graph replays number their launches identically, and which resets take the hand-off is reproducible.
// ------------------------------------------------------------------------
__device__ __forceinline__ uint32_t launch_index(const uint32_t* lctr) { return 1u + lctr[blockIdx.x]; }       // uniform address: scalar load
__device__ __forceinline__ void launch_done(uint32_t* lctr, uint32_t epoch) { if (threadIdx.x == 0) lctr[blockIdx.x] = epoch; }
__device__ __forceinline__ void stat_add(unsigned long long* stats, int which) {
  (void)__hip_atomic_fetch_add(stats + which, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------
// math building blocks
// ------------------------------------------------------------------------
// fp64: one wave per SIMD is bound by dependent-issue latency (~16 cycles per dependent
// fp64 op, tools/microbench_math.hip), so what matters is the DEPTH of each function:
// v_rcp/v_rsq seeds + Newton steps instead of the IEEE division / sqrt sequences,
// bounded-range Cody-Waite sincos and a single-division atan2, all with Estrin-scheme
// polynomials (log depth instead of Horner's linear chain).  Each is accurate to ~1 ulp
// (prototyped against numpy: <= 4.5e-16 absolute).
template <typename T> struct M;
template <> struct M<double> {
  static __device__ __forceinline__ double rcp_(double d) {
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    x = fma(fma(-d, x, 1.0), x, x);
    return x;
  }
  static __device__ __forceinline__ double div_(double n, double d) {
    double x = rcp_(d);
    double q = n * x;
    return fma(fma(-d, q, n), x, q);
  }
  static __device__ __forceinline__ double sqrt_(double a) {
    double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    double e = fma(-g, g, a);
    g = fma(e, h, g);
    return (a == 0.0) ? 0.0 : g;
  }
  // sin/cos kernels on |r| <= pi/4 (fdlibm coefficients), Estrin form
  static __device__ __forceinline__ void sincos_kernel_(double r, double* sn, double* cs) {
    double z = r * r, z2 = z * z, z4 = z2 * z2;
    double s01 = fma(z, 8.33333333332248946124e-03, -1.66666666666666324348e-01);
    double s23 = fma(z, 2.75573137070700676789e-06, -1.98412698298579493134e-04);
    double s45 = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    double ps = fma(z4, s45, fma(z2, s23, s01));
    *sn = fma(z * r, ps, r);
    double c01 = fma(z, -1.38888888888741095749e-03, 4.16666666666666019037e-02);
    double c23 = fma(z, -2.75573143513906633035e-07, 2.48015872894767294178e-05);
    double c45 = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    double pc = fma(z4, c45, fma(z2, c23, c01));
    *cs = fma(z2, pc, fma(z, -0.5, 1.0));
  }
  static __device__ __forceinline__ void sincos_(double x, double* s, double* c) {
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
    double k = ::rint(x * INV_PIO2);
    double r = fma(-k, PIO2_HI, x);
    r = fma(-k, PIO2_LO, r);
    double sn, cs;
    sincos_kernel_(r, &sn, &cs);
    int q = (int)k;
    double s0 = (q & 1) ? cs : sn, c0 = (q & 1) ? sn : cs;
    *s = (q & 2) ? -s0 : s0;
    *c = ((q + 1) & 2) ? -c0 : c0;
  }
  static __device__ __forceinline__ double sin_(double x) { double s, c; sincos_(x, &s, &c); return s; }
  // atan2 with ONE division: t = min/max in [0,1] is split at c in {0,1/4,1/2,3/4,1};
  // atan(t) = atan(c) + atan((u - c v)/(v + c u)), |z| <= 1/8, 9-term odd series.
  static __device__ __forceinline__ double atan2_(double y, double x) {
    double ax = ::fabs(x), ay = ::fabs(y);
    double u = ::fmin(ax, ay), v = ::fmax(ax, ay);
    double c = 0.0, tc = 0.0;
    c = (u > 0.125 * v) ? 0.25 : c;  tc = (u > 0.125 * v) ? 2.44978663126864143e-01 : tc;
    c = (u > 0.375 * v) ? 0.50 : c;  tc = (u > 0.375 * v) ? 4.63647609000806094e-01 : tc;
    c = (u > 0.625 * v) ? 0.75 : c;  tc = (u > 0.625 * v) ? 6.43501108793284371e-01 : tc;
    c = (u > 0.875 * v) ? 1.00 : c;  tc = (u > 0.875 * v) ? 7.85398163397448279e-01 : tc;
    double num = fma(-c, v, u), den = fma(c, u, v);
    double z = (den == 0.0) ? 0.0 : div_(num, den);
    double w = z * z, w2 = w * w, w4 = w2 * w2;
    double p01 = fma(w, -1.0 / 3.0, 1.0), p23 = fma(w, -1.0 / 7.0, 1.0 / 5.0);
    double p45 = fma(w, -1.0 / 11.0, 1.0 / 9.0), p67 = fma(w, -1.0 / 15.0, 1.0 / 13.0);
    double lo = fma(w2, p23, p01), hi = fma(w2, p67, p45);
    double p = fma(w4, fma(w4, 1.0 / 17.0, hi), lo);
    double r = fma(z, p, tc);
    r = (ay > ax) ? (0.5 * kPi - r) : r;
    r = (x < 0.0) ? (kPi - r) : r;
    return (y < 0.0) ? -r : r;
  }
  static __device__ __forceinline__ double asin_(double s) { return atan2_(s, sqrt_((1.0 - s) * (1.0 + s))); }
  static __device__ __forceinline__ double fabs_(double x) { return ::fabs(x); }
  static __device__ __forceinline__ double fmax_(double a, double b) { return ::fmax(a, b); }
  static __device__ __forceinline__ double log_(double x) { return ::log(x); }
};
template <> struct M<float> {
  static __device__ __forceinline__ float rcp_(float d) { return 1.0f / d; }
  static __device__ __forceinline__ float div_(float n, float d) { return n / d; }
  static __device__ __forceinline__ float sqrt_(float x) { return ::sqrtf(x); }
  static __device__ __forceinline__ void sincos_(float x, float* s, float* c) { ::sincosf(x, s, c); }
  static __device__ __forceinline__ float sin_(float x) { return ::sinf(x); }
  static __device__ __forceinline__ float atan2_(float y, float x) { return ::atan2f(y, x); }
  static __device__ __forceinline__ float asin_(float x) { return ::asinf(x); }
  static __device__ __forceinline__ float fabs_(float x) { return ::fabsf(x); }
  static __device__ __forceinline__ float fmax_(float a, float b) { return ::fmaxf(a, b); }
  static __device__ __forceinline__ float log_(float x) { return ::logf(x); }
};

// ------------------------------------------------------------------------
// cross-lane helpers for the 8-lanes-per-env mapping (DPP: no LDS round trip)
// ------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ double dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(dpp_i32<CTRL>(hi), dpp_i32<CTRL>(lo));
}
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
  return __int_as_float(dpp_i32<CTRL>(__float_as_int(v)));
}
template <int CTRL> __device__ __forceinline__ int dpp(int v) { return dpp_i32<CTRL>(v); }
// sum over the 8 lanes of a group; every lane ends with the bit-identical total
template <int G, typename T> __device__ __forceinline__ T group_sum(T v) {
  if (G == 8) {
    v += dpp<0xB1>(v);    // quad_perm [1,0,3,2]  (lane ^ 1)
    v += dpp<0x4E>(v);    // quad_perm [2,3,0,1]  (lane ^ 2)
    v += dpp<0x141>(v);   // row_half_mirror      (the other quad of the 8)
  }
  return v;
}
// min / bitwise-or over the 8 lanes of a group (same DPP pattern; every lane gets the result)
template <int G, typename T> __device__ __forceinline__ T group_min(T v) {
  if (G == 8) {
    T o = dpp<0xB1>(v); v = o < v ? o : v;
    o = dpp<0x4E>(v); v = o < v ? o : v;
    o = dpp<0x141>(v); v = o < v ? o : v;
  }
  return v;
}
template <int G> __device__ __forceinline__ uint32_t group_or(uint32_t v) {
  if (G == 8) {
    v |= (uint32_t)dpp_i32<0xB1>((int)v);
    v |= (uint32_t)dpp_i32<0x4E>((int)v);
    v |= (uint32_t)dpp_i32<0x141>((int)v);
  }
  return v;
}
// does any lane of my group have `pred` set?
template <int G> __device__ __forceinline__ bool group_any(bool pred) {
  if (G == 1) return pred;
  unsigned long long m = __ballot(pred);
  return ((m >> (threadIdx.x & ~(G - 1))) & ((1ull << G) - 1)) != 0ull;
}

// ---- Philox4x32-10, identical counter/key convention to the spec in DESIGN.md ----
enum { STREAM_SCENARIO = 0, STREAM_NOISE = 1 };
enum { J_WIND_BASE = 0, J_WIND_AMP = 3, J_WIND_PHASE = 6, J_THETA = 8, J_PHI = 16, J_DIST = 24,
       J_DUCK_X = 32, J_DUCK_Y = 33, J_DUCK_YAW = 34, J_OBST = 40 };

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
template <typename T>
__device__ __forceinline__ uint64_t rng_u64(const Params<T>& P, uint32_t genv, uint32_t ep, uint32_t stream, uint32_t j) {
  uint32_t o[4];
  philox4x32_10(j >> 1, ep, genv, stream, P.seed_lo, P.seed_hi, o);
  uint32_t lo = (j & 1) ? o[2] : o[0], hi = (j & 1) ? o[3] : o[1];
  return ((uint64_t)hi << 32) | lo;
}
template <typename T>
__device__ __forceinline__ double rng_uniform(const Params<T>& P, uint32_t genv, uint32_t ep, uint32_t j, double lo, double hi) {
  double u = (double)(rng_u64(P, genv, ep, STREAM_SCENARIO, j) >> 11) * (1.0 / 9007199254740992.0);
  return lo + (hi - lo) * u;
}
// two N(0,1) for Aviary step `astep`; both 64-bit words come from ONE Philox block
template <typename T>
__device__ __forceinline__ void rng_normal2(const Params<T>& P, uint32_t genv, uint32_t ep, uint32_t astep, T& z0, T& z1) {
  uint32_t o[4];
  philox4x32_10(astep, ep, genv, STREAM_NOISE, P.seed_lo, P.seed_hi, o);
  uint64_t a = ((uint64_t)o[1] << 32) | o[0], b = ((uint64_t)o[3] << 32) | o[2];
  double u1 = (double)((a >> 11) + 1) * (1.0 / 9007199254740992.0);
  double u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
  T r = M<T>::sqrt_((T)-2.0 * M<T>::log_((T)u1));
  T s, c;
  M<T>::sincos_((T)(2.0 * kPi * u2), &s, &c);
  z0 = r * c; z1 = r * s;
}

// ------------------------------------------------------------------------
// rigid state in registers
// ------------------------------------------------------------------------
template <typename T>
struct Rigid {
  T p[3], q[4], v[3], w[3];    // world-frame velocities, q = (x,y,z,w) body->world
  T act[FW_NUM_ACTUATORS];
};

// 2/|q|^2: q is a unit quaternion up to rounding, so a 4-term series in eps = |q|^2 - 1
// replaces the reciprocal; the exact path is kept for badly normalised input (set_state).
template <typename T>
__device__ __forceinline__ T two_over_norm2(T d) {
  T e = d - (T)1;
  T ser = (T)2 * ((T)1 + e * ((T)-1 + e * ((T)1 + e * ((T)-1 + e))));
  return (M<T>::fabs_(e) < (T)1e-4) ? ser : (T)2 * M<T>::rcp_(d);
}

template <typename T>
__device__ __forceinline__ void rot_from_quat(const T q[4], T m[9]) {
  T x = q[0], y = q[1], z = q[2], w = q[3];
  T d = x * x + y * y + z * z + w * w;
  T s = two_over_norm2<T>(d);
  T xs = x * s, ys = y * s, zs = z * s;
  T wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  m[0] = (T)1 - (yy + zz); m[1] = xy - wz;           m[2] = xz + wy;
  m[3] = xy + wz;           m[4] = (T)1 - (xx + zz); m[5] = yz - wx;
  m[6] = xz - wy;           m[7] = yz + wx;           m[8] = (T)1 - (xx + yy);
}
template <typename T> __device__ __forceinline__ void mv(const T m[9], const T v[3], T o[3]) {
  o[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  o[1] = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  o[2] = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
}
template <typename T> __device__ __forceinline__ void mtv(const T m[9], const T v[3], T o[3]) {
  o[0] = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  o[1] = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  o[2] = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
}
template <typename T> __device__ __forceinline__ void cross(const T a[3], const T b[3], T o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// np.interp(x,[x0,x1],[y0,y1]) with end clamping
template <typename T> __device__ __forceinline__ T interp2(T x, T x0, T x1, T y0, T y1) {
  T t = M<T>::div_(x - x0, x1 - x0);
  T y = y0 + (y1 - y0) * t;
  y = (x <= x0) ? y0 : y;
  y = (x >= x1) ? y1 : y;
  return y;
}

// One lifting surface: branch-free Khan&Nahon flat-plate model (pre- and post-stall
// evaluated on the same sincos, selected per lane => no wave divergence).
// cos(alpha), sin(alpha) are never formed: V*cos = v_f, V*sin = -v_l.
// Returns the surface's force f and its torque about the COM (r x f + pitching moment).
// SC = SurfC<T> (constants in registers / scalar loads) or volatile SurfC<T> (the 256-register build of the 8-lane mapping:
// this lane's surface sits in LDS and every use is a fresh ds_read -- 54 registers the allocator otherwise spilled to scratch).
template <typename T, typename SC>
__device__ __forceinline__ void surface_wrench(SC& S, T act, const T v_b[3], const T w_b[3],
                                               const T wind_b[3], T f[3], T tq[3]) {
  T wxr[3];
  const T spos[3] = { S.pos[0], S.pos[1], S.pos[2] };
  cross(w_b, spos, wxr);
  T vl0 = v_b[0] + wxr[0] - wind_b[0], vl1 = v_b[1] + wxr[1] - wind_b[1], vl2 = v_b[2] + wxr[2] - wind_b[2];
  T v_l = vl0 * S.lift[0] + vl1 * S.lift[1] + vl2 * S.lift[2];
  T v_f = vl0 * S.fwd[0] + vl1 * S.fwd[1] + vl2 * S.fwd[2];
  T V2 = v_f * v_f + v_l * v_l;
  T V = M<T>::sqrt_(V2);
  T alpha = M<T>::atan2_(-v_l, v_f);

  T defl = S.defl_scale * act;
  T dCl = S.k_dCl * act;
  T dClmax = S.ftc * dCl;
  T a0 = S.a0b - dCl * S.inv_Cl3;
  T asP = a0 + (S.ClmaxPb + dClmax) * S.inv_Cl3;
  T asN = a0 + (S.ClmaxNb + dClmax) * S.inv_Cl3;
  bool nostall = (asN < alpha) && (alpha < asP);

  // induced angle: linear pre-stall, np.interp'd to zero at +-pi/2 post-stall
  T Cl_lin = S.Cl3 * (alpha - a0);
  T ai_lin = Cl_lin * S.inv_piAR;
  const T hpi = (T)(0.5 * kPi);
  T ai_stP = S.Cl3 * (asP - a0) * S.inv_piAR;
  T ai_stN = S.Cl3 * (asN - a0) * S.inv_piAR;
  // np.interp on the active side only (one division): positive stall [asP, pi/2] -> [ai_stP, 0],
  // negative stall [-pi/2, asN] -> [0, ai_stN]
  const bool pos = alpha > (T)0;
  T ix0 = pos ? asP : -hpi, ix1 = pos ? hpi : asN;
  T iy0 = pos ? ai_stP : (T)0, iy1 = pos ? (T)0 : ai_stN;
  T ai_st = interp2<T>(alpha, ix0, ix1, iy0, iy1);
  T ai = nostall ? ai_lin : ai_st;
  T ae = alpha - a0 - ai;
  T sn, cs;
  M<T>::sincos_(ae, &sn, &cs);

  // one reciprocal serves both branches: 1/cos(ae) pre-stall, 1/(0.56+0.44|sin ae|) post-stall
  T asn = M<T>::fabs_(sn);
  T inv = M<T>::rcp_(nostall ? cs : ((T)0.56 + (T)0.44 * asn));
  // pre-stall
  T CT_a = S.Cd0 * cs;
  T CN_a = (Cl_lin + CT_a * sn) * inv;
  T CM_a = -CN_a * ((T)0.25 - (T)0.175 * ((T)1 - ((T)2 * ae) * (T)(1.0 / kPi)));
  // post-stall
  T Cd90 = ((T)-4.26e-2 * (defl * defl)) + ((T)2.1e-1 * defl) + (T)1.98;
  T CN_b = Cd90 * sn * (inv - S.k_exp);
  T CT_b = (T)0.5 * S.Cd0 * cs;
  T CM_b = -CN_b * ((T)0.25 - (T)0.175 * ((T)1 - ((T)2 * M<T>::fabs_(ae)) * (T)(1.0 / kPi)));

  T CN = nostall ? CN_a : CN_b;
  T CT = nostall ? CT_a : CT_b;
  T CM = nostall ? CM_a : CM_b;
  T Cl = nostall ? Cl_lin : (CN * cs - CT * sn);
  T Cd = CN * sn + CT * cs;

  // force_normal = Q*area*(Cl cos a + Cd sin a) = hra * V * (Cl v_f - Cd v_l), etc.
  T hV = S.hra * V;
  T Fn = hV * (Cl * v_f - Cd * v_l);
  T Fp = hV * (-Cl * v_l - Cd * v_f);
  T Mq = S.hra * V2 * CM * S.chord;
  f[0] = S.lift[0] * Fn + S.fwd[0] * Fp; f[1] = S.lift[1] * Fn + S.fwd[1] * Fp; f[2] = S.lift[2] * Fn + S.fwd[2] * Fp;
  T rxf[3];
  cross(spos, f, rxf);
  tq[0] = rxf[0] + Mq * S.tq[0]; tq[1] = rxf[1] + Mq * S.tq[1]; tq[2] = rxf[2] + Mq * S.tq[2];
}

// wind vector at time t (envs/fixedwing_envs/fixedwing_base_env.py:145-171)
template <typename T>
__device__ __forceinline__ void wind_at(const Params<T>& P, const T wb[3], const T wa[3], T phase, int32_t tick, T w[3]) {
  if (P.wind_mode == FW_WIND_OFF) { w[0] = w[1] = w[2] = (T)0; return; }
  if (P.wind_mode == FW_WIND_CONSTANT) { w[0] = wb[0]; w[1] = wb[1]; w[2] = wb[2]; return; }
  T t = (T)tick * P.inv_physics_hz;
  T s = M<T>::sin_(P.gust_omega * t + phase);
  w[0] = wb[0] + wa[0] * s; w[1] = wb[1] + wa[1] * s; w[2] = wb[2] + wa[2] * s;
}

// Gust phase carried as (sin, cos): one sincos when the clock is (re)set -- launch start, reset, shadow swap-in -- then a
// rotation by the constant per-tick advance (4 FMAs instead of a 45-instruction sin per tick; <= 8 rotations from an exact
// start, i.e. ~1e-15 from the literal sin(2 pi f t + phi) of fixedwing_base_env.py:167-171).
template <typename T>
__device__ __forceinline__ void gust_init(const Params<T>& P, T phase, int32_t tick, T g[2]) {
  g[0] = (T)0; g[1] = (T)1;
  if (P.wind_mode == FW_WIND_GUST_SINE) M<T>::sincos_(P.gust_omega * ((T)tick * P.inv_physics_hz) + phase, &g[0], &g[1]);
}
template <typename T>
__device__ __forceinline__ void gust_advance(const Params<T>& P, T g[2]) {
  const T s = g[0] * P.gust_cd + g[1] * P.gust_sd, c = g[1] * P.gust_cd - g[0] * P.gust_sd;
  g[0] = s; g[1] = c;
}
template <typename T>
__device__ __forceinline__ void wind_from_phase(const Params<T>& P, const T wb[3], const T wa[3], const T g[2], T w[3]) {
  if (P.wind_mode == FW_WIND_OFF) { w[0] = w[1] = w[2] = (T)0; return; }
  if (P.wind_mode == FW_WIND_CONSTANT) { w[0] = wb[0]; w[1] = wb[1]; w[2] = wb[2]; return; }
  w[0] = wb[0] + wa[0] * g[0]; w[1] = wb[1] + wa[1] * g[0]; w[2] = wb[2] + wa[2] * g[0];
}

// ------------------------------------------------------------------------
// Constants of the tick loop, gathered once per launch.
//   G = 1: they stay wave-uniform (SGPRs / scalar cache).
//   G = 8: loaded through a per-lane ("opaque") pointer so that they live in VGPRs for
//          the whole launch.  rocprofv3 showed the uniform version spending ~30 % of the
//          wave time in s_waitcnt behind in-loop scalar loads (98 SMEM per wave-step,
//          SGPR file exhausted by fp64 pairs); with 512 VGPRs available and one wave
//          per SIMD the vector file is the right home.
// ------------------------------------------------------------------------
template <typename T>
struct TickC {
  T dt_tau[FW_NUM_SURFACES];
  T motor_dt_tau, noise_ratio;
  T mF[3], mT[3];            // motor force / torque about the COM at throttle = 1
  T inv_mass, gravity, dt, hdt;
  T I[9], Iinv[9];
  T wind_force_coef;
  T cpt[3];                  // G = 8: this lane's collision point
  T cvalid;                  // 1 if this lane's collision point exists
};

__device__ __forceinline__ int opaque_zero() {
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}

// RESIDENT = true (wind-free waypoints kernel on the 8-lane mapping): the shared constants are loaded through a per-lane
// address and stay in VGPRs.  The wind / camera kernels keep far more per-env state in registers; there the shared
// constants stay wave-uniform (scalar loads), which takes ~80 VGPRs off the peak (their spills went 84 -> 38) at no cost
// in time (measured: waypoints + wind 27.2 us either way, ObjLock 41.6 -> 39.2 us).  This lane's surface is per-lane data
// in both cases.
template <typename T, int G, bool RESIDENT = (G == 8)>
__device__ __forceinline__ void load_tick_constants(const Params<T>* Pp, TickC<T>& C, SurfC<T>& mine, T& wmask) {
  const int sub = (G == 1) ? 0 : (int)(threadIdx.x & (G - 1));
  const Params<T>* Q = (G == 1 || !RESIDENT) ? Pp : Pp + opaque_zero();     // per-lane address => vector loads => VGPR residency
#pragma unroll
  for (int s = 0; s < FW_NUM_SURFACES; ++s) C.dt_tau[s] = Q->s[s].dt_tau;
  C.motor_dt_tau = Q->motor_dt_tau; C.noise_ratio = Q->noise_ratio;
#pragma unroll
  for (int k = 0; k < 3; ++k) { C.mF[k] = Q->m_force[k]; C.mT[k] = Q->m_wrench_t[k]; }
  C.inv_mass = Q->inv_mass; C.gravity = Q->gravity; C.dt = Q->dt; C.hdt = (T)0.5 * Q->dt;
#pragma unroll
  for (int k = 0; k < 9; ++k) { C.I[k] = Q->I[k]; C.Iinv[k] = Q->Iinv[k]; }
  C.wind_force_coef = Q->wind_force_coef;
  const int ci = (G == 1) ? 0 : sub;
#pragma unroll
  for (int k = 0; k < 3; ++k) C.cpt[k] = Q->coll[ci][k];
  C.cvalid = (ci < Q->n_coll) ? (T)1 : (T)0;
  mine = Q->s[(G == 1) ? 0 : min(sub, FW_NUM_SURFACES - 1)];
  wmask = (sub < FW_NUM_SURFACES) ? (T)1 : (T)0;
}

// rotation matrix of a UNIT quaternion (s = 2 exactly); the tick keeps q normalised
template <typename T>
__device__ __forceinline__ void rot_from_unit_quat(const T q[4], T m[9]) {
  T x = q[0], y = q[1], z = q[2], w = q[3];
  T xs = x + x, ys = y + y, zs = z + z;
  T wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  m[0] = (T)1 - (yy + zz); m[1] = xy - wz;           m[2] = xz + wy;
  m[3] = xy + wz;           m[4] = (T)1 - (xx + zz); m[5] = yz - wx;
  m[6] = xz - wy;           m[7] = yz + wx;           m[8] = (T)1 - (xx + yy);
}
// bring an externally supplied quaternion (set_state) onto the unit sphere
template <typename T>
__device__ __forceinline__ void normalize_quat(T q[4]) {
  T d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (M<T>::fabs_(d - (T)1) > (T)1e-12) {
    T inv = M<T>::rcp_(M<T>::sqrt_(d));
    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
  }
}

// G = 8: the control-surface actuator lags are lane-local during the ticks -- lane j carries the state of surface
// min(j, 4) and its command (`LaneAct`), instead of every lane updating all five and selecting its own afterwards.
// S.act[0..4] is only current at the sync points (lane_act_gather); the motor (S.act[5]) stays replicated.
template <typename T> struct LaneAct { T a, cmd; };
// (operands by value: selecting between loads of one array makes the compiler index the array dynamically -- in scratch)
template <typename T>
__device__ __forceinline__ T lane_pick5(T v0, T v1, T v2, T v3, T v4) {
  const int sub = threadIdx.x & 7;
  T a = v0;
  a = (sub == 1) ? v1 : a; a = (sub == 2) ? v2 : a; a = (sub == 3) ? v3 : a; a = (sub >= 4) ? v4 : a;
  return a;
}
template <typename T> __device__ __forceinline__ void lane_act_scatter(const Rigid<T>& S, LaneAct<T>& LA) {
  LA.a = lane_pick5<T>(S.act[0], S.act[1], S.act[2], S.act[3], S.act[4]);
}
template <typename T>
__device__ __forceinline__ void lane_act_gather(Rigid<T>& S, const LaneAct<T>& LA) {
  const int gbase = threadIdx.x & ~7;
#pragma unroll
  for (int k = 0; k < FW_NUM_SURFACES; ++k) S.act[k] = __shfl(LA.a, gbase | k, kWave);
}

// ---- one 1/240 s physics tick, in three stages ----
// (1) actuator lags
template <typename T>
__device__ __forceinline__ void tick_actuators(const TickC<T>& C, Rigid<T>& S, const T cmd[FW_NUM_ACTUATORS], T noise_z) {
#pragma unroll
  for (int s = 0; s < FW_NUM_SURFACES; ++s) S.act[s] += C.dt_tau[s] * (cmd[s] - S.act[s]);
  T thr = S.act[FW_NUM_SURFACES];
  thr += C.motor_dt_tau * (cmd[FW_NUM_SURFACES] - thr);
  thr += noise_z * thr * C.noise_ratio;
  S.act[FW_NUM_SURFACES] = thr;
}

// Exponential-map quaternion update (Bullet): q <- normalize(dq(w dt) (x) q).
// sin(th)/th and cos(th), th = |w| dt / 2 <= pi/8, are even polynomials in th^2, so no
// sqrt / sincos / division sits on the critical path; the angular-motion clamp
// (|w| dt > pi/4, i.e. > 188 rad/s) takes the literal formula on a rare branch.
template <typename T>
__device__ __forceinline__ void quat_integrate(const TickC<T>& C, Rigid<T>& S) {
  const T dt = C.dt, hdt = C.hdt;
  T w2 = S.w[0] * S.w[0] + S.w[1] * S.w[1] + S.w[2] * S.w[2];
  T x = w2 * hdt * hdt;                         // th^2
  T x2 = x * x, x4 = x2 * x2;
  T s01 = (T)1 + x * (T)(-1.0 / 6.0), s23 = (T)(1.0 / 120.0) + x * (T)(-1.0 / 5040.0);
  T s45 = (T)(1.0 / 362880.0) + x * (T)(-1.0 / 39916800.0), s67 = (T)(1.0 / 6227020800.0) + x * (T)(-1.0 / 1307674368000.0);
  T sinc = (s01 + s23 * x2) + (s45 + s67 * x2) * x4;
  T c01 = (T)1 + x * (T)-0.5, c23 = (T)(1.0 / 24.0) + x * (T)(-1.0 / 720.0);
  T c45 = (T)(1.0 / 40320.0) + x * (T)(-1.0 / 3628800.0), c67 = (T)(1.0 / 479001600.0) + x * (T)(-1.0 / 87178291200.0);
  T ch = (c01 + c23 * x2) + (c45 + c67 * x2 + (T)(1.0 / 20922789888000.0) * x4) * x4;
  T k = sinc * hdt;                              // sin(th)/|w|
  const T lim = (T)(0.25 * kPi);
  if (w2 * dt * dt > lim * lim) {                // ANGULAR_MOTION_THRESHOLD clamp (rare)
    T ang = M<T>::div_(lim, dt);
    T sh;
    M<T>::sincos_((T)0.5 * ang * dt, &sh, &ch);
    k = M<T>::div_(sh, ang);
  }
  T ax = S.w[0] * k, ay = S.w[1] * k, az = S.w[2] * k;
  T qx = S.q[0], qy = S.q[1], qz = S.q[2], qw = S.q[3];
  T nx = ch * qx + ax * qw + ay * qz - az * qy;
  T ny = ch * qy + ay * qw + az * qx - ax * qz;
  T nz = ch * qz + az * qw + ax * qy - ay * qx;
  T nw = ch * qw - ax * qx - ay * qy - az * qz;
  T n2 = (nx * nx + ny * ny) + (nz * nz + nw * nw);
  T e = n2 - (T)1;                               // 1/sqrt(1+e) series; exact path for un-normalised input
  T e2 = e * e;
  T inv = ((T)1 + e * (T)-0.5) + e2 * (((T)0.375 + e * (T)-0.3125) + e2 * (T)0.2734375);
  if (M<T>::fabs_(e) >= (T)1e-4) inv = M<T>::rcp_(M<T>::sqrt_(n2));
  S.q[0] = nx * inv; S.q[1] = ny * inv; S.q[2] = nz * inv; S.q[3] = nw * inv;
}

// Full tick.  R = R(S.q) on entry and on exit (carried across ticks: it is needed for the
// contact test of this tick and the body-frame velocities of the next one).
// G = 1: rolled loop over the 5 surfaces (constants by scalar loads at a wave-uniform index
//        -- unrolling makes hipcc hoist ~100 constants into SGPRs and spill).
// G = 8: `mine` holds this lane's surface constants in VGPRs, `wmask` zeroes lanes 5-7.
template <typename T, bool WIND, int G, typename SC>
__device__ __forceinline__ bool physics_tick(const Params<T>& P, const TickC<T>& C, Rigid<T>& S, T R[9],
                                             const T cmd[FW_NUM_ACTUATORS], T noise_z, const T wind[3],
                                             SC& mine, T wmask, LaneAct<T>& LA) {
  if (G == 8) {
    LA.a += mine.dt_tau * (LA.cmd - LA.a);                          // my surface
    T thr = S.act[FW_NUM_SURFACES];                                 // the motor, replicated
    thr += C.motor_dt_tau * (cmd[FW_NUM_SURFACES] - thr);
    thr += noise_z * thr * C.noise_ratio;
    S.act[FW_NUM_SURFACES] = thr;
  } else {
    tick_actuators<T>(C, S, cmd, noise_z);
  }
  T v_b[3], w_b[3], wind_b[3] = {(T)0, (T)0, (T)0};
  mtv(R, S.v, v_b);
  mtv(R, S.w, w_b);
  if (WIND && P.wind_coupling == FW_WIND_COUPLE_AIRSPEED) mtv(R, wind, wind_b);
  T F[3] = {(T)0, (T)0, (T)0}, Tq[3] = {(T)0, (T)0, (T)0};
  if (G == 8) {
    const T a_s = LA.a;
    T f[3], tq[3];
    surface_wrench<T, SC>(mine, a_s, v_b, w_b, wind_b, f, tq);
#pragma unroll
    for (int k = 0; k < 3; ++k) { F[k] = group_sum<8, T>(f[k] * wmask); Tq[k] = group_sum<8, T>(tq[k] * wmask); }
  } else {
#pragma unroll 1
    for (int s = 0; s < FW_NUM_SURFACES; ++s) {
      T a_s = (s == 0) ? S.act[0] : (s == 1) ? S.act[1] : (s == 2) ? S.act[2] : (s == 3) ? S.act[3] : S.act[4];
      T f[3], tq[3];
      surface_wrench<T, const SurfC<T>>(P.s[s], a_s, v_b, w_b, wind_b, f, tq);
#pragma unroll
      for (int k = 0; k < 3; ++k) { F[k] += f[k]; Tq[k] += tq[k]; }
    }
  }
  // motor
  {
    T thr = S.act[FW_NUM_SURFACES];
    T t2 = thr * thr;
#pragma unroll
    for (int k = 0; k < 3; ++k) { F[k] += t2 * C.mF[k]; Tq[k] += t2 * C.mT[k]; }
  }
  const T dt = C.dt;
  T Fw[3];
  mv(R, F, Fw);
  if (WIND && P.wind_coupling == FW_WIND_COUPLE_FORCE) {
    Fw[0] += C.wind_force_coef * wind[0]; Fw[1] += C.wind_force_coef * wind[1]; Fw[2] += C.wind_force_coef * wind[2];
  }
  T acc[3] = { Fw[0] * C.inv_mass, Fw[1] * C.inv_mass, Fw[2] * C.inv_mass - C.gravity };
  T Iw[3], rhs[3] = { Tq[0], Tq[1], Tq[2] }, al_b[3], al_w[3];
  mv(C.I, w_b, Iw);
  if (P.gyroscopic) {
    T g[3];
    cross(w_b, Iw, g);
    rhs[0] -= g[0]; rhs[1] -= g[1]; rhs[2] -= g[2];
  }
  mv(C.Iinv, rhs, al_b);
  mv(R, al_b, al_w);
#pragma unroll
  for (int k = 0; k < 3; ++k) { S.v[k] += acc[k] * dt; S.w[k] += al_w[k] * dt; }
#pragma unroll
  for (int k = 0; k < 3; ++k) S.p[k] += S.v[k] * dt;
  quat_integrate<T>(C, S);
  rot_from_unit_quat<T>(S.q, R);
  // contacts: third row of R(q_new) dotted with the body-fixed points
  bool contact = false;
  if (G == 8) {                // lane `sub` tests point `sub`; OR over the group
    T zc = S.p[2] + R[6] * C.cpt[0] + R[7] * C.cpt[1] + R[8] * C.cpt[2];
    contact = group_any<8>(C.cvalid != (T)0 && zc <= (T)0);
  } else {
    for (int i = 0; i < P.n_coll; ++i) {
      T zc = S.p[2] + R[6] * P.coll[i][0] + R[7] * P.coll[i][1] + R[8] * P.coll[i][2];
      contact |= (zc <= (T)0);
    }
  }
  return contact;
}

// Aviary.step() lives in fwsim_objlock.hpp (it also drives the ObjLock contacts / camera cadence).

// pybullet.getEulerFromQuaternion incl. the gimbal guard
template <typename T>
__device__ __forceinline__ bool euler_from_quat(const T q[4], T e[3]) {
  T x = q[0], y = q[1], z = q[2], w = q[3];
  T sarg = (T)-2 * (x * z - w * y);
  bool lock = (sarg <= (T)-0.99999) || (sarg >= (T)0.99999);
  if (lock) {
    bool neg = sarg <= (T)-0.99999;
    e[0] = (T)0;
    e[1] = neg ? (T)(-0.5 * kPi) : (T)(0.5 * kPi);
    e[2] = (T)2 * (neg ? M<T>::atan2_(x, -y) : M<T>::atan2_(-x, y));
  } else {
    T sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
    e[0] = M<T>::atan2_((T)2 * (y * z + w * x), squ - sqx - sqy + sqz);
    e[1] = M<T>::asin_(sarg);
    e[2] = M<T>::atan2_((T)2 * (x * y + w * z), squ + sqx - sqy - sqz);
  }
  return lock;
}
// The same, for a pass that all 8 lanes of an env's group run: lanes 0-2 evaluate one angle each (one atan2 in the
// instruction stream instead of three), the results are handed round by shuffle.  Same function, same arguments per
// angle => bit-identical to euler_from_quat.
template <typename T>
__device__ __forceinline__ bool euler_from_quat_lanes8(const T q[4], T e[3]) {
  const int lane = threadIdx.x, sub = lane & 7, gbase = lane & ~7;
  T x = q[0], y = q[1], z = q[2], w = q[3];
  T sarg = (T)-2 * (x * z - w * y);
  const bool lock = (sarg <= (T)-0.99999) || (sarg >= (T)0.99999);
  const bool neg = sarg <= (T)-0.99999;
  T sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
  T ya = (T)2 * (y * z + w * x), xa = squ - sqx - sqy + sqz;                         // roll
  if (sub == 1) { ya = sarg; xa = M<T>::sqrt_(((T)1 - sarg) * ((T)1 + sarg)); }     // pitch = asin(sarg)
  if (sub == 2) { ya = (T)2 * (x * y + w * z); xa = squ + sqx - sqy - sqz; }        // yaw
  if (lock && sub == 2) { ya = neg ? x : -x; xa = neg ? -y : y; }
  const T res = M<T>::atan2_(ya, xa);
#pragma unroll
  for (int k = 0; k < 3; ++k) e[k] = __shfl(res, gbase | k, kWave);
  if (lock) { e[0] = (T)0; e[1] = neg ? (T)(-0.5 * kPi) : (T)(0.5 * kPi); e[2] = (T)2 * e[2]; }
  return lock;
}
template <typename T>
__device__ __forceinline__ void quat_from_euler(const T e[3], T q[4]) {
  T sr, cr, sp, cp, sy, cy;
  M<T>::sincos_((T)0.5 * e[0], &sr, &cr);
  M<T>::sincos_((T)0.5 * e[1], &sp, &cp);
  M<T>::sincos_((T)0.5 * e[2], &sy, &cy);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}

// Observation writer.  W(k, value) stores element k of this env's row.
// Waypoints layout: attitude[att_dim] ++ ctx target deltas (zero padded).
//   The reference rebuilds the quaternion from the Euler angles before rotating
//   (fixedwing_base_env.py:288); away from the gimbal guard that round trip is
//   the identity on the rotation, so q itself is used and the round trip is
//   taken only on the (rare) guarded branch or when the quaternion is observed.
template <typename T, bool LANES8 = false, typename W>
__device__ __forceinline__ int write_obs_attitude(const Params<T>& P, const Rigid<T>& S, const T action[4], T R[9], W&& put) {
  rot_from_quat(S.q, R);
  T ang_vel[3], lin_vel[3], eul[3];
  mtv(R, S.w, ang_vel);
  mtv(R, S.v, lin_vel);
  bool lock = LANES8 ? euler_from_quat_lanes8(S.q, eul) : euler_from_quat(S.q, eul);
  T qrt[4] = { S.q[0], S.q[1], S.q[2], S.q[3] };
  if (lock || P.angle_repr == 1) {
    quat_from_euler(eul, qrt);
    rot_from_quat(qrt, R);
  }
  int o = 0;
  put(o++, ang_vel[0]); put(o++, ang_vel[1]); put(o++, ang_vel[2]);
  if (P.angle_repr == 0) { put(o++, eul[0]); put(o++, eul[1]); put(o++, eul[2]); }
  else { put(o++, qrt[0]); put(o++, qrt[1]); put(o++, qrt[2]); put(o++, qrt[3]); }
  put(o++, lin_vel[0]); put(o++, lin_vel[1]); put(o++, lin_vel[2]);
  put(o++, S.p[0]); put(o++, S.p[1]); put(o++, S.p[2]);
  put(o++, action[0]); put(o++, action[1]); put(o++, action[2]); put(o++, action[3]);
#pragma unroll
  for (int k = 0; k < FW_NUM_ACTUATORS; ++k) put(o++, S.act[k]);
  return o;
}
template <typename T, typename W>
__device__ __forceinline__ void write_obs(const Params<T>& P, const DevState<T>& D, int env, const Rigid<T>& S,
                                          const T action[4], int tgt_idx, W&& put) {
  T R[9];
  int o = write_obs_attitude<T, false>(P, S, action, R, put);
  for (int i = 0; i < P.ctx; ++i) {
    int t = tgt_idx + i;
    T d[3] = {(T)0, (T)0, (T)0}, b[3] = {(T)0, (T)0, (T)0};
    if (t < P.num_targets) {
      const T* tp = D.r + (size_t)(RF_TARGETS + 3 * t) * D.npad + env;
      d[0] = tp[0] - S.p[0]; d[1] = tp[D.npad] - S.p[1]; d[2] = tp[2 * (size_t)D.npad] - S.p[2];
      mtv(R, d, b);
    }
    put(o++, b[0]); put(o++, b[1]); put(o++, b[2]);
  }
}

// ---- SoA load/store of the register-resident part ----
template <typename T>
__device__ __forceinline__ void load_rigid(const DevState<T>& D, int env, Rigid<T>& S) {
  const T* b = D.r + env;
  const size_t n = D.npad;
#pragma unroll
  for (int k = 0; k < 3; ++k) { S.p[k] = b[(RF_POS + k) * n]; S.v[k] = b[(RF_VEL + k) * n]; S.w[k] = b[(RF_OMEGA + k) * n]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) S.q[k] = b[(RF_QUAT + k) * n];
#pragma unroll
  for (int k = 0; k < FW_NUM_ACTUATORS; ++k) S.act[k] = b[(RF_ACT + k) * n];
}
template <typename T>
__device__ __forceinline__ void store_rigid(const DevState<T>& D, int env, const Rigid<T>& S) {
  T* b = D.r + env;
  const size_t n = D.npad;
#pragma unroll
  for (int k = 0; k < 3; ++k) { b[(RF_POS + k) * n] = S.p[k]; b[(RF_VEL + k) * n] = S.v[k]; b[(RF_OMEGA + k) * n] = S.w[k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) b[(RF_QUAT + k) * n] = S.q[k];
#pragma unroll
  for (int k = 0; k < FW_NUM_ACTUATORS; ++k) b[(RF_ACT + k) * n] = S.act[k];
}

// Copy `count` consecutive SoA words of one env between two state arrays: the group's lanes
// take every G-th word and keep their loads in flight together (one memory round trip, not `count`).
template <typename T, int G>
__device__ __forceinline__ void copy_words(T* __restrict__ dst, const T* __restrict__ src, int base, int count, size_t n, int env) {
  const int sub = (G == 1) ? 0 : (int)(threadIdx.x & (G - 1));
  constexpr int B = 8;
#pragma unroll 1
  for (int k0 = sub; k0 < count; k0 += B * G) {
    T tmp[B];
#pragma unroll
    for (int m = 0; m < B; ++m) { const int k = k0 + m * G; if (k < count) tmp[m] = src[(size_t)(base + k) * n + env]; }
#pragma unroll
    for (int m = 0; m < B; ++m) { const int k = k0 + m * G; if (k < count) dst[(size_t)(base + k) * n + env] = tmp[m]; }
  }
}

// ------------------------------------------------------------------------
// reset of one env, split like the reference: begin_reset (+ scenario sampling)
// and end_reset; the warm-up Aviary steps in between are run by the caller's
// single tick loop (so the tick is inlined exactly once per kernel).
// ------------------------------------------------------------------------
// Returns the number of warm-up Aviary steps still to run (0 when the cached
// env-independent warm state could be copied).  G = 8: lane `sub` samples waypoint
// `sub` (num_targets <= 8 = group size); only the group leader stores wind.
// Scenario sampling of a reset (wind + waypoints).  Inside the step loop it is called out of line
// (sample_scenario below): the double-precision sincos / Philox live ranges would otherwise be added
// to the register budget of the loop (+80 VGPRs), and only the few waves that contain a reset run it.
template <typename T> struct Scenario { T wb[3], wa[3], wph, t_mine[3]; };
template <typename T, int G>
__device__ __forceinline__ void sample_scenario_inl(const Params<T>* Pp, T* r, size_t n, int env, uint32_t ep, Scenario<T>* out) {
  const Params<T>& P = *Pp;
  const uint32_t genv = (uint32_t)(P.env_offset + env);
  const int sub = (G == 1) ? 0 : (int)(threadIdx.x & (G - 1));
  T wb[3], wa[3], wphase;
  // wind: base(3), gust amp(3), phase -- fixedwing_base_env.py:139-165
#pragma unroll
  for (int k = 0; k < 3; ++k) { wb[k] = P.wind_base[k]; wa[k] = P.wind_amp[k]; }
  wphase = P.wind_phase;
  if (P.wind_mode != FW_WIND_OFF && P.wind_randomize) {
    for (int k = 0; k < 3; ++k)
      wb[k] = (T)rng_uniform<T>(P, genv, ep, J_WIND_BASE + k, P.wind_base_range[k][0], P.wind_base_range[k][1]);
    if (P.wind_mode == FW_WIND_GUST_SINE) {
      for (int k = 0; k < 3; ++k)
        wa[k] = (T)rng_uniform<T>(P, genv, ep, J_WIND_AMP + k, P.wind_amp_range[k][0], P.wind_amp_range[k][1]);
      if (P.wind_randomize_phase) wphase = (T)rng_uniform<T>(P, genv, ep, J_WIND_PHASE, 0.0, 2.0 * kPi);
    }
  }
  if (P.wind_mode != FW_WIND_OFF && sub == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { r[(RF_WIND + k) * n + env] = wb[k]; r[(RF_WIND + 3 + k) * n + env] = wa[k]; }
    r[(RF_WIND + 6) * n + env] = wphase;
  }
  T tm[3] = {(T)0, (T)0, (T)0};
  // WaypointHandler.reset: polar sampling (always in double, cast once)
  if (P.task != FW_TASK_OBJLOCK) {
    const int i0 = (G == 1) ? 0 : sub, i1 = (G == 1) ? P.num_targets : min(sub + 1, P.num_targets);
#pragma unroll 1
    for (int i = i0; i < i1; ++i) {
      double theta = rng_uniform<T>(P, genv, ep, J_THETA + i, 0.0, 2.0 * kPi);
      double phi = rng_uniform<T>(P, genv, ep, J_PHI + i, 0.0, 2.0 * kPi);
      double dist = rng_uniform<T>(P, genv, ep, J_DIST + i, 1.0, (double)P.spawn_hi);
      double sphi, cphi, sth, cth;
      M<double>::sincos_(phi, &sphi, &cphi);
      M<double>::sincos_(theta, &sth, &cth);
      double x = dist * sphi * cth, y = dist * sphi * sth, z = ::fabs(dist * cphi);
      z = z > (double)P.min_height ? z : (double)P.min_height;
      T* tp = r + (size_t)(RF_TARGETS + 3 * i) * n + env;
      tp[0] = (T)x; tp[n] = (T)y; tp[2 * n] = (T)z;
      if (G > 1 || i == 0) { tm[0] = (T)x; tm[1] = (T)y; tm[2] = (T)z; }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { out->wb[k] = wb[k]; out->wa[k] = wa[k]; out->t_mine[k] = tm[k]; }
  out->wph = wphase;
}

// out-of-line entry for call sites inside the step loop (the deferred reset of the wind-free kernel inlines it
// into the epilogue instead, where almost nothing is live)
template <typename T, int G>
__device__ __noinline__ void sample_scenario(const Params<T>* Pp, T* r, size_t n, int env, uint32_t ep, Scenario<T>* out) {
  sample_scenario_inl<T, G>(Pp, r, n, env, ep, out);
}

template <typename T, int G>
__device__ __forceinline__ int begin_reset(const Params<T>& P, const DevState<T>& D, int env, Rigid<T>& S, int32_t& tick,
                                           int32_t& episode, int32_t& num_reached, T wb[3], T wa[3], T& wphase,
                                           T t_mine[3] /* out: the waypoint this lane sampled (G = 8) / waypoint 0 (G = 1) */) {
  episode += 1;
  {
    Scenario<T> sc;
    sample_scenario<T, G>(&P, D.r, (size_t)D.npad, env, (uint32_t)episode, &sc);
#pragma unroll
    for (int k = 0; k < 3; ++k) { wb[k] = sc.wb[k]; wa[k] = sc.wa[k]; t_mine[k] = sc.t_mine[k]; }
    wphase = sc.wph;
  }
  num_reached = 0;
  // Aviary(): start pose + PyFlyt starting velocity, zero actuators
  if (P.warm_valid) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { S.p[k] = P.warm[k]; S.v[k] = P.warm[7 + k]; S.w[k] = P.warm[10 + k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) S.q[k] = P.warm[3 + k];
#pragma unroll
    for (int k = 0; k < FW_NUM_ACTUATORS; ++k) S.act[k] = P.warm[13 + k];
    tick = P.warm_ticks;
    return 0;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { S.p[k] = P.start_pos[k]; S.v[k] = P.start_vel[k]; S.w[k] = (T)0; }
#pragma unroll
  for (int k = 0; k < 4; ++k) S.q[k] = P.start_quat[k];
#pragma unroll
  for (int k = 0; k < FW_NUM_ACTUATORS; ++k) S.act[k] = (T)0;
  tick = 0;
  return P.warmup_aviary_steps;
}

// first waypoint of an episode regenerated from the RNG (G = 8, after an in-kernel warm-up:
// nothing is kept live across it and no lane reads back what another lane has just stored)
template <typename T>
__device__ __forceinline__ void first_target(const Params<T>& P, uint32_t genv, uint32_t ep, T t0[3]) {
  double theta = rng_uniform<T>(P, genv, ep, J_THETA, 0.0, 2.0 * kPi);
  double phi = rng_uniform<T>(P, genv, ep, J_PHI, 0.0, 2.0 * kPi);
  double dist = rng_uniform<T>(P, genv, ep, J_DIST, 1.0, (double)P.spawn_hi);
  double sphi, cphi, sth, cth;
  M<double>::sincos_(phi, &sphi, &cphi);
  M<double>::sincos_(theta, &sth, &cth);
  double z = ::fabs(dist * cphi);
  t0[0] = (T)(dist * sphi * cth); t0[1] = (T)(dist * sphi * sth);
  t0[2] = (T)(z > (double)P.min_height ? z : (double)P.min_height);
}

// end_reset -> compute_state: WaypointHandler distances (old = 0 -> new)
template <typename T, int G>
__device__ __forceinline__ T end_reset(const Params<T>& P, const DevState<T>& D, int env, int32_t episode, const Rigid<T>& S,
                                       const T* t_first = nullptr /* waypoint 0 of the episode, if the caller still holds it */) {
  T new_dist = (T)0;
  if (P.task != FW_TASK_OBJLOCK && P.num_targets > 0) {
    T t0[3];
    if (t_first) {
      t0[0] = t_first[0]; t0[1] = t_first[1]; t0[2] = t_first[2];
    } else if (G == 1) {
      const size_t n = D.npad;
      const T* tp = D.r + (size_t)RF_TARGETS * n + env;
      t0[0] = tp[0]; t0[1] = tp[n]; t0[2] = tp[2 * n];
    } else {
      first_target<T>(P, (uint32_t)(P.env_offset + env), (uint32_t)episode, t0);
    }
    T dx = t0[0] - S.p[0], dy = t0[1] - S.p[1], dz = t0[2] - S.p[2];
    new_dist = M<T>::sqrt_(dx * dx + dy * dy + dz * dz);
  }
  return new_dist;
}

// ------------------------------------------------------------------------
// LDS-staged, coalesced store of a [rows x D] observation tile
// ------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void flush_obs_tile(const T* tile, int ld, T* obs, int blk_env0, int rows_max, int n, int D) {
  // tile[row*ld + k]; global rows [blk_env0, blk_env0+rows) are one contiguous span of rows*D elements
  const int rows = min(rows_max, n - blk_env0);
  const int total = rows * D;
  T* dst = obs + (size_t)blk_env0 * D;
  for (int e = threadIdx.x; e < total; e += kWave) {
    int row = e / D, col = e - row * D;
    dst[e] = tile[row * ld + col];
  }
}

}  // namespace fwsim
