// fwsim_objlock.hpp -- device side of the ObjLock task (envs/fixedwing_objlock_env.py).
//
// The reference derives its vision features from PyBullet's rendered segmentation /
// depth images (:643-761).  Rendering is replaced by an ANALYTIC camera (build-owned,
// DESIGN.md section 2b): duck = sphere, obstacles = vertical cylinders, ground = plane
// z = 0; a "frame" is the 8 numbers the reference extracts from the images.  Everything
// downstream of the frame (feature vector, 3-deep history with float32 deltas, obstacle
// penalty, dense lock / approach shaping, strike) follows the reference line by line.
// The task state of one env lives in registers for the whole launch.
#pragma once
#include <type_traits>
#include "fwsim_device.hpp"

namespace fwsim {

constexpr int kHist = FW_VISION_HIST * FW_VISION_FEATS;   // 27

template <typename T>
struct ObjC {          // wave-uniform task constants
  T cam_f[3], cam_r[3], cam_d[3], cam_off[3];
  T focal, inv_focal, W, H, vmid, near_, far_;
  T inv_near, inv_far, db_c1;   // 1 / near, 1 / far, far / (far - near): depth-buffer value = db_c1 (1 - near / t)
  T duck_radius, half_dome;
  T obst_radius, obst_hmin, obst_hmax, safe_dist, avoid_scale, avoid_max;
  T k_dist, lock_radius, k_center, k_visible, k_area, lost_penalty, approach_clip, k_approach;
  T strike_dist, strike_reward, lock_step_reward;
  T switch_min_area;
  T reach_margin;      // largest |collision point| + slack: how far from the COM a contact can happen
  int32_t hold_steps, decay_steps, num_obstacles, camera_ratio_ticks, switch_min_seen;
  int32_t zrow_stride;  // floats per env of the LDS row buffer of the camera (8-lane mapping)
};

template <typename T>
struct ObjState {
  T duck[3];
  T lock_steps, prev_est, last_cx, last_cy, last_area, last_depth, since_seen, filled, frame_has;
  T frame[8];          // visible, cx, cy, area, depth_m, d_left, d_center, d_right
  float hist[kHist];   // float32 by construction (np.float32 feature vectors)
  float cur_z[3];      // combined task: obstacle-zone depths of the current feature vector (transient)
  uint32_t near_mask;  // obstacles the aircraft can touch during this agent step (transient, conservative) that are NOT in the LDS table
  int32_t near_n;      // ... and how many of them are (G = 8: obj_update_near_mask)
  int32_t nob;
  int32_t phase;       // combined task: bit0 duck_phase, bit1 post_waypoints
  T seen_consec;
#ifdef FW_PROFILE
  long long p_cap = 0; int p_ncap = 0;    // dev-only: cycles spent in camera captures, number of captures (tools/wave_profile.py)
  long long p_ph[6] = {0, 0, 0, 0, 0, 0};  // dev-only: phases of capture_body in capture steps with 5+ envs due, [5] = how many
  long long p_capm[4] = {0, 0, 0, 0};     // dev-only, wave-level: cycles (low 40 bits) | episodes << 40 of capture steps with 1 / 2 / 3-4 / 5+ envs due
#endif
};

template <typename T> __device__ __forceinline__ T f32r(T x) { return (T)(float)x; }

// HIST = false (combined task): the 27-float vision history is not part of that task (its observation has no duck_vision,
// envs/flatten_waypoint_env.py:68-70) -- it is neither loaded, kept in registers nor stored; its state rows stay zero
template <typename T, bool HIST = true>
__device__ __forceinline__ void obj_load(const DevState<T>& D, int env, ObjState<T>& O) {
  const T* b = D.r + (size_t)RF_TASK * D.npad + env;
  const size_t n = D.npad;
#pragma unroll
  for (int k = 0; k < 3; ++k) O.duck[k] = b[(FW_ST_DUCK_POS + k) * n];
  O.lock_steps = b[FW_ST_LOCK_STEPS * n]; O.prev_est = b[FW_ST_PREV_EST * n];
  O.last_cx = b[FW_ST_LAST_CX * n]; O.last_cy = b[FW_ST_LAST_CY * n]; O.last_area = b[FW_ST_LAST_AREA * n];
  O.last_depth = b[FW_ST_LAST_DEPTH * n]; O.since_seen = b[FW_ST_SINCE_SEEN * n]; O.filled = b[FW_ST_HIST_FILLED * n];
  O.frame_has = b[FW_ST_FRAME_HAS * n];
#pragma unroll
  for (int k = 0; k < 8; ++k) O.frame[k] = b[(FW_ST_FRAME + k) * n];
  if (HIST) {
#pragma unroll
    for (int k = 0; k < kHist; ++k) O.hist[k] = (float)b[(FW_ST_HIST + k) * n];
  }
  O.nob = (int32_t)b[FW_ST_NUM_OBST * n];
  O.phase = (int32_t)b[FW_ST_DUCK_PHASE * n]; O.seen_consec = b[FW_ST_SEEN_CONSEC * n];
}
template <typename T, bool HIST = true>
__device__ __forceinline__ void obj_store(const DevState<T>& D, int env, const ObjState<T>& O) {
  T* b = D.r + (size_t)RF_TASK * D.npad + env;
  const size_t n = D.npad;
#pragma unroll
  for (int k = 0; k < 3; ++k) b[(FW_ST_DUCK_POS + k) * n] = O.duck[k];
  b[FW_ST_LOCK_STEPS * n] = O.lock_steps; b[FW_ST_PREV_EST * n] = O.prev_est;
  b[FW_ST_LAST_CX * n] = O.last_cx; b[FW_ST_LAST_CY * n] = O.last_cy; b[FW_ST_LAST_AREA * n] = O.last_area;
  b[FW_ST_LAST_DEPTH * n] = O.last_depth; b[FW_ST_SINCE_SEEN * n] = O.since_seen; b[FW_ST_HIST_FILLED * n] = O.filled;
  b[FW_ST_FRAME_HAS * n] = O.frame_has;
#pragma unroll
  for (int k = 0; k < 8; ++k) b[(FW_ST_FRAME + k) * n] = O.frame[k];
  if (HIST) {
#pragma unroll
    for (int k = 0; k < kHist; ++k) b[(FW_ST_HIST + k) * n] = (T)O.hist[k];
  }
  b[FW_ST_NUM_OBST * n] = (T)O.nob;
  b[FW_ST_DUCK_PHASE * n] = (T)O.phase; b[FW_ST_SEEN_CONSEC * n] = O.seen_consec;
}

// _reset_duck_state :409-419
template <typename T, bool HIST = true>
__device__ __forceinline__ void obj_reset_state(ObjState<T>& O) {
  O.lock_steps = (T)0; O.prev_est = (T)-1; O.last_cx = (T)0.5; O.last_cy = (T)0.5; O.last_area = (T)0; O.last_depth = (T)0;
  O.since_seen = (T)60; O.filled = (T)0; O.frame_has = (T)0;
#pragma unroll
  for (int k = 0; k < 8; ++k) O.frame[k] = (T)0;
  if (HIST) {
#pragma unroll
    for (int k = 0; k < kHist; ++k) O.hist[k] = 0.0f;
  }
  O.phase = 0; O.seen_consec = (T)0;
}

// Obstacle list of episode `ep`: candidate i = (height, x, y) from its own three Philox blocks, kept unless `reject(x, y)`;
// the kept ones are stored in candidate order.  G = 8: lane j of the env's group draws candidates j, j + 8, j + 16 and the
// positions in the list come from ballots (60 Philox blocks: 9 per lane instead of 60 -- the start of an episode is the longest
// chunk of a worker wave, and with it of the launch); all 8 lanes of the group must call.  G = 1 / `leader`: one lane stores.
template <typename T, int G, typename Rej>
__device__ __forceinline__ int sample_obstacles(const Params<T>& P, const ObjC<T>& OC, const DevState<T>& D, int env, uint32_t ep,
                                                bool leader, Rej&& reject) {
  const uint32_t genv = (uint32_t)(P.env_offset + env);
  const double r = (double)OC.half_dome;
  T* ob = D.r + (size_t)(RF_TASK + FW_ST_OBST) * D.npad + env;
  const size_t n = D.npad;
  int nob = 0;
  if (G == 8) {
    const int sub = (int)(threadIdx.x & 7), gsh = (int)(threadIdx.x & (kWave - 1)) & ~7;
#pragma unroll 1
    for (int i0 = 0; i0 < OC.num_obstacles; i0 += 8) {
      const int i = i0 + sub;
      double hh = 0.0, x = 0.0, y = 0.0;
      bool keep = false;
      if (i < OC.num_obstacles) {
        hh = rng_uniform<T>(P, genv, ep, J_OBST + 3 * i + 0, (double)OC.obst_hmin, (double)OC.obst_hmax);
        x = rng_uniform<T>(P, genv, ep, J_OBST + 3 * i + 1, -r, r);
        y = rng_uniform<T>(P, genv, ep, J_OBST + 3 * i + 2, -r, r);
        keep = !reject(x, y);
      }
      const uint32_t gm = (uint32_t)(__ballot(keep) >> gsh) & 0xFFu;
      if (keep) {
        const int at = nob + __popc(gm & ((1u << sub) - 1u));
        ob[(3 * at + 0) * n] = (T)x; ob[(3 * at + 1) * n] = (T)y; ob[(3 * at + 2) * n] = (T)hh;
      }
      nob += __popc(gm);
    }
    for (int i = nob + sub; i < FW_MAX_OBSTACLES; i += 8) { ob[(3 * i + 0) * n] = (T)0; ob[(3 * i + 1) * n] = (T)0; ob[(3 * i + 2) * n] = (T)0; }
    return nob;
  }
#pragma unroll 1
  for (int i = 0; i < OC.num_obstacles; ++i) {
    double hh = rng_uniform<T>(P, genv, ep, J_OBST + 3 * i + 0, (double)OC.obst_hmin, (double)OC.obst_hmax);
    double x = rng_uniform<T>(P, genv, ep, J_OBST + 3 * i + 1, -r, r);
    double y = rng_uniform<T>(P, genv, ep, J_OBST + 3 * i + 2, -r, r);
    if (reject(x, y)) continue;
    if (leader) { ob[(3 * nob + 0) * n] = (T)x; ob[(3 * nob + 1) * n] = (T)y; ob[(3 * nob + 2) * n] = (T)hh; }
    ++nob;
  }
  if (leader)
    for (int i = nob; i < FW_MAX_OBSTACLES; ++i) { ob[(3 * i + 0) * n] = (T)0; ob[(3 * i + 1) * n] = (T)0; ob[(3 * i + 2) * n] = (T)0; }
  return nob;
}

// _spawn_duck :461-491, _spawn_obstacles :507-565.  Every lane computes the duck; the obstacle list is sampled by the env's lanes
// together (sample_obstacles).
template <typename T, int G>
__device__ __forceinline__ void obj_spawn_impl(const Params<T>& P, const ObjC<T>& OC, const DevState<T>& D, int env, uint32_t ep,
                                          bool leader, ObjState<T>& O) {
  const uint32_t genv = (uint32_t)(P.env_offset + env);
  const double r = (double)OC.half_dome;
  const double dx = rng_uniform<T>(P, genv, ep, J_DUCK_X, -r, r), dy = rng_uniform<T>(P, genv, ep, J_DUCK_Y, -r, r);
  O.duck[0] = (T)dx; O.duck[1] = (T)dy; O.duck[2] = (T)0.05;
  O.nob = sample_obstacles<T, G>(P, OC, D, env, ep, leader, [&](double x, double y) {
    const double ex = x - dx, ey = y - dy;
    return M<double>::sqrt_(ex * ex + ey * ey) < 10.0 || x * x + y * y < 100.0;
  });
}

// i-th waypoint of episode `ep`, regenerated from the RNG (WaypointHandler.reset polar sampling)
template <typename T>
__device__ __forceinline__ void nth_target(const Params<T>& P, uint32_t genv, uint32_t ep, int i, T t[3]) {
  double theta = rng_uniform<T>(P, genv, ep, J_THETA + i, 0.0, 2.0 * kPi);
  double phi = rng_uniform<T>(P, genv, ep, J_PHI + i, 0.0, 2.0 * kPi);
  double dist = rng_uniform<T>(P, genv, ep, J_DIST + i, 1.0, (double)P.spawn_hi);
  double sphi, cphi, sth, cth;
  M<double>::sincos_(phi, &sphi, &cphi);
  M<double>::sincos_(theta, &sth, &cth);
  double z = ::fabs(dist * cphi);
  t[0] = (T)(dist * sphi * cth); t[1] = (T)(dist * sphi * sth);
  t[2] = (T)(z > (double)P.min_height ? z : (double)P.min_height);
}

// combined task: duck at the last waypoint's x,y (envs/fixedwing_waypoint_objlock_env.py:394-436), obstacles
// with the origin rejection only (:452-503)
template <typename T, int G>
__device__ __forceinline__ void comb_spawn_impl(const Params<T>& P, const ObjC<T>& OC, const DevState<T>& D, int env, uint32_t ep,
                                           bool leader, ObjState<T>& O) {
  const uint32_t genv = (uint32_t)(P.env_offset + env);
  if (P.num_targets > 0) {
    T tl[3];
    nth_target<T>(P, genv, ep, P.num_targets - 1, tl);
    O.duck[0] = tl[0]; O.duck[1] = tl[1];
  } else { O.duck[0] = (T)10; O.duck[1] = (T)0; }
  O.duck[2] = (T)0.05;
  O.nob = sample_obstacles<T, G>(P, OC, D, env, ep, leader, [&](double x, double y) { return x * x + y * y < 100.0; });
}

// Out-of-line entry points of the spawners (cold path: keeps their live ranges out of the step loop's register budget).
template <typename T> struct Spawned { T duck[3]; int32_t nob; };
template <typename T, int TKIND, int G>
__device__ __noinline__ void spawn_task(const Params<T>* Pp, const ObjC<T>* OCp, T* r, int npad, int env, uint32_t ep, bool leader, Spawned<T>* out) {
  DevState<T> D; D.r = r; D.npad = npad;
  ObjState<T> O;
  if (TKIND == FW_TASK_OBJLOCK) obj_spawn_impl<T, G>(*Pp, *OCp, D, env, ep, leader, O); else comb_spawn_impl<T, G>(*Pp, *OCp, D, env, ep, leader, O);
  out->duck[0] = O.duck[0]; out->duck[1] = O.duck[1]; out->duck[2] = O.duck[2]; out->nob = O.nob;
}
template <typename T, int G>
__device__ __forceinline__ void obj_spawn(const Params<T>& P, const ObjC<T>& OC, const DevState<T>& D, int env, uint32_t ep, bool leader, ObjState<T>& O) {
  Spawned<T> sp;
  spawn_task<T, FW_TASK_OBJLOCK, G>(&P, &OC, D.r, D.npad, env, ep, leader, &sp);
  O.duck[0] = sp.duck[0]; O.duck[1] = sp.duck[1]; O.duck[2] = sp.duck[2]; O.nob = sp.nob;
}
template <typename T, int G>
__device__ __forceinline__ void comb_spawn(const Params<T>& P, const ObjC<T>& OC, const DevState<T>& D, int env, uint32_t ep, bool leader, ObjState<T>& O) {
  Spawned<T> sp;
  spawn_task<T, FW_TASK_WAYPOINT_OBJLOCK, G>(&P, &OC, D.r, D.npad, env, ep, leader, &sp);
  O.duck[0] = sp.duck[0]; O.duck[1] = sp.duck[1]; O.duck[2] = sp.duck[2]; O.nob = sp.nob;
}

// LDS map of the analytic camera on the 8-lane mapping (aliases the observation tile, which is only written after the step
// loop).  FRONT (every camera handle), bytes from the start of dynamic LDS:
//   8 x kSetWords words of T   constants of a lane set's env: row h//2 (pp, pq, qq, g0z, g1z, cam z; ground gk0, gk1; duck columns
//                              first, last) and duck mask (zc, xc, yc, k2, A, 1 / A, first row, rows)
//   8 x 4 words                duck mask accumulators of a set: count, sum x, sum y (exact sums: order-free), max 1 / t (bits)
//   40 x u32                   [0] slices, [1] chunks, [2] mask rows of the wave, [4 + 2 s], [5 + 2 s] covered columns of set s,
//                              [20 + 2 s], [21 + 2 s] duck columns on row h//2 of set s
// CYLINDERS (handles with num_obstacles > 0), behind it:
//   8 row buffers of zrow_stride words (1 / t of the nearest cylinder fragment per column, one per lane set)
//   8 tables of FW_MAX_OBSTACLES x kCtabWords (the screened cylinders of a set: cc, hh, op, oq, first column, last column, 1 / cc)
//   8 x 3 x chunks-per-third partial sums of the rows
//   u16 slice list: set << 10 | cylinder << 5 | index of the 32-column slice inside the cylinder's interval
//   u16 chunk list: set << 8 | third << 6 | index of the 32-column chunk inside (covered columns of the set) x (third)
//   the cylinders an env can touch during this agent step (obj_update_near_mask): 8 x u32 counts, 8 x kNearSlots x (x, y, height)
constexpr int kCtabWords = 7, kSetWords = 20, kSliceCols = 32, kNearSlots = 4, kCamU32 = 40;
__host__ __device__ inline int chunks_per_third(int res) { return ((res + 2) / 3 + kSliceCols - 1) / kSliceCols + 1; }
struct CamLds { size_t sconst, dacc, lu, zr, ctab, cpart, slist, clist, near_cnt, near_tab, total; };     // byte offsets
__host__ __device__ inline CamLds cam_lds(size_t word, int zrow_stride, int res, bool cylinders) {
  CamLds L;
  size_t o = 0;
  L.sconst = o; o += word * 8 * kSetWords;
  L.dacc = o; o += word * 8 * 4;
  L.lu = o; o += 4 * kCamU32;
  o = (o + 15) & ~(size_t)15;
  L.zr = L.ctab = L.cpart = L.slist = L.clist = L.near_cnt = L.near_tab = o;
  if (cylinders) {
    const size_t slices = (size_t)8 * FW_MAX_OBSTACLES * (size_t)((res + kSliceCols - 1) / kSliceCols);
    const size_t chunks = (size_t)8 * 3 * (size_t)chunks_per_third(res);
    L.zr = o; o += word * 8 * (size_t)zrow_stride;
    L.ctab = o; o += word * 8 * FW_MAX_OBSTACLES * kCtabWords;
    L.cpart = o; o += word * chunks;
    L.slist = o; o += 2 * slices;
    L.clist = o; o += 2 * chunks;
    o = (o + 15) & ~(size_t)15;
    L.near_cnt = o; o += 32;
    L.near_tab = o; o += word * 8 * kNearSlots * 3;
  }
  L.total = o;
  return L;
}
// Which cylinders can be touched during this agent step?  Conservative: horizontal distance of the COM to
// the axis below radius + reach of the airframe + 1.25 x the distance flown in one agent step (+ slack).
// The per-tick contact test then only visits these (usually none) instead of all 20.
// G = 8: lane j screens obstacles j, j+8, j+16; the masks are OR-ed across the group.
template <typename T, int G>
__device__ __forceinline__ void obj_update_near_mask(const Params<T>& P, const ObjC<T>& OC, const DevState<T>& D, int env,
                                                     ObjState<T>& O, const Rigid<T>& S) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const T* ob = D.r + (size_t)(RF_TASK + FW_ST_OBST) * D.npad + env;
  const size_t n = D.npad;
  const T speed = M<T>::sqrt_(S.v[0] * S.v[0] + S.v[1] * S.v[1] + S.v[2] * S.v[2]);
  const T t_step = (T)(P.step_ratio * P.ticks_per_aviary) * P.dt;
  const T reach = OC.obst_radius + OC.reach_margin + (speed + (T)5) * t_step * (T)1.25 + (T)0.5;
  uint32_t m = 0u;
  O.near_n = 0;
  if (G == 8 && OC.num_obstacles > 0) {
    // the first kNearSlots of them go to an LDS table of the env (x, y, height): the contact test of every tick then costs LDS
    // reads instead of a dependent memory round trip per cylinder; any further ones stay in the mask (read from memory)
    const int sub = (int)(threadIdx.x & 7), row = (int)(threadIdx.x & (kWave - 1)) >> 3;
    const CamLds L = cam_lds(sizeof(T), OC.zrow_stride, (int)OC.W, true);
    uint32_t* ncnt = reinterpret_cast<uint32_t*>(smem_raw + L.near_cnt) + row;
    T* ntab = reinterpret_cast<T*>(smem_raw + L.near_tab) + (size_t)row * kNearSlots * 3;
    if (sub == 0) *ncnt = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    T cx[3], cy[3];
#pragma unroll
    for (int slot = 0; slot < 3; ++slot) {                        // all loads in flight together
      const int o = sub + 8 * slot;
      cx[slot] = cy[slot] = (T)0;
      if (o < O.nob) { cx[slot] = ob[(3 * o) * n]; cy[slot] = ob[(3 * o + 1) * n]; }
    }
#pragma unroll
    for (int slot = 0; slot < 3; ++slot) {
      const int o = sub + 8 * slot;
      const T ox = S.p[0] - cx[slot], oy = S.p[1] - cy[slot];
      if (o < O.nob && ox * ox + oy * oy <= reach * reach) {
        const uint32_t pos = __hip_atomic_fetch_add(ncnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (pos < (uint32_t)kNearSlots) { T* e = ntab + pos * 3; e[0] = cx[slot]; e[1] = cy[slot]; e[2] = ob[(3 * o + 2) * n]; }
        else m |= 1u << o;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint32_t c = *ncnt;
    O.near_n = (int)(c < (uint32_t)kNearSlots ? c : (uint32_t)kNearSlots);
  } else {
    const int sub = (G == 1) ? 0 : (int)(threadIdx.x & (G - 1));
    for (int o = sub; o < O.nob; o += G) {
      T ox = S.p[0] - ob[(3 * o) * n], oy = S.p[1] - ob[(3 * o + 1) * n];
      if (ox * ox + oy * oy <= reach * reach) m |= 1u << o;
    }
  }
  O.near_mask = group_or<G>(m);
}

// is world point pw inside the duck sphere or a (near) obstacle cylinder?
template <typename T>
__device__ __forceinline__ bool obj_point_hit(const ObjC<T>& OC, const DevState<T>& D, int env, const ObjState<T>& O, const T pw[3]) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T dx = pw[0] - O.duck[0], dy = pw[1] - O.duck[1], dz = pw[2] - (O.duck[2] + OC.duck_radius);
  bool hit = dx * dx + dy * dy + dz * dz <= OC.duck_radius * OC.duck_radius;
  if (O.near_n > 0) {
    const T* ntab = reinterpret_cast<const T*>(smem_raw + cam_lds(sizeof(T), OC.zrow_stride, (int)OC.W, true).near_tab) +
                    (size_t)((int)(threadIdx.x & (kWave - 1)) >> 3) * kNearSlots * 3;
    for (int k = 0; k < O.near_n; ++k) {
      const T ox = pw[0] - ntab[3 * k], oy = pw[1] - ntab[3 * k + 1];
      hit |= (ox * ox + oy * oy <= OC.obst_radius * OC.obst_radius) && (pw[2] <= ntab[3 * k + 2]);
    }
  }
  const T* ob = D.r + (size_t)(RF_TASK + FW_ST_OBST) * D.npad + env;
  const size_t n = D.npad;
  uint32_t m = O.near_mask;
  while (m) {
    const int o = __ffs((int)m) - 1;
    m &= m - 1u;
    T ox = pw[0] - ob[(3 * o) * n], oy = pw[1] - ob[(3 * o + 1) * n];
    hit |= (ox * ox + oy * oy <= OC.obst_radius * OC.obst_radius) && (pw[2] <= ob[(3 * o + 2) * n]);
  }
  return hit;
}

// contacts of the body-fixed points with duck / cylinders (G = 8: lane `sub` owns point `sub`)
template <typename T, int G>
__device__ __forceinline__ bool obj_contacts(const Params<T>& P, const TickC<T>& C, const ObjC<T>& OC, const DevState<T>& D, int env,
                                             const ObjState<T>& O, const Rigid<T>& S, const T R[9]) {
  if (G == 8) {
    T pw[3];
    mv(R, C.cpt, pw);
    pw[0] += S.p[0]; pw[1] += S.p[1]; pw[2] += S.p[2];
    return group_any<8>(C.cvalid != (T)0 && obj_point_hit<T>(OC, D, env, O, pw));
  }
  bool hit = false;
  for (int i = 0; i < P.n_coll; ++i) {
    T pb[3] = { P.coll[i][0], P.coll[i][1], P.coll[i][2] }, pw[3];
    mv(R, pb, pw);
    pw[0] += S.p[0]; pw[1] += S.p[1]; pw[2] += S.p[2];
    hit |= obj_point_hit<T>(OC, D, env, O, pw);
  }
  return hit;
}

// One cylinder against a camera ray / the line of sight to the duck.
// Returns the hit parameter t (depth along the view axis for rays built with unit forward component) or +inf.
template <typename T>
__device__ __forceinline__ T cyl_hit(const ObjC<T>& OC, T cx, T cy, T hh, const T cam[3], const T dw[3]) {
  const T inf = (T)1e300;
  T ox = cam[0] - cx, oy = cam[1] - cy;
  T a = dw[0] * dw[0] + dw[1] * dw[1], b = (T)2 * (ox * dw[0] + oy * dw[1]), cc = ox * ox + oy * oy - OC.obst_radius * OC.obst_radius;
  if (a <= (T)0) return inf;
  T disc = b * b - (T)4 * a * cc;
  if (disc < (T)0) return inf;
  T t = M<T>::div_(-b - M<T>::sqrt_(disc), (T)2 * a);
  if (t <= (T)0) return inf;
  T z = cam[2] + t * dw[2];
  return (z < (T)0 || z > hh) ? inf : t;
}

// ------------------------------------------------------------------------------------------
// Camera.capture_image() replaced by an analytic render of the scene, and _compute_vision_features' image statistics
// (:662-743) computed on it -- the SAME functionals the reference applies to segImg / depthImg:
//   duck mask = pixels whose ray hits the sphere between the clip planes (none if a cylinder blocks the line of sight to
//   its centre);  cx, cy = mean(xs)/(w-1), mean(ys)/(h-1);  area = count/(h*w);  depth = metres(min depth-BUFFER value over
//   the mask);  d_left/center/right = metres(mean depth-BUFFER value (float32, as depthImg) of the non-duck pixels of the
//   thirds of row h//2), 0 for an empty third.
// Nothing is rasterised pixel by pixel where a closed form exists:
//   * a row of the duck mask is the interval between the roots of a quadratic (silhouette of the sphere on that row):
//     count, sum(x), sum(y) come from the interval's end points; the row's nearest fragment is one of the two pixels
//     around the closed-form continuous minimiser (the depth along a scan line across a sphere is unimodal);
//   * the ground's depth-buffer value is linear in the column; a cylinder covers the columns of one interval (the
//     directions inside its tangent cone), found in closed form (conservatively) and then tested pixel by pixel.
// G = 8: a capture step is run by the whole wave (obj_capture_wave): the envs that are due hand their pose to sets of 8-64
// lanes; the rows of all masks, the cylinder intervals (in 32-column slices) and the row sums (in 32-column chunks) are work
// lists in LDS dealt to the wave's lanes, combined through LDS atomics whose results do not depend on the order (max of
// 1 / t on the ordered bit pattern; exact sums of integers) or through slots added in a fixed order -- so a frame is the same
// bits whatever the neighbours do.  G = 1: one lane does all of it, pixel by pixel against every cylinder (the throughput
// mapping has no LDS to spare for 64 rows; it serves the camera tasks only without obstacles and above 16 384 envs).
// ------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T depthbuf_from_inv(const ObjC<T>& OC, T inv_t) {
  // depth-buffer value far (t - near) / (t (far - near)) = c1 (1 - near / t) of a fragment at view-axis depth t (clipped to
  // [near, far]); the analytic depth image is kept in the env dtype (PyBullet's is float32: a renderer property)
  inv_t = inv_t > OC.inv_near ? OC.inv_near : inv_t;
  inv_t = inv_t < OC.inv_far ? OC.inv_far : inv_t;
  return OC.db_c1 * ((T)1 - OC.near_ * inv_t);
}
template <typename T> __device__ __forceinline__ T depthbuf_to_meters(const ObjC<T>& OC, T d) {           // :691-696
  const T denom = OC.far_ - (OC.far_ - OC.near_) * d;
  return M<T>::fabs_(denom) < (T)1e-9 ? OC.far_ : M<T>::div_(OC.far_ * OC.near_, denom);
}
template <typename T> __device__ __forceinline__ T ceil_(T x) { return ::ceil(x); }
template <typename T> __device__ __forceinline__ T floor_(T x) { return ::floor(x); }

// 1 / t of the first hit of the ray cam + t dw (t = view-axis depth) with the cylinder (cx, cy, radius, [0, hh]); 0 = miss
template <typename T>
__device__ __forceinline__ T cyl_inv_t(const ObjC<T>& OC, T cx, T cy, T hh, const T cam[3], T dwx, T dwy, T dwz) {
  const T ox = cam[0] - cx, oy = cam[1] - cy;
  const T a = dwx * dwx + dwy * dwy, hb = ox * dwx + oy * dwy, cc = ox * ox + oy * oy - OC.obst_radius * OC.obst_radius;
  const T disc = hb * hb - a * cc;                       // (b^2 - 4 a c) / 4 with b = 2 hb
  if (!(a > (T)0) || disc < (T)0 || hb >= (T)0) return (T)0;      // t = (-hb - sqrt(disc)) / a > 0 needs hb < 0 (outside the cylinder)
  const T num = -hb - M<T>::sqrt_(disc);
  if (!(num > (T)0)) return (T)0;
  const T inv_t = M<T>::div_(a, num);
  const T z = cam[2] + M<T>::div_(num, a) * dwz;
  return (z < (T)0 || z > hh) ? (T)0 : inv_t;
}

// column interval [xlo, xhi] of row h//2 that a cylinder can cover (conservative: one pixel of slack, the per-pixel test
// decides), in float: the directions inside the tangent cone of its disc, o.d <= 0 and (o.d)^2 >= |d|^2 (L^2 - r^2), d = p + a q
__device__ __forceinline__ int cyl_columns(float ox, float oy, float r2, float px, float py, float qx, float qy, float u0, float F, float invF, float W) {
  const float cc = ox * ox + oy * oy - r2;
  if (!(cc > 0.f)) return 1;                                      // camera inside the cylinder (a contact ends the episode): nothing drawn
  const float op = ox * px + oy * py, oq = ox * qx + oy * qy;
  const float pp = px * px + py * py, pq = px * qx + py * qy, qq = qx * qx + qy * qy;
  const float A2 = oq * oq - qq * cc, B2 = op * oq - pq * cc, C2 = op * op - pp * cc;
  auto inside = [&](float a) { const float od = op + a * oq; return od <= 0.f && od * od >= (pp + a * (2.f * pq + a * qq)) * cc; };
  const float amin = (0.f - u0) * invF, amax = ((W - 1.f) - u0) * invF;
  float lo = 1.f, hi = 0.f;
  const float d2 = B2 * B2 - A2 * C2;
  if (d2 >= 0.f && fabsf(A2) > 1e-30f) {
    const float sq = sqrtf(d2), ia = 1.0f / A2;
    float r1 = (-B2 - sq) * ia, r2_ = (-B2 + sq) * ia;
    if (r1 > r2_) { const float t_ = r1; r1 = r2_; r2_ = t_; }
    if (inside(0.5f * (r1 + r2_))) { lo = r1; hi = r2_; }
    else if (inside(r1 - 1.f)) { lo = amin; hi = r1; }
    else if (inside(r2_ + 1.f)) { lo = r2_; hi = amax; }
  } else if (inside(0.f)) { lo = amin; hi = amax; }
  if (!(hi >= lo)) return 1;
  float fl = floorf(u0 + F * lo) - 2.f, fh = ceilf(u0 + F * hi) + 2.f;      // float rounding + one pixel of slack
  fl = fl < 0.f ? 0.f : fl; fh = fh > W - 1.f ? W - 1.f : fh;
  return (fh >= fl) ? ((int)fl | ((int)fh << 16)) : 1;            // packed lo | hi << 16; 1 = (lo 1, hi 0) = empty
}

// The capture of ONE env, run by a SET of `VG` lanes (G = 8: VG = 8, 16, 32 or 64 consecutive lanes of the wave, lane `vsub` of the
// set; G = 1: one lane): all of them are handed the owner env's pose / duck / obstacle count and compute the same scalars; rows of
// the duck mask, cylinders and pixels of row h//2 are dealt out modulo VG; partial statistics are combined inside the set (DPP
// inside 8 lanes, xor-butterflies above), after which every lane of the set holds the same `frame`.  `work` = false: a lane set
// without an env this time (it only takes part in the cross-lane steps).
template <typename T, int G, bool NOLEAD = false>
__device__ __forceinline__ void capture_body(const ObjC<T>& OC, const DevState<T>& D, int env, const T duck[3], int nob_in,
                                             const T Sp[3], const T R[9], int vsub, int VG, int erow, bool work, T frame[8],
                                             long long* ph = nullptr, bool ph_on = false /* dev-only: cycles per phase (FW_PROFILE) */) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
#ifdef FW_PROFILE
  long long ph_t = (long long)__builtin_readcyclecounter();
#define FW_PH(i) do { const long long t_ = (long long)__builtin_readcyclecounter(); if (ph_on) ph[i] += t_ - ph_t; ph_t = t_; } while (0)
#else
#define FW_PH(i) do { } while (0)
#endif
  const size_t n = D.npad;
  const int nob = work ? nob_in : 0;
  const int sub = vsub & (G - 1);                     // lane within its 8-lane group
  // reductions over the set (every lane gets the result)
  auto ssum = [&](T v) {
    v = group_sum<G, T>(v);
    if (G == 8) { if (VG >= 16) v += __shfl_xor(v, 8, kWave); if (VG >= 32) v += __shfl_xor(v, 16, kWave); if (VG >= 64) v += __shfl_xor(v, 32, kWave); }
    return v;
  };
  auto smin = [&](auto v) {
    v = group_min<G>(v);
    if (G == 8) {
      if (VG >= 16) { auto o = __shfl_xor(v, 8, kWave); v = o < v ? o : v; }
      if (VG >= 32) { auto o = __shfl_xor(v, 16, kWave); v = o < v ? o : v; }
      if (VG >= 64) { auto o = __shfl_xor(v, 32, kWave); v = o < v ? o : v; }
    }
    return v;
  };
  auto sor = [&](uint32_t v) {
    v = group_or<G>(v);
    if (G == 8) {
      if (VG >= 16) v |= (uint32_t)__shfl_xor((int)v, 8, kWave);
      if (VG >= 32) v |= (uint32_t)__shfl_xor((int)v, 16, kWave);
      if (VG >= 64) v |= (uint32_t)__shfl_xor((int)v, 32, kWave);
    }
    return v;
  };
  T cam[3];
  {
    T offw[3];
    mv(R, OC.cam_off, offw);
#pragma unroll
    for (int k = 0; k < 3; ++k) cam[k] = Sp[k] + offw[k];
  }
  const T* ob = D.r + (size_t)(RF_TASK + FW_ST_OBST) * n + env;
  const T W = OC.W, H = OC.H, F = OC.focal, invF = OC.inv_focal;
  const int Wi = (int)W, Hi = (int)H;
  const T u0 = (T)0.5 * (W - (T)1), v0 = (T)0.5 * (H - (T)1), Rd = OC.duck_radius;
  // ---- duck: sphere centre in camera coordinates (zc forward, xc right, yc down) ----
  T relw[3] = { duck[0] - cam[0], duck[1] - cam[1], duck[2] + Rd - cam[2] }, relb[3];
  mtv(R, relw, relb);
  const T zc = relb[0] * OC.cam_f[0] + relb[1] * OC.cam_f[1] + relb[2] * OC.cam_f[2];
  const T xc = relb[0] * OC.cam_r[0] + relb[1] * OC.cam_r[1] + relb[2] * OC.cam_r[2];
  const T yc = relb[0] * OC.cam_d[0] + relb[1] * OC.cam_d[1] + relb[2] * OC.cam_d[2];
  const T k2 = zc * zc + xc * xc + yc * yc - Rd * Rd;
  // row h//2 in world coordinates: dw(a) = g0 + a g1  (forward component 1 => the ray parameter is the view-axis depth)
  const int y_mid = Hi / 2, x_1 = Wi / 3, x_2 = (2 * Wi) / 3;
  const T bm = ((T)y_mid - v0) * invF;
  T g0[3], g1[3];
  {
    T db[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) db[k] = OC.cam_f[k] + bm * OC.cam_d[k];
    mv(R, db, g0);
    mv(R, OC.cam_r, g1);
  }
  const bool duck_front = (zc - Rd > OC.near_) && (zc - Rd < OC.far_);
  // ---- cylinders: line-of-sight occlusion of the duck + the columns each one can cover on row h//2 (lane-parallel) ----
  bool occluded = false;
  uint32_t vis = 0u;                          // G = 8: cylinders that can show on row h//2 (group-uniform after the OR below)
  // G = 8: LDS of this env -- the row buffer (1 / t of the nearest cylinder fragment per column; aliases the observation tile,
  // which is only written after the step loop) and, behind the 8 rows, a table of kCtabWords words per cylinder filled by the lane
  // that screened it: ox, oy, cc, hh, op, oq, first column, last column
  using UB = std::conditional_t<sizeof(T) == 8, unsigned long long, unsigned int>;      // T as ordered bits (positive values)
  const bool cyl_on = G == 8 && OC.num_obstacles > 0;   // (the handle's LDS holds the cylinder part of the camera map only then)
  const CamLds L = cam_lds(sizeof(T), OC.zrow_stride, Wi, cyl_on);
  UB* const zr_all = reinterpret_cast<UB*>(smem_raw + L.zr);
  UB* zr = zr_all + (size_t)erow * OC.zrow_stride;
  T* ctab_all = reinterpret_cast<T*>(smem_raw + L.ctab);
  T* ctab = ctab_all + (size_t)erow * (FW_MAX_OBSTACLES * kCtabWords);
  T* sconst_all = reinterpret_cast<T*>(smem_raw + L.sconst);
  T* dacc_all = reinterpret_cast<T*>(smem_raw + L.dacc);
  const int cpz = chunks_per_third(Wi);
  T* cpart_all = reinterpret_cast<T*>(smem_raw + L.cpart);                     // [set][third][chunk] partial sums of the row
  uint32_t* lu = reinterpret_cast<uint32_t*>(smem_raw + L.lu);
  uint16_t* slist = reinterpret_cast<uint16_t*>(smem_raw + L.slist);
  uint16_t* clist = reinterpret_cast<uint16_t*>(smem_raw + L.clist);
  const T pp = g0[0] * g0[0] + g0[1] * g0[1], pq = g0[0] * g1[0] + g0[1] * g1[1], qq = g1[0] * g1[0] + g1[1] * g1[1];
  if (G == 8) {
    if ((threadIdx.x & (kWave - 1)) == 0) { lu[0] = 0u; lu[1] = 0u; lu[2] = 0u; }
    if ((threadIdx.x & (kWave - 1)) < 8) sconst_all[(threadIdx.x & 7) * kSetWords + 17] = (T)0;     // no mask rows listed (also for sets without lanes this time)
    if (vsub == 0) {
      lu[4 + 2 * erow] = 0x7FFFFFFFu; lu[5 + 2 * erow] = 0u; lu[20 + 2 * erow] = 0x7FFFFFFFu; lu[21 + 2 * erow] = 0u;
      T* da = dacc_all + erow * 4;
      da[0] = da[1] = da[2] = da[3] = (T)0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (nob > 0) {
    if (G == 8) {
      T myc[3][3];
#pragma unroll
      for (int slot = 0; slot < 3; ++slot) {                      // all loads in flight together: one memory round trip
        const int o = vsub + VG * slot;
        myc[slot][0] = myc[slot][1] = myc[slot][2] = (T)0;
        if (o < nob) { myc[slot][0] = ob[(3 * o) * n]; myc[slot][1] = ob[(3 * o + 1) * n]; myc[slot][2] = ob[(3 * o + 2) * n]; }
      }
      const T r2 = OC.obst_radius * OC.obst_radius;
#pragma unroll
      for (int slot = 0; slot < 3; ++slot) {
        const int o = vsub + VG * slot;
        if (o < nob) {
          const T cx = myc[slot][0], cy = myc[slot][1], hh = myc[slot][2];
          if (duck_front) occluded |= cyl_inv_t<T>(OC, cx, cy, hh, cam, relw[0], relw[1], relw[2]) > (T)1;   // a hit at 0 < t < 1 of the segment camera -> sphere centre
          const T ox = cam[0] - cx, oy = cam[1] - cy;
          const int packed = cyl_columns((float)ox, (float)oy, (float)r2, (float)g0[0], (float)g0[1],
                                         (float)g1[0], (float)g1[1], (float)u0, (float)F, (float)invF, (float)W);
          const int xlo = packed & 0xFFFF, xhi = packed >> 16;
          if (xhi >= xlo) {
            vis |= 1u << o;
            T* e = ctab + o * kCtabWords;
            const T cc = ox * ox + oy * oy - r2;
            e[0] = cc; e[1] = hh;
            e[2] = ox * g0[0] + oy * g0[1]; e[3] = ox * g1[0] + oy * g1[1]; e[4] = (T)xlo; e[5] = (T)xhi; e[6] = M<T>::rcp_(cc);
            // the interval joins the wave's work list in slices of kSliceCols columns
            (void)__hip_atomic_fetch_min(lu + 4 + 2 * erow, (uint32_t)xlo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void)__hip_atomic_fetch_max(lu + 5 + 2 * erow, (uint32_t)xhi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int nsl = (xhi - xlo) / kSliceCols + 1;
            const uint32_t pos = __hip_atomic_fetch_add(lu, (uint32_t)nsl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (int i = 0; i < nsl; ++i) slist[pos + i] = (uint16_t)((erow << 10) | (o << 5) | i);
          }
        }
      }
      // one cross-lane step for both results of the screening
      const uint32_t pk = sor(vis | (occluded ? 0x80000000u : 0u));
      vis = pk & 0x7FFFFFFFu; occluded = (pk >> 31) != 0u;
      if (vsub == 0 && vis) {
        T* sc = sconst_all + erow * kSetWords;
        sc[0] = pp; sc[1] = pq; sc[2] = qq; sc[3] = g0[2]; sc[4] = g1[2]; sc[5] = cam[2];
      }
    } else {
#pragma unroll 1
      for (int o = 0; o < nob; ++o)
        if (duck_front) occluded |= cyl_inv_t<T>(OC, ob[(3 * o) * n], ob[(3 * o + 1) * n], ob[(3 * o + 2) * n], cam, relw[0], relw[1], relw[2]) > (T)1;
    }
  }
  FW_PH(0);                                                       // set-up + screening
  // ---- duck mask statistics ----
  T cnt = (T)0, sx = (T)0, sy = (T)0, itmax = (T)0;               // itmax = 1 / (nearest fragment depth)
  int mid_lo = 1 << 30, mid_hi = -1;                              // duck columns on row h//2
  const bool duck_ok = duck_front && !occluded;
  const bool straddle = zc + Rd >= OC.far_;                       // some fragments may lie beyond the far plane: test them one by one
  // 1 / t of the sphere hit of pixel direction (1, a, b) for a sphere at (zc_, xc_, yc_) with k2_ = |c|^2 - R^2; 0 = miss / clipped
  auto inv_hit_c = [&](T a, T b, T zc_, T xc_, T yc_, T k2_) {
    const T q = (T)1 + a * a + b * b, p = zc_ + a * xc_ + b * yc_, disc = p * p - q * k2_;
    if (disc < (T)0 || p <= (T)0) return (T)0;
    const T it = M<T>::div_(q, p - M<T>::sqrt_(disc));
    return (it < OC.inv_near && it > OC.inv_far) ? it : (T)0;     // near < t < far
  };
  auto inv_hit = [&](T a, T b) { return inv_hit_c(a, b, zc, xc, yc, k2); };
  // One row of a mask that lies wholly inside the far plane: the silhouette's interval [fx0, fx1] on image row y (empty:
  // fx1 < fx0), and 1 / t of the row's nearest fragment -- one of the two pixels around the closed-form minimiser of the scan
  // line's depth profile -- unless the row's continuous minimum cannot beat `best` (then 0)
  auto mask_row = [&](int y, T zc_, T xc_, T yc_, T k2_, T A_, T iA_, T best, T& fx0, T& fx1) {
    fx0 = (T)1; fx1 = (T)0;
    const T b = ((T)y - v0) * invF;
    const T e = zc_ + b * yc_, Bh = xc_ * e, Cq = e * e - ((T)1 + b * b) * k2_;
    const T Dd = Bh * Bh - A_ * Cq;
    if (Dd < (T)0) return (T)0;
    const T sq = M<T>::sqrt_(Dd);
    const T a_lo = (-Bh + sq) * iA_, a_hi = (-Bh - sq) * iA_;
    T f0 = ceil_<T>(u0 + F * a_lo), f1 = floor_<T>(u0 + F * a_hi);
    f0 = f0 < (T)0 ? (T)0 : f0; f1 = f1 > W - (T)1 ? W - (T)1 : f1;
    fx0 = f0; fx1 = f1;
    if (f1 < f0) return (T)0;
    const T rs = M<T>::rcp_(M<T>::sqrt_((T)1 + b * b));
    const T s0 = e * rs, rp = M<T>::sqrt_(M<T>::fmax_(s0 * s0 + xc_ * xc_ - k2_, (T)0));
    const T tmin_row = (s0 - rp) * rs;                            // view-axis depth of the row's nearest sphere point
    if (!(tmin_row * best < (T)1 + (T)1e-9)) return (T)0;
    const T xs = u0 + F * M<T>::div_(xc_, tmin_row);
    T xa = floor_<T>(xs); xa = xa < f0 ? f0 : (xa > f1 ? f1 : xa);
    T xb = xa + (T)1; xb = xb > f1 ? f1 : xb;
    const T ia = inv_hit_c((xa - u0) * invF, b, zc_, xc_, yc_, k2_), ib = inv_hit_c((xb - u0) * invF, b, zc_, xc_, yc_, k2_);
    return ia > ib ? ia : ib;
  };
  T A = (T)0, iA = (T)0;
  int y0 = 0, y1 = -1;
  if (duck_ok) {
    A = Rd * Rd - zc * zc - yc * yc;                              // < 0: the sphere is wholly in front of the near plane
    const T den = M<T>::rcp_(Rd * Rd - zc * zc);
    const T sqb = Rd * M<T>::sqrt_(M<T>::fmax_(zc * zc + yc * yc - Rd * Rd, (T)0));
    const T b_lo = (-zc * yc + sqb) * den, b_hi = (-zc * yc - sqb) * den;       // den < 0
    T fy0 = ceil_<T>(v0 + F * b_lo), fy1 = floor_<T>(v0 + F * b_hi);
    fy0 = fy0 < (T)0 ? (T)0 : fy0; fy1 = fy1 > H - (T)1 ? H - (T)1 : fy1;
    iA = M<T>::rcp_(A);
    y0 = (int)fy0; y1 = (work && fy1 >= fy0) ? (int)fy1 : -1;
  }
  if (G == 8) {
    // The rows of ALL sets' masks are dealt to the wave's 64 lanes (a near duck covers 10x the rows of a far one, and the
    // capture step lasts as long as its longest set).  Count, sum x, sum y are sums of integers and half-integers -- exact in
    // floating point, so LDS atomic adds in any order give the same bits; the nearest fragment is a max.
    const bool listed = duck_ok && !straddle && y1 >= y0;
    if (!NOLEAD) {
      if (vsub == 0) {
        T* sc = sconst_all + erow * kSetWords;
        int base = 0;
        if (listed) base = (int)__hip_atomic_fetch_add(lu + 2, (uint32_t)(y1 - y0 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        sc[10] = zc; sc[11] = xc; sc[12] = yc; sc[13] = k2; sc[14] = A; sc[15] = iA; sc[16] = (T)y0;
        sc[17] = listed ? (T)(y1 - y0 + 1) : (T)0; sc[18] = (T)base;
      }
    } else {
      // NOLEAD (the fw_collect_step instantiations): no `if (vsub == 0)` here.  Every lane of the set takes part: the position in
      // the list is claimed by an atomic whose operand is the set's row count in its first lane and 0 in the others, the first
      // lane's result is handed round, and all lanes store the same words.  In those builds the leader-only branch made hipcc 7.2
      // save registers that are live across the row loop at the top of its join block -- ahead of the exec restore, i.e. for the
      // leader only: the hazard tools/check_isa.py guards against (it showed up at this very join in three different builds).
      // The all-lane form costs 2.9 us (ObjLock) / 4.5 us (combined) per step, so the plain step kernels keep the branch.
      T* sc = sconst_all + erow * kSetWords;
      const uint32_t claim = (vsub == 0 && listed) ? (uint32_t)(y1 - y0 + 1) : 0u;
      int base = (int)__hip_atomic_fetch_add(lu + 2, claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      base = __shfl(base, (int)(threadIdx.x & (kWave - 1)) - vsub, kWave);
      sc[10] = zc; sc[11] = xc; sc[12] = yc; sc[13] = k2; sc[14] = A; sc[15] = iA; sc[16] = (T)y0;
      sc[17] = listed ? (T)(y1 - y0 + 1) : (T)0; sc[18] = listed ? (T)base : (T)0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int total_rows = (int)lu[2];
    if (total_rows > 0) {
      const int wl = (int)(threadIdx.x & (kWave - 1));
      T best = (T)0;                                              // this lane's running max within the current set (only steers which rows are evaluated)
      int best_of = -1;
#pragma unroll 1
      for (int i = wl; i < total_rows; i += kWave) {
        int es = 0;
#pragma unroll
        for (int q = 1; q < 8; ++q) {                             // the set whose [base, base + rows) holds item i
          const T* scq = sconst_all + q * kSetWords;
          const int bq = (int)scq[18], nq = (int)scq[17];
          es = (nq > 0 && i >= bq && i < bq + nq) ? q : es;
        }
        const T* sc = sconst_all + es * kSetWords;
        const int y = (int)sc[16] + (i - (int)sc[18]);
        if (es != best_of) { best = (T)0; best_of = es; }
        T fx0, fx1;
        const T it = mask_row(y, sc[10], sc[11], sc[12], sc[13], sc[14], sc[15], best, fx0, fx1);
        if (fx1 >= fx0) {
          const T nn = fx1 - fx0 + (T)1;
          T* da = dacc_all + es * 4;
          (void)__hip_atomic_fetch_add(da + 0, nn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          (void)__hip_atomic_fetch_add(da + 1, nn * (T)0.5 * (fx0 + fx1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          (void)__hip_atomic_fetch_add(da + 2, nn * (T)y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (y == y_mid) { lu[20 + 2 * es] = (uint32_t)(int)fx0; lu[21 + 2 * es] = (uint32_t)(int)fx1; }
          if (it > (T)0) {
            best = it > best ? it : best;
            UB bits;
            __builtin_memcpy(&bits, &it, sizeof(T));
            (void)__hip_atomic_fetch_max(reinterpret_cast<UB*>(da + 3), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (listed) {
      const T* da = dacc_all + erow * 4;
      cnt = da[0]; sx = da[1]; sy = da[2];
      UB bits = reinterpret_cast<const UB*>(da)[3];
      __builtin_memcpy(&itmax, &bits, sizeof(T));
      if (lu[20 + 2 * erow] != 0x7FFFFFFFu) { mid_lo = (int)lu[20 + 2 * erow]; mid_hi = (int)lu[21 + 2 * erow]; }
    }
  }
  if (duck_ok && (G == 1 || straddle)) {
    // one lane (G = 1), or a mask that straddles the far plane (its fragments are tested one by one; it is <= 7 pixels wide there)
#pragma unroll 1
    for (int y = y0 + vsub; y <= y1; y += VG) {
      if (!straddle) {
        T fx0, fx1;
        const T it = mask_row(y, zc, xc, yc, k2, A, iA, itmax, fx0, fx1);
        if (fx1 < fx0) continue;
        const T nn = fx1 - fx0 + (T)1;
        cnt += nn; sx += nn * (T)0.5 * (fx0 + fx1); sy += nn * (T)y;
        if (y == y_mid) { mid_lo = (int)fx0; mid_hi = (int)fx1; }
        itmax = it > itmax ? it : itmax;
      } else {
        const T b = ((T)y - v0) * invF;
        const T e = zc + b * yc, Bh = xc * e, Cq = e * e - ((T)1 + b * b) * k2;
        const T Dd = Bh * Bh - A * Cq;
        if (Dd < (T)0) continue;
        const T sq = M<T>::sqrt_(Dd);
        const T a_lo = (-Bh + sq) * iA, a_hi = (-Bh - sq) * iA;
        T fx0 = ceil_<T>(u0 + F * a_lo), fx1 = floor_<T>(u0 + F * a_hi);
        fx0 = fx0 < (T)0 ? (T)0 : fx0; fx1 = fx1 > W - (T)1 ? W - (T)1 : fx1;
        for (T x = fx0; x <= fx1; x += (T)1) {
          const T it = inv_hit((x - u0) * invF, b);
          if (it > (T)0) { cnt += (T)1; sx += x; sy += (T)y; itmax = it > itmax ? it : itmax; }
        }
      }
    }
    cnt = ssum(cnt); sx = ssum(sx); sy = ssum(sy);
    itmax = -smin(-itmax);
  }
  T visible = (T)0, cxn = (T)0, cyn = (T)0, area = (T)0, depth = (T)0;
  const bool duck_in = cnt > (T)0;
  if (duck_in) {
    visible = (T)1;
    cxn = M<T>::div_(M<T>::div_(sx, cnt), M<T>::fmax_((T)1, W - (T)1));
    cyn = M<T>::div_(M<T>::div_(sy, cnt), M<T>::fmax_((T)1, H - (T)1));
    area = M<T>::div_(cnt, M<T>::fmax_((T)1, H * W));
    depth = depthbuf_to_meters<T>(OC, depthbuf_from_inv<T>(OC, itmax));
  }
  frame[0] = visible; frame[1] = cxn; frame[2] = cyn; frame[3] = area; frame[4] = depth;
  FW_PH(1);                                                       // duck rows + statistics
  // ---- obstacle zones: mean depth-buffer value of the non-duck pixels of each third of row h//2 ----
  // Everything is accumulated as sum of clip(1 / t, 1 / far, 1 / near): the buffer value is c1 (1 - near / t), affine in 1 / t.
  //  * ground / sky: 1 / t = -(g0z + a g1z) / cam_z is LINEAR in the column, so the sum over a run of columns is an arithmetic
  //    series between the two columns where it meets the clip planes -- closed form, no per-pixel work.  A third minus the
  //    duck's columns is at most two runs: six (third, run) pairs, one per lane of the env's group.
  //  * cylinders: the nearest cylinder fragment of every covered pixel goes to an LDS row (max of 1 / t; lane j owns the
  //    pixels x = j mod 8, 4 independent pixels in flight per lane), then every lane adds max(cyl, ground) - ground of its
  //    pixels to the third's sum.  (G = 1: one lane, pixel by pixel against every cylinder.)
  const T inv_camz = cam[2] > (T)0 ? M<T>::rcp_(cam[2]) : (T)0;
  const T gk1 = -g1[2] * invF * inv_camz, gk0 = -g0[2] * inv_camz - u0 * gk1;       // ground: 1 / t at column x = gk0 + gk1 x
  auto git = [&](T x) { T it = gk0 + gk1 * x; it = it > OC.inv_near ? OC.inv_near : it; return it < OC.inv_far ? OC.inv_far : it; };
  // columns where the ground's 1 / t crosses the clip values
  T xf_ = (T)0, xn_ = (T)0;
  const bool sloped = gk1 != (T)0;
  if (sloped) { const T ig = M<T>::rcp_(gk1); xf_ = (OC.inv_far - gk0) * ig; xn_ = (OC.inv_near - gk0) * ig; }
  auto run_sum = [&](int p_, int q_, T& sum, int& cntr) {         // adds sum of clip(1/t) over the columns [p_, q_)
    if (q_ <= p_) return;
    const T p = (T)p_, q = (T)q_;
    cntr += q_ - p_;
    if (!sloped) { sum += (q - p) * git((T)0); return; }
    // linear part [u, w): far-clipped on one side, near-clipped on the other (which side depends on the sign of the slope)
    T u, w, nlo, nhi;                                             // nlo / nhi: clip value below u / from w on
    if (gk1 > (T)0) { u = ceil_<T>(xf_); w = floor_<T>(xn_) + (T)1; nlo = OC.inv_far; nhi = OC.inv_near; }
    else { u = ceil_<T>(xn_); w = floor_<T>(xf_) + (T)1; nlo = OC.inv_near; nhi = OC.inv_far; }
    u = u < p ? p : (u > q ? q : u); w = w < u ? u : (w > q ? q : w);
    const T m = w - u;
    sum += (u - p) * nlo + (q - w) * nhi + gk0 * m + gk1 * ((T)0.5 * m * (u + w - (T)1));
  };
  // (no runtime-indexed private arrays below: thirds are picked with selects)
  auto z_lo = [&](int z) { return z == 0 ? 0 : (z == 1 ? x_1 : x_2); };
  auto z_hi = [&](int z) { return z == 0 ? x_1 : (z == 1 ? x_2 : Wi); };
  const bool exact_duck = duck_in && straddle;                    // duck columns not one interval: pixel by pixel below
  const int dlo = (duck_in && !straddle) ? mid_lo : (1 << 30), dhi = (duck_in && !straddle) ? mid_hi : -1;
  T zsum[3] = { (T)0, (T)0, (T)0 };
  int zcnt[3] = { 0, 0, 0 };
  if (G == 8) {
    T msum = (T)0; int mcnt = 0;
    if (sub < 6) {
      const int z = sub >> 1, r = sub & 1;
      const int za = z_lo(z), zb = z_hi(z);
      // run 0: [za, min(zb, max(dlo, za)));  run 1: [max(za, dhi + 1), zb) if the duck starts before the third's end
      if (r == 0) run_sum(za, dlo < zb ? (dlo > za ? dlo : za) : zb, msum, mcnt);
      else if (dlo < zb) run_sum((dhi + 1) > za ? (dhi + 1) : za, zb, msum, mcnt);
    }
#pragma unroll
    for (int z = 0; z < 3; ++z) {
      zsum[z] = group_sum<G, T>((sub >> 1) == z && sub < 6 ? msum : (T)0);
      zcnt[z] = (int)group_sum<G, float>((sub >> 1) == z && sub < 6 ? (float)mcnt : 0.f);
    }
  } else {
#pragma unroll
    for (int z = 0; z < 3; ++z) {
      const int za = z_lo(z), zb = z_hi(z);
      run_sum(za, dlo < zb ? (dlo > za ? dlo : za) : zb, zsum[z], zcnt[z]);
      if (dlo < zb) run_sum((dhi + 1) > za ? (dhi + 1) : za, zb, zsum[z], zcnt[z]);
    }
  }
  if (exact_duck) {                                               // rare: take the duck's pixels of row h//2 out one by one
    T dsum[3] = { (T)0, (T)0, (T)0 }; int dcnt[3] = { 0, 0, 0 };
    for (int x = work ? vsub : Wi; x < Wi; x += VG) {
      if (inv_hit(((T)x - u0) * invF, bm) > (T)0) {
        const T v = git((T)x);
        dsum[0] += x < x_1 ? v : (T)0; dsum[1] += (x >= x_1 && x < x_2) ? v : (T)0; dsum[2] += x >= x_2 ? v : (T)0;
        dcnt[0] += x < x_1 ? 1 : 0; dcnt[1] += (x >= x_1 && x < x_2) ? 1 : 0; dcnt[2] += x >= x_2 ? 1 : 0;
      }
    }
#pragma unroll
    for (int z = 0; z < 3; ++z) { zsum[z] -= ssum(dsum[z]); zcnt[z] -= (int)ssum((T)dcnt[z]); }
  }
  {
    auto is_duck = [&](int x) {
      if (!duck_in) return false;
      if (!straddle) return x >= mid_lo && x <= mid_hi;
      return inv_hit(((T)x - u0) * invF, bm) > (T)0;
    };
    // 1 / t of a cylinder's fragment in the column of direction g0 + a g1 (0 = none).  t = (-hb - sqrt(disc)) / A and
    // hb^2 - disc = A cc, so 1 / t = (sqrt(disc) - hb) / cc: no division per pixel (cc = |o|^2 - r^2 > 0 is the cylinder's), and no
    // cancellation (hb < 0).  Branch-free: several of these are in flight per lane.
    auto cyl_it = [&](T a, T cc, T icc, T op, T oq, T hh, T pp_, T pq_, T qq_, T g0z, T g1z, T camz) {
      const T A = pp_ + a * ((T)2 * pq_ + a * qq_), hb = op + a * oq, disc = hb * hb - A * cc;
      bool ok = A > (T)0 && disc >= (T)0 && hb < (T)0;
      const T sq = M<T>::sqrt_(disc > (T)0 ? disc : (T)0);
      const T num = -hb - sq;
      const T zA = camz * A + num * (g0z + a * g1z);               // z of the hit times A
      ok = ok && num > (T)0 && zA >= (T)0 && zA <= hh * A;
      return ok ? (sq - hb) * icc : (T)0;
    };
    // what a pixel whose nearest cylinder fragment is at 1 / t = c adds to its third's sum of clip(1 / t): the ground value g
    // stays underneath (the pixel shows max(cylinder, ground))
    auto over_ground = [&](T c, T g) { const T hi = c > OC.inv_near ? OC.inv_near : c; return (hi > g ? hi : g) - g; };
    T csum[3] = { (T)0, (T)0, (T)0 };
    FW_PH(2);                                                     // ground runs
    if (cyl_on) {
      // (1) my set clears the covered columns of its row buffer
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const bool drawn = work && vis != 0u;                       // set-uniform
      const int xmin = drawn ? (int)lu[4 + 2 * erow] : 1, xmax = drawn ? (int)lu[5 + 2 * erow] : 0;
#pragma unroll 4
      for (int x = xmin + vsub; x <= xmax; x += VG) zr[x] = (UB)0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // (2) the WAVE draws all sets' intervals: slice s of the list goes to 8-lane group s mod 8, 4 pixels per lane in flight;
      // the nearest fragment of a pixel is a max of 1 / t, kept with an LDS atomic on the ordered bit pattern, so who draws
      // what -- and in which order -- does not change a bit of the result
      const uint32_t total = lu[0];
      const int grp = (int)(threadIdx.x & (kWave - 1)) >> 3, s8 = (int)threadIdx.x & 7;
#pragma unroll 1
      for (uint32_t si = (uint32_t)grp; si < total; si += 8u) {
        const uint32_t ent = slist[si];
        const int es = (int)(ent >> 10), eo = (int)(ent >> 5) & 31, ei = (int)ent & 31;
        const T* e = ctab_all + ((size_t)es * FW_MAX_OBSTACLES + eo) * kCtabWords;      // same address in the 8 lanes: LDS broadcast
        const T* sc = sconst_all + es * kSetWords;
        const T cc = e[0], hh = e[1], op = e[2], oq = e[3], icc = e[6];
        const int xhi = (int)e[5];
        const int x0 = (int)e[4] + ei * kSliceCols + s8;
        const T c0 = sc[0], c1 = sc[1], c2 = sc[2], c3 = sc[3], c4 = sc[4], c5 = sc[5];
        UB* zrow = zr_all + (size_t)es * OC.zrow_stride;
        T itu[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int xx = x0 + 8 * u;
          itu[u] = xx <= xhi ? cyl_it(((T)xx - u0) * invF, cc, icc, op, oq, hh, c0, c1, c2, c3, c4, c5) : (T)0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int xx = x0 + 8 * u;
          if (itu[u] > (T)0) {
            UB bits;
            __builtin_memcpy(&bits, &itu[u], sizeof(T));
            (void)__hip_atomic_fetch_max(zrow + xx, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      FW_PH(3);                                                   // clear + draw
      // (3) the row sums, again by the wave: the covered columns of a set, cut at the thirds, in chunks of 32 columns; chunk c of
      // the list goes to 8-lane group c mod 8 (4 pixels per lane), its sum to a slot that belongs to (set, third, chunk), and
      // the set's lanes 0..2 add the slots of their third in chunk order -- so a sum is built the same way whoever computed
      // its parts.  (A duck that straddles the far plane is not one run of columns: such a set sums its row itself.)
      const bool by_wave = drawn && !exact_duck;
      const int my_lo = vsub < 3 ? (xmin > z_lo(vsub) ? xmin : z_lo(vsub)) : 1, my_hi = vsub < 3 ? (xmax < z_hi(vsub) - 1 ? xmax : z_hi(vsub) - 1) : 0;
      const int my_nch = (by_wave && vsub < 3 && my_hi >= my_lo) ? (my_hi - my_lo) / kSliceCols + 1 : 0;
      if (my_nch > 0) {
        const uint32_t pos = __hip_atomic_fetch_add(lu + 1, (uint32_t)my_nch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (int i = 0; i < my_nch; ++i) clist[pos + i] = (uint16_t)((erow << 8) | (vsub << 6) | i);
      }
      if (by_wave && vsub == 0) {
        T* sc = sconst_all + erow * kSetWords;
        sc[6] = gk0; sc[7] = gk1; sc[8] = (T)dlo; sc[9] = (T)dhi;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const uint32_t totalc = lu[1];
#pragma unroll 1
      for (uint32_t ci = (uint32_t)grp; ci < totalc; ci += 8u) {
        const uint32_t ent = clist[ci];
        const int es = (int)(ent >> 8), ez = (int)(ent >> 6) & 3, ei = (int)ent & 63;
        const T* sc = sconst_all + es * kSetWords;
        const T k0 = sc[6], k1 = sc[7];
        const int e_dlo = (int)sc[8], e_dhi = (int)sc[9];
        const int e_min = (int)lu[4 + 2 * es], e_max = (int)lu[5 + 2 * es];
        const int lo = e_min > z_lo(ez) ? e_min : z_lo(ez), hi = e_max < z_hi(ez) - 1 ? e_max : z_hi(ez) - 1;
        const UB* zrow = zr_all + (size_t)es * OC.zrow_stride;
        const int x0 = lo + ei * kSliceCols + s8;
        T v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int xx = x0 + 8 * u;
          const UB bits = xx <= hi ? zrow[xx] : (UB)0;
          __builtin_memcpy(&v[u], &bits, sizeof(T));
        }
        T acc = (T)0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int xx = x0 + 8 * u;
          T g = k0 + k1 * (T)xx;
          g = g > OC.inv_near ? OC.inv_near : g; g = g < OC.inv_far ? OC.inv_far : g;
          const bool on = v[u] > (T)0 && !(xx >= e_dlo && xx <= e_dhi);
          acc += on ? over_ground(v[u], g) : (T)0;
        }
        acc = group_sum<G, T>(acc);
        if (s8 == 0) cpart_all[(es * 3 + ez) * cpz + ei] = acc;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (by_wave) {
        T zs = (T)0;
        for (int i = 0; i < my_nch; ++i) zs += cpart_all[(erow * 3 + vsub) * cpz + i];
        const int first = (int)(threadIdx.x & (kWave - 1)) & ~(VG - 1);
#pragma unroll
        for (int z = 0; z < 3; ++z) zsum[z] += __shfl(zs, first + z, kWave);
      } else if (drawn) {
#pragma unroll 1
        for (int x = xmin + vsub; x <= xmax; x += VG) {
          UB bits = zr[x];
          T v;
          __builtin_memcpy(&v, &bits, sizeof(T));
          if (v > (T)0 && !is_duck(x)) {
            const T d = over_ground(v, git((T)x));
            csum[0] += x < x_1 ? d : (T)0; csum[1] += (x >= x_1 && x < x_2) ? d : (T)0; csum[2] += x >= x_2 ? d : (T)0;
          }
        }
#pragma unroll
        for (int z = 0; z < 3; ++z) zsum[z] += ssum(csum[z]);
      }
    } else if (nob > 0) {
      const T r2 = OC.obst_radius * OC.obst_radius;
#pragma unroll 1
      for (int x = 0; x < Wi; ++x) {
        if (is_duck(x)) continue;
        const T a = ((T)x - u0) * invF;
        T c = (T)0;
        for (int o = 0; o < nob; ++o) {
          const T cx = ob[(3 * o) * n], cy = ob[(3 * o + 1) * n], hh = ob[(3 * o + 2) * n];
          const T ox = cam[0] - cx, oy = cam[1] - cy;
          const T cc = ox * ox + oy * oy - r2;
          const T ic = cc > (T)0 ? cyl_it(a, cc, M<T>::rcp_(cc), ox * g0[0] + oy * g0[1], ox * g1[0] + oy * g1[1], hh, pp, pq, qq, g0[2], g1[2], cam[2]) : (T)0;
          c = ic > c ? ic : c;
        }
        if (c > (T)0) {
          const T d = over_ground(c, git((T)x));
          csum[0] += x < x_1 ? d : (T)0; csum[1] += (x >= x_1 && x < x_2) ? d : (T)0; csum[2] += x >= x_2 ? d : (T)0;
        }
      }
#pragma unroll
      for (int z = 0; z < 3; ++z) zsum[z] += csum[z];
    }
  }
#pragma unroll
  for (int z = 0; z < 3; ++z) {
    // mean buffer value = c1 (1 - near * mean(1 / t));  float(np.mean(vals)) :718, then metres unless the mean is 0 :725-727
    const T mean = zcnt[z] > 0 ? OC.db_c1 * ((T)1 - OC.near_ * M<T>::div_(zsum[z], (T)zcnt[z])) : (T)0;
    frame[5 + z] = mean > (T)1e-12 ? depthbuf_to_meters<T>(OC, mean) : (T)0;       // (guard band of the `> 0.0` test: see the oracle)
  }
  FW_PH(4);                                                       // row sums + zone means
#undef FW_PH
}

// A capture step of the WAVE.  G = 8: the envs whose camera is due hand their pose to sets of 8 * k lanes, k = 8 / 4 / 2 / 2 / 1
// for 1 / 2 / 3 / 4 / 5+ envs due (only a few of the wave's 8 envs capture at the same sub-step, and the launch lasts as long as
// its slowest wave): set j = lanes [j VG, (j + 1) VG) works for the j-th due env, whichever env's state those lanes hold
// themselves; the frame travels back by shuffle.  Must be called by all 64 lanes (wave-uniform control flow).
template <typename T, int G, bool NOLEAD = false>
__device__ __forceinline__ void obj_capture_wave(const ObjC<T>& OC, const DevState<T>& D, bool due, int env, ObjState<T>& O,
                                                 const Rigid<T>& S, const T R[9]) {
  T fr[8];
  if (G != 8) {
    if (due) {
      capture_body<T, G, NOLEAD>(OC, D, env, O.duck, O.nob, S.p, R, 0, 1, 0, true, fr);
#pragma unroll
      for (int i = 0; i < 8; ++i) O.frame[i] = fr[i];
      O.frame_has = (T)1;
    }
    return;
  }
  const unsigned long long db = __ballot(due);
  if (db == 0ull) return;
  uint32_t dmask = 0u;                                  // bit g: the env of lanes 8 g .. 8 g + 7 is due (wave-uniform)
#pragma unroll
  for (int g = 0; g < 8; ++g) dmask |= (uint32_t)((db >> (8 * g)) & 1ull) << g;
  const int m = __popc(dmask);
  const int k = m == 1 ? 8 : (m == 2 ? 4 : (m <= 4 ? 2 : 1));
  const int VG = 8 * k;
  const int lane = (int)threadIdx.x & (kWave - 1), sub = lane & 7, grp = lane >> 3;
  const int j = lane / VG;
  const bool work = j < m;
  int og = 0;                                           // group of the j-th due env
  {
    uint32_t mm = dmask;
#pragma unroll
    for (int i = 0; i < 7; ++i) mm = (i < (work ? j : 0)) ? (mm & (mm - 1u)) : mm;
    og = __ffs((int)mm) - 1;
  }
  const int src = og * 8 + sub;
  T oduck[3], oSp[3], oR[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) { oduck[i] = __shfl(O.duck[i], src, kWave); oSp[i] = __shfl(S.p[i], src, kWave); }
#pragma unroll
  for (int i = 0; i < 9; ++i) oR[i] = __shfl(R[i], src, kWave);
  const int onob = __shfl(O.nob, src, kWave), oenv = __shfl(env, src, kWave);
#ifdef FW_PROFILE
  capture_body<T, G, NOLEAD>(OC, D, oenv, oduck, onob, oSp, oR, lane & (VG - 1), VG, j, work, fr, O.p_ph, m >= 5);   // (always the array itself: a pointer that may be null keeps it from living in registers)
  if (m >= 5) O.p_ph[5] += 1;
#else
  capture_body<T, G, NOLEAD>(OC, D, oenv, oduck, onob, oSp, oR, lane & (VG - 1), VG, j, work, fr);
#endif
  const int back = __popc(dmask & ((1u << grp) - 1u)) * VG + sub;      // a lane of the set that worked for my env
#pragma unroll
  for (int i = 0; i < 8; ++i) { const T v = __shfl(fr[i], back, kWave); if (due) O.frame[i] = v; }
  if (due) O.frame_has = (T)1;
}

// compute_state (:253-287) side effects: _compute_vision_features (:643-689) on the latest frame,
// then the history shift of _build_duck_vision_observation (:421-442)
template <typename T>
__device__ __forceinline__ void obj_compute_state(ObjState<T>& O) {
  T visible = (T)0, dl = (T)0, dc = (T)0, dr = (T)0;
  if (O.frame_has != (T)0) {
    dl = O.frame[5]; dc = O.frame[6]; dr = O.frame[7];
    if (O.frame[0] == (T)0) {
      O.since_seen = O.since_seen + (T)1 < (T)60 ? O.since_seen + (T)1 : (T)60;
    } else {
      O.last_cx = O.frame[1]; O.last_cy = O.frame[2]; O.last_area = O.frame[3]; O.last_depth = O.frame[4];
      O.since_seen = (T)0; visible = (T)1;
    }
  }
#pragma unroll
  for (int k = kHist - 1; k >= FW_VISION_FEATS; --k) O.hist[k] = O.hist[k - FW_VISION_FEATS];
  O.hist[0] = (float)visible; O.hist[1] = (float)O.last_cx; O.hist[2] = (float)O.last_cy; O.hist[3] = (float)O.last_area;
  O.hist[4] = (float)O.last_depth; O.hist[5] = (float)(O.since_seen / (T)60);
  O.hist[6] = (float)dl; O.hist[7] = (float)dc; O.hist[8] = (float)dr;
  O.filled = O.filled + (T)1 < (T)FW_VISION_HIST ? O.filled + (T)1 : (T)FW_VISION_HIST;
}

// combined task compute_state (:248-274): current feature vector (kept in hist[0..8]) + phase switching
template <typename T>
__device__ __forceinline__ void comb_compute_state(const ObjC<T>& OC, ObjState<T>& O, bool all_reached) {
  T visible = (T)0, dl = (T)0, dc = (T)0, dr = (T)0;
  if (O.frame_has != (T)0) {
    dl = O.frame[5]; dc = O.frame[6]; dr = O.frame[7];
    if (O.frame[0] == (T)0) {
      O.since_seen = O.since_seen + (T)1 < (T)60 ? O.since_seen + (T)1 : (T)60;
    } else {
      O.last_cx = O.frame[1]; O.last_cy = O.frame[2]; O.last_area = O.frame[3]; O.last_depth = O.frame[4];
      O.since_seen = (T)0; visible = (T)1;
    }
  }
  O.cur_z[0] = (float)dl; O.cur_z[1] = (float)dc; O.cur_z[2] = (float)dr;
  if (all_reached) {
    O.phase |= 2;
    if (!(O.phase & 1)) {
      const bool vis = visible > (T)0.5 && O.last_area >= OC.switch_min_area;
      O.seen_consec = vis ? O.seen_consec + (T)1 : (T)0;
      if (O.seen_consec >= (T)OC.switch_min_seen) O.phase |= 1;
    }
  } else {
    O.phase = 0;
  }
}

// obstacle penalty of the combined task (:347-380): full scale in the waypoint phase, half in the duck phase
template <typename T>
__device__ __forceinline__ void comb_obstacle_penalty(const ObjC<T>& OC, const ObjState<T>& O, T mult, T& rew) {
  T d_obs = (T)1e300; bool any = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) { T d = (T)O.cur_z[k]; if (d > (T)0 && d < (T)3.0e38) { any = true; d_obs = d < d_obs ? d : d_obs; } }
  if (any && OC.safe_dist > (T)0 && d_obs < OC.safe_dist) {
    T pen = M<T>::div_(OC.avoid_scale * mult * (OC.safe_dist - d_obs), OC.safe_dist);
    rew -= pen < OC.avoid_max ? pen : OC.avoid_max;
  }
}

// duck phase reward (:310-343); returns true on a strike
template <typename T>
__device__ __forceinline__ bool comb_duck_reward(const ObjC<T>& OC, int sparse, ObjState<T>& O, T& rew) {
  if (!(O.phase & 1)) return false;
  const T est = O.last_depth;
  if (!sparse && est > (T)0) rew += M<T>::rcp_(M<T>::fmax_(est, (T)2));
  if (O.last_cx > (T)0) {
    T dc = M<T>::sqrt_((O.last_cx - (T)0.5) * (O.last_cx - (T)0.5) + (O.last_cy - (T)0.5) * (O.last_cy - (T)0.5));
    if (dc < (T)0.35) { O.lock_steps += (T)1; rew += OC.lock_step_reward; } else O.lock_steps = (T)0;
  } else {
    O.lock_steps = (T)0;
  }
  if (O.prev_est >= (T)0 && est > (T)0) {
    T diff = O.prev_est - est;
    if (diff > (T)0) rew += diff * OC.k_approach;
  }
  O.prev_est = est;
  if (O.lock_steps >= (T)OC.hold_steps && est > (T)0 && est <= OC.strike_dist) { rew += OC.strike_reward; return true; }
  return false;
}

// combined task observation (local FlattenWaypointEnv over [remaining waypoints ..., duck], float64)
template <typename T, typename W>
__device__ __forceinline__ void comb_write_obs(const Params<T>& P, const DevState<T>& D, int env, const ObjState<T>& O,
                                               const Rigid<T>& S, const T action[4], int tgt_idx, W&& put) {
  T R[9];
  rot_from_quat(S.q, R);
  T ang_vel[3], lin_vel[3], eul[3];
  mtv(R, S.w, ang_vel);
  mtv(R, S.v, lin_vel);
  bool lock = euler_from_quat(S.q, eul);
  T qrt[4] = { S.q[0], S.q[1], S.q[2], S.q[3] };
  if (lock || P.angle_repr == 1) { quat_from_euler(eul, qrt); rot_from_quat(qrt, R); }
  int o = 0;
  put(o++, ang_vel[0]); put(o++, ang_vel[1]); put(o++, ang_vel[2]);
  if (P.angle_repr == 0) { put(o++, eul[0]); put(o++, eul[1]); put(o++, eul[2]); }
  else { put(o++, qrt[0]); put(o++, qrt[1]); put(o++, qrt[2]); put(o++, qrt[3]); }
  put(o++, lin_vel[0]); put(o++, lin_vel[1]); put(o++, lin_vel[2]);
  put(o++, S.p[0]); put(o++, S.p[1]); put(o++, S.p[2]);
  put(o++, action[0]); put(o++, action[1]); put(o++, action[2]); put(o++, action[3]);
#pragma unroll
  for (int k = 0; k < FW_NUM_ACTUATORS; ++k) put(o++, S.act[k]);
  for (int i = 0; i < P.ctx; ++i) {
    const int t = tgt_idx + i;
    T d[3] = {(T)0, (T)0, (T)0}, b[3] = {(T)0, (T)0, (T)0};
    if (t < P.num_targets) {
      const T* tp = D.r + (size_t)(RF_TARGETS + 3 * t) * D.npad + env;
      d[0] = tp[0] - S.p[0]; d[1] = tp[D.npad] - S.p[1]; d[2] = tp[2 * (size_t)D.npad] - S.p[2];
      mtv(R, d, b);
    } else if (t == P.num_targets) {          // the duck is the row after the last remaining waypoint (:234-246)
      d[0] = O.duck[0] - S.p[0]; d[1] = O.duck[1] - S.p[1]; d[2] = O.duck[2] - S.p[2];
      mtv(R, d, b);
    }
    put(o++, b[0]); put(o++, b[1]); put(o++, b[2]);
  }
}

// compute_term_trunc_reward (:296-372) after the base checks; returns true on a strike
template <typename T>
__device__ __forceinline__ bool obj_reward(const ObjC<T>& OC, int sparse, ObjState<T>& O, T dist_to_duck, T& rew) {
  const float* vis = O.hist;                      // newest feature vector
  {  // _apply_obstacle_avoidance_reward :376-407
    T d_obs = (T)1e300; bool any = false;
#pragma unroll
    for (int k = 6; k < 9; ++k) { T d = (T)vis[k]; if (d > (T)0 && d < (T)3.0e38) { any = true; d_obs = d < d_obs ? d : d_obs; } }
    if (any && OC.safe_dist > (T)0 && d_obs < OC.safe_dist) {
      T pen = M<T>::div_(OC.avoid_scale * (T)0.5 * (OC.safe_dist - d_obs), OC.safe_dist);
      rew -= pen < OC.avoid_max ? pen : OC.avoid_max;
    }
  }
  if (!sparse) {
    rew += M<T>::div_(OC.k_dist, M<T>::fmax_(dist_to_duck, (T)2));
    if (vis[0] > 0.5f) {
      T cx = (T)vis[1], cy = (T)vis[2], area = (T)vis[3], est = (T)vis[4];
      rew += OC.k_visible;
      rew += OC.k_area * M<T>::fmax_((T)0, area);
      T dcen = M<T>::sqrt_((cx - (T)0.5) * (cx - (T)0.5) + (cy - (T)0.5) * (cy - (T)0.5));
      T r_lock = M<T>::fmax_(OC.lock_radius, (T)1e-6);
      rew += OC.k_center * M<T>::fmax_((T)0, M<T>::div_(r_lock - dcen, r_lock));
      if (dcen < r_lock) {
        O.lock_steps = O.lock_steps + (T)1 < (T)OC.hold_steps ? O.lock_steps + (T)1 : (T)OC.hold_steps;
        rew += OC.lock_step_reward;
      } else {
        O.lock_steps = M<T>::fmax_(O.lock_steps - (T)OC.decay_steps, (T)0);
      }
      if (O.prev_est >= (T)0 && est > (T)0) {
        T diff = O.prev_est - est;
        if (OC.approach_clip > (T)0) { diff = diff > OC.approach_clip ? OC.approach_clip : diff; diff = diff < -OC.approach_clip ? -OC.approach_clip : diff; }
        rew += diff * OC.k_approach;
      }
      O.prev_est = est > (T)0 ? est : (T)-1;
    } else {
      if (O.lock_steps > (T)0) rew -= OC.lost_penalty;
      O.lock_steps = M<T>::fmax_(O.lock_steps - (T)OC.decay_steps, (T)0);
      O.prev_est = (T)-1;
    }
  }
  if (O.lock_steps >= (T)OC.hold_steps && dist_to_duck <= OC.strike_dist) { rew += OC.strike_reward; return true; }
  return false;
}

// FlattenObjLockEnv (envs/flatten_objlock_env.py:41-46): attitude ++ target_vector ++ duck_vision, all float32
template <typename T, typename W>
__device__ __forceinline__ void obj_write_obs(const Params<T>& P, const ObjState<T>& O, const Rigid<T>& S, const T action[4], W&& put) {
  T R[9];
  rot_from_quat(S.q, R);
  T ang_vel[3], lin_vel[3], eul[3];
  mtv(R, S.w, ang_vel);
  mtv(R, S.v, lin_vel);
  bool lock = euler_from_quat(S.q, eul);
  T qrt[4] = { S.q[0], S.q[1], S.q[2], S.q[3] };
  if (lock || P.angle_repr == 1) { quat_from_euler(eul, qrt); rot_from_quat(qrt, R); }
  int o = 0;
  put(o++, f32r(ang_vel[0])); put(o++, f32r(ang_vel[1])); put(o++, f32r(ang_vel[2]));
  if (P.angle_repr == 0) { put(o++, f32r(eul[0])); put(o++, f32r(eul[1])); put(o++, f32r(eul[2])); }
  else { put(o++, f32r(qrt[0])); put(o++, f32r(qrt[1])); put(o++, f32r(qrt[2])); put(o++, f32r(qrt[3])); }
  put(o++, f32r(lin_vel[0])); put(o++, f32r(lin_vel[1])); put(o++, f32r(lin_vel[2]));
  put(o++, f32r(S.p[0])); put(o++, f32r(S.p[1])); put(o++, f32r(S.p[2]));
#pragma unroll
  for (int k = 0; k < 4; ++k) put(o++, f32r(action[k]));
#pragma unroll
  for (int k = 0; k < FW_NUM_ACTUATORS; ++k) put(o++, f32r(S.act[k]));
  T d[3] = { O.duck[0] - S.p[0], O.duck[1] - S.p[1], O.duck[2] - S.p[2] }, tv[3];
  mtv(R, d, tv);
  put(o++, f32r(tv[0])); put(o++, f32r(tv[1])); put(o++, f32r(tv[2]));
#pragma unroll
  for (int k = 0; k < kHist; ++k) put(o++, (T)O.hist[k]);
  if (o + 4 > P.obs_dim) return;                       // duck_vision_use_deltas=False (:440-441): the row ends with the history
  const bool both = O.filled >= (T)2 && O.hist[0] > 0.5f && O.hist[FW_VISION_FEATS] > 0.5f;
#pragma unroll
  for (int k = 0; k < 4; ++k) put(o++, both ? (T)(O.hist[1 + k] - O.hist[FW_VISION_FEATS + 1 + k]) : (T)0);
}

// Aviary.step(): ticks_per_aviary ticks; returns any-contact.  (z0, z1) are the two ticks'
// motor-noise normals (zero for warm-up lanes: their throttle is exactly 0).  OBJ adds the
// duck / cylinder contacts per tick; the camera capture every physics_camera_ratio ticks
// (envs/fixedwing_objlock_env.py:631-641) is the caller's next call: obj_capture_step, by the whole wave.
template <typename T, bool WIND, int G, bool OBJ, typename SC>
__device__ __forceinline__ bool aviary_step(const Params<T>& P, const TickC<T>& C, const ObjC<T>& OC, const DevState<T>& D,
                                            int env, ObjState<T>& O, Rigid<T>& S, T R[9], const T cmd[FW_NUM_ACTUATORS],
                                            int32_t& tick, T z0, T z1, const T wb[3], const T wa[3], T gust[2],
                                            SC& mine, T wmask, LaneAct<T>& LA) {
  bool contact = false;
#pragma unroll 1
  for (int t = 0; t < P.ticks_per_aviary; ++t) {
    T wind[3] = {(T)0, (T)0, (T)0};
    if (WIND) wind_from_phase<T>(P, wb, wa, gust, wind);
    contact |= physics_tick<T, WIND, G>(P, C, S, R, cmd, (t & 1) ? z1 : z0, wind, mine, wmask, LA);
    if (OBJ) contact |= obj_contacts<T, G>(P, C, OC, D, env, O, S, R);
    tick += 1;
    if (WIND) gust_advance<T>(P, gust);
  }
  return contact;
}

// the capture that follows an Aviary step (`stepped`: this lane's env ran one); all lanes of the wave call it
template <typename T, int G, bool NOLEAD = false>
__device__ __forceinline__ void obj_capture_step(const ObjC<T>& OC, const DevState<T>& D, bool stepped, int env, ObjState<T>& O,
                                                 const Rigid<T>& S, const T R[9], int32_t tick) {
  const bool due = stepped && OC.camera_ratio_ticks > 0 && (tick % OC.camera_ratio_ticks) == 0;
#ifdef FW_PROFILE
  const long long c0 = (long long)__builtin_readcyclecounter();
#endif
  obj_capture_wave<T, G, NOLEAD>(OC, D, due, env, O, S, R);
#ifdef FW_PROFILE
  const long long c1 = (long long)__builtin_readcyclecounter();
  if (due) { O.p_cap += c1 - c0; O.p_ncap += 1; }
  if (G == 8) {
    const unsigned long long db = __ballot(due);
    int m = 0;
    for (int g = 0; g < 8; ++g) m += (int)((db >> (8 * g)) & 1ull);
    if (m > 0) { const int b = m == 1 ? 0 : (m == 2 ? 1 : (m <= 4 ? 2 : 3)); const long long add = (c1 - c0) + (1ll << 40);
      O.p_capm[0] += b == 0 ? add : 0; O.p_capm[1] += b == 1 ? add : 0; O.p_capm[2] += b == 2 ? add : 0; O.p_capm[3] += b == 3 ? add : 0; }
  }
#endif
}

// ------------------------------------------------------------------------------------------
// The capture wave (8-lane mapping, fw_step_kernel_obj_g8h; opt-in, FWSIM_CAPTURE_WAVE=1).  A step wave runs physics, camera and
// task logic one after the other.  Here the workgroup has a SECOND wave on another SIMD of the same CU that does the captures (and
// the tile's shadow work: with one wave per SIMD, step waves + capture waves are all 1024 SIMDs of the chip at 4096 envs): the step wave posts the poses of the envs that are due (LDS mailbox, two slots), goes on with the NEXT
// sub-step's physics, and only then collects the frame and runs the half of the task logic that reads it (fwsim.hip, step_body
// HELP).  Nothing is speculated: the only frame-dependent way a sub-step can end the agent step is a strike, and whether a strike
// is possible at all is known without the frame (lock counter one short of the hold, close enough); where it is -- or where a
// lane's agent step ends anyway -- the wave collects at once, as the one-wave kernel always does.  Same arithmetic in the same
// order as obj_capture_wave / the one-wave loop; the two kernels agree to rounding (1e-11 over thousands of steps: the compiler
// contracts FMAs differently at the two inlining sites), not bit for bit.
// Both waves of a workgroup are resident together (a workgroup is placed on its CU as a whole), so the waits below cannot
// starve; they are bounded all the same (a protocol error ends in FW_CTR_HELPER_TIMEOUTS, not in a hung queue).
// ------------------------------------------------------------------------------------------
constexpr int kMboxReqWords = 16;            // per env row: position [3], rotation [9], duck [3], (obstacle count | due << 16)
#ifndef FW_MBOX_SPIN_LOG2
#define FW_MBOX_SPIN_LOG2 22
#endif
constexpr uint32_t kMboxSpin = 1u << FW_MBOX_SPIN_LOG2;     // polls (each >= 64 cycles of s_sleep) before a wait gives up
template <typename T> struct Mbox {
  uint32_t* ctl;     // [0] requests posted  [1] requests served  [2] step wave is done  [3] (spare)
  T* req;            // [2 slots][8 rows][kMboxReqWords]
  T* resp;           // [2 slots][8 rows][8]
};
__host__ __device__ inline size_t mbox_bytes(size_t word) { return 16 + word * 2 * 8 * (size_t)(kMboxReqWords + 8); }
template <typename T> __device__ __forceinline__ Mbox<T> mbox_at(int off) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  Mbox<T> M;
  M.ctl = reinterpret_cast<uint32_t*>(smem_raw + off);
  M.req = reinterpret_cast<T*>(smem_raw + off + 16);
  M.resp = M.req + 2 * 8 * kMboxReqWords;
  return M;
}

// step wave: hand this sub-step's due envs to the capture wave.  Returns the request's sequence number, 0 if nobody was due.
// Wave-uniform call (all 64 lanes).
template <typename T>
__device__ __forceinline__ uint32_t cap_post(const Mbox<T>& M, uint32_t& n_posted, bool due, int row, bool leader,
                                             const ObjState<T>& O, const Rigid<T>& S, const T R[9]) {
  if (__ballot(due) == 0ull) return 0u;
  const uint32_t seq = ++n_posted;
  if (leader) {
    T* q = M.req + ((seq & 1u) * 8 + row) * kMboxReqWords;
    if (due) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { q[i] = S.p[i]; q[12 + i] = O.duck[i]; }
#pragma unroll
      for (int i = 0; i < 9; ++i) q[3 + i] = R[i];
    }
    *reinterpret_cast<int32_t*>(q + 15) = (O.nob & 0xFFFF) | (due ? 0x10000 : 0);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & (kWave - 1)) == 0) __hip_atomic_store(M.ctl + 0, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  return seq;
}

// step wave: wait for request `seq` (0 = there was none) and take the frames of the lanes that were due in it.  Wave-uniform call.
template <typename T>
__device__ __forceinline__ void cap_collect(const Mbox<T>& M, uint32_t seq, bool due, int row, ObjState<T>& O, unsigned long long* stats) {
  if (seq == 0u) return;
  uint32_t spins = 0;
  while (__hip_atomic_load(M.ctl + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < seq) {
    if (++spins > kMboxSpin) { if ((threadIdx.x & (kWave - 1)) == 0) stat_add(stats, FW_CTR_HELPER_TIMEOUTS); break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (due) {
    const T* r = M.resp + ((seq & 1u) * 8 + row) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) O.frame[i] = r[i];
    O.frame_has = (T)1;
  }
}

// the capture wave: one request (number `want`, which has been posted)
template <typename T>
__device__ __forceinline__ void capture_serve_one(const Mbox<T>& M, uint32_t want, const ObjC<T>& OC, const DevState<T>& D, int env0
#ifdef FW_PROFILE
                                                  , long long* p_busy, int* p_nreq     // dev-only: cycles in captures, requests served
#endif
                                                  ) {
#ifdef FW_PROFILE
  const long long c0 = (long long)__builtin_readcyclecounter();
#endif
  const int lane = (int)threadIdx.x & (kWave - 1), sub = lane & 7, row = lane >> 3;
  const T* q = M.req + ((want & 1u) * 8 + row) * kMboxReqWords;
  const int32_t word = *reinterpret_cast<const int32_t*>(q + 15);
  const bool due = (word & 0x10000) != 0;
  ObjState<T> O;
  Rigid<T> S;
  T R[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) { S.p[i] = due ? q[i] : (T)0; O.duck[i] = due ? q[12 + i] : (T)0; }
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = due ? q[3 + i] : (T)0;
  O.nob = word & 0xFFFF;
#pragma unroll
  for (int i = 0; i < 8; ++i) O.frame[i] = (T)0;
  O.frame_has = (T)0;
  obj_capture_wave<T, 8>(OC, D, due, env0 + row, O, S, R);
  if (due && sub == 0) {
    T* r = M.resp + ((want & 1u) * 8 + row) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = O.frame[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(M.ctl + 1, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef FW_PROFILE
  *p_busy += (long long)__builtin_readcyclecounter() - c0; *p_nreq += 1;
#endif
}
#ifdef FW_PROFILE
#define FW_CAP_PROF_ARGS , p_busy, p_nreq
#define FW_CAP_PROF_PARAMS , long long* p_busy, int* p_nreq
#else
#define FW_CAP_PROF_ARGS
#define FW_CAP_PROF_PARAMS
#endif

// ... and in order until the step wave says it is done
template <typename T>
__device__ __forceinline__ void capture_server(const Mbox<T>& M, uint32_t served, const ObjC<T>& OC, const DevState<T>& D, int env0 FW_CAP_PROF_PARAMS) {
  for (;;) {
    const uint32_t want = served + 1u;
    uint32_t spins = 0;
    bool go = false;
    for (;;) {
      if (__hip_atomic_load(M.ctl + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) { go = true; break; }
      if (__hip_atomic_load(M.ctl + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) break;   // (set after the last collect: nothing is outstanding)
      if (++spins > (kMboxSpin << 2)) break;
      __builtin_amdgcn_s_sleep(2);
    }
    if (!go) return;
    capture_serve_one<T>(M, want, OC, D, env0 FW_CAP_PROF_ARGS);
    served = want;
  }
}

}  // namespace fwsim
