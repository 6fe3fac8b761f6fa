/*
 * fwsim.h -- C ABI of the MI355X-native vectorised fixed-wing env step.
 *
 * This is the drop-in boundary for the hot path named by BASELINE.json:north_star:
 * N independent fixed-wing aircraft advanced in lockstep for one *agent step*
 * (= 4 Aviary steps = 8 physics ticks @240 Hz, with observation, reward,
 * termination/truncation and SB3-style auto-reset fused in).
 *
 * The reference (WdBlink/pyflyt-drone) has no FFI: its boundary is Python
 * duck typing (Gymnasium Env aggregated by SB3's VecEnv).  Each entry point
 * below names the reference interface it replaces (paths relative to the
 * reference repo root):
 *
 *   fw_create   <- gym.make(...)/FixedwingBaseEnv.__init__ x N inside SubprocVecEnv
 *                  envs/fixedwing_envs/fixedwing_base_env.py:21-106,
 *                  train/train_Fixedwing_Waypoints_v3.py:82-121,251
 *   fw_reset    <- Env.reset(): begin_reset/end_reset
 *                  envs/fixedwing_envs/fixedwing_base_env.py:193-257,
 *                  envs/fixedwing_objlock_env.py:177-251
 *   fw_step     <- Env.step(action) + the VecEnv worker's auto-reset
 *                  envs/fixedwing_envs/fixedwing_base_env.py:314-348
 *   fw_seed     <- VecEnv.seed(seed) / env.reset(seed=seed+rank)
 *                  train/train_Fixedwing_Waypoints_v3.py:119
 *   fw_get_state/fw_set_state <- (no reference twin) parity tests + checkpoints
 *   fw_get_counters <- (no reference twin) diagnostics of the auto-reset hand-off
 *   fw_render   <- Camera.capture_image() as consumed at envs/fixedwing_objlock_env.py:603-622 and replaced by a
 *                  network's mask at envs/fixedwing_envs/objlock_yolo_env.py:646-716
 *   fw_destroy  <- Env.close()  envs/fixedwing_envs/fixedwing_base_env.py:187-191
 *
 * Conventions
 *   - All I/O buffers of fw_reset/fw_step are DEVICE pointers owned by the
 *     caller (e.g. torch tensor .data_ptr()); nothing is retained past the call.
 *   - fw_get_state/fw_set_state take HOST pointers (double, canonical record).
 *   - Calls are asynchronous w.r.t. the host and ordered on `hip_stream`
 *     (a hipStream_t passed as void*; NULL = the legacy default stream).
 *   - A handle is not re-entrant.  Different handles are independent.
 *   - Return value: 0 = FW_OK, <0 = error enum; message via fw_last_error().
 *     Nothing throws across the ABI.
 *   - Element type T of action/obs/reward buffers is fw_config.dtype
 *     (FW_F64: double, the reference's arithmetic; FW_F32: float).
 */
#ifndef FWSIM_H
#define FWSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FW_ABI_VERSION 7

#define FW_NUM_SURFACES 5         /* left aileron, right aileron, h-tail, v-tail, main wing */
#define FW_NUM_ACTUATORS 6        /* 5 surfaces + throttle (aux_state order) */
#define FW_MAX_TARGETS 8
#define FW_MAX_COLLISION_PTS 8
#define FW_MAX_OBSTACLES 20
#define FW_VISION_FEATS 9         /* envs/fixedwing_objlock_env.py:745-761 */
#define FW_VISION_HIST 3          /* duck_vision_history_len default :69 */

/* error codes */
enum {
  FW_OK = 0,
  FW_EINVAL = -1,      /* bad argument / bad config (maps to ValueError) */
  FW_EHIP = -2,        /* HIP runtime error (no device, launch failure ...) */
  FW_ENOMEM = -3,
  FW_EVERSION = -4,    /* abi_version mismatch */
  FW_EUNSUPPORTED = -5
};

/* fw_config.task */
enum {
  FW_TASK_WAYPOINTS = 0,        /* PyFlyt/Fixedwing-Waypoints-v3 (train/train_Fixedwing_Waypoints_v3.py:100-110) */
  FW_TASK_OBJLOCK = 1,          /* envs/fixedwing_objlock_env.py */
  FW_TASK_WAYPOINT_OBJLOCK = 2  /* envs/fixedwing_waypoint_objlock_env.py */
};

/* fw_config.dtype */
enum { FW_F64 = 0, FW_F32 = 1 };

/* fw_config.wind_mode  (envs/fixedwing_envs/fixedwing_base_env.py:113-115) */
enum { FW_WIND_OFF = 0, FW_WIND_CONSTANT = 1, FW_WIND_GUST_SINE = 2 };

/* fw_config.wind_coupling: how the registered wind vector enters the dynamics.
 * Not in the reference (lives in un-vendored PyFlyt) -> explicit, build-owned. */
enum {
  FW_WIND_COUPLE_NONE = 0,
  FW_WIND_COUPLE_FORCE = 1,     /* world-frame force = wind_force_coef * w(t) on the base */
  FW_WIND_COUPLE_AIRSPEED = 2   /* w(t) subtracted from every surface's air velocity */
};

/* One lifting surface.  First 11 fields are my_models/fixedwing/fixewing.yaml:8-71
 * verbatim (angles in degrees); the geometry (units, link origin) is build-owned
 * because the URDF is not in the reference. */
typedef struct fw_surface_params {
  double Cl_alpha_2D;
  double chord;
  double span;
  double flap_to_chord;
  double eta;
  double alpha_0_base_deg;
  double alpha_stall_P_base_deg;
  double alpha_stall_N_base_deg;
  double Cd_0;
  double deflection_limit_deg;
  double tau;
  double lift_unit[3];     /* body frame */
  double forward_unit[3];  /* body frame */
  double pos[3];           /* link origin relative to the composite COM, body frame [m] */
} fw_surface_params;

/* my_models/fixedwing/fixewing.yaml:1-6 + geometry */
typedef struct fw_motor_params {
  double total_thrust;
  double thrust_coef;
  double torque_coef;
  double noise_ratio;
  double tau;
  double thrust_unit[3];
  double pos[3];
} fw_motor_params;

typedef struct fw_config {
  /* ---- integers ---- */
  int32_t abi_version;          /* must be FW_ABI_VERSION */
  int32_t task;                 /* FW_TASK_* */
  int32_t dtype;                /* FW_F64 | FW_F32 */
  int32_t angle_representation; /* 0 euler (attitude 12), 1 quaternion (13)  fixedwing_base_env.py:65-72 */
  int32_t agent_hz;             /* must divide 120              fixedwing_base_env.py:48-53 */
  int32_t physics_hz;           /* 240 */
  int32_t control_hz;           /* 120  => 2 ticks per Aviary step */
  int32_t warmup_aviary_steps;  /* 10                          fixedwing_base_env.py:254-255 */
  int32_t auto_reset;           /* 1: SB3 VecEnv worker semantics; 0: bare Gymnasium env */
  int32_t sparse_reward;
  int32_t num_targets;          /* <= FW_MAX_TARGETS */
  int32_t context_length;       /* flatten_waypoint_env.py:16 */
  int32_t wind_mode;            /* FW_WIND_* */
  int32_t wind_randomize_on_reset;
  int32_t wind_randomize_phase;
  int32_t wind_coupling;        /* FW_WIND_COUPLE_* */
  int32_t gyroscopic;           /* include  -w x (I w)  in the angular acceleration */
  int32_t n_collision_pts;      /* <= FW_MAX_COLLISION_PTS */
  int32_t num_obstacles;        /* <= FW_MAX_OBSTACLES (objlock tasks) */
  int32_t duck_camera_capture_interval_steps;
  int32_t duck_lock_hold_steps;
  int32_t duck_lock_decay_steps;
  int32_t duck_switch_min_consecutive_seen;
  int32_t camera_resolution;    /* square, pixels (analytic camera model) */
  int32_t duck_vision_no_deltas; /* ObjLock: 1 = duck_vision_use_deltas=False of the reference (envs/fixedwing_objlock_env.py:69-70, 440-441):
                                  * the observation ends with the 27 history values, without the 4 frame-to-frame deltas (52 wide, not 56).
                                  * (Round 5; one of the former reserved words, which callers zero: same layout, same ABI version.) */
  int32_t reserved_i[7];

  /* ---- env / task scalars ---- */
  double flight_dome_size;
  double max_duration_seconds;
  double goal_reach_distance;
  double waypoint_min_height;       /* 0.5  fixedwing_waypoint_objlock_env.py:103 */
  double waypoint_spawn_size;       /* dome used by the target sampler */
  double start_pos[3];
  double start_orn[3];              /* euler */
  double start_vel[3];              /* world frame, PyFlyt starting_velocity */

  /* ---- wind (fixedwing_base_env.py:108-173) ---- */
  double wind_enu_mps[3];
  double wind_enu_mps_range[3][2];
  double gust_amp_enu_mps[3];
  double gust_amp_enu_mps_range[3][2];
  double gust_freq_hz;
  double gust_phase_rad;
  double wind_force_coef;           /* N per (m/s) for FW_WIND_COUPLE_FORCE */

  /* ---- vehicle (build-owned: URDF is not in the reference) ---- */
  double mass;
  double inertia[6];                /* ixx iyy izz ixy ixz iyz, body frame about the COM */
  double gravity;                   /* 9.81, acts along -z world */
  double air_density;               /* 1.225 */
  double collision_pts[FW_MAX_COLLISION_PTS][3]; /* body-frame points; contact <=> world z <= 0 */
  double mixer[FW_NUM_ACTUATORS][4];/* mode-0: cmd = mixer * [roll,pitch,yaw,thrust01] */
  fw_surface_params surfaces[FW_NUM_SURFACES];
  fw_motor_params motor;

  /* ---- objlock tasks (envs/fixedwing_objlock_env.py:54-80) ---- */
  double duck_strike_distance_m;
  double duck_strike_reward;
  double duck_lock_step_reward;
  double duck_approach_reward_scale;
  double duck_global_scaling;
  double duck_distance_reward_scale;
  double duck_lock_center_radius;
  double duck_centering_reward_scale;
  double duck_visible_step_reward;
  double duck_area_reward_scale;
  double duck_lock_lost_penalty;
  double duck_approach_reward_clip_m;
  double duck_switch_min_area;
  double duck_radius_per_scale;     /* analytic duck = sphere of radius scale*this */
  double obstacle_radius;
  double obstacle_height_range[2];
  double obstacle_safe_distance_m;
  double obstacle_avoid_reward_scale;
  double obstacle_avoid_max_penalty;
  double camera_offset[3];          /* cockpit_fpv [0.8,0,0.12]  fixedwing_objlock_env.py:185 */
  double camera_angle_deg;          /* -5 */
  double camera_fov_deg;            /* 90 */
  double camera_near;               /* 0.1   fixedwing_objlock_env.py:692 */
  double camera_far;                /* 255.0 */
  double reserved_d[8];
} fw_config;

/* ---- canonical per-env state record (doubles; used by fw_get_state/fw_set_state) ---- */
enum {
  FW_S_POS = 0,        /* 3  world position */
  FW_S_QUAT = 3,       /* 4  x y z w, body->world */
  FW_S_VEL = 7,        /* 3  world-frame linear velocity */
  FW_S_OMEGA = 10,     /* 3  world-frame angular velocity */
  FW_S_ACT = 13,       /* 6  surface actuations (5) + throttle */
  FW_S_ACTION = 19,    /* 4  raw action as shown in the current observation (obs[12:16]) */
  FW_S_STEP_COUNT = 23,
  FW_S_TICK_COUNT = 24,/* physics ticks since reset (elapsed_time * physics_hz) */
  FW_S_EPISODE = 25,   /* episode index (RNG counter word) */
  FW_S_FLAGS = 26,     /* bit0 termination, bit1 truncation, bit2 collision, bit3 oob, bit4 env_complete,
                          bits 8-11: waypoint index seen by the last compute_state() (the returned obs) */
  FW_S_NUM_REACHED = 27,
  FW_S_NEW_DIST = 28,  /* WaypointHandler.new_distance */
  FW_S_WIND = 29,      /* 7  base[3], gust amp[3], phase */
  FW_S_EP_RETURN = 36,
  FW_S_TARGETS = 37,   /* FW_MAX_TARGETS x 3 (world) */
  FW_S_TASK = 61,      /* task-specific tail, see FW_ST_* */
  FW_STATE_DIM = 176
};

/* objlock tail (offsets from FW_S_TASK); mirrors the private fields of
 * envs/fixedwing_objlock_env.py:143-156 plus the latest analytic camera frame */
enum {
  FW_ST_DUCK_POS = 0,      /* 3 */
  FW_ST_LOCK_STEPS = 3,
  FW_ST_PREV_EST = 4,      /* _prev_est_dist_m, <0 encodes None */
  FW_ST_LAST_CX = 5,
  FW_ST_LAST_CY = 6,
  FW_ST_LAST_AREA = 7,
  FW_ST_LAST_DEPTH = 8,
  FW_ST_SINCE_SEEN = 9,
  FW_ST_HIST_FILLED = 10,
  FW_ST_FRAME_HAS = 11,    /* 1 once the camera has captured a frame this episode */
  FW_ST_FRAME = 12,        /* 8: visible, cx, cy, area, depth_m, d_left, d_center, d_right of the latest frame */
  FW_ST_HIST = 20,         /* FW_VISION_HIST x FW_VISION_FEATS = 27, newest first */
  FW_ST_DUCK_PHASE = 47,   /* combined env: bit0 duck_phase, bit1 post_waypoints */
  FW_ST_SEEN_CONSEC = 48,
  FW_ST_NUM_OBST = 49,
  FW_ST_OBST = 50,         /* FW_MAX_OBSTACLES x 3 (x, y, height) -> 110 of the 115 tail slots */
  FW_ST_DIM = 110
};

/* info_i32 columns written by fw_step (info of the step that just ran, i.e. of
 * the finished episode when the env was auto-reset). */
enum {
  FW_INFO_NUM_TARGETS_REACHED = 0,
  FW_INFO_COLLISION = 1,
  FW_INFO_OUT_OF_BOUNDS = 2,
  FW_INFO_ENV_COMPLETE = 3,
  FW_INFO_DUCK_STRIKE = 4,
  FW_INFO_IS_SUCCESS = 5,
  FW_INFO_EP_LEN = 6,     /* agent steps of the finished episode (valid when done) */
  FW_INFO_RESERVED = 7,
  FW_INFO_DIM = 8
};

/* fw_get_counters columns: how the auto-resets of fw_step were served so far (diagnostics; they let a test assert that
 * the background hand-off -- and not only its in-kernel fallback -- was exercised, e.g. under hipGraph replay). */
enum {
  FW_CTR_LAUNCHES = 0,       /* fw_step launches so far (the device-side launch index the hand-off protocol uses) */
  FW_CTR_RESETS = 1,         /* auto-resets performed inside fw_step */
  FW_CTR_SHADOW_HITS = 2,    /* ... served by a pre-simulated ("shadow") episode start: wind / camera tasks */
  FW_CTR_SCENARIO_HITS = 3,  /* ... served by pre-sampled waypoints: wind-free waypoints task */
  FW_CTR_FALLBACKS = 4,      /* ... served by the in-kernel sampler / warm-up */
  FW_CTR_HELPER_TIMEOUTS = 5,/* camera tasks, 8-lane mapping: waits of a step wave for its capture wave that gave up (a protocol error; 0 always) */
  FW_CTR_DIM = 8
};

typedef struct fw_env* fw_handle;

/* Size of fw_config as compiled into the library (binding self-check). */
int32_t fw_sizeof_config(void);
int32_t fw_abi_version(void);
int32_t fw_state_dim(void);      /* FW_STATE_DIM of the canonical state record */

/* Observation width D for a config (22/23 attitude + task part); <0 on error. */
int32_t fw_obs_dim(const fw_config* cfg);

/* Validate a config exactly like the reference constructors do.  On error
 * returns FW_EINVAL and writes a message (same wording class as the
 * reference's ValueError) into msg[msg_len]. */
int32_t fw_validate_config(const fw_config* cfg, char* msg, int32_t msg_len);

/* Create N envs on HIP device `device`.  global_env_offset is added to the
 * local env index to form the RNG key, so that a sharded job (rank r owns
 * [r*N, (r+1)*N)) draws the same scenarios as a single-device job. */
int32_t fw_create(const fw_config* cfg, int32_t num_envs, int32_t device,
                  uint64_t seed, int64_t global_env_offset, fw_handle* out);

/* Scenario of the episode a fw_reset starts, supplied by the caller instead of drawn by the env -- e.g. the targets /
 * duck / obstacles / wind a PyFlyt run sampled, so that reference traces can be replayed without
 * forging whole state records.  HOST arrays indexed by the handle's local env; any pointer may be NULL = keep the
 * env's own draw for that item.  Only the episode started by this call is affected (later auto-resets draw again). */
typedef struct fw_scenario {
  const double* targets;        /* [N][FW_MAX_TARGETS][3] world waypoints (rows >= num_targets ignored)     WaypointHandler.reset */
  const double* duck_pos;       /* [N][3] duck base position                         envs/fixedwing_objlock_env.py:472-491 */
  const double* obstacles;      /* [N][FW_MAX_OBSTACLES][3]  x, y, height            envs/fixedwing_objlock_env.py:528-565 */
  const int32_t* num_obstacles; /* [N]  cylinders actually placed (required with `obstacles`; clipped to fw_config.num_obstacles) */
  const double* wind_base;      /* [N][3]  wind_enu_mps                              envs/fixedwing_envs/fixedwing_base_env.py:135-143 */
  const double* gust_amp;       /* [N][3]  gust_amp_enu_mps                          :155-159 */
  const double* gust_phase;     /* [N]     gust_phase_rad                            :162-165 */
} fw_scenario;

/* Reset envs.  mask: device u8[N] (non-zero = reset) or NULL = all.
 * scenario: NULL, or the caller-supplied scenario of the new episodes (see fw_scenario; applied to the reset envs only).
 * obs_out: device T[N,D] or NULL.  Rows of non-reset envs are rewritten with
 * their current observation. */
int32_t fw_reset(fw_handle h, const uint8_t* mask, const fw_scenario* scenario, void* obs_out, void* hip_stream);

/* One agent step for all N envs.
 *   actions       T[N,4]  in [-1,1] (caller clips, as SB3 does)
 *   obs           T[N,D]  next observation (first obs of the new episode if auto-reset fired)
 *   reward        T[N]
 *   terminated    u8[N]
 *   truncated     u8[N]
 *   terminal_obs  T[N,D]  rows written only where terminated|truncated (may be NULL)
 *   info_i32      i32[N,FW_INFO_DIM] (may be NULL) */
int32_t fw_step(fw_handle h, const void* actions, void* obs, void* reward,
                uint8_t* terminated, uint8_t* truncated, void* terminal_obs,
                int32_t* info_i32, void* hip_stream);

/* Re-key the RNG: subsequent resets use (seed, global env id, episode=0...). */
int32_t fw_seed(fw_handle h, uint64_t seed);

/* Canonical state records, HOST memory, double[N, FW_STATE_DIM]. Synchronous. */
int32_t fw_get_state(fw_handle h, double* state_out);
int32_t fw_set_state(fw_handle h, const double* state_in);

/* Diagnostic counters, HOST uint64[FW_CTR_DIM] (see FW_CTR_*).  Synchronous. */
int32_t fw_get_counters(fw_handle h, uint64_t* out);

/* Recompute the observation from the current state (no dynamics), device T[N,D]. */
int32_t fw_observe(fw_handle h, void* obs_out, void* hip_stream);

/* FPV image of every env's CURRENT pose, for a network front end (the reference's camera hands the env segImg / depthImg,
 * envs/fixedwing_objlock_env.py:603-622; its CNN path replaces segImg by a detector's mask,
 * envs/fixedwing_envs/objlock_yolo_env.py:646-716).  The scene is the analytic one the vision features of the objlock tasks are
 * functionals of: out = device float32 [N][2][res][res], channel 0 = duck mask (0 / 1), channel 1 =
 * depth-buffer value in [0, 1] of the nearest fragment (sky 1.0); same camera (offset, tilt, FOV), focal length scaled to `res`.
 * FW_EUNSUPPORTED for the waypoints task (no camera). */
int32_t fw_render(fw_handle h, int32_t res, float* out, void* hip_stream);

/* ---- rollout-collector helpers (the caller side of the path: SB3's OnPolicyAlgorithm /
 * RolloutBuffer / VecNormalize as driven by train/train_Fixedwing_Waypoints_v3.py:260,293-337).
 * Stateless, float32 device buffers laid out [T, N] (time-major, env contiguous). ---- */

/* Generalised advantage estimation, SB3 RolloutBuffer.compute_returns_and_advantage:
 *   delta_t = r_t + gamma * V_{t+1} * (1 - start_{t+1}) - V_t
 *   A_t     = delta_t + gamma * lambda * (1 - start_{t+1}) * A_{t+1},   returns = A + V
 * with V_T = last_values, start_T = last_dones.  episode_starts[t,n] = 1 iff step t is the
 * first of an episode.  One lane per env, T sequential steps, coalesced across envs. */
int32_t fw_gae(const float* rewards, const float* values, const float* episode_starts,
               const float* last_values, const float* last_dones, float* advantages, float* returns,
               int32_t T, int32_t N, float gamma, float gae_lambda, void* hip_stream);

/* Episode bookkeeping of an evaluation loop (SB3 evaluate_policy; evaluate.ReplayedEvaluation): after a vec-step, add `reward` [N]
 * (env dtype) to cur_rew and 1 to cur_len; where terminated | truncated and counts[i] < targets[i], record the episode (reward, length,
 * the vec-step index, the info row) in slot counts[i] of env i ([N, E] buffers; info may be NULL) and count it; clear the accumulators of
 * finished episodes; advance step_ctr[0].  One launch, all buffers on the device. */
int32_t fw_eval_track(const void* reward, int32_t reward_is_f64, const uint8_t* terminated, const uint8_t* truncated, const int32_t* info,
                      int32_t info_dim, const int64_t* targets, int64_t* counts, double* cur_rew, int64_t* cur_len, int64_t* step_ctr,
                      double* fin_rew, int64_t* fin_len, int64_t* fin_step, int32_t* fin_info, int32_t N, int32_t E, void* hip_stream);

/* VecNormalize step (SB3 VecNormalize.step_wait + RunningMeanStd.update, Chan et al. merge),
 * fused: one pass over obs[N,D] (env dtype T_in = double|float per `in_is_f64`) that
 *   (a) if `update` != 0 merges the batch moments into (mean[D], var[D], count[1]) (double),
 *   (b) writes clip((obs - mean) / sqrt(var + eps), +-clip) as float32 to obs_out[N,D]
 *       using the UPDATED statistics, as SB3 does.
 * mean/var/count are device double buffers owned by the caller (checkpointable).
 * workspace: device buffer of fw_normalize_obs_workspace_bytes(D) bytes owned by the caller (needed when update != 0):
 * the per-block partial sums live there, so concurrent callers / streams share nothing and nothing is ever allocated
 * on the launch path (safe under hipGraph capture).
 * batch_acc (may be NULL): device double[2 D + 1]; when update != 0 the batch's column sums, sums of squares and row count
 * are ADDED to it.  A job sharded over GPUs all-reduces these accumulators once per rollout (RCCL) and re-derives the
 * statistics of all envs from them (collective 2) -- no collective sits between two env steps. */
int64_t fw_normalize_obs_workspace_bytes(int32_t D);
int32_t fw_normalize_obs(const void* obs, int32_t in_is_f64, int32_t N, int32_t D, double* mean, double* var,
                         double* count, int32_t update, float clip, float eps, float* obs_out, void* workspace,
                         double* batch_acc, void* hip_stream);

/* PPO minibatch updates (SB3 PPO.train() inner loop; train/train_Fixedwing_Waypoints_v3.py:293-310) for the
 * reference's MlpPolicy (separate 64-64 tanh nets for pi and V, 4-dim diagonal Gaussian, log_std parameter),
 * fused into ONE kernel that walks `n_minibatches` consecutive minibatches: forward, clipped-surrogate +
 * value + entropy loss, backward, global grad-norm clipping and Adam, weights resident in LDS.
 * Flat layout of params (float32, fw_ppo_param_count(obs_dim) elements):
 *   for net in (pi, vf): W1[Dp][64] b1[64] W2[64][64] b2[64] Wo[64][KO] bo[KO]   (KO = 4, 1; W = Linear.weight^T;
 *   Dp = obs_dim rounded up to even, the pad row is zero), then log_std[4].
 * Adam moments mom_m / mom_v (float32, fw_ppo_moment_count() elements each) are in the kernel's "slot" order, in
 * which every lane's elements are contiguous: fw_ppo_moment_map(obs_dim, out) gives the flat parameter index of
 * each slot (-1 for padding and for the unused second half of the buffers), which is all a caller needs to
 * move them to and from its optimiser.  The slot order belongs to a library build (round 3 moved Wo's moments): persist
 * moments in the optimiser's own order, never in slot order.
 * obs[S,obs_dim], act[S,4], old_logp[S], adv[S], ret[S]: the rollout buffer (float32, device);
 * perm[n_minibatches * batch_size]: sample indices of consecutive minibatches (int32, device).
 * batch_size must be a multiple of 16, obs_dim <= 64.  loss_acc[3] (may be NULL) accumulates the per-minibatch
 * mean policy loss, value loss and entropy loss.  The caller advances its Adam step count by n_minibatches.
 * workspace: caller-owned device buffer of >= fw_ppo_update_workspace_bytes(n_minibatches, batch_size, obs_dim) bytes: exchange
 * words, the gradient hand-off buffer and the PACKED ROWS of this call -- a parallel pre-pass (one workgroup per minibatch, all
 * CUs) writes every minibatch's rows in walking order, observation | action | old log-prob, normalised advantage, return,
 * (obs_dim rounded up to 4) + 8 floats each, so that the sequential kernel reads each 16- / 32- / 64-sample pass as one contiguous block
 * (144 B per sample and epoch for 28 observations: 189 MB for 20 epochs x 65 536 samples).  Two learners never share it.
 * The learner-side entry points run on the device their buffers live on, whatever the thread's current device.
 * The call runs on 2, 4, 8 or 16 workgroups: (policy, value) x 1, 2, 4 or 8 workgroups per network that share every minibatch (128
 * samples: 8 x 16, 64: 4 x 16) and exchange gradients (and, from four on, updated weights) once per minibatch; all of them end the call
 * with the same bits.
 * Failure inside the launch: the workgroups of a call wait for each other two or three times per minibatch, every wait
 * bounded.  A wait that runs out raises FW_PPO_ST_* in the workspace's status word and the workgroups leave.  `params`, `mom_m`
 * and `mom_v` are written back only behind a verdict every workgroup waits for at the end of the call (the last one to get through
 * its final minibatch reads the status word and says "write" or "do not"): a workgroup that gave up never arrives, so in that case
 * NOBODY writes and the three buffers are exactly as they were before the call.  The one exception is a verdict wait that itself
 * runs out (FW_PPO_ST_COMMIT): some workgroups may then have written their share.  The contract is therefore: status 0 = the
 * buffers hold the result; any other status = treat `params`, `mom_m`, `mom_v` as UNDEFINED and restore them from the caller's own
 * copies (rollout.FusedPpoUpdate reloads them from the module / optimiser, which it only overwrites after status 0).
 * fw_ppo_update_status reads
 * the word after the call (it synchronises `hip_stream`): 0 = the call ran to its end.  `paths` (may be NULL) receives which
 * exchanges went through an XCD's shared L2 rather than device-scope accesses: bit 2 b = workgroup b's gradient swap, bit
 * 2 b + 1 = its norm exchange, b = 2 * part + net (part = the workgroup's index inside its network).  Environment (read per call):
 * FWSIM_PPO_SPLIT=CHxN forces the cut (CH = 16 / 32 / 64 samples per pass, N = 1 / 2 / 4 / 8 workgroups per network); FWSIM_PPO_RS=0 makes
 * (up to four) workgroups swap whole gradients all-to-all instead of reduce-scatter + weight all-gather; FWSIM_PPO_NO_L2_SWAP=1 forces the
 * device-scope form of every exchange (same arithmetic: results must be bit-identical -- tests/test_protocols_gpu.py);
 * FWSIM_SPIN_LOG2=k shrinks every wait's budget to 2^k polls (tests provoke the timeout with it). */
#define FW_PPO_ST_IDS 1u    /* the workgroups never found each other at the start of the call */
#define FW_PPO_ST_SWAP 2u   /* gradient swap between the workgroups of a network */
#define FW_PPO_ST_NORM 4u   /* gradient-norm exchange between the policy and the value workgroup */
#define FW_PPO_ST_COMMIT 8u /* the closing verdict did not arrive: some workgroups may have written their results, others not */
typedef struct fw_ppo_hyper {
  float lr, clip_range, ent_coef, vf_coef, max_grad_norm, beta1, beta2, eps;
  float adv_mean, adv_std;      /* used when norm_adv == 2 */
  int32_t norm_adv;             /* 0: off, 1: per minibatch (SB3 default), 2: with the given statistics */
  int32_t step0;                /* Adam steps taken before this call */
} fw_ppo_hyper;
int32_t fw_ppo_param_count(int32_t obs_dim);
int32_t fw_ppo_moment_count(void);
int32_t fw_ppo_moment_map(int32_t obs_dim, int32_t* flat_index_of_slot /* host, [fw_ppo_moment_count()] */);
int64_t fw_ppo_update_workspace_bytes(int32_t n_minibatches, int32_t batch_size, int32_t obs_dim);
int32_t fw_ppo_update(float* params, float* mom_m, float* mom_v, const float* obs, const float* act,
                      const float* old_logp, const float* adv, const float* ret, const int32_t* perm,
                      int32_t n_minibatches, int32_t batch_size, int32_t obs_dim, const fw_ppo_hyper* hyper,
                      float* loss_acc, void* workspace, int64_t workspace_bytes, void* hip_stream);
int32_t fw_ppo_update_status(const void* workspace, int64_t workspace_bytes, uint32_t* status_out, uint32_t* paths_out, void* hip_stream);

/* Rollout collection between two env steps (SB3 OnPolicyAlgorithm.collect_rollouts + VecNormalize reward path,
 * train/train_Fixedwing_Waypoints_v3.py:260,293-310), for the same MlpPolicy / flat parameter image as fw_ppo_update.
 * fw_policy_act: obs[N,obs_dim] (normalised, float32) -> for `nets` bit 0 (policy): act_raw[N,4] = mean + sigma * z
 *   (z ~ N(0,1) from Philox(rng[0] = seed; rng[1] = draw counter, global env id = env_offset + row); mean if
 *   `deterministic`), logp[N], act_env[N,4] = clip(act_raw, +-1) in the env dtype (the fw_step input), and obs copied to
 *   obs_copy (rollout buffer, may be NULL); for bit 1 (value): value[N].
 * fw_rollout_post: VecNormalize.step_wait's reward path + the bootstrap of truncated episodes:
 *   returns = returns * gamma + reward; running variance of `returns` (if training && norm_reward);
 *   rew_out = clip(reward / sqrt(var + epsilon), +-clip_reward) (+ gamma * tvalue where truncated && !terminated);
 *   start_out = terminated | truncated; returns = 0 where done; rng[1] += 1 (rng may be NULL);
 *   ret_acc (may be NULL): device double[3], the tracker batch's (sum, sum of squares, count) are added to it (the
 *   reward-side twin of fw_normalize_obs's batch_acc). */
int32_t fw_policy_act(const float* params, const float* obs, int32_t N, int32_t obs_dim, int32_t nets, int32_t deterministic,
                      const uint64_t* rng, int64_t env_offset, float* obs_copy, float* act_raw, void* act_env,
                      int32_t act_is_f64, float* logp, float* value, void* hip_stream);
/* V(normalised terminal_observation) for the envs whose episode was truncated but not terminated (the only ones SB3
 * bootstraps): terminal_obs[N,obs_dim] is the env's raw buffer (fw_step's terminal_obs), normalised on load with
 * (mean, var, clip, eps) exactly like fw_normalize_obs; value[i] is written for every row of a 64-row block that
 * contains such an env and left untouched elsewhere. */
int32_t fw_policy_terminal_value(const float* params, const void* terminal_obs, int32_t obs_is_f64, int32_t N, int32_t obs_dim,
                                 const double* mean, const double* var, float clip, float eps, const uint8_t* terminated,
                                 const uint8_t* truncated, float* value, void* hip_stream);
int32_t fw_rollout_post(const void* reward, int32_t rew_is_f64, const uint8_t* terminated, const uint8_t* truncated,
                        const float* tvalue, double* returns, double* ret_mean, double* ret_var, double* ret_count,
                        int32_t N, int32_t training, int32_t norm_reward, double gamma, float clip_reward, float epsilon,
                        float* rew_out, float* start_out, uint64_t* rng, double* ret_acc, void* hip_stream);

/* The same collection in THREE launches per vec-step (fw_collect_act -> fw_step -> fw_collect_stats) instead of seven:
 * fw_collect_act  = fw_policy_act reading the env's RAW observation buffer and normalising it on load with (obs_mean, obs_var,
 *   clip_obs, eps_obs) -- the normalised rows still go to obs_copy -- whose value block also FINALISES THE PREVIOUS STEP for its
 *   rows when prev_reward != NULL: rew_out = clip(prev_reward / sqrt(ret_var + eps_reward), +-clip_reward) (if norm_reward)
 *   + gamma * V(normalised prev_terminal_obs) where prev_truncated && !prev_terminated (a second pass through the value network in
 *   the blocks that hold such a row), start_out = prev_terminated | prev_truncated.  It runs BEFORE the next fw_step, while the
 *   env's reward / flag / terminal-observation buffers still hold the previous step.
 * fw_collect_stats = VecNormalize.step_wait's statistics after an env step, one launch: observation moments -> (obs_mean, obs_var,
 *   obs_count) if update_obs; returns = returns * gamma + reward, their moments -> (ret_mean, ret_var, ret_count) if update_ret,
 *   returns = 0 where the episode ended; rng[1] += 1 (rng may be NULL); obs_acc / ret_acc as batch_acc / ret_acc above.
 *   workspace: caller-owned, fw_collect_stats_workspace_bytes(D) bytes, zero-initialised once. */
int32_t fw_collect_act(const float* params, const void* raw_obs, int32_t obs_is_f64, int32_t N, int32_t obs_dim, const double* obs_mean,
                       const double* obs_var, float clip_obs, float eps_obs, int32_t nets, int32_t deterministic, const uint64_t* rng,
                       int64_t env_offset, float* obs_copy, float* act_raw, void* act_env, int32_t act_is_f64, float* logp, float* value,
                       const void* prev_reward, const uint8_t* prev_terminated, const uint8_t* prev_truncated, const void* prev_terminal_obs,
                       const double* ret_var, int32_t norm_reward, float clip_reward, float eps_reward, float gamma, float* rew_out,
                       float* start_out, void* hip_stream);
int64_t fw_collect_stats_workspace_bytes(int32_t D);
int32_t fw_collect_stats(const void* obs, int32_t obs_is_f64, int32_t N, int32_t D, double* obs_mean, double* obs_var, double* obs_count,
                         int32_t update_obs, const void* reward, int32_t rew_is_f64, const uint8_t* terminated, const uint8_t* truncated,
                         double* returns, double* ret_mean, double* ret_var, double* ret_count, int32_t update_ret, double gamma,
                         uint64_t* rng, void* workspace, double* obs_acc, double* ret_acc, void* hip_stream);

/* The same collection in ONE launch per vec-step: the policy / value forward of step t (fw_collect_act's work, by "act waves" at
 * the front of the grid), the env step (fw_step's waves, which wait for their actions inside the launch) and the statistics
 * of VecNormalize.step_wait.  Semantics and arithmetic are those of fw_collect_act -> fw_step -> fw_collect_stats called with the
 * same buffers (statistics to ~1e-12: another summation order; normalised observations within an ulp of double before the
 * float32 rounding): `obs`, `reward`, `terminated`, `truncated`, `terminal_obs` hold the PREVIOUS step on entry (obs = the
 * observation to act on) and this step on return; rew_out / start_out (both or neither) receive the finalisation of the
 * previous step.  The batch sums of a step are folded by "fold waves" at the end of its own launch and merged into the
 * statistics buffers by the NEXT fw_collect_step (which needs them first) -- or by fw_collect_finish: call it after the last
 * step of a rollout, before anything else reads (obs_mean, obs_var, obs_count) / the return statistics.  Serves handles on the 8-lanes-per-env mapping, either build (FW_EUNSUPPORTED
 * otherwise -- one lane per env, i.e. beyond 24 576 waypoint envs per GPU: use the three calls).  workspace: caller-owned, fw_collect_step_workspace_bytes(h) bytes, one per handle, prepared ONCE with
 * fw_collect_workspace_init (stream-ordered; not inside a graph that is replayed).
 * Failure inside a launch: the waves of the grid wait for waves in front of them (step waves for their actions, fold waves for
 * the partial sums, the merge wave for the act waves), every wait bounded.  A wait that runs out does not stop the launch --
 * the wave goes on with what it has (zero actions, partial sums, an early commit) -- but raises FW_COLLECT_ST_* in the
 * workspace's status word, the uint32 at (fw_collect_step_workspace_bytes(h) - 64 + 12); the word is sticky until
 * fw_collect_workspace_init.  Non-zero means: everything collected since the last zero reading is void.  Read it at least once
 * per rollout -- fw_collect_status (synchronises `hip_stream`) or a copy of that word; rollout.PPO raises RuntimeError.
 * FWSIM_SPIN_LOG2=k (environment, read per call) shrinks the budgets to 2^k polls: tests provoke the timeouts with it. */
#define FW_COLLECT_ST_ACTIONS 1u  /* a step wave gave up waiting for its actions (it stepped with zeros) */
#define FW_COLLECT_ST_FOLD 2u     /* a fold wave summed without every partial, or never saw the merge wave finish */
#define FW_COLLECT_ST_MERGE 4u    /* the merge wave committed the statistics before every act wave had read the old ones */
#define FW_COLLECT_ST_NOINIT 8u   /* the workspace was never passed to fw_collect_workspace_init */
#define FW_COLLECT_ST_EPOCH 16u   /* the merge wave never saw the launch's index published */
#define FW_COLLECT_ST_NANACT 32u  /* the policy produced a NaN action (diverged weights or statistics): the env stepped with -1 in its place */
typedef struct fw_collect_args {
  const float* params;                     /* flat parameter image (fw_ppo_update layout) */
  double *obs_mean, *obs_var, *obs_count;  /* VecNormalize observation statistics (in/out) */
  double *returns;                         /* [N] discounted-return tracker (in/out) */
  double *ret_mean, *ret_var, *ret_count;  /* its running statistics (in/out) */
  double *obs_acc, *ret_acc;               /* sharded jobs: batch-sum accumulators (may be NULL) */
  uint64_t* rng;                           /* [2] seed, draw counter (advanced by one) */
  float *obs_copy, *act_raw, *logp, *value;/* rollout-buffer rows of the step being acted (obs_copy may be NULL) */
  void* act_env;                           /* T[N,4] the clipped actions the envs step with (scratch owned by the caller, one per handle;
                                            * between launches every word is NaN = "not there yet": the library fills a new buffer itself) */
  float *rew_out, *start_out;              /* finalisation of the previous step: both or neither */
  void *obs, *reward;                      /* env buffers, as fw_step (in: previous step, out: this step) */
  uint8_t *terminated, *truncated;
  void* terminal_obs;
  int32_t* info_i32;                       /* may be NULL */
  void* workspace; int64_t workspace_bytes;
  void* trace;                             /* NULL, or device int64[(grid size) * 8]: per-workgroup wall-clock stamps (diagnostic, tools/trace_collect.py) */
  double gamma;
  float clip_obs, eps_obs, clip_reward, eps_reward;
  int32_t update_obs, update_ret, norm_reward, deterministic;
} fw_collect_args;
int64_t fw_collect_step_workspace_bytes(fw_handle h);
int32_t fw_collect_workspace_init(fw_handle h, void* workspace, int64_t workspace_bytes, void* hip_stream);
int32_t fw_collect_step(fw_handle h, const fw_collect_args* a, void* hip_stream);
int32_t fw_collect_finish(fw_handle h, const fw_collect_args* a, void* hip_stream);   /* statistics buffers, flags and workspace of `a` only */
int32_t fw_collect_status(fw_handle h, const void* workspace, int64_t workspace_bytes, uint32_t* status_out, void* hip_stream);
/* The end of a rollout of T fw_collect_step launches in ONE more launch instead of fw_collect_finish + a value forward + a
 * normalisation + fw_gae: merges the last step's statistics; V(final observation) -> a->value [N] (SB3's last values); the
 * normalised final observation -> a->obs_copy (may be NULL); finalises step T - 1 (a->rew_out = row T - 1 of `rewards`,
 * a->start_out = the episode starts after the last step: fw_gae's last_dones); then SB3's
 * RolloutBuffer.compute_returns_and_advantage over the [T, N] float32 buffers (fw_gae's arithmetic, each value wave for its
 * own rows).  `a` as for fw_collect_step (obs / reward / terminated / truncated / terminal_obs = the outputs of the last
 * step; the policy-side buffers are not used). */
typedef struct fw_collect_close_args {
  const float *rewards, *values, *episode_starts;   /* [T, N] rollout buffers */
  float *adv, *ret;                                  /* [T, N] out */
  int32_t T;
  float gae_gamma, gae_lambda;
} fw_collect_close_args;
int32_t fw_collect_close(fw_handle h, const fw_collect_args* a, const fw_collect_close_args* c, void* hip_stream);

int32_t fw_num_envs(fw_handle h);
/* Lane mapping of this handle's step kernels: 8 = eight lanes of a wavefront share one env (latency mapping, one wave per SIMD),
 * 16 = the same mapping built for two waves per SIMD (256 registers), 1 = one lane per env (throughput mapping); chosen by
 * fw_create from the env count.  Diagnostic; results do not depend on it. */
int32_t fw_lanes_per_env(fw_handle h);
/* 1 when this handle's fw_step workgroups carry a capture wave (camera tasks on the 8-lane mapping, opt-in: environment variable
 * FWSIM_CAPTURE_WAVE=1 at fw_create; csrc/fwsim_objlock.hpp "The capture wave"), else 0.  Diagnostic; results agree with the
 * one-wave kernel to rounding. */
int32_t fw_capture_wave(fw_handle h);
const char* fw_last_error(fw_handle h); /* h may be NULL: last create/validate error */
int32_t fw_destroy(fw_handle h);

#ifdef __cplusplus
}
#endif
#endif /* FWSIM_H */
