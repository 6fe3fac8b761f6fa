// fwsim.hip -- kernels + C-ABI host side of libfwsim_hip.so (gfx950 only).
// See include/fwsim.h for the contract and fwsim_device.hpp for the device code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "fwsim_device.hpp"
#include "fwsim_rollout.hpp"
#include "fwsim_ppo.hpp"
#include "fwsim_collect.hpp"
#include "fwsim_fused.hpp"
#include "fwsim_objlock.hpp"
#include "fwsim_render.hpp"

using namespace fwsim;

// ======================================================================
// kernels
// ======================================================================

// K0: mark every env as an un-reset shell (step() before reset() is inert).
template <typename T>
__global__ void fw_init_kernel(DevState<T> D, int tile) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= D.npad) return;
  for (int f = 0; f < RF_COUNT; ++f) D.r[tile_index(tile, RF_COUNT, f, env)] = (T)0;
  D.r[tile_index(tile, RF_COUNT, RF_QUAT + 3, env)] = (T)1;
  for (int f = 0; f < IF_COUNT; ++f) D.i[tile_index(tile, IF_COUNT, f, env)] = 0;
  D.i[tile_index(tile, IF_COUNT, IF_FLAGS, env)] = FL_TERM;
  D.i[tile_index(tile, IF_COUNT, IF_EPISODE, env)] = -1;
}

// Kw: one lane integrates the wind-free warm-up once; resets then copy it.
template <typename T>
__global__ void fw_warm_kernel(Params<T>* Pm) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const Params<T>& P = *Pm;
  Rigid<T> S;
  for (int k = 0; k < 3; ++k) { S.p[k] = P.start_pos[k]; S.v[k] = P.start_vel[k]; S.w[k] = (T)0; }
  for (int k = 0; k < 4; ++k) S.q[k] = P.start_quat[k];
  for (int k = 0; k < FW_NUM_ACTUATORS; ++k) S.act[k] = (T)0;
  T cmd0[FW_NUM_ACTUATORS] = {(T)0, (T)0, (T)0, (T)0, (T)0, (T)0};
  T wind0[3] = {(T)0, (T)0, (T)0};
  int ticks = P.warmup_aviary_steps * P.ticks_per_aviary;
  TickC<T> C; SurfC<T> mine; T wmask;
  load_tick_constants<T, 1>(Pm, C, mine, wmask);
  T R[9];
  normalize_quat<T>(S.q);
  rot_from_unit_quat<T>(S.q, R);
  LaneAct<T> LA; LA.a = (T)0; LA.cmd = (T)0;      // (unused by the one-lane-per-env tick)
  for (int t = 0; t < ticks; ++t) (void)physics_tick<T, false, 1>(P, C, S, R, cmd0, (T)0, wind0, mine, wmask, LA);
  for (int k = 0; k < 3; ++k) { Pm->warm[k] = S.p[k]; Pm->warm[7 + k] = S.v[k]; Pm->warm[10 + k] = S.w[k]; }
  for (int k = 0; k < 4; ++k) Pm->warm[3 + k] = S.q[k];
  for (int k = 0; k < FW_NUM_ACTUATORS; ++k) Pm->warm[13 + k] = S.act[k];
  Pm->warm_ticks = ticks;
  // the observation every cached reset returns, up to its target deltas
  const T a0[4] = {(T)0, (T)0, (T)0, (T)0};
  T Rw[9];
  (void)write_obs_attitude<T, false>(P, S, a0, Rw, [&](int k, T v) { Pm->warm_obs[k] = v; });
  for (int k = 0; k < 9; ++k) Pm->warm_R[k] = Rw[k];
}

// Prefetch that stays where it is written.  What a resetting env copies in the epilogue is only USED inside the divergent reset
// block, and LLVM sinks a plain load down to its only use: the round trip the prefetch was meant to hide then sits inside that
// block (tools/wave_profile.py: ~2.3 k cycles in the waves that reset -- the slowest waves of nearly every launch -- whatever was
// "requested" ahead of the observation pass).  Volatile loads are no way out (the backend waits for each at once: 20.1 -> 24.5 us
// per step).  So the load is issued by hand: `pf_issue` emits it here, `pf_wait` (one s_waitcnt for the lot, tied to the values
// through "+v") comes right before the first use.  The compiler's own vmcnt bookkeeping stays conservative-correct: loads return
// in order, extra outstanding loads only make its waits longer.
__device__ __forceinline__ void pf_issue(double& dst, const double* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
__device__ __forceinline__ void pf_issue(float& dst, const float* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
template <typename T> __device__ __forceinline__ void pf_wait(T& a, T& b, T& c, T& d) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory"); }
template <typename T> __device__ __forceinline__ void pf_tie(T& a, T& b, T& c, T& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
template <typename T> __device__ __forceinline__ T vload(const T* p) { return *p; }

// action shown in the observation: src 0 = this step's input, 1 = stored (stale), 2 = zeros
template <typename T, bool COH = false>
__device__ __forceinline__ void load_action(const DevState<T>& D, const T* actions, int env, int src, T a[4]) {
  const T* ap = actions + (size_t)env * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    T cur = COH ? ld_coherent(ap + k) : ap[k], old = D.r[(size_t)(RF_ACTION + k) * D.npad + env];
    a[k] = (src == 0) ? cur : (src == 1) ? old : (T)0;
  }
}

// ------------------------------------------------------------------------------------------
// Background warm-up ("shadow") -- used when the reset warm-up cannot be cached (wind acts on
// the dynamics, or the task has a camera).  In the latency regime the launch time is the time
// of its slowest wave, and a wave that has to sample a scenario and integrate the 10 warm-up
// Aviary steps of a reset in-kernel takes several times as long as its neighbours -- with
// thousands of envs most launches contain one (rocprofv3: 27 us min, 96 us max, 61 us mean).
// Instead every env owns a shadow copy of its state holding the START of its NEXT episode.
// Extra workgroups of the same launch (blocks >= the step blocks; they land on SIMDs the
// latency mapping leaves idle) build it in chunks no longer than a stepping wave -- launch 1:
// scenario sampling, launches 2-4: <= step_ratio warm-up Aviary steps each -- and a reset
// swaps the finished shadow in (a copy).  An episode is a pure function of (seed, env,
// episode), so the result equals the in-kernel path (to rounding: the compiler contracts
// FMAs differently at the two inlining sites), which stays as the fallback while a shadow is
// not ready (episodes shorter than ~5 steps).
// Protocol (kernel-boundary visibility only, one writer per word, 8-byte words so nothing
// tears): the env's own lanes never touch the shadow arrays; they post `sreq` = (episode
// wanted, launch index).  The worker ignores a request carrying the current launch index
// (it may be half a launch old), owns the shadow arrays and `sdone` = (episode built, launch
// index, progress); the env's lanes accept a finished shadow only if it was finished in an
// EARLIER launch.
// ------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ DevState<T> shadow_view(const DevState<T>& D) {
  DevState<T> V = D; V.r = D.rs; V.i = nullptr; return V;
}
__device__ __forceinline__ unsigned long long pack_done(uint32_t ep, uint32_t epoch, int done) {
  return ((unsigned long long)ep << 32) | ((unsigned long long)(epoch & 0xFFFFFFu) << 8) | (unsigned long long)(done & 0xFF);
}

// worker: one chunk of work on the shadows of the envs of block `blk` (at most `max_chunk` warm-up Aviary steps; 0 = step_ratio)
template <typename T, int G, int TKIND>
__device__ __forceinline__ void shadow_worker(const Params<T>* __restrict__ Pp, const ObjC<T>* __restrict__ OCp, DevState<T> D, int blk, int max_chunk = 0) {
  constexpr bool OBJ = TKIND == FW_TASK_OBJLOCK, COMB = TKIND == FW_TASK_WAYPOINT_OBJLOCK, HASOBJ = OBJ || COMB;
  constexpr int EPW = kWave / G;
  const Params<T>& P = *Pp;
  const ObjC<T>& OC = *OCp;
  const int lane = threadIdx.x & (kWave - 1), sub = (G == 1) ? 0 : (lane & (G - 1)), row = lane / G;     // (the capture wave of a two-wave workgroup works here too)
  const bool leader = sub == 0;
  const int env = blk * EPW + row;
  const bool active = env < D.n;
  const int envc = active ? env : D.n - 1;
  const size_t n = D.npad;
  const int total = P.warmup_aviary_steps + 1;                  // progress 1 = scenario sampled, then one per warm-up step
  const unsigned long long req = D.sreq[envc], dn = D.sdone[envc];
  const uint32_t target = (uint32_t)(req >> 32);
  const bool fresh = active && (uint32_t)req != D.epoch && req != ~0ull;     // a request of an earlier launch
  int done = (int)(dn & 0xFF);
  const bool begin = fresh && (uint32_t)(dn >> 32) != target;   // new episode wanted: (re)start
  int left = (fresh && !begin) ? min(total - done, max_chunk > 0 ? max_chunk : P.step_ratio) : 0;
  if (__ballot(begin || left > 0) == 0ull) return;              // nothing to do in this wave: the common case
  const DevState<T> V = shadow_view<T>(D);
  if (begin) {
    Rigid<T> S0; int32_t tick0 = 0, episode = (int32_t)target - 1, nr = 0; T wb0[3], wa0[3], wph0;
    T tm0[3]; (void)begin_reset<T, G>(P, V, env, S0, tick0, episode, nr, wb0, wa0, wph0, tm0);     // shadow_on implies !warm_valid
    if (HASOBJ) {
      ObjState<T> O0;
      obj_reset_state<T, OBJ>(O0);
      if (OBJ) obj_spawn<T, G>(P, OC, V, env, target, leader, O0); else comb_spawn<T, G>(P, OC, V, env, target, leader, O0);
      if (leader) obj_store<T, OBJ>(V, env, O0);
    }
    if (leader) { store_rigid<T>(V, env, S0); D.is[env] = 0; D.sdone[env] = pack_done(target, D.epoch, 1); }
  }
  if (__ballot(left > 0) == 0ull) return;
  TickC<T> C; SurfC<T> mine; T wmask;
  load_tick_constants<T, G, false>(Pp, C, mine, wmask);
  Rigid<T> S;
  load_rigid<T>(V, envc, S);
  T R[9];
  rot_from_unit_quat<T>(S.q, R);
  int32_t tick = D.is[envc];
  T wb[3], wa[3], wph;
#pragma unroll
  for (int k = 0; k < 3; ++k) { wb[k] = V.r[(RF_WIND + k) * n + envc]; wa[k] = V.r[(RF_WIND + 3 + k) * n + envc]; }
  wph = V.r[(RF_WIND + 6) * n + envc];
  T gust[2];
  gust_init<T>(P, wph, tick, gust);
  ObjState<T> O;
  if (HASOBJ) { obj_load<T, OBJ>(V, envc, O); O.near_mask = 0u; O.near_n = 0; }
  const T cmd0[FW_NUM_ACTUATORS] = {(T)0, (T)0, (T)0, (T)0, (T)0, (T)0};
  LaneAct<T> LA; LA.cmd = (T)0; LA.a = (T)0;
  if (G == 8) lane_act_scatter<T>(S, LA);
  const int chunk = left;
#pragma unroll 1
  while (__ballot(left > 0) != 0ull) {
    const bool stepped = left > 0;
    if (stepped) {
      (void)aviary_step<T, true, G, HASOBJ>(P, C, OC, V, envc, O, S, R, cmd0, tick, (T)0, (T)0, wb, wa, gust, mine, wmask, LA);
      left -= 1;
    }
    if (HASOBJ) obj_capture_step<T, G>(OC, V, stepped, envc, O, S, R, tick);
  }
  if (G == 8) lane_act_gather<T>(S, LA);
  if (chunk > 0 && done + chunk == total) {
    // warm-up complete: end_reset()'s first compute_state and the observation reset() returns (zero action) are part of the
    // shadow, so that the reset that takes it is a copy (state rows, observation row) and nothing else.  The rows of the
    // shadow the worker never writes (stored action, episode return) are zero from the allocation: what a fresh episode holds.
    T nd = (T)0;
    if (OBJ) obj_compute_state<T>(O);
    else {
      T t0[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) t0[k] = V.r[(size_t)(RF_TARGETS + k) * n + envc];
      nd = end_reset<T, G>(P, V, envc, (int32_t)target, S, t0);
      if (COMB) comb_compute_state<T>(OC, O, P.num_targets == 0);
    }
    if (leader) {
      const T a0[4] = {(T)0, (T)0, (T)0, (T)0};
      T* orow = D.sobs + (size_t)env * P.obs_dim;
      if (OBJ) obj_write_obs<T>(P, O, S, a0, [&](int k, T v) { orow[k] = v; });
      else if (COMB) comb_write_obs<T>(P, V, env, O, S, a0, 0, [&](int k, T v) { orow[k] = v; });
      else write_obs<T>(P, V, env, S, a0, 0, [&](int k, T v) { orow[k] = v; });
      V.r[RF_NEW_DIST * n + env] = nd;
    }
  }
  if (chunk > 0 && leader) {
    store_rigid<T>(V, env, S);
    D.is[env] = tick;
    if (HASOBJ) obj_store<T, OBJ>(V, env, O);
    D.sdone[env] = pack_done(target, D.epoch, done + chunk);
  }
}

// Wind-free waypoints: the reset itself is cached (warm state, its observation), what remains per env is sampling the next
// episode's waypoints (3 Philox blocks + 2 sincos per waypoint).  The same hand-off as above lets a worker block do that
// ahead of time into the shadow's waypoint rows; the reset then only copies 3 words per lane (requested at launch start).
template <typename T, int G>
__device__ __forceinline__ void scenario_worker(const Params<T>* __restrict__ Pp, DevState<T> D, int blk) {
  constexpr int EPW = kWave / G;
  const int lane = threadIdx.x, sub = (G == 1) ? 0 : (lane & (G - 1)), row = lane / G;
  const int env = blk * EPW + row;
  const bool active = env < D.n;
  const int envc = active ? env : D.n - 1;
  const unsigned long long req = D.sreq[envc], dn = D.sdone[envc];
  const uint32_t target = (uint32_t)(req >> 32);
  const bool fresh = active && (uint32_t)req != D.epoch && req != ~0ull;          // a request of an earlier launch
  const bool begin = fresh && (uint32_t)(dn >> 32) != target;
  if (__ballot(begin) == 0ull) return;
  if (begin) {
    const Params<T>& P = *Pp;
    Scenario<T> sc;
    sample_scenario_inl<T, G>(Pp, D.rs, (size_t)D.npad, env, target, &sc);
    // ... and the observation the new episode starts with (cached attitude block ++ deltas of the fresh waypoints, zero padded)
    // plus its first distance: the reset that takes this hand-off is then a copy (state = the cached warm state)
    T* orow = D.sobs + (size_t)env * P.obs_dim;
    for (int k = sub; k < P.att_dim; k += G) orow[k] = Pp->warm_obs[k];
    if (G > 1) {
      if (sub < P.ctx) {
        T d[3] = {(T)0, (T)0, (T)0}, b[3] = {(T)0, (T)0, (T)0};
        if (sub < P.num_targets) {
#pragma unroll
          for (int k = 0; k < 3; ++k) d[k] = sc.t_mine[k] - P.warm[k];
          mtv(P.warm_R, d, b);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) orow[P.att_dim + 3 * sub + k] = b[k];
      }
      if (sub == 0) {
        T nd = (T)0;
        if (P.num_targets > 0) {
          const T dx = sc.t_mine[0] - P.warm[0], dy = sc.t_mine[1] - P.warm[1], dz = sc.t_mine[2] - P.warm[2];
          nd = M<T>::sqrt_(dx * dx + dy * dy + dz * dz);
        }
        D.rs[(size_t)RF_NEW_DIST * D.npad + env] = nd;
      }
    }
    if (sub == 0) D.sdone[env] = pack_done(target, D.epoch, 1);
  }
}

// Dev-only per-wave cycle accounting (tools/wave_profile.py builds a second library with -DFW_PROFILE).
#ifdef FW_PROFILE
#define FWP_NOW() ((long long)__builtin_readcyclecounter())
#define FWP(...) __VA_ARGS__
constexpr int kProfSlots = 256, kProfWords = 12;
#ifdef FW_PROFILE_PHASES
constexpr bool getenv_ph = true;     // the words of the wave split carry the phases of capture_body instead (tools/wave_profile.py --phases)
#else
constexpr bool getenv_ph = false;
#endif
#else
#define FWP(...)
#endif

// Per-lane phase of the fused step state machine.
enum Phase : int { PH_STEP = 0, PH_WARM = 1, PH_DONE = 2 };

// K1: one agent step, everything fused.  One wave per workgroup; the wave holds 64/G envs
// (G lanes per env, see fwsim_device.hpp).
//
// The kernel is a per-env state machine around ONE inlined Aviary step:
//   PH_STEP  the env runs its (<= step_ratio) sub-steps; after each one the
//            reference's compute_state / compute_term_trunc_reward are applied;
//   PH_WARM  the env finished its episode mid-launch and (wind acting on the
//            dynamics) is integrating its reset warm-up with a zero setpoint --
//            concurrently with neighbours that are still in PH_STEP;
//   PH_DONE  results latched; the lanes idle until the wave-uniform exit.
// Outputs are latched in registers and stored once at the end.
// GENERAL = wind is on (per-env wind registers, and -- if it acts on the dynamics -- the
// PH_WARM path).  The wind-free instantiation (the headline config) carries none of that.
// OBJ = the ObjLock task (duck / analytic camera / vision shaping, fwsim_objlock.hpp); its state rides in registers.
// WPE = waves per SIMD the instantiation is built for (G = 8 only): 1 = the whole register file (359 registers: shared tick
// constants resident in VGPRs), 2 = capped at 256 registers so that two waves share a SIMD -- the constants then stay
// wave-uniform (scalar loads); used between 8 192 and 65 536 envs, where the 1-wave build would run its waves in two rounds.
// COLLECT = the step waves of fw_collect_step (fwsim_fused.hpp): block indices are offset by the act waves in front, the
// actions are waited for and read coherently, and the epilogue carries the statistics of VecNormalize.step_wait.
// HELP = the workgroup has a capture wave (fwsim_objlock.hpp, "The capture wave"): camera tasks on the 8-lane mapping.
template <typename T, bool GENERAL, int G, int TKIND, int WPE = 1, bool COLLECT = false, bool HELP = false>
__device__ __forceinline__
void step_body(const Params<T>* __restrict__ Pp, const ObjC<T>* __restrict__ OCp, DevState<T> Dg,
               const T* __restrict__ actions, T* __restrict__ obs, T* __restrict__ reward,
               uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated, T* __restrict__ terminal_obs,
               int32_t* __restrict__ info, const CollectArgs* __restrict__ CAp = nullptr) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);
  const Params<T>& P = *Pp;
  constexpr bool OBJ = TKIND == FW_TASK_OBJLOCK;              // duck only
  constexpr bool COMB = TKIND == FW_TASK_WAYPOINT_OBJLOCK;    // waypoints, then the duck
  constexpr bool HASOBJ = OBJ || COMB;
  constexpr int EPW = kWave / G;                     // envs per wave
  // Wind-free waypoints (the headline config): the episode start is the cached warm state for every env, so an
  // auto-reset is deferred to the epilogue -- the obs pass there writes the terminal observation, the new episode's
  // observation is the cached attitude block + the deltas of the freshly sampled waypoints.  No second obs pass, no
  // sampling inside the step loop, and the waves that contain a reset finish with the others.
  constexpr bool DEFER = !GENERAL && TKIND == FW_TASK_WAYPOINTS;
  const int nblk = (Dg.npad + EPW - 1) / EPW;        // step blocks; blocks beyond are shadow workers
  FWP(const long long p_t0 = FWP_NOW(); long long p_reset = 0, p_avi = 0, p_task = 0, p_r1 = 0, p_r2 = 0, p_r3 = 0; int p_nreset = 0, p_nhit = 0;)
  // XCD-aware block -> env-block map (G = 8): a wave touches 64 B of every SoA row, i.e. half a 128-B L2 line, and
  // consecutive workgroups are dealt round-robin to the 8 XCDs, whose L2s are private -- with the identity map both
  // halves of every line are fetched by two different L2s (PMC: 2x the algorithmic read traffic).  Give each XCD a
  // contiguous range of env blocks instead.  (Speed only: nothing depends on where a block really lands.)
  const int bx = (int)blockIdx.x - (COLLECT ? CAp->n_act : 0);       // my index among the step (+ worker) workgroups
  const int wg = bx % nblk;
  const int blk = (G == 8 && (nblk & 7) == 0) ? (wg & 7) * (nblk >> 3) + (wg >> 3) : wg;
  const DevState<T> D = tile_view<T, EPW>(Dg, blk);  // this wave's tile: row stride EPW (a compile-time constant), global env ids
  if ((GENERAL || DEFER) && bx >= nblk) {
    if (GENERAL) shadow_worker<T, G, TKIND>(Pp, OCp, D, blk); else scenario_worker<T, G>(Pp, D, blk);
    FWP(if (D.prof && threadIdx.x == 0) {
      long long* w = D.prof + ((size_t)(D.epoch % kProfSlots) * 2 * nblk + blockIdx.x) * kProfWords;
      w[0] = FWP_NOW() - p_t0; })
    return;
  }
  const int lane = threadIdx.x;
  const int sub = (G == 1) ? 0 : (lane & (G - 1));   // my lane within the env's group
  const int row = lane / G;                          // env slot within the wave
  const bool leader = sub == 0;
  const int env0 = blk * EPW;
  const int env = env0 + row;
  const bool active = env < D.n;
  const int envc = active ? env : D.n - 1;           // inactive lanes shadow the last env and never store
  const size_t n = D.npad;
  const int Dobs = P.obs_dim;
  const int ld = Dobs + 1;

  // Loads first, in the order their results are needed (vmcnt retires in order): the counters that feed
  // the RNG and the target window, then the rigid state, then the launch-resident constants.
  int32_t step_count = D.i[IF_STEP * n + envc];
  int32_t tick = D.i[IF_TICK * n + envc];
  int32_t episode = D.i[IF_EPISODE * n + envc];
  int32_t flags = D.i[IF_FLAGS * n + envc];
  int32_t num_reached = D.i[IF_NUM_REACHED * n + envc];
  // G = 8, wind-free waypoints: lane j of the group keeps waypoint j (num_targets <= 8) and component j & 3 of the
  // action; the current waypoint, the obs deltas and the obs action are fetched by shuffle -- no dependent loads.
  constexpr bool LANE_T = DEFER && G == 8;
  T tmine[3] = {(T)0, (T)0, (T)0};
  T a_keep = (T)0;
  if (LANE_T) {
    if (sub < P.num_targets) {
#pragma unroll
      for (int k = 0; k < 3; ++k) tmine[k] = D.r[(size_t)(RF_TARGETS + 3 * sub + k) * n + envc];
    }
    if (!COLLECT) a_keep = actions[(size_t)envc * 4 + (sub & 3)];
  }
  Rigid<T> S;
  load_rigid<T>(D, envc, S);
  T new_dist = D.r[RF_NEW_DIST * n + envc];
  T ep_return = D.r[RF_EP_RETURN * n + envc];
  // tick constants + this lane's lifting surface (G = 8: resident in VGPRs for the whole launch)
  TickC<T> C; SurfC<T> mine_regs; T wmask;
  load_tick_constants<T, G, DEFER && G == 8 && WPE == 1>(Pp, C, mine_regs, wmask);
  // (Tried for WPE = 2: this lane's surface constants in LDS, read field by field through a volatile reference -- 54 registers
  // fewer to hold, but the allocator spilled as much elsewhere and the ds_reads sit on the tick's critical path: 40.6 -> 50.9 us
  // at 16 384 envs.  surface_wrench / physics_tick keep the template parameter that made the experiment a four-line change.)
  SurfC<T>& mine = mine_regs;

  normalize_quat<T>(S.q);
  T R[9];
  rot_from_unit_quat<T>(S.q, R);
  ObjState<T> O;
  const ObjC<T>& OC = *OCp;
  if (HASOBJ) { obj_load<T, OBJ>(D, envc, O); obj_update_near_mask<T, G>(P, OC, D, envc, O, S); }
  int32_t out_strike = 0;
  // shadow bookkeeping (kernel-boundary hand-off, see shadow_* above)
  unsigned long long sh_req = ~0ull, sh_done = 0ull;
  if ((GENERAL || DEFER) && D.shadow_on) { sh_req = D.sreq[envc]; sh_done = D.sdone[envc]; }
  T wb[3] = {(T)0, (T)0, (T)0}, wa[3] = {(T)0, (T)0, (T)0}, wphase = (T)0;
  if (GENERAL) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { wb[k] = D.r[(RF_WIND + k) * n + envc]; wa[k] = D.r[(RF_WIND + 3 + k) * n + envc]; }
    wphase = D.r[(RF_WIND + 6) * n + envc];
  }
  T gust[2] = {(T)0, (T)1};                          // sin / cos of the gust phase at `tick`
  if (GENERAL) gust_init<T>(P, wphase, tick, gust);
  // fixedwing_base_env.py:325-331
  T rew = (T)-0.1;
  T cmd[FW_NUM_ACTUATORS];
  // COLLECT: the actions of this launch come from the act waves in front of the grid; everything above was loaded while they
  // worked.  Coherent loads from here on: the rows were written (write-through) by a wave of another XCD in this same launch.
  double c_ret = 0.0;                                // COLLECT: my env's discounted-return tracker, fetched while the wave waits anyway
  if (COLLECT) {
    if (active && leader) c_ret = ld_sc1(CAp->S.returns + env);
    collect_wait_actions(*CAp, Dg.epoch, env0, min(EPW, Dg.n - env0));
  }
  {
    const T* ap = actions + (size_t)envc * 4;
    T a4[4];
    if (COLLECT) collect_load_actions<T>(*CAp, ap, a4);
    else {
#pragma unroll
      for (int k = 0; k < 4; ++k) a4[k] = ap[k];
    }
    if (COLLECT && LANE_T) a_keep = (sub & 3) == 0 ? a4[0] : (sub & 3) == 1 ? a4[1] : (sub & 3) == 2 ? a4[2] : a4[3];
    const T sp[4] = { a4[0], a4[1], a4[2], a4[3] * (T)0.5 + (T)0.5 };
#pragma unroll
    for (int c = 0; c < FW_NUM_ACTUATORS; ++c)
      cmd[c] = P.mixer[c][0] * sp[0] + P.mixer[c][1] * sp[1] + P.mixer[c][2] * sp[2] + P.mixer[c][3] * sp[3];
  }
  // G = 8: my surface's actuator state and command stay lane-local during the ticks (see LaneAct)
  LaneAct<T> LA; LA.a = (T)0; LA.cmd = (T)0;
  T cmd_mine = (T)0;
  if (G == 8) { lane_act_scatter<T>(S, LA); cmd_mine = lane_pick5<T>(cmd[0], cmd[1], cmd[2], cmd[3], cmd[4]); LA.cmd = cmd_mine; }

  // current and next waypoint stay in registers (no L2 round trip per sub-step)
  T tcur[3] = {(T)0, (T)0, (T)0}, tnext[3] = {(T)0, (T)0, (T)0};
  const int gbase = lane & ~(G - 1);                  // first lane of my group
  if (LANE_T) {
#pragma unroll
    for (int k = 0; k < 3; ++k) tcur[k] = __shfl(tmine[k], gbase | (num_reached & (G - 1)), kWave);
  } else if (!OBJ) {
    const int i0 = min(num_reached, FW_MAX_TARGETS - 1), i1 = min(num_reached + 1, FW_MAX_TARGETS - 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      tcur[k] = D.r[(size_t)(RF_TARGETS + 3 * i0 + k) * n + envc];
      tnext[k] = D.r[(size_t)(RF_TARGETS + 3 * i1 + k) * n + envc];
    }
  }

  const uint32_t genv = (uint32_t)(P.env_offset + envc);
  // An env that is already done (bare-Gymnasium mode) runs no sub-step: the reference then
  // returns its stale self.state, i.e. the previous action and target view.
  const bool done_at_entry = (flags & (FL_TERM | FL_TRUNC)) != 0;
  int tgt_obs = (flags >> FL_TGT_SHIFT) & 15;        // target index the last compute_state() saw
  flags &= FL_MASK;
  // which action the returned observation shows: 0 = this step's, 1 = the stored (stale) one,
  // 2 = zeros (fresh reset).  Re-read at the end instead of held in registers across the loop.
  int act_src = done_at_entry ? 1 : 0;

  // Motor noise.  G = 8: lane j of the group draws the normals of the step's j-th Aviary
  // step up front (8 lanes -> up to 8 sub-steps, in parallel); they are fetched by shuffle.
  const uint32_t astep0 = (uint32_t)(tick / P.ticks_per_aviary);
  const bool pre_noise = (G == 8) && P.has_noise && (P.step_ratio <= G);
  T nz0 = (T)0, nz1 = (T)0;
  if (pre_noise && !done_at_entry) rng_normal2<T>(P, genv, (uint32_t)episode, astep0 + (uint32_t)sub, nz0, nz1);

  int phase = active ? PH_STEP : PH_DONE;
  // STASH: the outputs of the step (reward, flags, info) wait in LDS -- four words per env behind the tile / the camera's map --
  // from the latch point to the END of the epilogue.  Stored at the latch point (round 2: to keep them out of registers) they
  // sat in front of the epilogue's loads (stored action, waypoints of the observation): vector memory operations retire in order
  // on gfx9, so those loads paid the stores' acknowledgement, ~2 k cycles in every wave.  The wind-free headline kernel has no such
  // loads (actions and waypoints ride in registers) and keeps storing at the latch point.
  // (The camera kernels would gain the same way -- but with the stash their plain builds trip the compiler hazard of section 4
  // of DESIGN.md at the mask-row join, and the hazard-free form of that join costs 3-4 us; they keep the latch-point stores.)
  constexpr bool STASH = (!DEFER && !HASOBJ) || COLLECT;
  double* latch = STASH ? reinterpret_cast<double*>(smem_raw + Dg.stash_off) : nullptr;        // [EPW][4]: reward, done, (num_reached, flags), (strike, step_count)
  int it = 0, warm_left = 0;
  // HELP: the request of the previous sub-step that is still with the capture wave (0 = none), and what this lane owes it:
  // bit 0 the frame-dependent half of that sub-step's task logic, bit 1 a frame, bit 2 "no collision / not out of bounds then",
  // bit 3 (combined) "all waypoints were reached then"; the distance to the duck of that sub-step (ObjLock's reward reads it)
  const Mbox<T> MB = mbox_at<T>(HELP ? Dg.mbox_off : 0);
  uint32_t n_posted = 0, pend_seq = 0;
  int pend = 0;
  T dist_keep = (T)0;
  bool resetting = false;                            // DEFER: auto-reset pending for the epilogue
  bool step_over = active && done_at_entry;          // nothing to simulate: finalise immediately

  FWP(const long long p_t1 = FWP_NOW();)
#pragma unroll 1
  for (;;) {
    FWP(const long long p_a = FWP_NOW();)
    if (phase == PH_STEP && step_over) {
      // ---- end of env.step(): :346, outputs, SB3 worker auto-reset ----
      step_count += 1;
      ep_return += rew;
      phase = PH_DONE;
      if (STASH && leader) {
        latch[4 * row] = (double)rew; latch[4 * row + 1] = (flags & (FL_TERM | FL_TRUNC)) ? 1.0 : 0.0;
        latch[4 * row + 2] = __hiloint2double(num_reached, flags); latch[4 * row + 3] = __hiloint2double(out_strike, step_count);
      }
      // DEFER: the outputs of the step leave for memory here (nothing is kept live across the rest of the loop for them)
      if (!(STASH && !DEFER) && leader) {     // (COLLECT && DEFER stashes for the statistics tail and stores here as well)
        reward[env] = rew;
        terminated[env] = (uint8_t)((flags & FL_TERM) ? 1 : 0);
        truncated[env] = (uint8_t)((flags & FL_TRUNC) ? 1 : 0);
        if (info) {
          int4* ip = reinterpret_cast<int4*>(info + (size_t)env * FW_INFO_DIM);
          ip[0] = make_int4(num_reached, (flags & FL_COLLISION) ? 1 : 0, (flags & FL_OOB) ? 1 : 0, (flags & FL_COMPLETE) ? 1 : 0);
          ip[1] = make_int4(out_strike, OBJ ? out_strike : 0, step_count, 0);
        }
      }
      if (DEFER) { resetting = (flags & (FL_TERM | FL_TRUNC)) && P.auto_reset; FWP(if (resetting) p_nreset += 1;) }
      // GENERAL: a reset whose pre-simulated episode is ready is a copy, done in the epilogue (the observation pass there writes
      // the terminal observation); only a reset without one runs here, with its warm-up in the loop
      const bool take_shadow = GENERAL && (flags & (FL_TERM | FL_TRUNC)) && P.auto_reset && D.shadow_on &&
                               (int)(sh_done & 0xFF) == P.warmup_aviary_steps + 1 && (uint32_t)(sh_done >> 32) == (uint32_t)(episode + 1) &&
                               (uint32_t)((sh_done >> 8) & 0xFFFFFFu) != (D.epoch & 0xFFFFFFu);
      if (take_shadow) { resetting = true; FWP(p_nreset += 1; p_nhit += 1;) }
      if (!DEFER && !take_shadow && (flags & (FL_TERM | FL_TRUNC)) && P.auto_reset) {
        if (G == 8) lane_act_gather<T>(S, LA);          // the terminal observation shows all six actuators
        if (terminal_obs && leader) {
          T* trow = terminal_obs + (size_t)env * Dobs;
          T act_t[4];
          load_action<T, COLLECT>(D, actions, env, act_src, act_t);
          if (OBJ) obj_write_obs<T>(P, O, S, act_t, [&](int k, T v) { trow[k] = v; });
          else if (COMB) comb_write_obs<T>(P, D, env, O, S, act_t, tgt_obs, [&](int k, T v) { trow[k] = v; });
          else write_obs<T>(P, D, env, S, act_t, tgt_obs, [&](int k, T v) { trow[k] = v; });
        }
        T t_mine[3] = {(T)0, (T)0, (T)0};            // first waypoint of the new episode
        FWP(const long long p_ra = FWP_NOW(); p_r1 += p_ra - p_a;)
        {
          if (leader) stat_add(D.stats, FW_CTR_FALLBACKS);
          warm_left = begin_reset<T, G>(P, D, env, S, tick, episode, num_reached, wb, wa, wphase, t_mine);
          if (G > 1) {                                          // the group's lane 0 sampled waypoint 0
            const int src = lane & ~(G - 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) t_mine[k] = __shfl(t_mine[k], src, kWave);
          }
          if (HASOBJ) {
            obj_reset_state<T, OBJ>(O);
            if (OBJ) obj_spawn<T, G>(P, OC, D, env, (uint32_t)episode, leader, O);
            else comb_spawn<T, G>(P, OC, D, env, (uint32_t)episode, leader, O);
          }
        }
        if (G > 1 && warm_left > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the warm-up's camera reads the obstacles
        if (HASOBJ) obj_update_near_mask<T, G>(P, OC, D, env, O, S);                          // new position, new cylinders
        if (GENERAL) gust_init<T>(P, wphase, tick, gust);                                  // new clock, new phase
        if (G == 8) lane_act_scatter<T>(S, LA);                                             // new episode's actuator state
        FWP(const long long p_rb = FWP_NOW(); p_r2 += p_rb - p_ra;)
        step_count = 0; flags = 0; ep_return = (T)0; tgt_obs = 0;
        act_src = 2;
        if (leader) stat_add(D.stats, FW_CTR_RESETS);
        FWP(p_nreset += 1;)
        rot_from_unit_quat<T>(S.q, R);
        if (GENERAL && warm_left > 0) phase = PH_WARM;
        else if (OBJ) obj_compute_state<T>(O);                 // end_reset(): first compute_state of the episode
        else {
          new_dist = end_reset<T, G>(P, D, env, episode, S, t_mine);
          if (COMB) comb_compute_state<T>(OC, O, P.num_targets == 0);
        }
        FWP(p_r3 += FWP_NOW() - p_rb;)
      }
    }
    FWP(const long long p_b = FWP_NOW(); p_reset += p_b - p_a;)
    if (__ballot(phase != PH_DONE) == 0ull) break;   // wave-uniform exit
    // noise of this iteration's Aviary step (all stepping envs of the wave are at sub-step `it`)
    T z0 = (T)0, z1 = (T)0;
    if (pre_noise) {
      const int src = (lane & ~(G - 1)) | (it & (G - 1));
      z0 = __shfl(nz0, src, kWave); z1 = __shfl(nz1, src, kWave);
    }
    const bool stepped = phase != PH_DONE;
    const bool stepping = stepped && (!GENERAL || phase == PH_STEP);
    bool contact = false;
    if (stepped) {
      if (P.has_noise && !pre_noise && stepping)
        rng_normal2<T>(P, genv, (uint32_t)episode, (uint32_t)(tick / P.ticks_per_aviary), z0, z1);
      if (GENERAL) {
        T c_eff[FW_NUM_ACTUATORS];
#pragma unroll
        for (int c = 0; c < FW_NUM_ACTUATORS; ++c) c_eff[c] = stepping ? cmd[c] : (T)0;
        z0 = stepping ? z0 : (T)0; z1 = stepping ? z1 : (T)0;
        LA.cmd = stepping ? cmd_mine : (T)0;
        contact = aviary_step<T, true, G, HASOBJ>(P, C, OC, D, env, O, S, R, c_eff, tick, z0, z1, wb, wa, gust, mine, wmask, LA);   // :339
      } else {
        contact = aviary_step<T, false, G, HASOBJ>(P, C, OC, D, env, O, S, R, cmd, tick, z0, z1, wb, wa, gust, mine, wmask, LA);    // :339
      }
    }
    if constexpr (HELP) {
      // ---- the camera on the capture wave, one sub-step behind the physics ----
      const bool due = stepped && OC.camera_ratio_ticks > 0 && (tick % OC.camera_ratio_ticks) == 0;
      FWP(const long long p_c0 = FWP_NOW(); p_avi += p_c0 - p_b;)
      const uint32_t seq_cur = cap_post<T>(MB, n_posted, due, row, leader, O, S, R);
      FWP(p_r1 += FWP_NOW() - p_c0;)
      // the frame-independent half of THIS sub-step's task logic first: it reads the state as the physics left it; rew / flags
      // of the previous sub-step's other half are complete only below, and the order of their updates is the reference's:
      // (previous: frame half) comes before (this: -100 assignments), see the two halves
      FWP(const long long p_w0 = FWP_NOW();)
      cap_collect<T>(MB, pend_seq, (pend & 2) != 0, row, O, D.stats);
      FWP(p_task += FWP_NOW() - p_w0;)
      pend_seq = 0;
      auto frame_half = [&](int keep) {             // compute_state + the rewards that read the frame (fixedwing_objlock_env.py:253-287, 296-372)
        if (OBJ) {
          obj_compute_state<T>(O);
          if ((keep & 4) && obj_reward<T>(OC, P.sparse, O, dist_keep, rew)) { flags |= FL_TERM | FL_COMPLETE; out_strike = 1; }
        } else {
          comb_compute_state<T>(OC, O, (keep & 8) != 0);
          if (keep & 4) {
            if (!(keep & 8)) comb_obstacle_penalty<T>(OC, O, (T)1, rew);
            else {
              comb_obstacle_penalty<T>(OC, O, (T)0.5, rew);
              if (comb_duck_reward<T>(OC, P.sparse, O, rew)) { flags |= FL_TERM | FL_COMPLETE; out_strike = 1; }
            }
          }
        }
      };
      FWP(const long long p_f0 = FWP_NOW();)
      if (pend & 1) frame_half(pend);
      FWP(const long long p_f1 = FWP_NOW(); p_r2 += p_f1 - p_f0;)
      pend = 0;
      int keep = 0;
      bool now = false;                             // this lane needs its frame before the next sub-step's physics
      if (stepped) {
        if (stepping) {
          keep = 1 | (due ? 2 : 0);
          int nleft = 0;
          T old_dist = new_dist;
          if (COMB) {
            nleft = P.num_targets - num_reached;
            if (nleft > 0) {
              T dx = tcur[0] - S.p[0], dy = tcur[1] - S.p[1], dz = tcur[2] - S.p[2];
              new_dist = M<T>::sqrt_(dx * dx + dy * dy + dz * dz);
            }
            tgt_obs = num_reached;
            if (nleft == 0) keep |= 8;
          }
          if (step_count > P.max_steps) flags |= FL_TRUNC;
          if (contact) { rew = (T)-100; flags |= FL_COLLISION | FL_TERM; }
          if (S.p[0] * S.p[0] + S.p[1] * S.p[1] + S.p[2] * S.p[2] > P.dome * P.dome) { rew = (T)-100; flags |= FL_OOB | FL_TERM; }
          bool can_strike = false;
          if (!(flags & (FL_COLLISION | FL_OOB))) {
            keep |= 4;
            if (OBJ) {
              T dx = O.duck[0] - S.p[0], dy = O.duck[1] - S.p[1], dz = O.duck[2] - S.p[2];
              dist_keep = M<T>::sqrt_(dx * dx + dy * dy + dz * dz);
              // obj_reward: a strike needs lock_steps >= hold_steps AFTER this sub-step's (at most +1) update, and the distance
              can_strike = O.lock_steps + (T)1 >= (T)OC.hold_steps && dist_keep <= OC.strike_dist;
            } else if (nleft > 0) {
              if (!P.sparse) {
                T progress = (old_dist != (T)0) ? (old_dist - new_dist) : (T)0;
                rew += M<T>::fmax_((T)3 * progress, (T)0);
                rew += M<T>::rcp_(new_dist);
              }
              if (new_dist < P.reach) {
                rew = (T)100;
                num_reached += 1;
                if (num_reached == P.num_targets) flags &= ~(FL_TERM | FL_TRUNC);          // :297-300
                const int i1 = min(num_reached + 1, FW_MAX_TARGETS - 1);
#pragma unroll
                for (int k = 0; k < 3; ++k) { tcur[k] = tnext[k]; tnext[k] = D.r[(size_t)(RF_TARGETS + 3 * i1 + k) * n + env]; }
              }
            } else {
              flags &= ~FL_TERM;                                                            // :306
              // comb_duck_reward: a strike needs the duck phase (which this sub-step's frame may switch on) and the lock counter
              can_strike = O.lock_steps + (T)1 >= (T)OC.hold_steps;
            }
          }
          step_over = (it + 1 >= P.step_ratio) || (flags & (FL_TERM | FL_TRUNC));
          now = can_strike || step_over;
        } else {
          now = true;                               // warm-up of an in-kernel reset: its last sub-step reads the frame
        }
      }
      auto finish_now = [&]() {                     // the rest of this sub-step for a lane whose frame (if it is due one) is in O
        if (stepping) {
          frame_half(keep);
          step_over = (it + 1 >= P.step_ratio) || (flags & (FL_TERM | FL_TRUNC));
        } else {
          warm_left -= 1;
          if (warm_left == 0) {
            if (OBJ) obj_compute_state<T>(O);
            else { new_dist = end_reset<T, G>(P, D, env, episode, S); comb_compute_state<T>(OC, O, P.num_targets == 0); }
            phase = PH_DONE;
          }
        }
      };
      // Wait here if somebody's agent step ends at this sub-step (or could, by a strike) -- ObjLock: ... AND that lane reads a
      // frame of this very sub-step; a lane that ends on the frame it already has finishes at once and the others' frames come
      // back behind the next sub-step's physics.  (The combined kernel keeps the simpler rule: with the finer one its build
      // trips the spill-before-exec-restore check of tools/check_isa.py.)
      FWP(const long long p_f2 = FWP_NOW(); p_r3 += p_f2 - p_f1;)
      bool wait_here;
      if constexpr (OBJ) wait_here = __ballot(now && due) != 0ull; else wait_here = __ballot(now) != 0ull;
      if (wait_here) {
        FWP(const long long p_w1 = FWP_NOW();)
        cap_collect<T>(MB, seq_cur, due, row, O, D.stats);
        FWP(p_task += FWP_NOW() - p_w1; p_nhit += 1;)
        FWP(const long long p_f3 = FWP_NOW();)
        if (stepped) finish_now();               // ... and the sub-step is finished for everybody, as in the one-wave kernel
        FWP(p_r2 += FWP_NOW() - p_f3;)
      } else {
        FWP(const long long p_f3 = FWP_NOW();)
        if constexpr (OBJ) { if (now) finish_now(); else pend = keep; }
        else pend = keep;
        FWP(p_r2 += FWP_NOW() - p_f3;)
        pend_seq = seq_cur;
      }
    } else {
    if (HASOBJ) obj_capture_step<T, G, COLLECT>(OC, D, stepped, envc, O, S, R, tick);     // the camera, by the whole wave
    FWP(const long long p_c = FWP_NOW(); p_avi += p_c - p_b;)
    if (stepped) {
      if (stepping && OBJ) {
        obj_compute_state<T>(O);                                                                   // :342
        // compute_base_term_trunc_reward(): :296-312
        if (step_count > P.max_steps) flags |= FL_TRUNC;
        if (contact) { rew = (T)-100; flags |= FL_COLLISION | FL_TERM; }
        if (S.p[0] * S.p[0] + S.p[1] * S.p[1] + S.p[2] * S.p[2] > P.dome * P.dome) { rew = (T)-100; flags |= FL_OOB | FL_TERM; }
        if (!(flags & (FL_COLLISION | FL_OOB))) {                                                  // :293-294
          T dx = O.duck[0] - S.p[0], dy = O.duck[1] - S.p[1], dz = O.duck[2] - S.p[2];
          if (obj_reward<T>(OC, P.sparse, O, M<T>::sqrt_(dx * dx + dy * dy + dz * dz), rew)) {
            flags |= FL_TERM | FL_COMPLETE; out_strike = 1;
          }
        }
        step_over = (it + 1 >= P.step_ratio) || (flags & (FL_TERM | FL_TRUNC));
      } else if (stepping && COMB) {
        // compute_state() :197-276
        const int nleft = P.num_targets - num_reached;
        const T old_dist = new_dist;
        if (nleft > 0) {
          T dx = tcur[0] - S.p[0], dy = tcur[1] - S.p[1], dz = tcur[2] - S.p[2];
          new_dist = M<T>::sqrt_(dx * dx + dy * dy + dz * dz);
        }
        tgt_obs = num_reached;
        comb_compute_state<T>(OC, O, nleft == 0);
        // compute_term_trunc_reward() :278-343
        if (step_count > P.max_steps) flags |= FL_TRUNC;
        if (contact) { rew = (T)-100; flags |= FL_COLLISION | FL_TERM; }
        if (S.p[0] * S.p[0] + S.p[1] * S.p[1] + S.p[2] * S.p[2] > P.dome * P.dome) { rew = (T)-100; flags |= FL_OOB | FL_TERM; }
        if (!(flags & (FL_COLLISION | FL_OOB))) {
          if (nleft > 0) {
            if (!P.sparse) {
              T progress = (old_dist != (T)0) ? (old_dist - new_dist) : (T)0;
              rew += M<T>::fmax_((T)3 * progress, (T)0);
              rew += M<T>::rcp_(new_dist);
            }
            if (new_dist < P.reach) {
              rew = (T)100;
              num_reached += 1;
              if (num_reached == P.num_targets) flags &= ~(FL_TERM | FL_TRUNC);          // :297-300
              const int i1 = min(num_reached + 1, FW_MAX_TARGETS - 1);
#pragma unroll
              for (int k = 0; k < 3; ++k) { tcur[k] = tnext[k]; tnext[k] = D.r[(size_t)(RF_TARGETS + 3 * i1 + k) * n + env]; }
            }
            comb_obstacle_penalty<T>(OC, O, (T)1, rew);
          } else {
            flags &= ~FL_TERM;                                                            // :306
            comb_obstacle_penalty<T>(OC, O, (T)0.5, rew);
            if (comb_duck_reward<T>(OC, P.sparse, O, rew)) { flags |= FL_TERM | FL_COMPLETE; out_strike = 1; }
          }
        }
        step_over = (it + 1 >= P.step_ratio) || (flags & (FL_TERM | FL_TRUNC));
      } else if (stepping) {
        // compute_state(): WaypointHandler.distance_to_targets side effects
        const int nleft = P.num_targets - num_reached;
        const T old_dist = new_dist;
        if (P.task != FW_TASK_OBJLOCK && nleft > 0) {
          T dx = tcur[0] - S.p[0], dy = tcur[1] - S.p[1], dz = tcur[2] - S.p[2];
          new_dist = M<T>::sqrt_(dx * dx + dy * dy + dz * dz);
        }
        tgt_obs = num_reached;
        // compute_base_term_trunc_reward(): :296-312
        if (step_count > P.max_steps) flags |= FL_TRUNC;
        if (contact) { rew = (T)-100; flags |= FL_COLLISION | FL_TERM; }
        if (S.p[0] * S.p[0] + S.p[1] * S.p[1] + S.p[2] * S.p[2] > P.dome * P.dome) { rew = (T)-100; flags |= FL_OOB | FL_TERM; }
        // waypoint reward (upstream FixedwingWaypointsEnv; mirrored at fixedwing_waypoint_objlock_env.py:286-294)
        if (P.task != FW_TASK_OBJLOCK && nleft > 0) {
          if (!P.sparse) {
            T progress = (old_dist != (T)0) ? (old_dist - new_dist) : (T)0;
            rew += M<T>::fmax_((T)3 * progress, (T)0);
            rew += M<T>::rcp_(new_dist);
          }
          if (new_dist < P.reach) {
            rew = (T)100;
            num_reached += 1;
            if (num_reached == P.num_targets) flags |= FL_TRUNC | FL_COMPLETE;
            if (LANE_T) {                                                   // advance_targets(): the next waypoint's lane hands it over
#pragma unroll
              for (int k = 0; k < 3; ++k) tcur[k] = __shfl(tmine[k], gbase | (num_reached & (G - 1)), kWave);
            } else {
              const int i1 = min(num_reached + 1, FW_MAX_TARGETS - 1);     // advance_targets(): shift the register window
#pragma unroll
              for (int k = 0; k < 3; ++k) { tcur[k] = tnext[k]; tnext[k] = D.r[(size_t)(RF_TARGETS + 3 * i1 + k) * n + env]; }
            }
          }
        }
        step_over = (it + 1 >= P.step_ratio) || (flags & (FL_TERM | FL_TRUNC));     // :334-337
      } else {
        warm_left -= 1;
        if (warm_left == 0) {
          if (OBJ) obj_compute_state<T>(O);
          else { new_dist = end_reset<T, G>(P, D, env, episode, S); if (COMB) comb_compute_state<T>(OC, O, P.num_targets == 0); }
          phase = PH_DONE;
        }
      }
      FWP(p_task += FWP_NOW() - p_c;)
    }
    }
    it += 1;
  }
  FWP(const long long p_t2 = FWP_NOW();)
  // G = 8: waypoints sampled by sibling lanes during an in-launch reset are read back below
  if (G > 1 && !DEFER) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  if (G == 8) lane_act_gather<T>(S, LA);             // S.act[0..4] current again for the observation and the state store
  // DEFER: the waypoint a worker block sampled ahead of time for my next episode, requested only by resetting envs and only
  // now -- its latency hides behind the observation pass below
  T tpre[3] = {(T)0, (T)0, (T)0};
  const bool pre = DEFER && resetting && D.shadow_on && (int)(sh_done & 0xFF) == 1 && (uint32_t)(sh_done >> 32) == (uint32_t)(episode + 1) &&
                   (uint32_t)((sh_done >> 8) & 0xFFFFFFu) != (D.epoch & 0xFFFFFFu);
  if (pre) {      // (every lane issues: rows >= num_targets of the shadow are zero and are not stored back)
#pragma unroll
    for (int k = 0; k < 3; ++k) pf_issue(tpre[k], &D.rs[(size_t)(RF_TARGETS + 3 * sub + k) * n + env]);
  }
  // ... and so is the cached attitude block of the observation a reset returns (lane `sub` holds its words sub, sub + G, ...)
  constexpr bool WO = DEFER && G == 8;               // (one lane per env: the block is copied straight from memory below)
  constexpr int kWO = WO ? 32 / G : 1;
  T wo[kWO];
  if (WO && resetting) {
#pragma unroll
    for (int j = 0; j < kWO; ++j) { const int k = sub + j * G; wo[j] = (k < P.att_dim) ? vload(&Pp->warm_obs[k]) : (T)0; }
  }

  // (pre: the worker also left the first observation row and the first distance: requested here as well)
  constexpr int kDOW = 4;                            // words per lane of the first observation row (obs_dim <= 32: every waypoints config up to ctx 3)
  T dow[kDOW], dnd = (T)0;
  if (WO && pre && Dobs <= kDOW * G) {
#pragma unroll
    for (int j = 0; j < kDOW; ++j) { const int k = sub + j * G; dow[j] = (T)0; if (k < Dobs) pf_issue(dow[j], &Dg.sobs[(size_t)env * Dobs + k]); }
    pf_issue(dnd, &D.rs[(size_t)RF_NEW_DIST * n + env]);
  }

  // GENERAL: the rows of a shadow that is being taken, fetched by the env's G lanes before the observation pass (their round
  // trip hides behind it): word w of the [RF_COUNT] state column goes to lane w % G; the first observation likewise
  constexpr int kRows = HASOBJ ? RF_COUNT : RF_TASK;  // the waypoint task has no task tail
  constexpr int kCW = GENERAL ? (kRows + G - 1) / G : 1, kOW = GENERAL ? (kMaxObs + G - 1) / G : 1;
  constexpr bool SWAP_REGS = GENERAL && G == 8;       // G = 1: one lane would hold all 171 words; it copies through memory below
  T cw[kCW], ow[kOW];
  int32_t tick_new = 0;
  // PF_GEN: the same hand-issued form for the shadow take-over of the wind / camera kernels.  Measured on the wind kernel: 23.35 ->
  // 23.49 us (15 more values live across the observation pass cost as much as the sunk loads), so it stays off; the camera
  // kernels sit at the edge of the register file and were not tried.
  constexpr bool PF_GEN = false;
  if (GENERAL && resetting) {
    if (SWAP_REGS) {
#pragma unroll
      for (int j = 0; j < kCW; ++j) {
        const int w = sub + j * G; cw[j] = (T)0;
        if (w < kRows) { if (PF_GEN) pf_issue(cw[j], &D.rs[(size_t)w * n + env]); else cw[j] = D.rs[(size_t)w * n + env]; }
      }
#pragma unroll
      for (int j = 0; j < kOW; ++j) {
        const int k = sub + j * G; ow[j] = (T)0;
        if (k < Dobs) { if (PF_GEN) pf_issue(ow[j], &Dg.sobs[(size_t)env * Dobs + k]); else ow[j] = Dg.sobs[(size_t)env * Dobs + k]; }
      }
    }
    tick_new = D.is[env];
  }
  const int ep_want = episode + 1 + ((GENERAL && resetting) ? 1 : 0);     // a reset taken below asks for the episode after next
  if ((GENERAL || DEFER) && D.shadow_on && active && leader && (uint32_t)(sh_req >> 32) != (uint32_t)ep_want)
    D.sreq[env] = ((unsigned long long)(uint32_t)ep_want << 32) | (unsigned long long)D.epoch;   // ask for the next episode
  T act_obs[4] = {(T)0, (T)0, (T)0, (T)0};
  if (LANE_T) {
    // every lane runs the pass (same issue cost as one lane), the leader stores; action and waypoints come by shuffle
#pragma unroll
    for (int k = 0; k < 4; ++k) act_obs[k] = __shfl(a_keep, gbase | k, kWave);
    if (act_src == 1) {                              // bare-Gymnasium stale view: the stored action
#pragma unroll
      for (int k = 0; k < 4; ++k) act_obs[k] = D.r[(size_t)(RF_ACTION + k) * n + envc];
    }
    T Ro[9];
    int o = write_obs_attitude<T, true>(P, S, act_obs, Ro, [&](int k, T v) { if (leader) tile[row * ld + k] = v; });
#pragma unroll 1
    for (int i = 0; i < P.ctx; ++i) {
      const int t = tgt_obs + i;
      T tw[3], d[3], b[3] = {(T)0, (T)0, (T)0};
#pragma unroll
      for (int k = 0; k < 3; ++k) tw[k] = __shfl(tmine[k], gbase | (t & (G - 1)), kWave);
      if (t < P.num_targets) {
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] = tw[k] - S.p[k];
        mtv(Ro, d, b);
      }
      if (leader) { tile[row * ld + o] = b[0]; tile[row * ld + o + 1] = b[1]; tile[row * ld + o + 2] = b[2]; }
      o += 3;
    }
  } else if (active && leader) {
    load_action<T, COLLECT>(D, actions, env, act_src, act_obs);
    if (OBJ) obj_write_obs<T>(P, O, S, act_obs, [&](int k, T v) { tile[row * ld + k] = v; });
    else if (COMB) comb_write_obs<T>(P, D, env, O, S, act_obs, tgt_obs, [&](int k, T v) { tile[row * ld + k] = v; });
    else write_obs<T>(P, D, env, S, act_obs, tgt_obs, [&](int k, T v) { tile[row * ld + k] = v; });
  }
  FWP(const long long p_e0 = FWP_NOW();)
  if (DEFER && pre) {
    // the hand-issued prefetches: ONE wait, here -- behind the observation pass and AHEAD of the terminal-observation stores below
    // (vmcnt counts stores too: a wait placed after them pays their acknowledgement, ~2 k cycles) -- tied to every value
    pf_wait(tpre[0], tpre[1], tpre[2], dnd);
    if (WO) pf_tie(dow[0], dow[1], dow[2], dow[3]);
  }
  if (DEFER && resetting) {                          // group-uniform: all G lanes of the env take part
    // the row just written is the terminal observation: move it out before the new episode's row replaces it
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (terminal_obs) {
      T* trow = terminal_obs + (size_t)env * Dobs;
      for (int k = sub; k < Dobs; k += G) trow[k] = tile[row * ld + k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    FWP(const long long p_e1 = FWP_NOW(); p_r1 = p_e1 - p_e0;)      // obs pass end -> terminal observation copied out
    episode += 1;
    if (leader) { stat_add(D.stats, FW_CTR_RESETS); stat_add(D.stats, pre ? FW_CTR_SCENARIO_HITS : FW_CTR_FALLBACKS); }
    Scenario<T> sc;
    // (the obs pass above read the old waypoints; they are overwritten here)
    if (pre) {                                         // sampled ahead of time by a worker block of an earlier launch
#pragma unroll
      for (int k = 0; k < 3; ++k) sc.t_mine[k] = tpre[k];
      if (sub < P.num_targets) {
#pragma unroll
        for (int k = 0; k < 3; ++k) D.r[(size_t)(RF_TARGETS + 3 * sub + k) * n + env] = tpre[k];
      }
    } else {
      sample_scenario_inl<T, G>(Pp, D.r, n, env, (uint32_t)episode, &sc);
    }
    const int nt = min(P.ctx, P.num_targets);
    T t0[3] = {(T)0, (T)0, (T)0};
    const bool copied = WO && pre && Dobs <= kDOW * G;    // the worker left the whole first observation row: nothing to compute
    if (copied) {
#pragma unroll
      for (int j = 0; j < kDOW; ++j) { const int k = sub + j * G; if (k < Dobs) tile[row * ld + k] = dow[j]; }
    }
#pragma unroll 1
    for (int i = 0; i < (copied ? 0 : P.ctx); ++i) {
      T tw[3] = {(T)0, (T)0, (T)0};
      if (G > 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) tw[k] = __shfl(sc.t_mine[k], (lane & ~(G - 1)) | (i & (G - 1)), kWave);    // lane i of the group sampled waypoint i
      } else if (i < nt) {
#pragma unroll
        for (int k = 0; k < 3; ++k) tw[k] = D.r[(size_t)(RF_TARGETS + 3 * i + k) * n + env];                      // own stores, program order
      }
      T d[3] = {(T)0, (T)0, (T)0}, b[3] = {(T)0, (T)0, (T)0};
      if (i < nt) {
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] = tw[k] - P.warm[k];
        mtv(P.warm_R, d, b);
        if (i == 0) { t0[0] = tw[0]; t0[1] = tw[1]; t0[2] = tw[2]; }
      }
      if (leader) { tile[row * ld + P.att_dim + 3 * i] = b[0]; tile[row * ld + P.att_dim + 3 * i + 1] = b[1]; tile[row * ld + P.att_dim + 3 * i + 2] = b[2]; }
    }
    FWP(const long long p_e2 = FWP_NOW(); p_r2 = p_e2 - p_e1;)      // new waypoints stored, their deltas in the tile
    if (WO && !copied) {
#pragma unroll
      for (int j = 0; j < kWO; ++j) { const int k = sub + j * G; if (k < P.att_dim) tile[row * ld + k] = wo[j]; }   // the env's lanes share the cached attitude block
    }
    if (!copied) for (int k = sub + (WO ? kWO * G : 0); k < P.att_dim; k += G) tile[row * ld + k] = Pp->warm_obs[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { S.p[k] = P.warm[k]; S.v[k] = P.warm[7 + k]; S.w[k] = P.warm[10 + k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) S.q[k] = P.warm[3 + k];
#pragma unroll
    for (int k = 0; k < FW_NUM_ACTUATORS; ++k) S.act[k] = P.warm[13 + k];
    tick = P.warm_ticks;
    step_count = 0; flags = 0; ep_return = (T)0; tgt_obs = 0; num_reached = 0;
    new_dist = copied ? dnd : (T)0;
    if (P.num_targets > 0 && !copied) {
      if (G == 1 && nt == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) t0[k] = D.r[(size_t)(RF_TARGETS + k) * n + env];
      } else if (G > 1 && nt == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) t0[k] = __shfl(sc.t_mine[k], lane & ~(G - 1), kWave);
      }
      T dx = t0[0] - S.p[0], dy = t0[1] - S.p[1], dz = t0[2] - S.p[2];
      new_dist = M<T>::sqrt_(dx * dx + dy * dy + dz * dz);                 // end_reset -> compute_state
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) act_obs[k] = (T)0;
    FWP(p_r3 = FWP_NOW() - p_e2;)                                   // cached block + warm state + first distance
  }
  if (PF_GEN && resetting) {                         // ONE wait for the hand-issued loads, ahead of the terminal-observation stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < kCW; ++j) asm volatile("" : "+v"(cw[j]));
#pragma unroll
    for (int j = 0; j < kOW; ++j) asm volatile("" : "+v"(ow[j]));
  }
  if (GENERAL && resetting) {                        // group-uniform: all G lanes of the env take part
    // the row just written is the terminal observation: move it out, then the new episode's row and state replace the old
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (terminal_obs) {
      T* trow = terminal_obs + (size_t)env * Dobs;
      for (int k = sub; k < Dobs; k += G) trow[k] = tile[row * ld + k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (SWAP_REGS) {
#pragma unroll
      for (int j = 0; j < kOW; ++j) { const int k = sub + j * G; if (k < Dobs) tile[row * ld + k] = ow[j]; }
#pragma unroll
      for (int j = 0; j < kCW; ++j) { const int w = sub + j * G; if (w < kRows) D.r[(size_t)w * n + env] = cw[j]; }
    } else {
      for (int k = 0; k < Dobs; ++k) tile[row * ld + k] = Dg.sobs[(size_t)env * Dobs + k];
      copy_words<T, G>(D.r, D.rs, 0, kRows, n, env);
    }
    if (leader) { stat_add(D.stats, FW_CTR_RESETS); stat_add(D.stats, FW_CTR_SHADOW_HITS); }
    episode += 1; tick = tick_new;
    step_count = 0; flags = 0; tgt_obs = 0; num_reached = 0;
  }
  if (active && leader) {
    if (!(GENERAL && resetting)) {
      store_rigid<T>(D, env, S);
      if (HASOBJ) obj_store<T, OBJ>(D, env, O);
#pragma unroll
      for (int k = 0; k < 4; ++k) D.r[(RF_ACTION + k) * n + env] = act_obs[k];
      D.r[RF_NEW_DIST * n + env] = new_dist;
      D.r[RF_EP_RETURN * n + env] = ep_return;
    }
    D.i[IF_STEP * n + env] = step_count;
    D.i[IF_TICK * n + env] = tick;
    D.i[IF_EPISODE * n + env] = episode;
    D.i[IF_FLAGS * n + env] = flags | (tgt_obs << FL_TGT_SHIFT);
    D.i[IF_NUM_REACHED * n + env] = num_reached;
  }
  if (STASH && !DEFER && active && leader) {         // the outputs parked at the latch point: behind every load of the epilogue
    const double w2 = latch[4 * row + 2], w3 = latch[4 * row + 3];
    const int o_nr = __double2hiint(w2), o_fl = __double2loint(w2), o_st = __double2hiint(w3), o_sc = __double2loint(w3);
    reward[env] = (T)latch[4 * row];
    terminated[env] = (uint8_t)((o_fl & FL_TERM) ? 1 : 0);
    truncated[env] = (uint8_t)((o_fl & FL_TRUNC) ? 1 : 0);
    if (info) {
      int4* ip = reinterpret_cast<int4*>(info + (size_t)env * FW_INFO_DIM);
      ip[0] = make_int4(o_nr, (o_fl & FL_COLLISION) ? 1 : 0, (o_fl & FL_OOB) ? 1 : 0, (o_fl & FL_COMPLETE) ? 1 : 0);
      ip[1] = make_int4(o_st, OBJ ? o_st : 0, o_sc, 0);
    }
  }
  if (HELP) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }   // (the capture wave takes no part: no s_barrier)
  else __syncthreads();
  // (COLLECT: the partial sums first -- the fold waves at the end of the launch are waiting for them, nobody for the observation rows)
  if (COLLECT) collect_stats_tail<T>(*CAp, D.epoch, tile, ld, min(EPW, D.n - env0), wg, nblk, active && leader, latch[4 * row], latch[4 * row + 1] != 0.0, env, c_ret);
  flush_obs_tile<T>(tile, ld, obs, env0, EPW, D.n, Dobs);
  if (COLLECT && active && leader) collect_clear_actions<T>(const_cast<T*>(actions) + (size_t)env * 4);      // "not there yet" for the next launch
  FWP(long long p_capmax = HASOBJ ? O.p_cap : 0; const int p_ncapw = HASOBJ ? __popcll(__ballot(leader && O.p_ncap > 0)) : 0;)
  FWP(if (D.prof) {
    for (int o = 32; o > 0; o >>= 1) p_capmax = max(p_capmax, (long long)__shfl_xor((long long)p_capmax, o, kWave));
    const int nr = __popcll(__ballot(leader && p_nreset > 0)), nh = __popcll(__ballot(leader && p_nhit > 0));
    for (int o = 32; o > 0; o >>= 1) {               // the reset split is per lane (divergent region): wave max
      p_r1 = max(p_r1, (long long)__shfl_xor((long long)p_r1, o, kWave));
      p_r2 = max(p_r2, (long long)__shfl_xor((long long)p_r2, o, kWave));
      p_r3 = max(p_r3, (long long)__shfl_xor((long long)p_r3, o, kWave));
    }
    if (lane == 0) {
      long long* w = D.prof + ((size_t)(D.epoch % kProfSlots) * 2 * nblk + blockIdx.x) * kProfWords;
      const long long t3 = FWP_NOW();
      w[0] = t3 - p_t0; w[1] = p_t1 - p_t0; w[2] = p_reset; w[3] = p_avi; w[4] = p_task; w[5] = t3 - p_t2; w[6] = it | (nr << 8) | (nh << 16); w[7] = p_t0; w[8] = p_r1; w[9] = p_r2; w[10] = p_r3;
      w[11] = p_capmax | ((long long)p_ncapw << 48);
      if (HELP) { w[8] = p_nhit; w[9] = n_posted; w[7] = p_r1; w[10] = p_r2; w[11] = p_r3; }
      else if (HASOBJ) { w[8] = O.p_capm[0]; w[9] = O.p_capm[1]; w[10] = O.p_capm[2]; w[7] = O.p_capm[3];
        if (getenv_ph) { w[1] = O.p_ph[0]; w[3] = O.p_ph[1]; w[4] = O.p_ph[2]; w[5] = O.p_ph[3]; w[9] = O.p_ph[4]; w[6] = O.p_ph[5]; } }     // (the reset split is unused by these kernels)
    } })
}

#ifndef FW_G1_WAVES
#define FW_G1_WAVES 2   // env-per-lane mapping: cap registers for >= 2 waves/SIMD (TLP hides the scalar-load and fp64 latency)
#endif
#define FW_STEP_ARGS const Params<T>* __restrict__ Pp, const ObjC<T>* __restrict__ OCp, DevState<T> D,                  \
    const T* __restrict__ actions, T* __restrict__ obs, T* __restrict__ reward, uint8_t* __restrict__ terminated,        \
    uint8_t* __restrict__ truncated, T* __restrict__ terminal_obs, int32_t* __restrict__ info
#define FW_STEP_PASS Pp, OCp, D, actions, obs, reward, terminated, truncated, terminal_obs, info
// every step kernel: launch index from the workgroup's own device-side counter, advanced when the workgroup is finished
#define FW_STEP_RUN(...) do { D.epoch = launch_index(D.lctr); step_body<__VA_ARGS__>(FW_STEP_PASS); launch_done(D.lctr, D.epoch); } while (0)
// latency mapping (8 lanes per env): one wave per SIMD by construction, let the allocator use the whole file
template <typename T, bool GENERAL>
__global__ __launch_bounds__(kWave) void fw_step_kernel_g8(FW_STEP_ARGS) { FW_STEP_RUN(T, GENERAL, 8, FW_TASK_WAYPOINTS); }
// ... and the same mapping capped at 256 registers: two waves per SIMD (8 192 < N <= 65 536 envs)
template <typename T, bool GENERAL>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fw_step_kernel_g8w2(FW_STEP_ARGS) { FW_STEP_RUN(T, GENERAL, 8, FW_TASK_WAYPOINTS, 2); }
// throughput mapping (one lane per env)
template <typename T, bool GENERAL>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(FW_G1_WAVES, FW_G1_WAVES)))
void fw_step_kernel_g1(FW_STEP_ARGS) { FW_STEP_RUN(T, GENERAL, 1, FW_TASK_WAYPOINTS); }
// ObjLock task (always the GENERAL path: its training config has wind)
template <typename T, int TKIND>
__global__ __launch_bounds__(kWave) void fw_step_kernel_obj_g8(FW_STEP_ARGS) { FW_STEP_RUN(T, true, 8, TKIND); }
template <typename T, int TKIND>
__global__ __launch_bounds__(kWave) void fw_step_kernel_obj_g1(FW_STEP_ARGS) { FW_STEP_RUN(T, true, 1, TKIND); }
#ifndef FW_HELP_CHUNK
#define FW_HELP_CHUNK 2      // warm-up Aviary steps of shadow work a capture wave does per launch
#endif
// ... with a capture wave beside every step wave (fwsim_objlock.hpp, "The capture wave"): 128 threads, wave 0 steps, wave 1 serves
template <typename T, int TKIND>
__global__ __launch_bounds__(2 * kWave) void fw_step_kernel_obj_g8h(FW_STEP_ARGS) {
  const Mbox<T> MB = mbox_at<T>(D.mbox_off);
  if (threadIdx.x < 4) MB.ctl[threadIdx.x] = 0u;
  __syncthreads();                                   // the ONE s_barrier both waves meet at: the mailbox words are zero
  if (threadIdx.x >= kWave) {
    // The grid of this kernel is the step workgroups alone -- with one wave per SIMD (the whole register file each) 512 step waves
    // + 512 capture waves fill the chip, and separate worker workgroups would run BEHIND them (measured: + 15 us).  The capture
    // wave is the shadow worker of its tile as well: one warm-up Aviary step per launch (instead of step_ratio), at the start,
    // while the step wave is in its prologue and first sub-step and cannot have posted anything yet.
    const int nblk = (D.npad + 7) / 8, bx = (int)blockIdx.x % nblk;
    const int blk = ((nblk & 7) == 0) ? (bx & 7) * (nblk >> 3) + (bx >> 3) : bx;       // (step_body's XCD-aware map)
    D.epoch = launch_index(D.lctr);
    const DevState<T> V = tile_view<T, 8>(D, blk);
    // (Tried: the mailbox looked at in front of every warm-up Aviary step of the shadow work, requests first -- combined 60.2 ->
    // 61.5 us: the wave is short of time, not badly ordered.)
    FWP(const long long h0 = FWP_NOW(); long long hcap = 0; int hreq = 0;)
    if (D.shadow_on) shadow_worker<T, 8, TKIND>(Pp, OCp, V, blk, FW_HELP_CHUNK);
    FWP(const long long h1 = FWP_NOW();)
    capture_server<T>(MB, 0u, *OCp, V, blk * 8 FWP(, &hcap, &hreq));
    FWP(if (D.prof && threadIdx.x == kWave) {
      long long* w = D.prof + ((size_t)(D.epoch % kProfSlots) * 2 * nblk + nblk + blockIdx.x) * kProfWords;
      w[0] = (h1 - h0) + hcap; w[1] = h1 - h0; w[2] = hcap; w[3] = hreq; w[4] = FWP_NOW() - h0; })
    return;
  }
  D.epoch = launch_index(D.lctr);
  step_body<T, true, 8, TKIND, 1, false, true>(FW_STEP_PASS);
  launch_done(D.lctr, D.epoch);
  if (threadIdx.x == 0) __hip_atomic_store(MB.ctl + 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);     // every path of wave 0 ends here
}

// fw_collect_step (fwsim_fused.hpp): act waves in front, then the step waves of the same step kernels with COLLECT on.  The
// 8-lane mapping at one wave per SIMD only (the collector's regime: thousands of envs per GPU).
#define FW_COLLECT_RUN(T_, ...) do {                                                                                       \
    if ((int)blockIdx.x < CA.n_act) {                                                                                  \
      /* launch index = the word of a step workgroup of MY chunk: it cannot finish (and advance its word) before I publish */ \
      const int nblk_ = (D.npad + 7) / 8, c_ = min((int)blockIdx.x >> 1, CA.n_chunks - 1);                              \
      const int blk_ = min(c_ * kCRows / 8, nblk_ - 1);                                                                 \
      const int wg_ = ((nblk_ & 7) == 0) ? (blk_ % (nblk_ >> 3)) * 8 + blk_ / (nblk_ >> 3) : blk_;   /* inverse of the XCD-aware map */ \
      collect_act_wave<T_>(CA, 1u + D.lctr[wg_]);                                                                          \
      return;                                                                                                           \
    }                                                                                                                   \
    if (CA.trace && threadIdx.x == 0) CA.trace[(size_t)blockIdx.x * 8] = collect_now();                                 \
    const int bx_ = (int)blockIdx.x - CA.n_act;                                                                         \
    if (bx_ >= CA.n_workers) {                                                                                          \
      collect_fold_wave(CA, bx_ - CA.n_workers);                                                                        \
      if (CA.trace && threadIdx.x == 0) CA.trace[(size_t)blockIdx.x * 8 + 7] = collect_now();                           \
      return;                                                                                                           \
    }                                                                                                                   \
    D.epoch = 1u + D.lctr[bx_];                                                                                         \
    step_body<T_, __VA_ARGS__>(FW_STEP_PASS, &CA);                                                                     \
    if (threadIdx.x == 0) D.lctr[bx_] = D.epoch;                                                                        \
    if (CA.trace && threadIdx.x == 0) CA.trace[(size_t)blockIdx.x * 8 + 7] = collect_now();                             \
  } while (0)
template <typename T, bool GENERAL>
__global__ __launch_bounds__(kWave) void fw_collect_kernel_g8(FW_STEP_ARGS, const CollectArgs CA) { FW_COLLECT_RUN(T, GENERAL, 8, FW_TASK_WAYPOINTS, 1, true); }
template <typename T, int TKIND>
__global__ __launch_bounds__(kWave) void fw_collect_kernel_obj_g8(FW_STEP_ARGS, const CollectArgs CA) { FW_COLLECT_RUN(T, true, 8, TKIND, 1, true); }
// ... and for the two-waves-per-SIMD build of the waypoints kernels (8 192 < N <= 24 576 envs: 12 288 with wind), so that those
// env counts keep the one-launch collector (round 3 sent them to the three-launch one)
template <typename T, bool GENERAL>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fw_collect_kernel_g8w2(FW_STEP_ARGS, const CollectArgs CA) { FW_COLLECT_RUN(T, GENERAL, 8, FW_TASK_WAYPOINTS, 2, true); }

// Caller-supplied scenario of the episodes a fw_reset starts (fw_scenario, uploaded to device memory by the host side;
// doubles indexed by the handle's local env, null = keep the env's own draw).
struct ScenOv {
  const double *targets, *duck, *obst, *nob, *wind_base, *gust_amp, *gust_phase;
};

// K2: reset (masked) + observation.  Same single-tick-site structure (warm-up only).
template <typename T, int G, int TKIND>
__global__ __launch_bounds__(kWave)
void fw_reset_kernel(const Params<T>* __restrict__ Pp, const ObjC<T>* __restrict__ OCp, DevState<T> Dg,
                     const uint8_t* __restrict__ mask, T* __restrict__ obs, int do_reset, ScenOv ov) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);
  const Params<T>& P = *Pp;
  constexpr bool OBJ = TKIND == FW_TASK_OBJLOCK;
  constexpr bool COMB = TKIND == FW_TASK_WAYPOINT_OBJLOCK;
  constexpr bool HASOBJ = OBJ || COMB;
  constexpr int EPW = kWave / G;
  const int lane = threadIdx.x;
  const int sub = (G == 1) ? 0 : (lane & (G - 1));
  const int row = lane / G;
  const bool leader = sub == 0;
  const DevState<T> D = tile_view<T, EPW>(Dg, (int)blockIdx.x);
  const int env0 = blockIdx.x * EPW;
  const int env = env0 + row;
  const bool active = env < D.n;
  const int envc = active ? env : D.n - 1;
  const size_t n = D.npad;
  const int Dobs = P.obs_dim;
  const int ld = Dobs + 1;
  TickC<T> C; SurfC<T> mine; T wmask;
  load_tick_constants<T, G, false>(Pp, C, mine, wmask);

  Rigid<T> S;
  load_rigid<T>(D, envc, S);
  ObjState<T> O;
  const ObjC<T>& OC = *OCp;
  if (HASOBJ) obj_load<T, OBJ>(D, envc, O);
  T action[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) action[k] = D.r[(RF_ACTION + k) * n + envc];
  int32_t num_reached = D.i[IF_NUM_REACHED * n + envc];
  int tgt_obs = (D.i[IF_FLAGS * n + envc] >> FL_TGT_SHIFT) & 15;
  const bool resetting = active && do_reset && (!mask || mask[envc]);
  int32_t tick = 0, episode = D.i[IF_EPISODE * n + envc];
  T new_dist = (T)0, wb[3] = {(T)0, (T)0, (T)0}, wa[3] = {(T)0, (T)0, (T)0}, wphase = (T)0;
  int warm_left = 0;
  if (resetting) {
    T t_mine[3] = {(T)0, (T)0, (T)0};
    warm_left = begin_reset<T, G>(P, D, env, S, tick, episode, num_reached, wb, wa, wphase, t_mine);
    // ---- caller-supplied scenario: replaces what begin_reset drew, before anything depends on it ----
    if (P.wind_mode != FW_WIND_OFF && (ov.wind_base || ov.gust_amp || ov.gust_phase)) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (ov.wind_base) wb[k] = (T)ov.wind_base[3 * (size_t)env + k];
        if (ov.gust_amp) wa[k] = (T)ov.gust_amp[3 * (size_t)env + k];
      }
      if (ov.gust_phase) wphase = (T)ov.gust_phase[env];
      if (leader) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { D.r[(RF_WIND + k) * n + env] = wb[k]; D.r[(RF_WIND + 3 + k) * n + env] = wa[k]; }
        D.r[(RF_WIND + 6) * n + env] = wphase;
      }
    }
    T t_last[3] = {(T)0, (T)0, (T)0};
    if (!OBJ && ov.targets && P.num_targets > 0) {
      if (leader)
        for (int t = 0; t < P.num_targets; ++t)
#pragma unroll
          for (int k = 0; k < 3; ++k) D.r[(size_t)(RF_TARGETS + 3 * t + k) * n + env] = (T)ov.targets[((size_t)env * FW_MAX_TARGETS + t) * 3 + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        t_mine[k] = (T)ov.targets[(size_t)env * FW_MAX_TARGETS * 3 + k];                              // waypoint 0, for end_reset
        t_last[k] = (T)ov.targets[((size_t)env * FW_MAX_TARGETS + (P.num_targets - 1)) * 3 + k];
      }
    }
    if (HASOBJ) {
      obj_reset_state<T, OBJ>(O);
      if (OBJ) obj_spawn<T, G>(P, OC, D, env, (uint32_t)episode, leader, O); else comb_spawn<T, G>(P, OC, D, env, (uint32_t)episode, leader, O);
      if (COMB && ov.targets && P.num_targets > 0) { O.duck[0] = t_last[0]; O.duck[1] = t_last[1]; }   // the duck sits under the last waypoint
      if (ov.duck) {
#pragma unroll
        for (int k = 0; k < 3; ++k) O.duck[k] = (T)ov.duck[3 * (size_t)env + k];
      }
      if (ov.obst && ov.nob) {
        int nob = (int)ov.nob[env];
        nob = nob < 0 ? 0 : (nob > OC.num_obstacles ? OC.num_obstacles : nob);           // never more than the config (and its LDS row) allows
        if (leader)
          for (int o = 0; o < FW_MAX_OBSTACLES; ++o)
#pragma unroll
            for (int k = 0; k < 3; ++k)
              D.r[(size_t)(RF_TASK + FW_ST_OBST + 3 * o + k) * n + env] = o < nob ? (T)ov.obst[((size_t)env * FW_MAX_OBSTACLES + o) * 3 + k] : (T)0;
        O.nob = nob;
      }
      if (G > 1) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    if (HASOBJ) obj_update_near_mask<T, G>(P, OC, D, envc, O, S);
    if (warm_left == 0) {
      if (OBJ) obj_compute_state<T>(O);
      else { new_dist = end_reset<T, G>(P, D, env, episode, S, ov.targets ? t_mine : nullptr); if (COMB) comb_compute_state<T>(OC, O, P.num_targets == 0); }
    }
  }
  // waypoint 0 of a supplied scenario, for the end_reset that follows an in-kernel warm-up
  T t_first[3] = {(T)0, (T)0, (T)0};
  const bool have_first = resetting && !OBJ && ov.targets && P.num_targets > 0;
  if (have_first) {
#pragma unroll
    for (int k = 0; k < 3; ++k) t_first[k] = (T)ov.targets[(size_t)env * FW_MAX_TARGETS * 3 + k];
  }
  const T cmd0[FW_NUM_ACTUATORS] = {(T)0, (T)0, (T)0, (T)0, (T)0, (T)0};
  T R[9];
  normalize_quat<T>(S.q);
  rot_from_unit_quat<T>(S.q, R);
  T gust[2];
  gust_init<T>(P, wphase, tick, gust);
  LaneAct<T> LA; LA.cmd = (T)0; LA.a = (T)0;
  if (G == 8) lane_act_scatter<T>(S, LA);
#pragma unroll 1
  while (__ballot(warm_left > 0) != 0ull) {
    const bool stepped = warm_left > 0;
    if (stepped) (void)aviary_step<T, true, G, HASOBJ>(P, C, OC, D, env, O, S, R, cmd0, tick, (T)0, (T)0, wb, wa, gust, mine, wmask, LA);
    if (HASOBJ) obj_capture_step<T, G>(OC, D, stepped, env, O, S, R, tick);
    if (stepped) {
      warm_left -= 1;
      if (warm_left == 0) {
        if (OBJ) obj_compute_state<T>(O);
        else { new_dist = end_reset<T, G>(P, D, env, episode, S, have_first ? t_first : nullptr); if (COMB) comb_compute_state<T>(OC, O, P.num_targets == 0); }
      }
    }
  }
  if (G > 1) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  if (G == 8) lane_act_gather<T>(S, LA);
  if (resetting) {
    action[0] = action[1] = action[2] = action[3] = (T)0;
    tgt_obs = 0;
    if (leader) {
      store_rigid<T>(D, env, S);
#pragma unroll
      for (int k = 0; k < 4; ++k) D.r[(RF_ACTION + k) * n + env] = (T)0;
      D.r[RF_NEW_DIST * n + env] = new_dist;
      D.r[RF_EP_RETURN * n + env] = (T)0;
      D.i[IF_STEP * n + env] = 0;
      D.i[IF_TICK * n + env] = tick;
      D.i[IF_EPISODE * n + env] = episode;
      D.i[IF_FLAGS * n + env] = 0;
      D.i[IF_NUM_REACHED * n + env] = num_reached;
      if (HASOBJ) obj_store<T, OBJ>(D, env, O);
    }
  }
  if (obs) {
    if (active && leader) {
      if (OBJ) obj_write_obs<T>(P, O, S, action, [&](int k, T v) { tile[row * ld + k] = v; });
      else if (COMB) comb_write_obs<T>(P, D, env, O, S, action, tgt_obs, [&](int k, T v) { tile[row * ld + k] = v; });
      else write_obs<T>(P, D, env, S, action, tgt_obs, [&](int k, T v) { tile[row * ld + k] = v; });
    }
    __syncthreads();
    flush_obs_tile<T>(tile, ld, obs, env0, EPW, D.n, Dobs);
  }
}

// ======================================================================
// host side
// ======================================================================
namespace {

constexpr int kG8MaxEnvs = 16384;   // measured crossover on MI355X: 8 lanes/env wins up to 2^14 envs (51 vs 70 us), loses at 2^15 (93 vs 76 us)
thread_local std::string g_err;

struct HostDerived {
  double area, aspect, Cl3, a0b, asPb, asNb, theta_f, tau_f, tq[3];
};

int validate(const fw_config* c, std::string& msg) {
  char buf[256];
  auto fail = [&](int code) { msg = buf; return code; };
  if (!c) { snprintf(buf, sizeof buf, "null config"); return fail(FW_EINVAL); }
  if (c->abi_version != FW_ABI_VERSION) { snprintf(buf, sizeof buf, "abi_version %d != %d", c->abi_version, FW_ABI_VERSION); return fail(FW_EVERSION); }
  if (c->agent_hz <= 0 || 120 % c->agent_hz != 0) {
    int lowest = c->agent_hz > 0 ? (int)(120 / ((int)(120 / c->agent_hz) + 1)) : 1;
    int highest = (c->agent_hz > 0 && c->agent_hz <= 120) ? (int)(120 / (int)(120 / c->agent_hz)) : 120;
    snprintf(buf, sizeof buf, "`agent_hz` must be round denominator of 120, try %d or %d.", lowest, highest);
    return fail(FW_EINVAL);
  }
  if (c->angle_representation != 0 && c->angle_representation != 1) {
    snprintf(buf, sizeof buf, "angle_representation must be either `euler` or `quaternion`, not %d", c->angle_representation);
    return fail(FW_EINVAL);
  }
  if (c->wind_mode < FW_WIND_OFF || c->wind_mode > FW_WIND_GUST_SINE) { snprintf(buf, sizeof buf, "Unsupported wind mode: %d", c->wind_mode); return fail(FW_EINVAL); }
  if (c->wind_mode != FW_WIND_OFF && c->wind_randomize_on_reset) {
    for (int i = 0; i < 3; ++i) {
      if (!(c->wind_enu_mps_range[i][0] <= c->wind_enu_mps_range[i][1])) { snprintf(buf, sizeof buf, "Invalid wind_enu_mps_range"); return fail(FW_EINVAL); }
      if (!(c->gust_amp_enu_mps_range[i][0] <= c->gust_amp_enu_mps_range[i][1])) { snprintf(buf, sizeof buf, "Invalid gust_amp_enu_mps_range"); return fail(FW_EINVAL); }
    }
  }
  if (c->task < FW_TASK_WAYPOINTS || c->task > FW_TASK_WAYPOINT_OBJLOCK) { snprintf(buf, sizeof buf, "unknown task %d", c->task); return fail(FW_EINVAL); }
  if (c->dtype != FW_F64 && c->dtype != FW_F32) { snprintf(buf, sizeof buf, "unknown dtype %d", c->dtype); return fail(FW_EINVAL); }
  if (c->num_targets < 0 || c->num_targets > FW_MAX_TARGETS) { snprintf(buf, sizeof buf, "num_targets must be in [0,%d]", FW_MAX_TARGETS); return fail(FW_EINVAL); }
  if (c->task != FW_TASK_OBJLOCK && (c->context_length < 0 || c->context_length > FW_MAX_TARGETS + 1)) { snprintf(buf, sizeof buf, "bad context_length"); return fail(FW_EINVAL); }
  if (c->n_collision_pts < 0 || c->n_collision_pts > FW_MAX_COLLISION_PTS) { snprintf(buf, sizeof buf, "bad n_collision_pts"); return fail(FW_EINVAL); }
  if (c->num_obstacles < 0 || c->num_obstacles > FW_MAX_OBSTACLES) { snprintf(buf, sizeof buf, "bad num_obstacles"); return fail(FW_EINVAL); }
  if (c->task != FW_TASK_WAYPOINTS && (c->camera_resolution < 0 || c->camera_resolution > 1024)) { snprintf(buf, sizeof buf, "camera_resolution must be in [1, 1024]"); return fail(FW_EINVAL); }
  if (c->physics_hz <= 0 || c->control_hz <= 0 || c->physics_hz % c->control_hz != 0) { snprintf(buf, sizeof buf, "physics_hz must be a multiple of control_hz"); return fail(FW_EINVAL); }
  if (!(c->mass > 0.0)) { snprintf(buf, sizeof buf, "mass must be > 0"); return fail(FW_EINVAL); }
  if (c->wind_coupling < FW_WIND_COUPLE_NONE || c->wind_coupling > FW_WIND_COUPLE_AIRSPEED) { snprintf(buf, sizeof buf, "bad wind_coupling"); return fail(FW_EINVAL); }
  return FW_OK;
}

int obs_dim_of(const fw_config* c) {
  int att = (c->angle_representation == 0 ? 12 : 13) + 4 + 6;
  if (c->task == FW_TASK_OBJLOCK) return att + 3 + FW_VISION_FEATS * FW_VISION_HIST + (c->duck_vision_no_deltas ? 0 : 4);
  return att + 3 * c->context_length;
}

bool invert3(const double m[9], double inv[9]) {
  double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  if (det == 0.0) return false;
  double id = 1.0 / det;
  inv[0] = (m[4] * m[8] - m[5] * m[7]) * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  inv[3] = (m[5] * m[6] - m[3] * m[8]) * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  inv[6] = (m[3] * m[7] - m[4] * m[6]) * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return true;
}

// Fold fw_config into the wave-uniform constant block (all derivations in double).
template <typename T>
bool build_params(const fw_config& c, uint64_t seed, int64_t env_offset, Params<T>& P, std::string& err) {
  std::memset(&P, 0, sizeof P);
  const double dt = 1.0 / (double)c.physics_hz;
  const double d2r = kPi / 180.0;
  for (int s = 0; s < FW_NUM_SURFACES; ++s) {
    const fw_surface_params& sp = c.surfaces[s];
    SurfC<T>& o = P.s[s];
    double area = sp.chord * sp.span, AR = sp.span / sp.chord;
    double Cl3 = sp.Cl_alpha_2D * (AR / (AR + ((2.0 * (AR + 4.0)) / (AR + 2.0))));
    double theta_f = std::acos(2.0 * sp.flap_to_chord - 1.0);
    double tau_f = 1.0 - ((theta_f - std::sin(theta_f)) / kPi);
    double a0b = sp.alpha_0_base_deg * d2r, asPb = sp.alpha_stall_P_base_deg * d2r, asNb = sp.alpha_stall_N_base_deg * d2r;
    o.dt_tau = (T)(dt / sp.tau);
    for (int k = 0; k < 3; ++k) { o.lift[k] = (T)sp.lift_unit[k]; o.fwd[k] = (T)sp.forward_unit[k]; o.pos[k] = (T)sp.pos[k]; }
    const double* L = sp.lift_unit; const double* F = sp.forward_unit;
    o.tq[0] = (T)(L[1] * F[2] - L[2] * F[1]); o.tq[1] = (T)(L[2] * F[0] - L[0] * F[2]); o.tq[2] = (T)(L[0] * F[1] - L[1] * F[0]);
    o.hra = (T)(0.5 * c.air_density * area);
    o.chord = (T)sp.chord;
    o.Cl3 = (T)Cl3; o.inv_Cl3 = (T)(1.0 / Cl3); o.inv_piAR = (T)(1.0 / (kPi * AR));
    o.a0b = (T)a0b;
    o.defl_scale = (T)(sp.deflection_limit_deg * d2r);
    o.k_dCl = (T)(Cl3 * tau_f * sp.eta * sp.deflection_limit_deg * d2r);
    o.ClmaxPb = (T)(Cl3 * (asPb - a0b)); o.ClmaxNb = (T)(Cl3 * (asNb - a0b));
    o.ftc = (T)sp.flap_to_chord; o.Cd0 = (T)sp.Cd_0;
    o.k_exp = (T)(0.41 * (1.0 - std::exp(-17.0 / AR)));
  }
  const double max_rpm = std::sqrt(c.motor.total_thrust / c.motor.thrust_coef);
  P.motor_dt_tau = (T)(dt / c.motor.tau);
  P.noise_ratio = (T)c.motor.noise_ratio;
  P.has_noise = c.motor.noise_ratio != 0.0;
  for (int k = 0; k < 3; ++k) {
    P.m_force[k] = (T)(max_rpm * max_rpm * c.motor.thrust_coef * c.motor.thrust_unit[k]);
    P.m_torque[k] = (T)(max_rpm * max_rpm * c.motor.torque_coef * c.motor.thrust_unit[k]);
    P.m_pos[k] = (T)c.motor.pos[k];
  }
  {
    const double f[3] = { max_rpm * max_rpm * c.motor.thrust_coef * c.motor.thrust_unit[0], max_rpm * max_rpm * c.motor.thrust_coef * c.motor.thrust_unit[1], max_rpm * max_rpm * c.motor.thrust_coef * c.motor.thrust_unit[2] };
    const double* r = c.motor.pos;
    const double tq[3] = { max_rpm * max_rpm * c.motor.torque_coef * c.motor.thrust_unit[0], max_rpm * max_rpm * c.motor.torque_coef * c.motor.thrust_unit[1], max_rpm * max_rpm * c.motor.torque_coef * c.motor.thrust_unit[2] };
    P.m_wrench_t[0] = (T)(r[1] * f[2] - r[2] * f[1] + tq[0]);
    P.m_wrench_t[1] = (T)(r[2] * f[0] - r[0] * f[2] + tq[1]);
    P.m_wrench_t[2] = (T)(r[0] * f[1] - r[1] * f[0] + tq[2]);
  }
  for (int a = 0; a < FW_NUM_ACTUATORS; ++a) for (int k = 0; k < 4; ++k) P.mixer[a][k] = (T)c.mixer[a][k];
  P.inv_mass = (T)(1.0 / c.mass); P.gravity = (T)c.gravity;
  const double* I = c.inertia;
  double m[9] = { I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2] }, mi[9];
  if (!invert3(m, mi)) { err = "singular inertia"; return false; }
  for (int k = 0; k < 9; ++k) { P.I[k] = (T)m[k]; P.Iinv[k] = (T)mi[k]; }
  for (int i = 0; i < FW_MAX_COLLISION_PTS; ++i) for (int k = 0; k < 3; ++k) P.coll[i][k] = (T)c.collision_pts[i][k];
  P.n_coll = c.n_collision_pts; P.gyroscopic = c.gyroscopic;
  P.dt = (T)dt; P.inv_physics_hz = (T)dt; P.physics_hz_T = (T)c.physics_hz;
  P.dome = (T)c.flight_dome_size; P.reach = (T)c.goal_reach_distance;
  P.min_height = (T)c.waypoint_min_height; P.spawn_hi = (T)(c.waypoint_spawn_size * 0.9);
  {
    double hr = 0.5 * c.start_orn[0], hp = 0.5 * c.start_orn[1], hy = 0.5 * c.start_orn[2];
    double cr = std::cos(hr), sr = std::sin(hr), cp = std::cos(hp), sp = std::sin(hp), cy = std::cos(hy), sy = std::sin(hy);
    P.start_quat[0] = (T)(sr * cp * cy - cr * sp * sy); P.start_quat[1] = (T)(cr * sp * cy + sr * cp * sy);
    P.start_quat[2] = (T)(cr * cp * sy - sr * sp * cy); P.start_quat[3] = (T)(cr * cp * cy + sr * sp * sy);
  }
  for (int k = 0; k < 3; ++k) {
    P.start_pos[k] = (T)c.start_pos[k]; P.start_vel[k] = (T)c.start_vel[k];
    P.wind_base[k] = (T)c.wind_enu_mps[k]; P.wind_amp[k] = (T)c.gust_amp_enu_mps[k];
    for (int j = 0; j < 2; ++j) { P.wind_base_range[k][j] = c.wind_enu_mps_range[k][j]; P.wind_amp_range[k][j] = c.gust_amp_enu_mps_range[k][j]; }
  }
  P.wind_phase = (T)c.gust_phase_rad; P.gust_omega = (T)(2.0 * kPi * c.gust_freq_hz); P.wind_force_coef = (T)c.wind_force_coef;
  P.gust_sd = (T)std::sin((double)P.gust_omega * (double)P.inv_physics_hz); P.gust_cd = (T)std::cos((double)P.gust_omega * (double)P.inv_physics_hz);
  P.wind_mode = c.wind_mode; P.wind_randomize = c.wind_randomize_on_reset; P.wind_randomize_phase = c.wind_randomize_phase;
  P.wind_coupling = (c.wind_mode == FW_WIND_OFF) ? FW_WIND_COUPLE_NONE : c.wind_coupling;
  P.task = c.task; P.angle_repr = c.angle_representation;
  P.att_dim = (c.angle_representation == 0 ? 12 : 13) + 4 + 6;
  P.obs_dim = obs_dim_of(&c); P.ctx = c.context_length;
  P.num_targets = c.num_targets; P.sparse = c.sparse_reward; P.auto_reset = c.auto_reset;
  P.max_steps = (int32_t)(c.agent_hz * c.max_duration_seconds);
  P.step_ratio = 120 / c.agent_hz;
  P.ticks_per_aviary = c.physics_hz / c.control_hz;
  P.warmup_aviary_steps = c.warmup_aviary_steps;
  // The warm-up is env-independent iff wind cannot act on the dynamics (throttle
  // stays exactly 0 under a zero setpoint, so motor noise multiplies 0).
  // ObjLock: the camera may capture during the warm-up (cadence), so it is always integrated in-kernel.
  P.warm_valid = (P.wind_coupling == FW_WIND_COUPLE_NONE && c.task == FW_TASK_WAYPOINTS) ? 1 : 0;   // (the duck cannot be in contact during warm-up: it spawns >= start height away only by chance; contacts there are ignored by the reference too)
  P.seed_lo = (uint32_t)seed; P.seed_hi = (uint32_t)(seed >> 32);
  P.env_offset = env_offset;
  if (P.obs_dim > kMaxObs) { err = "obs_dim exceeds kMaxObs"; return false; }
  return true;
}

}  // namespace

struct fw_env {
  fw_config cfg;
  int32_t n = 0, npad = 0, device = 0;
  int32_t lanes_per_env = 1;    // 1: throughput mapping, 8: latency mapping (see fwsim_device.hpp)
  int32_t g8_waves = 1;         // 8-lane mapping, waypoints task: waves per SIMD the step kernel is built for (1 | 2)
  int32_t capture_wave = 0;     // 8-lane mapping, camera tasks: fw_step workgroups carry a capture wave (fw_step_kernel_obj_g8h)
  uint64_t seed = 0;
  int64_t env_offset = 0;
  void* params_dev = nullptr;   // Params<T>
  void* objc_dev = nullptr;     // ObjC<T>
  void* r_dev = nullptr;        // T[RF_COUNT][npad]
  int32_t* i_dev = nullptr;     // i32[IF_COUNT][npad]
  void* rs_dev = nullptr;       // shadow T[RF_COUNT][npad]   (only when shadow_on)
  int32_t* is_dev = nullptr;    // shadow ticks i32[npad]
  void* sobs_dev = nullptr;     // shadow first observations T[npad][obs_dim]
  unsigned long long* sreq_dev = nullptr;   // shadow requests / progress words, u64[npad] each
  unsigned long long* sdone_dev = nullptr;
  uint32_t* lctr_dev = nullptr;             // device-side launch index, one word per workgroup (fwsim_device.hpp: launch_index)
  unsigned long long* stats_dev = nullptr;  // hand-off counters (fw_get_counters)
  double* scen_dev = nullptr;               // staging of a caller-supplied fw_scenario (allocated on first use)
  long long* prof_dev = nullptr; // FW_PROFILE builds only
  int32_t shadow_on = 0;        // background warm-up of the next episode (see shadow_* kernels)
  const void* collect_act_seen = nullptr;   // the action buffer fw_collect_step last filled with "not there yet"
  std::string err;
};

#define HIP_TRY(h, expr)                                                                  \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      std::string m_ = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
      if (h) (h)->err = m_; else g_err = m_;                                              \
      return FW_EHIP;                                                                     \
    }                                                                                     \
  } while (0)

namespace {
struct DeviceGuard {
  int prev = -1; bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) { switched = hipSetDevice(dev) == hipSuccess; }
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// The learner-side entry points have no handle: the device is the one their buffers live on (never the thread's current
// device), and everything they launch runs under a DeviceGuard for it.
int device_of(const void* p) {
  hipPointerAttribute_t a;
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (p && hipPointerGetAttributes(&a, p) == hipSuccess && a.device >= 0) return a.device;
  (void)hipGetLastError();                     // clear the sticky error of a failed query
  return cur;
}
// Opt an LDS-resident learner kernel into `bytes` of dynamic LDS: done when a device first sees a size larger than any
// before (i.e. on the first call per observation width), never again on the launch path.
int ensure_learner_lds(int dev, int which /*0: fw_ppo_update, 1: fw_policy_act*/, size_t bytes);

template <typename T> size_t tile_bytes(const fw_env* h);
template <typename T> size_t step_lds_bytes(const fw_env* h);
template <typename T> DevState<T> dev_state(fw_env* h) {
  DevState<T> D; D.r = (T*)h->r_dev; D.i = h->i_dev; D.n = h->n; D.npad = h->npad;
  D.rs = (T*)h->rs_dev; D.is = h->is_dev; D.sobs = (T*)h->sobs_dev; D.sreq = h->sreq_dev; D.sdone = h->sdone_dev; D.epoch = 0; D.shadow_on = h->shadow_on;
  D.lctr = h->lctr_dev; D.stats = h->stats_dev;
  D.stash_off = (int32_t)((tile_bytes<T>(h) + 15) & ~(size_t)15);
  D.mbox_off = (int32_t)((step_lds_bytes<T>(h) + 15) & ~(size_t)15);
  FWP(D.prof = h->prof_dev;)
  return D;
}
int invalidate_shadow(fw_env* h) {
  if (!h->shadow_on) return FW_OK;
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipMemset(h->sreq_dev, 0xFF, sizeof(unsigned long long) * h->npad));    // ~0 = nothing requested
  HIP_TRY(h, hipMemset(h->sdone_dev, 0xFF, sizeof(unsigned long long) * h->npad));   // episode 0xFFFFFFFF never matches
  HIP_TRY(h, hipMemset(h->is_dev, 0, sizeof(int32_t) * h->npad));
  return FW_OK;
}

// every device buffer of a handle, freed once and nulled (fw_destroy and the error path of fw_create)
void free_device_buffers(fw_env* h) {
  void** bufs[] = { &h->params_dev, &h->objc_dev, &h->r_dev, (void**)&h->i_dev, &h->rs_dev, (void**)&h->is_dev, &h->sobs_dev,
                    (void**)&h->sreq_dev, (void**)&h->sdone_dev, (void**)&h->lctr_dev, (void**)&h->stats_dev, (void**)&h->scen_dev,
                    (void**)&h->prof_dev };
  for (void** b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
}

// words per env of the camera's LDS row buffer: the width padded so that the 4 envs of a half-wave start on different banks
inline int zrow_stride_of(int res) { return ((res + 31) / 32) * 32 + 8; }

// LDS bytes: the padded [64/G][D+1] observation tile; the camera tasks on the 8-lane mapping alias it (in time) with the
// row buffer of the analytic camera (1 / t of the nearest cylinder fragment per column), 8 envs x zrow_stride words
template <typename T> size_t tile_bytes(const fw_env* h) {
  size_t b = sizeof(T) * (size_t)(kWave / h->lanes_per_env) * (size_t)(obs_dim_of(&h->cfg) + 1);
  if (h->cfg.task != FW_TASK_WAYPOINTS && h->lanes_per_env == 8) {
    const int res = h->cfg.camera_resolution > 0 ? h->cfg.camera_resolution : 128;
    b = std::max(b, cam_lds(sizeof(T), zrow_stride_of(res), res, h->cfg.num_obstacles > 0).total);   // camera map (fwsim_objlock.hpp): 2 KB without, 48 KB with cylinders at 480 columns
  }
  return b;
}
// dynamic LDS of a step launch: the tile (or the camera's map) + the output stash of the envs of a wave
template <typename T> size_t step_lds_bytes(const fw_env* h) { return ((tile_bytes<T>(h) + 15) & ~(size_t)15) + sizeof(double) * 4 * (size_t)(kWave / h->lanes_per_env); }
// ... of a step launch whose workgroups carry a capture wave: + the mailbox
template <typename T> size_t step_lds_bytes_h(const fw_env* h) { return ((step_lds_bytes<T>(h) + 15) & ~(size_t)15) + mbox_bytes(sizeof(T)); }
inline dim3 grid_of(const fw_env* h) { return dim3((unsigned)(h->npad / (kWave / h->lanes_per_env))); }

// ObjLock task constants (analytic camera axes, shaping coefficients of envs/fixedwing_objlock_env.py:54-80)
template <typename T>
void build_objc(const fw_config& c, ObjC<T>& O) {
  std::memset(&O, 0, sizeof O);
  const double th = c.camera_angle_deg * kPi / 180.0;
  const double f[3] = { std::cos(th), 0.0, std::sin(th) }, r[3] = { 0.0, -1.0, 0.0 };
  const double d[3] = { f[1] * r[2] - f[2] * r[1], f[2] * r[0] - f[0] * r[2], f[0] * r[1] - f[1] * r[0] };
  for (int k = 0; k < 3; ++k) { O.cam_f[k] = (T)f[k]; O.cam_r[k] = (T)r[k]; O.cam_d[k] = (T)d[k]; O.cam_off[k] = (T)c.camera_offset[k]; }
  const int res = c.camera_resolution > 0 ? c.camera_resolution : 128;
  O.W = O.H = (T)res; O.vmid = (T)(res / 2);
  O.focal = (T)(0.5 * res / std::tan(0.5 * c.camera_fov_deg * kPi / 180.0));
  O.inv_focal = (T)(1.0 / (0.5 * res / std::tan(0.5 * c.camera_fov_deg * kPi / 180.0)));
  O.near_ = (T)c.camera_near; O.far_ = (T)c.camera_far;
  O.inv_near = (T)(1.0 / c.camera_near); O.inv_far = (T)(1.0 / c.camera_far); O.db_c1 = (T)(c.camera_far / (c.camera_far - c.camera_near));
  O.zrow_stride = zrow_stride_of(res);
  O.duck_radius = (T)(c.duck_radius_per_scale * c.duck_global_scaling); O.half_dome = (T)(c.flight_dome_size / 2.0);
  O.obst_radius = (T)c.obstacle_radius; O.obst_hmin = (T)c.obstacle_height_range[0]; O.obst_hmax = (T)c.obstacle_height_range[1];
  O.safe_dist = (T)c.obstacle_safe_distance_m; O.avoid_scale = (T)c.obstacle_avoid_reward_scale; O.avoid_max = (T)c.obstacle_avoid_max_penalty;
  O.k_dist = (T)c.duck_distance_reward_scale; O.lock_radius = (T)c.duck_lock_center_radius; O.k_center = (T)c.duck_centering_reward_scale;
  O.k_visible = (T)c.duck_visible_step_reward; O.k_area = (T)c.duck_area_reward_scale; O.lost_penalty = (T)c.duck_lock_lost_penalty;
  O.approach_clip = (T)c.duck_approach_reward_clip_m; O.k_approach = (T)c.duck_approach_reward_scale;
  O.strike_dist = (T)c.duck_strike_distance_m; O.strike_reward = (T)c.duck_strike_reward; O.lock_step_reward = (T)c.duck_lock_step_reward;
  O.hold_steps = c.duck_lock_hold_steps; O.decay_steps = c.duck_lock_decay_steps; O.num_obstacles = c.num_obstacles;
  O.camera_ratio_ticks = (c.physics_hz / c.control_hz) * c.duck_camera_capture_interval_steps;
  O.switch_min_area = (T)c.duck_switch_min_area; O.switch_min_seen = c.duck_switch_min_consecutive_seen;
  {
    double rm = 0.0;
    for (int i = 0; i < c.n_collision_pts; ++i) { const double* q = c.collision_pts[i]; rm = std::max(rm, std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2])); }
    O.reach_margin = (T)(rm + 0.25);
  }
}

template <typename T>
int upload_params(fw_env* h) {
  Params<T> P;
  if (!build_params<T>(h->cfg, h->seed, h->env_offset, P, h->err)) return FW_EINVAL;
  if (!h->params_dev) HIP_TRY(h, hipMalloc(&h->params_dev, sizeof(Params<T>)));
  HIP_TRY(h, hipMemcpy(h->params_dev, &P, sizeof P, hipMemcpyHostToDevice));
  ObjC<T> OC;
  build_objc<T>(h->cfg, OC);
  if (!h->objc_dev) HIP_TRY(h, hipMalloc(&h->objc_dev, sizeof(ObjC<T>)));
  HIP_TRY(h, hipMemcpy(h->objc_dev, &OC, sizeof OC, hipMemcpyHostToDevice));
  if (P.warm_valid) {
    hipLaunchKernelGGL(fw_warm_kernel<T>, dim3(1), dim3(kWave), 0, 0, (Params<T>*)h->params_dev);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipDeviceSynchronize());
  }
  return FW_OK;
}

template <typename T>
int create_T(fw_env* h) {
  const size_t npad = (size_t)h->npad;
  HIP_TRY(h, hipMalloc(&h->r_dev, sizeof(T) * RF_COUNT * npad));
  HIP_TRY(h, hipMalloc((void**)&h->i_dev, sizeof(int32_t) * IF_COUNT * npad));
  int rc = upload_params<T>(h);
  if (rc != FW_OK) return rc;
  const size_t nctr = 2 * (size_t)grid_of(h).x;              // step workgroups + (possibly) as many workers
  HIP_TRY(h, hipMalloc((void**)&h->lctr_dev, sizeof(uint32_t) * nctr));
  HIP_TRY(h, hipMemset(h->lctr_dev, 0, sizeof(uint32_t) * nctr));
  HIP_TRY(h, hipMalloc((void**)&h->stats_dev, sizeof(unsigned long long) * FW_CTR_DIM));
  HIP_TRY(h, hipMemset(h->stats_dev, 0, sizeof(unsigned long long) * FW_CTR_DIM));
  // shadow warm-up whenever the reset warm-up cannot be cached (wind acting on the dynamics, camera tasks)
  const bool cached = (h->cfg.task == FW_TASK_WAYPOINTS) &&
                      (h->cfg.wind_mode == FW_WIND_OFF || h->cfg.wind_coupling == FW_WIND_COUPLE_NONE);
  h->shadow_on = (!cached && h->cfg.auto_reset && !getenv("FWSIM_NO_SHADOW")) ? 1 : 0;
  // wind-free waypoints on the latency mapping: workers only pre-sample the next episode's waypoints (scenario_worker)
  if (h->cfg.task == FW_TASK_WAYPOINTS && h->cfg.wind_mode == FW_WIND_OFF && h->cfg.auto_reset && h->lanes_per_env == 8 &&
      !getenv("FWSIM_NO_SHADOW"))
    h->shadow_on = 2;
  if (h->shadow_on) {
    HIP_TRY(h, hipMalloc(&h->rs_dev, sizeof(T) * RF_COUNT * npad));
    HIP_TRY(h, hipMemset(h->rs_dev, 0, sizeof(T) * RF_COUNT * npad));
    HIP_TRY(h, hipMalloc((void**)&h->is_dev, sizeof(int32_t) * npad));
    HIP_TRY(h, hipMalloc(&h->sobs_dev, sizeof(T) * (size_t)npad * (size_t)obs_dim_of(&h->cfg)));
    HIP_TRY(h, hipMemset(h->sobs_dev, 0, sizeof(T) * (size_t)npad * (size_t)obs_dim_of(&h->cfg)));
    HIP_TRY(h, hipMalloc((void**)&h->sreq_dev, sizeof(unsigned long long) * npad));
    HIP_TRY(h, hipMalloc((void**)&h->sdone_dev, sizeof(unsigned long long) * npad));
    rc = invalidate_shadow(h);
    if (rc != FW_OK) return rc;
  }
  // the camera's LDS map outgrows what a workgroup gets without asking from ~700 columns on.  The attribute belongs to the
  // (device, kernel) pair, not to the handle: keep the maximum ever asked for and only ever raise it, so a later, smaller
  // handle cannot lower the cap under an earlier one
  if (const size_t lds = step_lds_bytes_h<T>(h); lds > 48 * 1024) {
    if (lds > 160 * 1024) { h->err = "camera_resolution x num_obstacles needs more LDS than a CU has"; return FW_EINVAL; }
    const int which = (h->cfg.task == FW_TASK_OBJLOCK ? 0 : 1) + (sizeof(T) == 8 ? 0 : 2);
    static size_t have[64][4] = {};
    if (h->device < 64 && lds > have[h->device][which]) {
      if (h->cfg.task == FW_TASK_OBJLOCK) {
        HIP_TRY(h, hipFuncSetAttribute((const void*)fw_step_kernel_obj_g8<T, FW_TASK_OBJLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute((const void*)fw_step_kernel_obj_g8h<T, FW_TASK_OBJLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute((const void*)fw_reset_kernel<T, 8, FW_TASK_OBJLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      } else {
        HIP_TRY(h, hipFuncSetAttribute((const void*)fw_step_kernel_obj_g8<T, FW_TASK_WAYPOINT_OBJLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute((const void*)fw_step_kernel_obj_g8h<T, FW_TASK_WAYPOINT_OBJLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute((const void*)fw_reset_kernel<T, 8, FW_TASK_WAYPOINT_OBJLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      }
      have[h->device][which] = lds;
    }
  }
  hipLaunchKernelGGL(fw_init_kernel<T>, dim3((h->npad + 255) / 256), dim3(256), 0, 0, dev_state<T>(h), kWave / h->lanes_per_env);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipDeviceSynchronize());
  return FW_OK;
}

#define FW_LAUNCH_STEP(KERNEL)                                                                                   \
  hipLaunchKernelGGL((KERNEL), step_grid, dim3(kWave), step_lds_bytes<T>(h), st, (const Params<T>*)h->params_dev,    \
                     (const ObjC<T>*)h->objc_dev, dev_state<T>(h), (const T*)actions, (T*)obs, (T*)reward, term,  \
                     trunc, (T*)tobs, info)
#define FW_LAUNCH_STEP_H(KERNEL)   /* step wave + capture wave per workgroup */                                      \
  hipLaunchKernelGGL((KERNEL), step_grid, dim3(2 * kWave), step_lds_bytes_h<T>(h), st, (const Params<T>*)h->params_dev, \
                     (const ObjC<T>*)h->objc_dev, dev_state<T>(h), (const T*)actions, (T*)obs, (T*)reward, term,  \
                     trunc, (T*)tobs, info)

template <typename T>
int step_T(fw_env* h, const void* actions, void* obs, void* reward, uint8_t* term, uint8_t* trunc, void* tobs,
           int32_t* info, hipStream_t st) {
  const bool general = h->cfg.wind_mode != FW_WIND_OFF;
  const bool g8 = h->lanes_per_env == 8;
  dim3 step_grid = grid_of(h);
  const bool two_wave = g8 && h->capture_wave && h->cfg.task != FW_TASK_WAYPOINTS;
  if (h->shadow_on && !two_wave) step_grid.x *= 2;          // second half of the grid = shadow workers (two-wave workgroups: the capture wave is the worker)
  if (h->cfg.task == FW_TASK_OBJLOCK) {
    if (g8 && h->capture_wave) FW_LAUNCH_STEP_H((fw_step_kernel_obj_g8h<T, FW_TASK_OBJLOCK>));
    else if (g8) FW_LAUNCH_STEP((fw_step_kernel_obj_g8<T, FW_TASK_OBJLOCK>)); else FW_LAUNCH_STEP((fw_step_kernel_obj_g1<T, FW_TASK_OBJLOCK>));
  } else if (h->cfg.task == FW_TASK_WAYPOINT_OBJLOCK) {
    if (g8 && h->capture_wave) FW_LAUNCH_STEP_H((fw_step_kernel_obj_g8h<T, FW_TASK_WAYPOINT_OBJLOCK>));
    else if (g8) FW_LAUNCH_STEP((fw_step_kernel_obj_g8<T, FW_TASK_WAYPOINT_OBJLOCK>)); else FW_LAUNCH_STEP((fw_step_kernel_obj_g1<T, FW_TASK_WAYPOINT_OBJLOCK>));
  } else if (general) {
    if (g8 && h->g8_waves == 2) FW_LAUNCH_STEP((fw_step_kernel_g8w2<T, true>));
    else if (g8) FW_LAUNCH_STEP((fw_step_kernel_g8<T, true>)); else FW_LAUNCH_STEP((fw_step_kernel_g1<T, true>));
  } else {
    if (g8 && h->g8_waves == 2) FW_LAUNCH_STEP((fw_step_kernel_g8w2<T, false>));
    else if (g8) FW_LAUNCH_STEP((fw_step_kernel_g8<T, false>)); else FW_LAUNCH_STEP((fw_step_kernel_g1<T, false>));
  }
  HIP_TRY(h, hipGetLastError());
  return FW_OK;
}

#define FW_LAUNCH_RESET(KERNEL)                                                                                   \
  hipLaunchKernelGGL((KERNEL), grid_of(h), dim3(kWave), tile_bytes<T>(h), st, (const Params<T>*)h->params_dev,    \
                     (const ObjC<T>*)h->objc_dev, dev_state<T>(h), mask, (T*)obs, do_reset, ov)

template <typename T>
int reset_T(fw_env* h, const uint8_t* mask, void* obs, int do_reset, hipStream_t st, ScenOv ov = ScenOv{}) {
  const bool g8 = h->lanes_per_env == 8;
  if (h->cfg.task == FW_TASK_OBJLOCK) {
    if (g8) FW_LAUNCH_RESET((fw_reset_kernel<T, 8, FW_TASK_OBJLOCK>)); else FW_LAUNCH_RESET((fw_reset_kernel<T, 1, FW_TASK_OBJLOCK>));
  } else if (h->cfg.task == FW_TASK_WAYPOINT_OBJLOCK) {
    if (g8) FW_LAUNCH_RESET((fw_reset_kernel<T, 8, FW_TASK_WAYPOINT_OBJLOCK>)); else FW_LAUNCH_RESET((fw_reset_kernel<T, 1, FW_TASK_WAYPOINT_OBJLOCK>));
  } else {
    if (g8) FW_LAUNCH_RESET((fw_reset_kernel<T, 8, FW_TASK_WAYPOINTS>)); else FW_LAUNCH_RESET((fw_reset_kernel<T, 1, FW_TASK_WAYPOINTS>));
  }
  HIP_TRY(h, hipGetLastError());
  return FW_OK;
}

// canonical record <-> SoA field maps
struct FieldMap { int rec; int rf; int count; };
const FieldMap kRealMap[] = {
  {FW_S_POS, RF_POS, 3}, {FW_S_QUAT, RF_QUAT, 4}, {FW_S_VEL, RF_VEL, 3}, {FW_S_OMEGA, RF_OMEGA, 3},
  {FW_S_ACT, RF_ACT, 6}, {FW_S_ACTION, RF_ACTION, 4}, {FW_S_NEW_DIST, RF_NEW_DIST, 1}, {FW_S_WIND, RF_WIND, 7},
  {FW_S_EP_RETURN, RF_EP_RETURN, 1}, {FW_S_TARGETS, RF_TARGETS, 3 * FW_MAX_TARGETS},
  {FW_S_TASK, RF_TASK, FW_STATE_DIM - FW_S_TASK},
};
const FieldMap kIntMap[] = {
  {FW_S_STEP_COUNT, IF_STEP, 1}, {FW_S_TICK_COUNT, IF_TICK, 1}, {FW_S_EPISODE, IF_EPISODE, 1},
  {FW_S_FLAGS, IF_FLAGS, 1}, {FW_S_NUM_REACHED, IF_NUM_REACHED, 1},
};

template <typename T>
int get_state_T(fw_env* h, double* out) {
  const size_t npad = (size_t)h->npad;
  const int tile = kWave / h->lanes_per_env;
  std::vector<T> r(RF_COUNT * npad);
  std::vector<int32_t> iv(IF_COUNT * npad);
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipMemcpy(r.data(), h->r_dev, sizeof(T) * r.size(), hipMemcpyDeviceToHost));
  HIP_TRY(h, hipMemcpy(iv.data(), h->i_dev, sizeof(int32_t) * iv.size(), hipMemcpyDeviceToHost));
  for (int e = 0; e < h->n; ++e) {
    double* rec = out + (size_t)e * FW_STATE_DIM;
    std::memset(rec, 0, sizeof(double) * FW_STATE_DIM);
    for (const FieldMap& f : kRealMap) for (int k = 0; k < f.count; ++k) rec[f.rec + k] = (double)r[tile_index(tile, RF_COUNT, f.rf + k, e)];
    for (const FieldMap& f : kIntMap) rec[f.rec] = (double)iv[tile_index(tile, IF_COUNT, f.rf, e)];
  }
  return FW_OK;
}

template <typename T>
int set_state_T(fw_env* h, const double* in) {
  const size_t npad = (size_t)h->npad;
  const int tile = kWave / h->lanes_per_env;
  std::vector<T> r(RF_COUNT * npad, (T)0);
  std::vector<int32_t> iv(IF_COUNT * npad, 0);
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipMemcpy(r.data(), h->r_dev, sizeof(T) * r.size(), hipMemcpyDeviceToHost));
  HIP_TRY(h, hipMemcpy(iv.data(), h->i_dev, sizeof(int32_t) * iv.size(), hipMemcpyDeviceToHost));
  for (int e = 0; e < h->n; ++e) {
    const double* rec = in + (size_t)e * FW_STATE_DIM;
    for (const FieldMap& f : kRealMap) for (int k = 0; k < f.count; ++k) r[tile_index(tile, RF_COUNT, f.rf + k, e)] = (T)rec[f.rec + k];
    for (const FieldMap& f : kIntMap) iv[tile_index(tile, IF_COUNT, f.rf, e)] = (int32_t)rec[f.rec];
  }
  HIP_TRY(h, hipMemcpy(h->r_dev, r.data(), sizeof(T) * r.size(), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(h->i_dev, iv.data(), sizeof(int32_t) * iv.size(), hipMemcpyHostToDevice));
  return FW_OK;
}
}  // namespace

namespace {
int ensure_learner_lds(int dev, int which, size_t bytes) {
  static size_t have[64][2] = {};
  if (dev < 0 || dev >= 64) { g_err = "device index out of range"; return FW_EINVAL; }
  if (bytes <= have[dev][which]) return FW_OK;
  if (which == 0) {
    const void* fns[] = {(const void*)fw_ppo_update_kernel<64, 0>, (const void*)fw_ppo_update_kernel<32, 0>, (const void*)fw_ppo_update_kernel<16, 0>,
                         (const void*)fw_ppo_update_kernel<64, 4>, (const void*)fw_ppo_update_kernel<32, 4>, (const void*)fw_ppo_update_kernel<16, 4>,
                         (const void*)fw_ppo_update_kernel<64, 8>, (const void*)fw_ppo_update_kernel<32, 8>, (const void*)fw_ppo_update_kernel<16, 8>};
    for (const void* fn : fns) HIP_TRY((fw_env*)nullptr, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  } else {
    HIP_TRY((fw_env*)nullptr, hipFuncSetAttribute((const void*)fw_policy_act_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  }
  have[dev][which] = bytes;
  return FW_OK;
}
}  // namespace

// ---- fw_collect_step: one launch per vec-step (fwsim_fused.hpp) ----
namespace {
struct CollectWs { size_t part1, tot, flag_p, flag_v, sync, total; };
CollectWs collect_ws(const fw_env* h) {
  const size_t PW = 2 * (size_t)obs_dim_of(&h->cfg) + 2, nblk = (size_t)h->npad / 8, nch = ((size_t)h->n + kCRows - 1) / kCRows;
  CollectWs w; size_t o = 0;
  w.part1 = o; o += sizeof(double) * PW * kCGroups * ((nblk + kCGroups - 1) / kCGroups);
  w.tot = o; o += sizeof(double) * PW;
  w.flag_p = o; o += sizeof(unsigned int) * nch;
  w.flag_v = o; o += sizeof(unsigned int) * nch;
  o = (o + 63) & ~(size_t)63;
  w.sync = o; o += sizeof(unsigned int) * 16;                 // the last 64 bytes of the workspace: CS_* words (status = word 3)
  w.total = o;
  return w;
}
template <typename T>
int collect_step_T(fw_env* h, CollectArgs& CA, const void* actions, void* obs, void* reward, uint8_t* term, uint8_t* trunc, void* tobs,
                   int32_t* info, hipStream_t st) {
  const size_t lds = std::max(step_lds_bytes<T>(h), collect_act_lds_bytes(CA.A.D));
  if (lds > 160 * 1024) { h->err = "fw_collect_step: the networks do not fit the LDS next to the step kernel's tile"; return FW_EINVAL; }
  const bool w2 = h->g8_waves == 2, windy = h->cfg.wind_mode != FW_WIND_OFF, f32 = sizeof(T) != 8;       // (w2: waypoints task only, fw_create)
  const void* fn = h->cfg.task == FW_TASK_OBJLOCK ? (const void*)fw_collect_kernel_obj_g8<T, FW_TASK_OBJLOCK>
                 : h->cfg.task == FW_TASK_WAYPOINT_OBJLOCK ? (const void*)fw_collect_kernel_obj_g8<T, FW_TASK_WAYPOINT_OBJLOCK>
                 : windy ? (w2 ? (const void*)fw_collect_kernel_g8w2<T, true> : (const void*)fw_collect_kernel_g8<T, true>)
                         : (w2 ? (const void*)fw_collect_kernel_g8w2<T, false> : (const void*)fw_collect_kernel_g8<T, false>);
  if (lds > 48 * 1024) {                       // opt in once per (device, kernel, size class): only ever raised
    static size_t have[64][12] = {};
    const int which = w2 ? 8 + (windy ? 0 : 1) + (f32 ? 2 : 0)
                         : (h->cfg.task == FW_TASK_OBJLOCK ? 0 : h->cfg.task == FW_TASK_WAYPOINT_OBJLOCK ? 1 : windy ? 2 : 3) + (f32 ? 4 : 0);
    if (h->device < 64 && lds > have[h->device][which]) {
      HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      have[h->device][which] = lds;
    }
  }
  CA.n_workers = (int32_t)(grid_of(h).x * (h->shadow_on ? 2u : 1u));
  dim3 grid((unsigned)(CA.n_act + CA.n_workers + 2 * CA.A.D + 2));          // + one fold wave per partial-sum word
#define FW_LAUNCH_COLLECT(KERNEL)                                                                                  \
  hipLaunchKernelGGL((KERNEL), grid, dim3(kWave), lds, st, (const Params<T>*)h->params_dev, (const ObjC<T>*)h->objc_dev, \
                     dev_state<T>(h), (const T*)actions, (T*)obs, (T*)reward, term, trunc, (T*)tobs, info, CA)
  if (h->cfg.task == FW_TASK_OBJLOCK) FW_LAUNCH_COLLECT((fw_collect_kernel_obj_g8<T, FW_TASK_OBJLOCK>));
  else if (h->cfg.task == FW_TASK_WAYPOINT_OBJLOCK) FW_LAUNCH_COLLECT((fw_collect_kernel_obj_g8<T, FW_TASK_WAYPOINT_OBJLOCK>));
  else if (windy) { if (w2) FW_LAUNCH_COLLECT((fw_collect_kernel_g8w2<T, true>)); else FW_LAUNCH_COLLECT((fw_collect_kernel_g8<T, true>)); }
  else { if (w2) FW_LAUNCH_COLLECT((fw_collect_kernel_g8w2<T, false>)); else FW_LAUNCH_COLLECT((fw_collect_kernel_g8<T, false>)); }
  HIP_TRY(h, hipGetLastError());
  return FW_OK;
}
}  // namespace


// ======================================================================
// C ABI
// ======================================================================
extern "C" {

int32_t fw_sizeof_config(void) { return (int32_t)sizeof(fw_config); }
int32_t fw_abi_version(void) { return FW_ABI_VERSION; }
int32_t fw_state_dim(void) { return FW_STATE_DIM; }
int32_t fw_obs_dim(const fw_config* cfg) { return cfg ? obs_dim_of(cfg) : FW_EINVAL; }

int32_t fw_validate_config(const fw_config* cfg, char* msg, int32_t msg_len) {
  std::string m;
  int rc = validate(cfg, m);
  if (rc != FW_OK && msg && msg_len > 0) snprintf(msg, (size_t)msg_len, "%s", m.c_str());
  return rc;
}

int32_t fw_create(const fw_config* cfg, int32_t num_envs, int32_t device, uint64_t seed, int64_t global_env_offset,
                  fw_handle* out) {
  if (!cfg || !out || num_envs <= 0) { g_err = "bad arguments"; return FW_EINVAL; }
  int rc = validate(cfg, g_err);
  if (rc != FW_OK) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device available (this library has no CPU fallback)"; return FW_EHIP; }
  if (device < 0 || device >= ndev) { g_err = "device index out of range"; return FW_EINVAL; }
  fw_env* h = new (std::nothrow) fw_env();
  if (!h) return FW_ENOMEM;
  h->cfg = *cfg; h->n = num_envs; h->npad = (num_envs + kWave - 1) / kWave * kWave;
  h->device = device; h->seed = seed; h->env_offset = global_env_offset;
  // Lane mapping, by measured crossover (tools/crossover.py, profiles/r03_crossover.txt; us per eager step, fp64, median of 7 x 20):
  //   wind-free waypoints   N:  8192  12288  16384  24576  32768  65536     waypoints + wind   N:  4096  6144  8192  12288  16384
  //     8 lanes, 1 wave/SIMD    25.6   42.3   46.8   64.2   83.5  150.4                              25.5  34.8  48.8   58.1   69.8
  //     8 lanes, 2 waves/SIMD   28.0   36.8   39.1   54.0   68.1  149.6                              35.1  38.2  41.2   54.3   74.7
  //     1 lane                  57.6   58.5   59.6   63.9   66.8   80.9                              62.3  63.1  63.1   66.0   67.0
  //   8 lanes per env split an env's five lifting surfaces over lanes (latency mapping: 4096 envs are 512 waves instead of
  //   64); its full-register-file build holds ONE wave per SIMD, so above 8192 envs (1024 waves) a second round starts -- there
  //   the build capped at 256 registers (two waves per SIMD) takes over, and one lane per env above that.  The wind kernels
  //   run as many worker waves again, which lowers their thresholds.
  // FWSIM_LANES_PER_ENV=1|8 and FWSIM_G8_WAVES=1|2 override (parity tests, the benchmark sweep).
  // Camera tasks with obstacles stay on the 8-lane mapping at every size: there the cylinders are drawn by the wave from LDS
  // work lists; one lane per env tests every pixel against every cylinder (combined, 20 cylinders: 330 us vs 7.8 ms per
  // step at 32 768 envs).
  const bool cyl_camera = cfg->task != FW_TASK_WAYPOINTS && cfg->num_obstacles > 0;
  const bool wp = cfg->task == FW_TASK_WAYPOINTS, windy = wp && cfg->wind_mode != FW_WIND_OFF;
  const int one_wave_max = windy ? 6144 : 8192;                 // up to here the one-wave-per-SIMD build wins
  const int g8_max = !wp ? kG8MaxEnvs : (windy ? 12288 : 24576);   // ... and up to here the 8-lane mapping
  const bool fits8 = cfg->num_targets <= 8 && cfg->n_collision_pts <= 8;
  h->lanes_per_env = ((num_envs <= g8_max || cyl_camera) && fits8) ? 8 : 1;
  if (const char* ev = getenv("FWSIM_LANES_PER_ENV")) {
    int v = atoi(ev);
    if (v == 1 || (v == 8 && fits8)) h->lanes_per_env = v;
  }
  h->g8_waves = (h->lanes_per_env == 8 && wp && num_envs > one_wave_max) ? 2 : 1;
  if (const char* ev = getenv("FWSIM_G8_WAVES")) {
    int v = atoi(ev);
    if ((v == 1 || v == 2) && h->lanes_per_env == 8 && wp) h->g8_waves = v;
  }
  // Camera tasks on the 8-lane mapping, opt-in (FWSIM_CAPTURE_WAVE=1): a second wave per step workgroup takes the captures
  // (fwsim_objlock.hpp, "The capture wave").  Off by default: measured in round 5 it shortens the MEAN wave (ObjLock - 6 %,
  // combined - 8 %) but not the launch, which lasts as long as its slowest wave (CHANGELOG round 5).
  h->capture_wave = 0;
  if (const char* ev = getenv("FWSIM_CAPTURE_WAVE")) { if (atoi(ev) != 0 && h->lanes_per_env == 8 && !wp) h->capture_wave = 1; }
  DeviceGuard g(device);
  rc = (cfg->dtype == FW_F64) ? create_T<double>(h) : create_T<float>(h);
  if (rc != FW_OK) {
    g_err = h->err;
    free_device_buffers(h);
    delete h;
    return rc;
  }
  *out = h;
  return FW_OK;
}

int32_t fw_reset(fw_handle h, const uint8_t* mask, const fw_scenario* scenario, void* obs_out, void* hip_stream) {
  if (!h) return FW_EINVAL;
  DeviceGuard g(h->device);
  hipStream_t st = (hipStream_t)hip_stream;
  ScenOv ov{};
  if (scenario) {
    if (scenario->obstacles && !scenario->num_obstacles) { h->err = "fw_scenario: obstacles without num_obstacles"; return FW_EINVAL; }
    // stage the host arrays in one device buffer owned by the handle (stream-ordered copies, then the reset kernel)
    const size_t N = (size_t)h->n;
    const size_t sz[7] = { N * FW_MAX_TARGETS * 3, N * 3, N * FW_MAX_OBSTACLES * 3, N, N * 3, N * 3, N };
    size_t total = 0;
    for (size_t v : sz) total += v;
    if (!h->scen_dev) HIP_TRY(h, hipMalloc((void**)&h->scen_dev, sizeof(double) * total));
    std::vector<double> nobd;
    if (scenario->num_obstacles) { nobd.resize(N); for (size_t i = 0; i < N; ++i) nobd[i] = (double)scenario->num_obstacles[i]; }
    const double* src[7] = { scenario->targets, scenario->duck_pos, scenario->obstacles, scenario->num_obstacles ? nobd.data() : nullptr,
                             scenario->wind_base, scenario->gust_amp, scenario->gust_phase };
    const double* dst[7];
    size_t off = 0;
    for (int k = 0; k < 7; ++k) {
      dst[k] = nullptr;
      if (src[k]) {
        HIP_TRY(h, hipMemcpyAsync(h->scen_dev + off, src[k], sizeof(double) * sz[k], hipMemcpyHostToDevice, st));
        dst[k] = h->scen_dev + off;
      }
      off += sz[k];
    }
    HIP_TRY(h, hipStreamSynchronize(st));          // the host arrays (and `nobd`) may go away as soon as we return
    ov.targets = dst[0]; ov.duck = dst[1]; ov.obst = dst[2]; ov.nob = dst[3]; ov.wind_base = dst[4]; ov.gust_amp = dst[5]; ov.gust_phase = dst[6];
    if (!ov.obst) ov.nob = nullptr;
  }
  return (h->cfg.dtype == FW_F64) ? reset_T<double>(h, mask, obs_out, 1, st, ov) : reset_T<float>(h, mask, obs_out, 1, st, ov);
}

int32_t fw_observe(fw_handle h, void* obs_out, void* hip_stream) {
  if (!h || !obs_out) { if (h) h->err = "obs_out is NULL"; return FW_EINVAL; }
  DeviceGuard g(h->device);
  hipStream_t st = (hipStream_t)hip_stream;
  return (h->cfg.dtype == FW_F64) ? reset_T<double>(h, nullptr, obs_out, 0, st) : reset_T<float>(h, nullptr, obs_out, 0, st);
}

int32_t fw_step(fw_handle h, const void* actions, void* obs, void* reward, uint8_t* terminated, uint8_t* truncated,
                void* terminal_obs, int32_t* info_i32, void* hip_stream) {
  if (!h) return FW_EINVAL;
  if (!actions || !obs || !reward || !terminated || !truncated) { h->err = "actions/obs/reward/terminated/truncated must be non-NULL"; return FW_EINVAL; }
  DeviceGuard g(h->device);
  hipStream_t st = (hipStream_t)hip_stream;
  return (h->cfg.dtype == FW_F64)
             ? step_T<double>(h, actions, obs, reward, terminated, truncated, terminal_obs, info_i32, st)
             : step_T<float>(h, actions, obs, reward, terminated, truncated, terminal_obs, info_i32, st);
}

int32_t fw_render(fw_handle h, int32_t res, float* out, void* hip_stream) {
  if (!h) return FW_EINVAL;
  if (!out) { h->err = "fw_render: out is NULL"; return FW_EINVAL; }
  if (h->cfg.task == FW_TASK_WAYPOINTS) { h->err = "fw_render: the waypoints task has no camera"; return FW_EUNSUPPORTED; }
  if (res < 1 || res > 1024) { h->err = "fw_render: res must be in [1, 1024]"; return FW_EINVAL; }
  DeviceGuard g(h->device);
  const fw_config& c = h->cfg;
  RenderC K;
  const double th = c.camera_angle_deg * kPi / 180.0;
  const double f[3] = { std::cos(th), 0.0, std::sin(th) }, r[3] = { 0.0, -1.0, 0.0 };
  const double d[3] = { f[1] * r[2] - f[2] * r[1], f[2] * r[0] - f[0] * r[2], f[0] * r[1] - f[1] * r[0] };
  for (int k = 0; k < 3; ++k) { K.cam_f[k] = f[k]; K.cam_r[k] = r[k]; K.cam_d[k] = d[k]; K.cam_off[k] = c.camera_offset[k]; }
  K.tan_half_fov = std::tan(0.5 * c.camera_fov_deg * (kPi / 180.0));
  K.near_ = c.camera_near; K.far_ = c.camera_far; K.duck_radius = c.duck_radius_per_scale * c.duck_global_scaling; K.obst_radius = c.obstacle_radius;
  K.inv_near = 1.0 / K.near_; K.inv_far = 1.0 / K.far_; K.db_c1 = K.far_ / (K.far_ - K.near_);
  const int tile = kWave / h->lanes_per_env;
  hipStream_t st = (hipStream_t)hip_stream;
  // LDS behind the kernel's static tables: the image-plane coordinate of every pixel column / row, and from 64 x 64 pixels on a stage
  // of 2048 pixels per channel through which the image leaves as whole rows (16 KB: eight workgroups per CU keep their place).
  // Measured at 4096 envs, stage against direct stores: 32 x 32 28.1 / 26.8 us (a 1024-pixel image is two store instructions per lane
  // either way: the stage only adds its barrier), 64 x 64 62.6 / 63.5, 128 x 128 223 / 273
  int stage_px = (res >= 64 && res <= 128) ? 2048 : 0;
  if (const char* e = std::getenv("FWSIM_RENDER_STAGE")) { const int v = std::atoi(e); if (v >= 0 && v <= 4096) stage_px = v & ~3; }       // (measurement knob)
  if (((stage_px / res) & ~15) == 0) stage_px = 0;
  const size_t lds = sizeof(double) * (size_t)((res + 1) & ~1) + 2 * sizeof(float) * (size_t)stage_px;
  int threads = 256;                                     // (measured at 4096 x 32 x 32 on the 1 / t kernel with 16 x 16 tiles: 64 threads per env 52.7 us, 128: 50.8, 256: 48.2)
  if (const char* e = std::getenv("FWSIM_RENDER_THREADS")) { const int v = std::atoi(e); if (v == 64 || v == 128 || v == 256) threads = v; }   // (measurement knob)
#ifdef FW_RENDER_PROF
  static long long* rprof = nullptr; static size_t rprof_n = 0;
  const size_t rp_n = (size_t)h->n * 4 * 8;
  if (rprof_n < rp_n) { if (rprof) (void)hipFree(rprof); HIP_TRY(h, hipMalloc((void**)&rprof, rp_n * sizeof(long long))); rprof_n = rp_n; }
  HIP_TRY(h, hipMemsetAsync(rprof, 0, rp_n * sizeof(long long), st));
  HIP_TRY(h, hipMemcpyToSymbolAsync(HIP_SYMBOL(g_render_prof), &rprof, sizeof(rprof), 0, hipMemcpyHostToDevice, st));
#endif
  if (c.dtype == FW_F64) {
    if (stage_px) hipLaunchKernelGGL((fw_render_kernel<double, true>), dim3((unsigned)h->n), dim3(threads), lds, st, (const double*)h->r_dev, tile, h->n, threads / 64, K, res, out, stage_px);
    else hipLaunchKernelGGL((fw_render_kernel<double, false>), dim3((unsigned)h->n), dim3(threads), lds, st, (const double*)h->r_dev, tile, h->n, threads / 64, K, res, out, 0);
  } else {
    if (stage_px) hipLaunchKernelGGL((fw_render_kernel<float, true>), dim3((unsigned)h->n), dim3(threads), lds, st, (const float*)h->r_dev, tile, h->n, threads / 64, K, res, out, stage_px);
    else hipLaunchKernelGGL((fw_render_kernel<float, false>), dim3((unsigned)h->n), dim3(threads), lds, st, (const float*)h->r_dev, tile, h->n, threads / 64, K, res, out, 0);
  }
  HIP_TRY(h, hipGetLastError());
#ifdef FW_RENDER_PROF
  if (std::getenv("FWSIM_RENDER_PROF_DUMP")) {       // per-wave cycle stamps -> mean phase lengths, set-up waves and the others apart
    HIP_TRY(h, hipStreamSynchronize(st));
    std::vector<long long> P(rp_n);
    HIP_TRY(h, hipMemcpy(P.data(), rprof, rp_n * sizeof(long long), hipMemcpyDeviceToHost));
    const int nw = threads / 64;
    double su[6] = {0}, ot[6] = {0}; long long nsu = 0, not_ = 0; long long t0min = -1, t5max = 0; double life_su = 0, life_ot = 0;
    for (int e = 0; e < h->n; ++e) for (int w = 0; w < nw; ++w) {
      const long long* q = &P[((size_t)e * 4 + w) * 8];
      if (!q[0] || !q[5]) continue;
      if (t0min < 0 || q[0] < t0min) t0min = q[0];
      if (q[5] > t5max) t5max = q[5];
      const bool is_su = (w == (e & (nw - 1))) || (w == ((e + 1) & (nw - 1)));
      double* d = is_su ? su : ot;
      if (is_su) { d[0] += (double)(q[1] - q[0]); d[1] += (double)(q[2] - q[1]); ++nsu; life_su += (double)(q[5] - q[0]); }
      else { d[0] += 0; d[1] += (double)(q[2] - q[0]); ++not_; life_ot += (double)(q[5] - q[0]); }
      d[2] += (double)(q[3] - q[2]); d[3] += (double)(q[4] - q[3]); d[4] += (double)(q[5] - q[4]);
    }
    std::fprintf(stderr, "fw_render wave profile (cycles of the counter, mean per wave; res %d, %d envs, %d waves per env, stage %d)\n", res, h->n, nw, stage_px);
    std::fprintf(stderr, "  set-up waves (%lld): start->gather in %.0f, set-up arithmetic %.0f, at the barrier %.0f, constants %.0f, strips %.0f, life %.0f\n",
                 nsu, su[0] / nsu, su[1] / nsu, su[2] / nsu, su[3] / nsu, su[4] / nsu, life_su / nsu);
    std::fprintf(stderr, "  other waves  (%lld): start->barrier arrival %.0f, at the barrier %.0f, constants %.0f, strips %.0f, life %.0f\n",
                 not_, ot[1] / not_, ot[2] / not_, ot[3] / not_, ot[4] / not_, life_ot / not_);
    std::fprintf(stderr, "  first stamp to last stamp of the launch: %lld\n", t5max - t0min);
  }
#endif
  return FW_OK;
}

int32_t fw_seed(fw_handle h, uint64_t seed) {
  if (!h) return FW_EINVAL;
  DeviceGuard g(h->device);
  h->seed = seed;
  HIP_TRY(h, hipDeviceSynchronize());
  int rc = (h->cfg.dtype == FW_F64) ? upload_params<double>(h) : upload_params<float>(h);
  if (rc != FW_OK) return rc;
  {
    // episode counters back to -1 (the next reset starts episode 0 of the new seed); rows are tiled, so go through a host copy
    const int tile = kWave / h->lanes_per_env;
    std::vector<int32_t> iv((size_t)IF_COUNT * h->npad);
    HIP_TRY(h, hipMemcpy(iv.data(), h->i_dev, sizeof(int32_t) * iv.size(), hipMemcpyDeviceToHost));
    for (int e = 0; e < h->npad; ++e) iv[tile_index(tile, IF_COUNT, IF_EPISODE, e)] = -1;
    HIP_TRY(h, hipMemcpy(h->i_dev, iv.data(), sizeof(int32_t) * iv.size(), hipMemcpyHostToDevice));
  }
  return invalidate_shadow(h);                 // shadows were drawn with the old seed
}

int32_t fw_get_state(fw_handle h, double* state_out) {
  if (!h || !state_out) return FW_EINVAL;
  DeviceGuard g(h->device);
  return (h->cfg.dtype == FW_F64) ? get_state_T<double>(h, state_out) : get_state_T<float>(h, state_out);
}

int32_t fw_set_state(fw_handle h, const double* state_in) {
  if (!h || !state_in) return FW_EINVAL;
  DeviceGuard g(h->device);
  return (h->cfg.dtype == FW_F64) ? set_state_T<double>(h, state_in) : set_state_T<float>(h, state_in);
}

int32_t fw_get_counters(fw_handle h, uint64_t* out) {
  if (!h || !out) return FW_EINVAL;
  DeviceGuard g(h->device);
  unsigned long long w[FW_CTR_DIM];
  uint32_t launches = 0;
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipMemcpy(w, h->stats_dev, sizeof w, hipMemcpyDeviceToHost));
  HIP_TRY(h, hipMemcpy(&launches, h->lctr_dev, sizeof launches, hipMemcpyDeviceToHost));
  for (int i = 0; i < FW_CTR_DIM; ++i) out[i] = w[i];
  out[FW_CTR_LAUNCHES] = launches;
  return FW_OK;
}

int32_t fw_gae(const float* rewards, const float* values, const float* episode_starts, const float* last_values,
               const float* last_dones, float* advantages, float* returns, int32_t T, int32_t N, float gamma,
               float gae_lambda, void* hip_stream) {
  if (!rewards || !values || !episode_starts || !last_values || !last_dones || !advantages || !returns || T <= 0 || N <= 0) {
    g_err = "fw_gae: bad arguments"; return FW_EINVAL;
  }
  DeviceGuard g(device_of(rewards));
  hipLaunchKernelGGL(fw_gae_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, rewards, values,
                     episode_starts, last_values, last_dones, advantages, returns, T, N, gamma, gae_lambda);
  HIP_TRY((fw_env*)nullptr, hipGetLastError());
  return FW_OK;
}

int32_t fw_eval_track(const void* reward, int32_t reward_is_f64, const uint8_t* terminated, const uint8_t* truncated, const int32_t* info,
                      int32_t info_dim, const int64_t* targets, int64_t* counts, double* cur_rew, int64_t* cur_len, int64_t* step_ctr,
                      double* fin_rew, int64_t* fin_len, int64_t* fin_step, int32_t* fin_info, int32_t N, int32_t E, void* hip_stream) {
  if (!reward || !terminated || !truncated || !targets || !counts || !cur_rew || !cur_len || !step_ctr || !fin_rew || !fin_len || !fin_step ||
      N <= 0 || E <= 0 || (info && (info_dim <= 0 || !fin_info))) { g_err = "fw_eval_track: bad arguments"; return FW_EINVAL; }
  EvalTrackArgs A;
  A.reward = reward; A.reward_is_f64 = reward_is_f64; A.terminated = terminated; A.truncated = truncated; A.info = info; A.info_dim = info_dim;
  A.targets = targets; A.counts = counts; A.cur_rew = cur_rew; A.cur_len = cur_len; A.step_ctr = step_ctr;
  A.fin_rew = fin_rew; A.fin_len = fin_len; A.fin_step = fin_step; A.fin_info = fin_info; A.N = N; A.E = E;
  DeviceGuard g(device_of(reward));
  hipLaunchKernelGGL(fw_eval_track_kernel, dim3(1), dim3(256), 0, (hipStream_t)hip_stream, A);
  HIP_TRY((fw_env*)nullptr, hipGetLastError());
  return FW_OK;
}

int64_t fw_normalize_obs_workspace_bytes(int32_t D) { return D > 0 ? (int64_t)sizeof(double) * 64 * 2 * D : FW_EINVAL; }

int32_t fw_normalize_obs(const void* obs, int32_t in_is_f64, int32_t N, int32_t D, double* mean, double* var,
                         double* count, int32_t update, float clip, float eps, float* obs_out, void* workspace,
                         double* batch_acc, void* hip_stream) {
  if (!obs || !mean || !var || !count || !obs_out || N <= 0 || D <= 0 || D > 256) { g_err = "fw_normalize_obs: bad arguments"; return FW_EINVAL; }
  if (update && !workspace) { g_err = "fw_normalize_obs: update needs a workspace of fw_normalize_obs_workspace_bytes(D) bytes"; return FW_EINVAL; }
  hipStream_t st = (hipStream_t)hip_stream;
  DeviceGuard g(device_of(obs));
  if (update) {
    // per-block column sums land in the CALLER's workspace: nothing here is shared between callers or streams
    double* scratch = (double*)workspace;
    const int nblocks = N >= 64 * 64 ? 64 : (N + 63) / 64;       // >= 64 rows per block
    if (in_is_f64) hipLaunchKernelGGL(fw_obs_moments_kernel<double>, dim3(nblocks), dim3(256), 0, st, (const double*)obs, N, D, scratch);
    else hipLaunchKernelGGL(fw_obs_moments_kernel<float>, dim3(nblocks), dim3(256), 0, st, (const float*)obs, N, D, scratch);
    hipLaunchKernelGGL(fw_obs_merge_kernel, dim3(1), dim3(256), 0, st, scratch, nblocks, N, D, mean, var, count, batch_acc);
  }
  const int total = N * D;
  if (in_is_f64) hipLaunchKernelGGL(fw_obs_normalize_kernel<double>, dim3((total + 255) / 256), dim3(256), 0, st, (const double*)obs, total, D, mean, var, clip, eps, obs_out);
  else hipLaunchKernelGGL(fw_obs_normalize_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)obs, total, D, mean, var, clip, eps, obs_out);
  HIP_TRY((fw_env*)nullptr, hipGetLastError());
  return FW_OK;
}

int32_t fw_ppo_param_count(int32_t obs_dim) { return obs_dim > 0 ? ppo_total_params((obs_dim + 1) & ~1) : FW_EINVAL; }
int32_t fw_ppo_moment_count(void) { return 2 * kPMomentSlots; }   // (second half: unused since round 4 -- the moments live in registers during a call; kept for the ABI's buffer size)
int32_t fw_ppo_moment_map(int32_t obs_dim, int32_t* flat_index_of_slot) {
  if (obs_dim <= 0 || obs_dim > 64 || !flat_index_of_slot) { g_err = "fw_ppo_moment_map: bad arguments"; return FW_EINVAL; }
  ppo_moment_map(obs_dim, flat_index_of_slot);
  for (int i = kPMomentSlots; i < 2 * kPMomentSlots; ++i) flat_index_of_slot[i] = -1;
  return FW_OK;
}

// Dev knob: FWSIM_SPIN_LOG2=k bounds every in-grid wait (fw_collect_step, fw_ppo_update) to 2^k polls instead of its default --
// tests use it to provoke the timeout paths and assert that the status words surface on the host side.
static long long spin_budget(long long dflt) {
  if (const char* e = getenv("FWSIM_SPIN_LOG2")) { const int k = atoi(e); if (k >= 0 && k < 40) return std::min(dflt, 1ll << k); }
  return dflt;
}

// workspace layout of fw_ppo_update: [0, kPpoWords x 8) exchange words (kPpoWordPaths: which exchanges shared an L2, kPpoWordStatus: include/fwsim.h) |
// gradient hand-off buffer | the packed rows of every minibatch, in walking order (fw_ppo_pack_kernel)
// (the words first -- everything that is polled -- and the data a whole 64 KB behind them: the blocks that wait spin on these lines with
// loads past their L1, and whatever part of the L2 serves the words should not also serve the first partial's tiles)
static constexpr size_t kPpoWsXch = 65536;
static_assert(kPpoWords * sizeof(unsigned long long) <= kPpoWsXch, "exchange words");
static constexpr size_t kPpoWsGx = sizeof(float) * (4 * kPMaxSplit * (size_t)kPGxSlots);      // [parity][net][part]: gradient partial + weight share
int64_t fw_ppo_update_workspace_bytes(int32_t n_minibatches, int32_t batch_size, int32_t obs_dim) {
  if (n_minibatches <= 0 || batch_size <= 0 || obs_dim <= 0 || obs_dim > 64) return FW_EINVAL;
  return (int64_t)(kPpoWsXch + kPpoWsGx + sizeof(float) * (size_t)n_minibatches * (size_t)batch_size * (size_t)ppo_pack_width(obs_dim));
}

int32_t fw_ppo_update(float* params, float* mom_m, float* mom_v, const float* obs, const float* act, const float* old_logp,
                      const float* adv, const float* ret, const int32_t* perm, int32_t n_minibatches, int32_t batch_size,
                      int32_t obs_dim, const fw_ppo_hyper* hyper, float* loss_acc, void* workspace, int64_t workspace_bytes,
                      void* hip_stream) {
  static_assert(sizeof(fw_ppo_hyper) == sizeof(PpoHyper), "fw_ppo_hyper layout");
  if (!params || !mom_m || !mom_v || !obs || !act || !old_logp || !adv || !ret || !perm || !hyper || n_minibatches <= 0) {
    g_err = "fw_ppo_update: bad arguments"; return FW_EINVAL;
  }
  if (batch_size <= 0 || batch_size % 16 != 0) { g_err = "fw_ppo_update: batch_size must be a multiple of 16"; return FW_EINVAL; }
  if (obs_dim <= 0 || obs_dim > 64) { g_err = "fw_ppo_update: obs_dim must be in [1, 64]"; return FW_EINVAL; }
  if (!workspace || workspace_bytes < fw_ppo_update_workspace_bytes(n_minibatches, batch_size, obs_dim)) {
    g_err = "fw_ppo_update: workspace smaller than fw_ppo_update_workspace_bytes(n_minibatches, batch_size, obs_dim)"; return FW_EINVAL;
  }
  const size_t lds = ppo_lds_bytes(obs_dim);
  if (lds > 160 * 1024) { g_err = "fw_ppo_update: networks do not fit the 160 KB of LDS"; return FW_EINVAL; }
  hipStream_t st = (hipStream_t)hip_stream;
  const int dev = device_of(params);
  DeviceGuard g(dev);
  int rc = ensure_learner_lds(dev, 0, lds);
  if (rc != FW_OK) return rc;
  // everything the blocks exchange lives in the caller's workspace: two learners (or two streams) never share a word
  unsigned long long* xch = (unsigned long long*)workspace;
  float* gx = (float*)((char*)workspace + kPpoWsXch);
  float* packed = (float*)((char*)workspace + kPpoWsXch + kPpoWsGx);
  HIP_TRY((fw_env*)nullptr, hipMemsetAsync(xch, 0, kPpoWords * sizeof(unsigned long long), st));
  PpoArgs A;
  A.params = params; A.mom_m = mom_m; A.mom_v = mom_v; A.packed = packed;
  A.n_mb = n_minibatches; A.B = batch_size; A.D = obs_dim; A.loss_acc = loss_acc; A.xch = xch; A.gx = gx;
  A.spin = spin_budget(kPpoSpin);
  A.flags = 0;
  if (const char* e = getenv("FWSIM_PPO_NO_L2_SWAP")) if (atoi(e) != 0) A.flags |= PPO_FLAG_NO_L2_SWAP;
  if (const char* e = getenv("FWSIM_PPO_WRITER")) if (!strcmp(e, "last")) A.flags |= PPO_FLAG_WRITER_LAST;
  std::memcpy(&A.H, hyper, sizeof A.H);
  if (batch_size <= 1 && A.H.norm_adv == 1) A.H.norm_adv = 0;      // SB3 skips the normalisation of single-sample minibatches
  // the parallel pre-pass: every minibatch's rows, in walking order, advantages normalised (one workgroup per minibatch, all CUs)
  PpoPackArgs P;
  P.obs = obs; P.act = act; P.old_logp = old_logp; P.adv = adv; P.ret = ret; P.perm = perm; P.B = batch_size; P.D = obs_dim;
  P.norm_adv = A.H.norm_adv; P.adv_mean = A.H.adv_mean; P.adv_std = A.H.adv_std; P.out = packed;
  hipLaunchKernelGGL(fw_ppo_pack_kernel, dim3(n_minibatches), dim3(256), 0, st, P);
  // four / eight blocks per network: gradient tiles by reduce-scatter, updated weights by all-gather (FWSIM_PPO_RS=0: all-to-all, as for two
  // blocks -- written for up to four, so the cut is then chosen among round 4's)
  bool rs_env = true;
  if (const char* e = getenv("FWSIM_PPO_RS")) if (atoi(e) == 0) rs_env = false;
  PpoSplit cut = ppo_split(batch_size, rs_env ? kPMaxSplit : 4);      // samples per pass and blocks per network (128 samples: 16 x 8, 64: 16 x 4)
  if (const char* e = getenv("FWSIM_PPO_SPLIT")) {  // dev knob "CHxN" (64x2 = round 3's cut, 32x4 = round 4's for 128 samples): A / B measurements, tests of every form
    int ch = 0, ns = 0;
    if (sscanf(e, "%dx%d", &ch, &ns) == 2 && (ch == 16 || ch == 32 || ch == 64) && (ns == 1 || ns == 2 || ns == 4 || (ns == 8 && rs_env)) && batch_size % ch == 0 && batch_size / ch >= ns) {
      cut.ch = ch; cut.nsplit = ns;
    } else { g_err = "fw_ppo_update: FWSIM_PPO_SPLIT must be CHxN with CH in {16, 32, 64}, N in {1, 2, 4, 8} (8: not with FWSIM_PPO_RS=0), N chunks of CH samples in a minibatch"; return FW_EINVAL; }
  }
  const int ns = cut.nsplit >= 4 && rs_env ? cut.nsplit : 0;      // the kernel's NS: 0 = all-to-all swap of whole partials
  const dim3 grid(16 * cut.nsplit);                 // (every 8th block works -- see the kernel)
#define FW_PPO_LAUNCH(CH_, NS_) hipLaunchKernelGGL((fw_ppo_update_kernel<CH_, NS_>), grid, dim3(kPThreads), lds, st, A)
  if (cut.ch == 64) { if (ns == 8) FW_PPO_LAUNCH(64, 8); else if (ns == 4) FW_PPO_LAUNCH(64, 4); else FW_PPO_LAUNCH(64, 0); }
  else if (cut.ch == 32) { if (ns == 8) FW_PPO_LAUNCH(32, 8); else if (ns == 4) FW_PPO_LAUNCH(32, 4); else FW_PPO_LAUNCH(32, 0); }
  else { if (ns == 8) FW_PPO_LAUNCH(16, 8); else if (ns == 4) FW_PPO_LAUNCH(16, 4); else FW_PPO_LAUNCH(16, 0); }
#undef FW_PPO_LAUNCH
  HIP_TRY((fw_env*)nullptr, hipGetLastError());
  return FW_OK;
}

int32_t fw_ppo_update_status(const void* workspace, int64_t workspace_bytes, uint32_t* status_out, uint32_t* paths_out, void* hip_stream) {
  if (!workspace || workspace_bytes < (int64_t)kPpoWsXch || !status_out) { g_err = "fw_ppo_update_status: bad arguments"; return FW_EINVAL; }
  DeviceGuard g(device_of(workspace));
  unsigned long long w[2] = {0ull, 0ull};
  HIP_TRY((fw_env*)nullptr, hipMemcpyAsync(w, (const unsigned long long*)workspace + kPpoWordPaths, sizeof w, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
  HIP_TRY((fw_env*)nullptr, hipStreamSynchronize((hipStream_t)hip_stream));
  static_assert(kPpoWordStatus == kPpoWordPaths + 1 && kPpoWordStatus < kPpoWords, "exchange-word layout");
  static_assert(FW_PPO_ST_IDS == PPO_ST_IDS && FW_PPO_ST_SWAP == PPO_ST_SWAP && FW_PPO_ST_NORM == PPO_ST_NORM && FW_PPO_ST_COMMIT == PPO_ST_COMMIT, "status bits of include/fwsim.h");
  if (paths_out) *paths_out = (uint32_t)w[0];
  *status_out = (uint32_t)w[1];
  return FW_OK;
}

int32_t fw_policy_act(const float* params, const float* obs, int32_t N, int32_t obs_dim, int32_t nets, int32_t deterministic,
                      const uint64_t* rng, int64_t env_offset, float* obs_copy, float* act_raw, void* act_env, int32_t act_is_f64,
                      float* logp, float* value, void* hip_stream) {
  if (!params || !obs || N <= 0 || obs_dim <= 0 || obs_dim > 64 || (nets & ~3) || !nets) { g_err = "fw_policy_act: bad arguments"; return FW_EINVAL; }
  if ((nets & 1) && (!act_raw || !act_env || !logp || (!deterministic && !rng))) { g_err = "fw_policy_act: policy outputs missing"; return FW_EINVAL; }
  if ((nets & 2) && !value) { g_err = "fw_policy_act: value output missing"; return FW_EINVAL; }
  const size_t lds = act_lds_bytes(obs_dim);
  const int dev = device_of(params);
  DeviceGuard g(dev);
  if (int rc = ensure_learner_lds(dev, 1, lds)) return rc;
  ActArgs A;
  std::memset(&A, 0, sizeof A);
  A.params = params; A.obs = obs; A.N = N; A.D = obs_dim; A.nets = nets; A.deterministic = deterministic; A.act_is_f64 = act_is_f64;
  A.rng = rng; A.env_offset = env_offset; A.obs_copy = obs_copy; A.act_raw = act_raw; A.act_env = act_env; A.logp = logp; A.value = value;
  A.raw = nullptr; A.raw_is_f64 = 0; A.mean = A.var = nullptr; A.clip = A.eps = 0.f; A.terminated = A.truncated = nullptr;
  hipLaunchKernelGGL(fw_policy_act_kernel, dim3((N + kPChunk - 1) / kPChunk, 2), dim3(kPThreads), lds, (hipStream_t)hip_stream, A);
  HIP_TRY((fw_env*)nullptr, hipGetLastError());
  return FW_OK;
}

int32_t fw_collect_act(const float* params, const void* raw_obs, int32_t obs_is_f64, int32_t N, int32_t obs_dim, const double* obs_mean,
                       const double* obs_var, float clip_obs, float eps_obs, int32_t nets, int32_t deterministic, const uint64_t* rng,
                       int64_t env_offset, float* obs_copy, float* act_raw, void* act_env, int32_t act_is_f64, float* logp, float* value,
                       const void* prev_reward, const uint8_t* prev_terminated, const uint8_t* prev_truncated, const void* prev_terminal_obs,
                       const double* ret_var, int32_t norm_reward, float clip_reward, float eps_reward, float gamma, float* rew_out,
                       float* start_out, void* hip_stream) {
  if (!params || !raw_obs || !obs_mean || !obs_var || N <= 0 || obs_dim <= 0 || obs_dim > 64 || (nets & ~3) || !nets) { g_err = "fw_collect_act: bad arguments"; return FW_EINVAL; }
  if ((nets & 1) && (!act_raw || !act_env || !logp || (!deterministic && !rng))) { g_err = "fw_collect_act: policy outputs missing"; return FW_EINVAL; }
  if ((nets & 2) && !value) { g_err = "fw_collect_act: value output missing"; return FW_EINVAL; }
  if (prev_reward && (!(nets & 2) || !prev_terminated || !prev_truncated || !prev_terminal_obs || !ret_var || !rew_out || !start_out)) {
    g_err = "fw_collect_act: finalising the previous step needs the value network and all of its buffers"; return FW_EINVAL;
  }
  const size_t lds = act_lds_bytes(obs_dim);
  const int dev = device_of(params);
  DeviceGuard g(dev);
  if (int rc = ensure_learner_lds(dev, 1, lds)) return rc;
  ActArgs A;
  std::memset(&A, 0, sizeof A);
  A.params = params; A.N = N; A.D = obs_dim; A.nets = nets; A.deterministic = deterministic; A.act_is_f64 = act_is_f64;
  A.rng = rng; A.env_offset = env_offset; A.obs_copy = obs_copy; A.act_raw = act_raw; A.act_env = act_env; A.logp = logp; A.value = value;
  A.raw = raw_obs; A.raw_is_f64 = obs_is_f64; A.mean = obs_mean; A.var = obs_var; A.clip = clip_obs; A.eps = eps_obs;
  A.prev_reward = prev_reward; A.prev_term = prev_terminated; A.prev_trunc = prev_truncated; A.prev_tobs = prev_terminal_obs;
  A.ret_var = ret_var; A.norm_reward = norm_reward; A.clip_reward = clip_reward; A.rew_eps = eps_reward; A.gamma = gamma;
  A.rew_out = rew_out; A.start_out = start_out;
  hipLaunchKernelGGL(fw_policy_act_kernel, dim3((N + kPChunk - 1) / kPChunk, 2), dim3(kPThreads), lds, (hipStream_t)hip_stream, A);
  HIP_TRY((fw_env*)nullptr, hipGetLastError());
  return FW_OK;
}

int64_t fw_collect_step_workspace_bytes(fw_handle h) { return h ? (int64_t)collect_ws(h).total : FW_EINVAL; }

// checks and argument block shared by fw_collect_step and fw_collect_close
static int32_t collect_fill(fw_handle h, const fw_collect_args* a, const char* who, bool close, CollectArgs& CA) {
  if (h->lanes_per_env != 8) {
    h->err = std::string(who) + " serves the 8-lanes-per-env mapping (what fw_create picks for waypoints up to 24576 envs per GPU, 12288 with wind, and "
             "for the camera tasks up to 16384 envs or at any size with obstacles); use fw_collect_act / fw_step / fw_collect_stats";
    return FW_EUNSUPPORTED;
  }
  const int D = obs_dim_of(&h->cfg), N = h->n;
  if (D > 62) { h->err = std::string(who) + ": obs_dim must be <= 62"; return FW_EINVAL; }
  const bool pol = close || (a->act_raw && a->act_env && a->logp && (a->deterministic || a->rng));
  if (!a->params || !a->obs_mean || !a->obs_var || !a->obs_count || !a->returns || !a->ret_mean || !a->ret_var || !a->ret_count ||
      !pol || !a->value || !a->obs || !a->reward || !a->terminated || !a->truncated || !a->terminal_obs ||
      (!a->rew_out != !a->start_out) || (close && !a->rew_out)) { h->err = std::string(who) + ": missing buffers"; return FW_EINVAL; }
  const CollectWs W = collect_ws(h);
  if (!a->workspace || a->workspace_bytes < (int64_t)W.total) { h->err = std::string(who) + ": workspace smaller than fw_collect_step_workspace_bytes()"; return FW_EINVAL; }
  const int f64 = h->cfg.dtype == FW_F64;
  std::memset(&CA, 0, sizeof CA);
  CA.n_chunks = (N + kCRows - 1) / kCRows;
  CA.n_act = (2 * CA.n_chunks + 1 + 7) & ~7;                 // act waves + the merge wave, padded: the step waves keep their XCD alignment
  CA.nblk = (int32_t)(h->npad / 8);
  ActArgs& A = CA.A;
  A.params = a->params; A.N = N; A.D = D; A.nets = close ? 2 : 3; A.deterministic = a->deterministic; A.act_is_f64 = f64;
  A.rng = a->rng; A.env_offset = h->env_offset; A.obs_copy = a->obs_copy; A.act_raw = a->act_raw; A.act_env = a->act_env; A.logp = a->logp; A.value = a->value;
  A.raw = a->obs; A.raw_is_f64 = f64; A.mean = a->obs_mean; A.var = a->obs_var; A.clip = a->clip_obs; A.eps = a->eps_obs;
  if (a->rew_out) {
    A.prev_reward = a->reward; A.prev_term = a->terminated; A.prev_trunc = a->truncated; A.prev_tobs = a->terminal_obs;
    A.rew_out = a->rew_out; A.start_out = a->start_out;
  }
  A.ret_var = a->ret_var; A.norm_reward = a->norm_reward; A.clip_reward = a->clip_reward; A.rew_eps = a->eps_reward; A.gamma = (float)a->gamma;
  StatsArgs& S = CA.S;
  S.obs = a->obs; S.obs_is_f64 = f64; S.N = N; S.D = D; S.mean = a->obs_mean; S.var = a->obs_var; S.count = a->obs_count; S.update_obs = a->update_obs;
  S.reward = a->reward; S.rew_is_f64 = f64; S.terminated = a->terminated; S.truncated = a->truncated; S.returns = a->returns;
  S.ret_mean = a->ret_mean; S.ret_var = a->ret_var; S.ret_count = a->ret_count; S.update_ret = a->update_ret; S.gamma = a->gamma; S.rng = a->rng;
  S.obs_acc = a->obs_acc; S.ret_acc = a->ret_acc;
  char* ws = (char*)a->workspace;
  CA.part1 = (double*)(ws + W.part1); CA.tot = (double*)(ws + W.tot);
  CA.flag_p = (unsigned int*)(ws + W.flag_p); CA.flag_v = (unsigned int*)(ws + W.flag_v);
  CA.sync = (unsigned int*)(ws + W.sync);
  CA.trace = close ? nullptr : (long long*)a->trace;
  CA.spin = (int32_t)spin_budget(kCollectSpin);
  return FW_OK;
}

int32_t fw_collect_step(fw_handle h, const fw_collect_args* a, void* hip_stream) {
  if (!h || !a) return FW_EINVAL;
  CollectArgs CA;
  if (int32_t rc = collect_fill(h, a, "fw_collect_step", false, CA)) return rc;
  DeviceGuard g(h->device);
  hipStream_t st = (hipStream_t)hip_stream;
  if (h->collect_act_seen != a->act_env) {
    // a new action buffer: every word "not there yet" (NaN) -- the step waves see their actions replace it and put it back at the
    // end of a launch.  (Stream-ordered; harmless if a captured graph replays it: between launches the words are NaN anyway.)
    const size_t bytes = (size_t)h->n * 4 * (h->cfg.dtype == FW_F64 ? 8 : 4);
    HIP_TRY(h, hipMemsetAsync(a->act_env, 0xFF, bytes, st));      // all-ones words: a NaN in float32 and in float64
    h->collect_act_seen = a->act_env;
  }
  return h->cfg.dtype == FW_F64 ? collect_step_T<double>(h, CA, a->act_env, a->obs, a->reward, a->terminated, a->truncated, a->terminal_obs, a->info_i32, st)
                                : collect_step_T<float>(h, CA, a->act_env, a->obs, a->reward, a->terminated, a->truncated, a->terminal_obs, a->info_i32, st);
}

int32_t fw_collect_close(fw_handle h, const fw_collect_args* a, const fw_collect_close_args* c, void* hip_stream) {
  if (!h || !a || !c) return FW_EINVAL;
  CollectArgs CA;
  if (int32_t rc = collect_fill(h, a, "fw_collect_close", true, CA)) return rc;
  if (!c->rewards || !c->values || !c->episode_starts || !c->adv || !c->ret || c->T <= 0) { h->err = "fw_collect_close: missing GAE buffers"; return FW_EINVAL; }
  if (a->rew_out != c->rewards + (size_t)(c->T - 1) * h->n) { h->err = "fw_collect_close: rew_out must be row T - 1 of the rewards buffer"; return FW_EINVAL; }
  DeviceGuard g(h->device);
  CloseArgs GA;
  GA.rewards = c->rewards; GA.values = c->values; GA.episode_starts = c->episode_starts; GA.adv = c->adv; GA.ret = c->ret;
  GA.T = c->T; GA.gamma = c->gae_gamma; GA.lam = c->gae_lambda;
  const size_t lds = collect_act_lds_bytes(CA.A.D);
  hipStream_t st = (hipStream_t)hip_stream;
  if (h->cfg.dtype == FW_F64) hipLaunchKernelGGL(fw_collect_close_kernel<double>, dim3((unsigned)CA.n_chunks + 1), dim3(64), lds, st, CA, GA);
  else hipLaunchKernelGGL(fw_collect_close_kernel<float>, dim3((unsigned)CA.n_chunks + 1), dim3(64), lds, st, CA, GA);
  HIP_TRY(h, hipGetLastError());
  return FW_OK;
}

int32_t fw_collect_workspace_init(fw_handle h, void* workspace, int64_t workspace_bytes, void* hip_stream) {
  if (!h) return FW_EINVAL;
  if (h->lanes_per_env != 8) { h->err = "fw_collect_workspace_init: this handle's lane mapping has no fw_collect_step"; return FW_EUNSUPPORTED; }
  const CollectWs W = collect_ws(h);
  if (!workspace || workspace_bytes < (int64_t)W.total) { h->err = "fw_collect_workspace_init: workspace smaller than fw_collect_step_workspace_bytes()"; return FW_EINVAL; }
  DeviceGuard g(h->device);
  hipStream_t st = (hipStream_t)hip_stream;
  char* ws = (char*)workspace;
  HIP_TRY(h, hipMemsetAsync(ws, 0, W.total, st));
  const size_t n = (W.tot - W.part1) / sizeof(double);
  hipLaunchKernelGGL(fw_collect_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (double*)(ws + W.part1), n, (unsigned int*)(ws + W.sync));
  HIP_TRY(h, hipGetLastError());
  return FW_OK;
}

int32_t fw_collect_finish(fw_handle h, const fw_collect_args* a, void* hip_stream) {
  if (!h || !a) return FW_EINVAL;
  if (h->lanes_per_env != 8) { h->err = "fw_collect_finish: this handle's lane mapping has no fw_collect_step"; return FW_EUNSUPPORTED; }
  if (!a->obs_mean || !a->obs_var || !a->obs_count || !a->ret_mean || !a->ret_var || !a->ret_count) { h->err = "fw_collect_finish: missing statistics buffers"; return FW_EINVAL; }
  const CollectWs W = collect_ws(h);
  if (!a->workspace || a->workspace_bytes < (int64_t)W.total) { h->err = "fw_collect_finish: workspace smaller than fw_collect_step_workspace_bytes()"; return FW_EINVAL; }
  DeviceGuard g(h->device);
  CollectArgs CA;
  std::memset(&CA, 0, sizeof CA);
  CA.n_chunks = (h->n + kCRows - 1) / kCRows; CA.nblk = (int32_t)(h->npad / 8);
  StatsArgs& S = CA.S;
  S.N = h->n; S.D = obs_dim_of(&h->cfg); S.mean = a->obs_mean; S.var = a->obs_var; S.count = a->obs_count; S.update_obs = a->update_obs;
  S.ret_mean = a->ret_mean; S.ret_var = a->ret_var; S.ret_count = a->ret_count; S.update_ret = a->update_ret; S.obs_acc = a->obs_acc; S.ret_acc = a->ret_acc;
  char* ws = (char*)a->workspace;
  CA.part1 = (double*)(ws + W.part1); CA.tot = (double*)(ws + W.tot); CA.sync = (unsigned int*)(ws + W.sync);
  hipLaunchKernelGGL(fw_collect_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, CA);
  HIP_TRY(h, hipGetLastError());
  return FW_OK;
}

int32_t fw_collect_status(fw_handle h, const void* workspace, int64_t workspace_bytes, uint32_t* status_out, void* hip_stream) {
  if (!h || !status_out) return FW_EINVAL;
  const CollectWs W = collect_ws(h);
  if (!workspace || workspace_bytes < (int64_t)W.total) { h->err = "fw_collect_status: workspace smaller than fw_collect_step_workspace_bytes()"; return FW_EINVAL; }
  static_assert(FW_COLLECT_ST_ACTIONS == CS_ST_ACTIONS && FW_COLLECT_ST_FOLD == CS_ST_FOLD && FW_COLLECT_ST_MERGE == CS_ST_MERGE &&
                FW_COLLECT_ST_NOINIT == CS_ST_NOINIT && FW_COLLECT_ST_EPOCH == CS_ST_EPOCH && FW_COLLECT_ST_NANACT == CS_ST_NANACT, "status bits of include/fwsim.h");
  DeviceGuard g(h->device);
  uint32_t w = 0;
  HIP_TRY(h, hipMemcpyAsync(&w, (const char*)workspace + W.sync + sizeof(uint32_t) * CS_STATUS, sizeof w, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
  HIP_TRY(h, hipStreamSynchronize((hipStream_t)hip_stream));
  *status_out = w;
  return FW_OK;
}

static int collect_stats_blocks(int N) { return N >= 64 * 64 ? 64 : (N + 63) / 64; }
int64_t fw_collect_stats_workspace_bytes(int32_t D) { return D > 0 ? (int64_t)(sizeof(double) * 64 * (2 * (size_t)D + 2) + 64) : FW_EINVAL; }

int32_t fw_collect_stats(const void* obs, int32_t obs_is_f64, int32_t N, int32_t D, double* obs_mean, double* obs_var, double* obs_count,
                         int32_t update_obs, const void* reward, int32_t rew_is_f64, const uint8_t* terminated, const uint8_t* truncated,
                         double* returns, double* ret_mean, double* ret_var, double* ret_count, int32_t update_ret, double gamma,
                         uint64_t* rng, void* workspace, double* obs_acc, double* ret_acc, void* hip_stream) {
  if (!obs || !obs_mean || !obs_var || !obs_count || !reward || !terminated || !truncated || !returns || !ret_mean || !ret_var || !ret_count ||
      !workspace || N <= 0 || D <= 0 || D > 256) { g_err = "fw_collect_stats: bad arguments"; return FW_EINVAL; }
  DeviceGuard g(device_of(obs));
  StatsArgs A;
  A.obs = obs; A.obs_is_f64 = obs_is_f64; A.N = N; A.D = D; A.mean = obs_mean; A.var = obs_var; A.count = obs_count; A.update_obs = update_obs;
  A.reward = reward; A.rew_is_f64 = rew_is_f64; A.terminated = terminated; A.truncated = truncated; A.returns = returns;
  A.ret_mean = ret_mean; A.ret_var = ret_var; A.ret_count = ret_count; A.update_ret = update_ret; A.gamma = gamma; A.rng = rng;
  A.part = (double*)workspace; A.ticket = (unsigned int*)((char*)workspace + sizeof(double) * 64 * (2 * (size_t)D + 2));
  A.obs_acc = obs_acc; A.ret_acc = ret_acc;
  const int nb = collect_stats_blocks(N);
  if (obs_is_f64) hipLaunchKernelGGL(fw_collect_stats_kernel<double>, dim3(nb), dim3(256), 0, (hipStream_t)hip_stream, A);
  else hipLaunchKernelGGL(fw_collect_stats_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)hip_stream, A);
  HIP_TRY((fw_env*)nullptr, hipGetLastError());
  return FW_OK;
}

int32_t fw_policy_terminal_value(const float* params, const void* terminal_obs, int32_t obs_is_f64, int32_t N, int32_t obs_dim,
                                 const double* mean, const double* var, float clip, float eps, const uint8_t* terminated,
                                 const uint8_t* truncated, float* value, void* hip_stream) {
  if (!params || !terminal_obs || !mean || !var || !terminated || !truncated || !value || N <= 0 || obs_dim <= 0 || obs_dim > 64) {
    g_err = "fw_policy_terminal_value: bad arguments"; return FW_EINVAL;
  }
  const size_t lds = act_lds_bytes(obs_dim);
  const int dev = device_of(params);
  DeviceGuard g(dev);
  if (int rc = ensure_learner_lds(dev, 1, lds)) return rc;
  ActArgs A;
  std::memset(&A, 0, sizeof A);
  A.params = params; A.N = N; A.D = obs_dim; A.nets = 2; A.deterministic = 1; A.value = value;
  A.raw = terminal_obs; A.raw_is_f64 = obs_is_f64; A.mean = mean; A.var = var; A.clip = clip; A.eps = eps;
  A.terminated = terminated; A.truncated = truncated;
  hipLaunchKernelGGL(fw_policy_act_kernel, dim3((N + kPChunk - 1) / kPChunk, 2), dim3(kPThreads), lds, (hipStream_t)hip_stream, A);
  HIP_TRY((fw_env*)nullptr, hipGetLastError());
  return FW_OK;
}

int32_t fw_rollout_post(const void* reward, int32_t rew_is_f64, const uint8_t* terminated, const uint8_t* truncated, const float* tvalue,
                        double* returns, double* ret_mean, double* ret_var, double* ret_count, int32_t N, int32_t training,
                        int32_t norm_reward, double gamma, float clip_reward, float epsilon, float* rew_out, float* start_out,
                        uint64_t* rng, double* ret_acc, void* hip_stream) {
  if (!reward || !terminated || !truncated || !tvalue || !returns || !ret_mean || !ret_var || !ret_count || !rew_out || !start_out || N <= 0) {
    g_err = "fw_rollout_post: bad arguments"; return FW_EINVAL;
  }
  DeviceGuard g(device_of(returns));
  PostArgs A;
  A.reward = reward; A.rew_is_f64 = rew_is_f64; A.terminated = terminated; A.truncated = truncated; A.tvalue = tvalue; A.returns = returns;
  A.ret_mean = ret_mean; A.ret_var = ret_var; A.ret_count = ret_count; A.N = N; A.training = training; A.norm_reward = norm_reward;
  A.gamma = gamma; A.clip_reward = clip_reward; A.epsilon = epsilon; A.rew_out = rew_out; A.start_out = start_out; A.rng = rng; A.ret_acc = ret_acc;
  hipLaunchKernelGGL(fw_rollout_post_kernel, dim3(1), dim3(1024), 0, (hipStream_t)hip_stream, A);
  HIP_TRY((fw_env*)nullptr, hipGetLastError());
  return FW_OK;
}

int32_t fw_num_envs(fw_handle h) { return h ? h->n : FW_EINVAL; }
int32_t fw_capture_wave(fw_handle h) { return h ? h->capture_wave : FW_EINVAL; }
int32_t fw_lanes_per_env(fw_handle h) { return h ? (h->lanes_per_env == 8 && h->g8_waves == 2 ? 16 : h->lanes_per_env) : FW_EINVAL; }

const char* fw_last_error(fw_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

int32_t fw_destroy(fw_handle h) {
  if (!h) return FW_EINVAL;
  DeviceGuard g(h->device);
  (void)hipDeviceSynchronize();
  free_device_buffers(h);
  delete h;
  return FW_OK;
}

#ifdef FW_PROFILE
// Dev-only (not part of include/fwsim.h): per-wave cycle accounting ring of the last 256 launches.
// Layout long long[256][2 * blocks][8]; returns the number of step blocks.
int32_t fw_debug_profile(fw_handle h, long long* host_out, int32_t enable) {
  if (!h) return FW_EINVAL;
  DeviceGuard g(h->device);
  const size_t nblk = grid_of(h).x, bytes = sizeof(long long) * kProfSlots * 2 * nblk * kProfWords;
  (void)hipDeviceSynchronize();
  if (enable && !h->prof_dev) { (void)hipMalloc((void**)&h->prof_dev, bytes); (void)hipMemset(h->prof_dev, 0, bytes); }
  if (host_out && h->prof_dev) (void)hipMemcpy(host_out, h->prof_dev, bytes, hipMemcpyDeviceToHost);
  return (int32_t)nblk;
}
int32_t fw_debug_epoch(fw_handle h) { uint64_t c[FW_CTR_DIM]; return (h && fw_get_counters(h, c) == FW_OK) ? (int32_t)c[FW_CTR_LAUNCHES] + 1 : 0; }
#endif

}  // extern "C"
