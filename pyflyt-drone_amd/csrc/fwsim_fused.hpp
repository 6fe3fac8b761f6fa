// fw_collect_step: ONE launch per vec-step of the rollout collector (SB3 OnPolicyAlgorithm.collect_rollouts +
// VecNormalize.step_wait around Env.step, train/train_Fixedwing_Waypoints_v3.py:260,293-310).
//
// Round 2 ran a vec-step as three dependent launches, fw_collect_act -> fw_step -> fw_collect_stats: 48.0 us of which 20 are
// physics -- each of the two small kernels pays ~8 us of dependent-launch latency (barrier packet, the producer's L2
// write-back, cold L2s on the XCDs that did not write the data) for < 1 us of work.  Here the three are one grid:
//
//   blocks [0, 2 n_chunks)      "act waves": one wave per (16-row chunk, network).  The arithmetic of fw_collect_act (raw
//                               observation normalised on load, 64-64 tanh MLP, Philox / Box-Muller sampling, log-prob, the
//                               rollout-buffer rows; the value wave also finalises the PREVIOUS step: reward normalisation,
//                               truncation bootstrap, episode starts) on v_mfma_f32_16x16x4_f32 tiles whose weight operands
//                               come from global memory straight into registers.  The policy wave writes its 16 clipped
//                               actions through over the NaN the step waves left there (they announce themselves); its word
//                               flag_p[chunk] = launch index, raised early, says "old statistics read".  The value wave
//                               publishes flag_v[chunk] as soon as it has READ everything the env step is about to overwrite
//                               (observations, rewards, flags, terminal observations).
//   block 2 n_chunks            the "merge wave": writes the updated statistics back for the caller (see below).
//   blocks [n_act, n_act+nblk)  the env step waves of fw_step (step_body<..., COLLECT = true>): they load their state, then
//                               wait -- bounded -- for the value wave's word of their chunk and for their actions (coherent
//                               loads until no word is NaN), and run the step; in their tail they leave 2 D + 2 partial sums
//                               and put the NaN back.
//   blocks beyond               the shadow / scenario workers of fw_step, unchanged;
//   the last 2 D + 2 blocks     "fold waves", one per partial-sum word (collect_fold_wave).
// Workgroups are dispatched in block order and every wave waits only for waves in front of it (act waves for nobody, step
// waves for act waves, fold waves for step waves and the merge wave), so every word a wave waits for belongs to a wave that
// is already running or done: no deadlock whatever the residency.  Every wait is bounded (CollectArgs::spin); one that runs
// out -- it never should; in-order dispatch is true in practice and promised nowhere -- raises a bit of the sticky status word
// (CS_ST_*, include/fwsim.h FW_COLLECT_ST_*) and the wave goes on with what it finds, so that the grid always drains.  The
// product reads the word once per rollout (rollout.PPO.check_collect_status: the graph's last node copies it to pinned memory)
// and raises before the rollout is used; tests provoke the timeouts with FWSIM_SPIN_LOG2 and assert exactly that.
//
// The statistics of VecNormalize.step_wait (observation moments, discounted-return tracker) need a reduction over ALL envs
// between the env step and the next policy forward.  Versions of this round, in order (waypoints, 4096 envs, us per
// vec-step with the per-rollout launches; three launches: 48.0):
//   * fold in the step waves' tail (last wave of 8 groups, then the last group): +36 us -- every hop between waves of
//     different XCDs is a write-through store, its acknowledgement, a ticket and a coherent load, ~3 us each, six in a row;
//   * plain partial stores, folded by the act waves of the NEXT launch in front of their own work (self-announcing totals):
//     49.1 -- the fold sat on the critical path of every act wave (statistics known 9.4 us into the launch);
//   * fold waves at the END of the launch that produced the partials (now): a step wave writes its partials through, a
//     slot announces itself by no longer holding the sentinel the fold wave put back after the previous fold, fold wave w sums
//     word w over all step waves in a fixed association (a slot that has arrived stays in its register: the pass that sees the
//     last partial reads only what was still missing) and stores the total with a plain store -- the next launch's act waves
//     and merge wave read 2 D + 2 plain words and derive the statistics THIS step is normalised with, each for itself (same
//     arithmetic, same bits everywhere); the merge wave writes them to the caller's buffers after every act wave has announced
//     that it has read the old ones.  fw_collect_finish merges the last step of a rollout.  46.3;
//   * 16-row act waves on 16x16x4 tiles instead of 32-row ones on 32x32x2 (half the MFMA passes, tanh and normalisation on
//     the path to the published actions, twice the waves -- there are SIMDs to spare while the step waves wait): 40.3;
//   * weight operands from global memory into registers, column tiles = columns 4 r + t (one dwordx4 per weight row): 39.1;
//   * a lane keeps one column of the input tile (statistics in two registers, no index division, no LDS staging), the
//     statistics merge with one reciprocal per channel instead of four divisions: 38.4;
//   * hidden layers as tile pairs, half of the tanh epilogue between the second pair's MFMAs: 37.9;
//   * fw_collect_close instead of four launches at the end of a rollout (value waves: last values, finalisation, GAE): 37.6;
//   * the weight operands fetched at once only by the first two waves per XCD and network, by the others when they are about
//     to need them (lines the L2 holds by then): 37.1;
//   * no shared readers counter: the merge wave watches the act waves' publishing words.  One atomic per act wave on one word
//     cost every later wait on the wave's memory counter its turn at that word (2.5 us on the XCDs served last), and, moved
//     to the end of the waves, 1.5-4 us per hand-off to the step waves polling next to it: 36.05 (113.6 M env-steps/s);
//   * actions that announce themselves (a step wave leaves NaN in its env's action row at the end of a launch; a clipped action
//     is never NaN) instead of a store wait and a flag the step waves poll before loading them: 34.85;
//   * the step waves' partial sums before their observation rows: 34.7; the closing launch's GAE rows fetched eight at a time:
//     34.3 (119.3 M env-steps/s).
// Timeline of a launch (tools/trace_collect.py, profiles/r03_collect_step_trace.txt): statistics known 2.6 us after an act
// wave starts, inputs normalised and in LDS +2.2, weight operands +0.15, forward 3.3, sampling and the action stores 1.0:
// actions out at 9.7 (mean) / 10.8 us (last), in the step waves' registers at 10.5 / 12.0, step waves done at 28.6, partials
// 29.5, totals and launch end 31.9 us (rocprofv3: 33.0 us per launch).  A bare hand-off between two waves costs 0.36 us inside
// an XCD and 0.41 us across two (tools/microbench_xcd.hip); in the grid each dependent hop measures ~1 us.
#pragma once
#include "fwsim_collect.hpp"

namespace fwsim {

constexpr int kCRows = 16;            // rows (envs) per act wave
constexpr int kCGroups = 8;           // partial-sum rows are grouped by workgroup index mod 8 (the XCD a step wave runs on)
enum { CS_PART_EPOCH = 0, CS_FOLDED_EPOCH = 1, CS_READERS = 2, CS_STATUS = 3, CS_INIT = 4, CS_MERGED = 5 };
constexpr unsigned int kCollectInitMagic = 0xF01DC0DEu;  // left in sync[CS_INIT] by fw_collect_workspace_init
// a total that has not been published yet: a quiet NaN no sum can produce
__device__ __forceinline__ double collect_sentinel() { return __longlong_as_double(0x7FF8C0DEC0DE0001ll); }
__device__ __forceinline__ bool collect_is_sentinel(double v) { return __double_as_longlong(v) == 0x7FF8C0DEC0DE0001ll; }

struct CollectArgs {
  int32_t n_act;                      // workgroups in front of the step waves: 2 per chunk + the merge wave, padded to a multiple of 8
  int32_t n_chunks;                   // ceil(N / 32)
  int32_t nblk;                       // step workgroups
  int32_t n_workers;                  // step + shadow / scenario workgroups; the fold waves follow them
  ActArgs A;                          // as fw_collect_act
  StatsArgs S;                        // as fw_collect_stats (S.obs / S.part / S.ticket unused)
  unsigned int* flag_p;               // [n_chunks] policy wave: actions of launch `epoch` are in act_env
  unsigned int* flag_v;               // [n_chunks] value wave: inputs of launch `epoch` have been read
  double* part1;                      // [2 D + 2][8][ceil(nblk / 8)] partial sums of the step waves: word-major; a slot nobody has written since the last fold holds a sentinel
  double* tot;                        // [2 D + 2] totals of the pending step (written by the fold waves of the launch that produced it)
  unsigned int* sync;                 // [16] CS_*: launch index of the partials in part1 / of the last merge, readers counter (closing launch), status bits, init mark, launch index of the last merge wave
  long long* trace;                   // null, or [grid][8] wall-clock stamps (10 ns ticks) per workgroup: tools/trace_collect.py
  int32_t spin;                       // polls a bounded wait may take before it gives up (kCollectSpin; FWSIM_SPIN_LOG2 shrinks it so that tests can provoke a timeout)
};
constexpr int kCollectSpin = 1 << 21; // ~ seconds: far beyond any healthy launch
// CS_STATUS bits (0 = every wait of every launch since fw_collect_workspace_init was answered).  A non-zero word means
// a launch went on with what it had -- zero actions, partial sums, an early commit -- and everything collected since is
// void: the caller must drop the rollout and re-initialise the workspace (rollout.PPO raises).
enum { CS_ST_ACTIONS = 1, CS_ST_FOLD = 2, CS_ST_MERGE = 4, CS_ST_NOINIT = 8, CS_ST_EPOCH = 16, CS_ST_NANACT = 32 };

// fw_collect_close: the GAE scan the value waves run for their own rows once the last values are known
struct CloseArgs {
  const float *rewards, *values, *episode_starts;   // [T, N] rollout buffers (row T - 1 of `rewards` is finalised by this launch)
  float *adv, *ret;                                  // [T, N]
  int32_t T;
  float gamma, lam;
};

__device__ __forceinline__ unsigned int ld_flag(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long collect_now() { return (long long)__builtin_amdgcn_s_memrealtime(); }
template <typename T> __device__ __forceinline__ T ld_coherent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void st_coherent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

inline size_t collect_act_lds_bytes(int D) {
  const int Dk = D <= 32 ? 32 : 64, ldx = Dk + 1;
  return sizeof(float) * (2 * (size_t)kCRows * ldx + 2 * (size_t)kCRows * kPLdh + kCRows * 4 + 4);
}

using f32x4 = __attribute__((ext_vector_type(4))) float;

// The weights of one network as this lane's B operands, fetched from global memory straight into registers in the operand
// layout (the parameter image is L2-resident and every act wave reads all of it exactly once: staging it in LDS first cost a
// 33 KB LDS write, its barrier and a second read).  Column tile t of a layer is the columns 4 r + t, r = 0 .. 15 (any partition
// of the 64 columns into four tiles of 16 serves the MFMA as long as operands, biases and results agree on it): a lane's four
// tiles are four CONSECUTIVE floats of a weight row -- one dwordx4 load.  w1[s][t] = W1[4 s + q][4 r + t] (zero beyond the
// image's rows), w2[s][t] = W2[4 s + q][4 r + t], wo[s] = Wo[4 s + q][r] (zero for r >= KO), biases of the lane's columns.
struct ActWeights { float w1[16][4], w2[16][4], wo[16], b1[4], b2[4], bo; };
__device__ __forceinline__ void act_load_weights(ActWeights& Wt, const float* __restrict__ params, int oW1, int Dp, int Dk, int KO) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int ob1 = oW1 + Dp * kPH, oW2 = ob1 + kPH, ob2 = oW2 + kPH * kPH, oWo = ob2 + kPH, obo = oWo + kPH * KO;
  const float4* p1 = reinterpret_cast<const float4*>(params + oW1 + q * kPH + 4 * r);
  const float4* p2 = reinterpret_cast<const float4*>(params + oW2 + q * kPH + 4 * r);
  const float* po = params + oWo + q * KO + (r < KO ? r : 0);
#pragma unroll
  for (int s_ = 0; s_ < 16; ++s_) {
    const float4 a = (4 * s_ < Dk && 4 * s_ + q < Dp) ? p1[s_ * kPH] : make_float4(0.f, 0.f, 0.f, 0.f);      // (rows 4 s_ + q: kPH floats = kPH / 4 float4 apart, x 4 rows)
    const float4 b = p2[s_ * kPH];
    Wt.w1[s_][0] = a.x; Wt.w1[s_][1] = a.y; Wt.w1[s_][2] = a.z; Wt.w1[s_][3] = a.w;
    Wt.w2[s_][0] = b.x; Wt.w2[s_][1] = b.y; Wt.w2[s_][2] = b.z; Wt.w2[s_][3] = b.w;
    Wt.wo[s_] = r < KO ? po[s_ * 4 * KO] : 0.f;
  }
  {
    const float4 a = *reinterpret_cast<const float4*>(params + ob1 + 4 * r), b = *reinterpret_cast<const float4*>(params + ob2 + 4 * r);
    Wt.b1[0] = a.x; Wt.b1[1] = a.y; Wt.b1[2] = a.z; Wt.b1[3] = a.w;
    Wt.b2[0] = b.x; Wt.b2[1] = b.y; Wt.b2[2] = b.z; Wt.b2[3] = b.w;
  }
  Wt.bo = r < KO ? params[obo + (r < KO ? r : 0)] : 0.f;
}

// 16 rows through one network, one wave, on v_mfma_f32_16x16x4_f32 (lane l holds A[l % 16][l / 16], B[l / 16][l % 16] and the
// four results D[4 (l / 16) + v][l % 16]): X[16, Dk] -> tanh -> H1 -> tanh -> H2 -> head: out[16, 4] (KO columns used).  Dk =
// the input width padded to 32 or 64 (zero columns of X against zero operands).  Half the rows of a 32x32x2 tile per
// wave means half the MFMA passes, half the tanh and half the input normalisation on the path to the published actions, for
// twice the act waves -- they are the critical path of the launch, and there are SIMDs to spare while the step waves wait.
// The A operands of four k-steps are fetched from LDS together and the four column tiles' accumulators interleave.
// One hidden layer, Hout = tanh(A W + b): the four column tiles in two pairs -- the two accumulators of a pair alternate, so a
// tile's next MFMA never waits for its previous one -- with the tanh epilogue of the FIRST pair written between the MFMAs of
// the second: vector instructions issue in the shadow of a running MFMA, so half of the epilogue costs nothing.
template <int S>
__device__ __forceinline__ void act_layer(const float* a, const float (&w)[16][4], const float (&b)[4], float* Hout) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  float av[S];
#pragma unroll
  for (int i = 0; i < S; ++i) av[i] = a[4 * i];
  float* h = Hout + 4 * q * kPLdh + 4 * r;
  f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    f32x4 c0 = {b[2 * pr], b[2 * pr], b[2 * pr], b[2 * pr]}, c1 = {b[2 * pr + 1], b[2 * pr + 1], b[2 * pr + 1], b[2 * pr + 1]};
#pragma unroll
    for (int i = 0; i < S; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], w[i][2 * pr], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], w[i][2 * pr + 1], c1, 0, 0, 0);
      if (pr == 1 && (i + 1) % (S / 8) == 0) {                          // eight elements of the first pair over the S steps
        const int e = (i + 1) / (S / 8) - 1, v = e & 3;
        h[v * kPLdh + (e >> 2)] = ppo_tanh(e < 4 ? p0[v] : p1[v]);
      }
    }
    p0 = c0; p1 = c1;
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) { h[v * kPLdh + 2] = ppo_tanh(p0[v]); h[v * kPLdh + 3] = ppo_tanh(p1[v]); }
}
__device__ __forceinline__ void act_forward_wave(const ActWeights& Wt, const float* X, float* H1, float* H2, float* out, int KO, int Dk, int ldx) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  if (Dk == 32) act_layer<8>(X + r * ldx + q, Wt.w1, Wt.b1, H1); else act_layer<16>(X + r * ldx + q, Wt.w1, Wt.b1, H1);
  __syncthreads();
  act_layer<16>(H1 + r * kPLdh + q, Wt.w2, Wt.b2, H2);
  __syncthreads();
  {
    f32x4 c = {Wt.bo, Wt.bo, Wt.bo, Wt.bo};
    const float* a = H2 + r * kPLdh + q;
    float av[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) av[i] = a[4 * i];
#pragma unroll
    for (int i = 0; i < 16; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], Wt.wo[i], c, 0, 0, 0);
    if (r < KO) {
#pragma unroll
      for (int v = 0; v < 4; ++v) out[(4 * q + v) * 4 + r] = c[v];
    }
  }
  __syncthreads();
}

// Running statistics after the pending step: (old statistics) (+) (batch sums), SB3's RunningMeanStd.update_from_moments.
struct CollectMerged { double mean, var; };
// (one division: 1 / (cnt + n); the batch moments use inv_n = 1 / n computed once per wave -- within an ulp or two of double of the
// quotients update_from_moments writes, ~1e-16 of a statistic; eight divisions on the critical path of every act wave otherwise)
__device__ __forceinline__ CollectMerged collect_chan(double om, double ov, double cnt, double cs, double cs2, double n, double inv_n) {
  const double bm = cs * inv_n;
  double bv = cs2 * inv_n - bm * bm;                                 // population variance, as np.var
  bv = bv < 0 ? 0 : bv;
  const double delta = bm - om, tot = cnt + n, inv_tot = 1.0 / tot;
  const double m2 = ov * cnt + bv * n + delta * delta * cnt * n * inv_tot;
  CollectMerged r; r.mean = om + delta * n * inv_tot; r.var = m2 * inv_tot;
  return r;
}

// What every wave in front of the step waves does first: read the statistics the previous launches left and the totals of the
// pending step, and derive the statistics THIS step is normalised with.  Lane d < D ends with column d's (mean, var);
// `ret_var` is wave-uniform.
struct CollectStats { double mean, var, cnt, ret_mean, ret_var, ret_cnt; double cs, cs2, r1, r2; bool pend_obs, pend_ret; unsigned int part_ep; };
// (in two halves, so that a caller can put other loads between the statistics' loads and their first use)
__device__ __forceinline__ void collect_front_load(const CollectArgs& CA, CollectStats& Q) {
  const StatsArgs& S = CA.S;
  const int lane = threadIdx.x & 63, D = S.D;
  const unsigned int part_ep = CA.sync[CS_PART_EPOCH], folded_ep = CA.sync[CS_FOLDED_EPOCH];
  Q.mean = 0.0; Q.var = 1.0;
  if (lane < D) { Q.mean = S.mean[lane]; Q.var = S.var[lane]; }
  Q.cnt = S.count[0]; Q.ret_mean = S.ret_mean[0]; Q.ret_var = S.ret_var[0]; Q.ret_cnt = S.ret_count[0];
  // the totals of the pending step: folded by the fold waves at the END of the launch that produced them (collect_fold_wave),
  // i.e. plain memory by now
  const int dcol = lane < D ? lane : 0;
  Q.cs = CA.tot[lane == 63 ? 2 * D : dcol];
  Q.cs2 = CA.tot[lane == 63 ? 2 * D + 1 : D + dcol];
  const bool pend = part_ep != folded_ep;
  Q.part_ep = part_ep;
  Q.pend_obs = pend && S.update_obs; Q.pend_ret = pend && S.update_ret;
}
__device__ __forceinline__ void collect_front_merge(const CollectArgs& CA, bool count_me, CollectStats& Q) {
  const StatsArgs& S = CA.S;
  const int lane = threadIdx.x & 63, D = S.D;
  // the old statistics are in registers: tell the merge wave (it overwrites them only after every wave in front has said so)
  asm volatile("" :: "v"(Q.mean), "v"(Q.var), "v"(Q.cnt), "v"(Q.ret_mean), "v"(Q.ret_var), "v"(Q.ret_cnt), "v"(Q.cs), "v"(Q.cs2), "s"(Q.part_ep), "s"((int)Q.pend_obs), "s"((int)Q.pend_ret) : "memory");
  if (count_me && lane == 0) (void)__hip_atomic_fetch_add(CA.sync + CS_READERS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  Q.r1 = __shfl(Q.cs, 63, 64); Q.r2 = __shfl(Q.cs2, 63, 64);
  const double n = (double)S.N, inv_n = 1.0 / n;
  if (Q.pend_obs) {
    const CollectMerged m = collect_chan(Q.mean, Q.var, Q.cnt, Q.cs, Q.cs2, n, inv_n);
    if (lane < D) { Q.mean = m.mean; Q.var = m.var; }
    Q.cnt += n;
  }
  if (Q.pend_ret) {
    const CollectMerged m = collect_chan(Q.ret_mean, Q.ret_var, Q.ret_cnt, Q.r1, Q.r2, n, inv_n);
    Q.ret_mean = m.mean; Q.ret_var = m.var; Q.ret_cnt += n;
  }
}
__device__ __forceinline__ void collect_front_stats(const CollectArgs& CA, bool count_me, CollectStats& Q) {
  collect_front_load(CA, Q);
  collect_front_merge(CA, count_me, Q);
}

// A fold wave (one of the 2 D + 2 workgroups at the very END of the grid): waits until every step wave has left its partial of
// word w -- a slot is there as soon as it is not the sentinel this wave itself put back after the previous fold -- sums the slots in
// a fixed association (lane l adds its slots l, l + 64, ... in ascending order, then an xor-butterfly over the lanes), stores the total (plain: the next launch reads it) and restores the sentinels.
// It waits only for workgroups in front of it (dispatch is in block order): no residency can deadlock it.
__device__ __forceinline__ void collect_fold_wave(const CollectArgs& CA, int w) {
  const int lane = threadIdx.x & 63, PW = 2 * CA.S.D + 2;
  if (w >= PW) return;
  const int mstride = (CA.nblk + kCGroups - 1) / kCGroups, slots = kCGroups * mstride;
  double* row = CA.part1 + (size_t)w * slots;
  // slot i belongs to step workgroup (i / mstride) + 8 (i % mstride); groups with fewer members leave their last slot unused
  auto used = [&](int i) { const int g = i / mstride, m = i - g * mstride; return g + kCGroups * m < CA.nblk; };
  // a slot that has arrived stays in its register: the pass that sees the last partial reads only what was still missing.
  // kS slots per lane and pass = 1024 step workgroups (8192 envs) -- the whole row where the one-wave-per-SIMD mapping is the
  // fast one; rows beyond that (camera tasks up to 16 384 envs, any N with cylinders) take further passes of the same kind, and
  // lane l still adds its slots l, l + 64, ... in ascending order.
  constexpr int kS = 16;
  double v = 0.0;
  bool ok = true;
  int budget = CA.spin;
  for (int base = 0; base < slots; base += 64 * kS) {
    double x[kS];
    unsigned int missing = 0;
#pragma unroll
    for (int k = 0; k < kS; ++k) { x[k] = 0.0; const int i = base + lane + 64 * k; if (i < slots && used(i)) missing |= 1u << k; }
    bool here = false;
    for (; budget > 0; --budget) {
#pragma unroll
      for (int k = 0; k < kS; ++k) if ((missing >> k) & 1u) x[k] = ld_sc1(row + base + lane + 64 * k);
#pragma unroll
      for (int k = 0; k < kS; ++k) if (((missing >> k) & 1u) && !collect_is_sentinel(x[k])) missing &= ~(1u << k);
      if (__ballot(missing != 0u) == 0ull) { here = true; break; }
      __builtin_amdgcn_s_sleep(8);                   // (~0.2 us: 58 waves polling 512 slots each are traffic the step waves share the fabric with)
    }
    ok = ok && here;
#pragma unroll
    for (int k = 0; k < kS; ++k) v += ((missing >> k) & 1u) ? 0.0 : x[k];   // ascending slot order
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  // every act wave has read the totals of the step before (its step waves have finished), the merge wave maybe not yet: it
  // leaves the launch's index in CS_MERGED when it has (the index itself: any chunk's word, all published by now)
  if (ok) {
    ok = false;
    const unsigned int e = ld_flag(CA.flag_p);
    for (int it = 0; it < CA.spin; ++it) {
      if (ld_flag(CA.sync + CS_MERGED) == e) { ok = true; break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  if (lane == 0) CA.tot[w] = v;
  for (int i = lane; i < slots; i += 64) row[i] = collect_sentinel();
  if (!ok && lane == 0) atomicOr(CA.sync + CS_STATUS, (unsigned int)CS_ST_FOLD);
}

// writes the statistics of `Q` (already merged) to the caller's buffers and closes the pending fold
__device__ __forceinline__ void collect_commit_stats(const CollectArgs& CA, const CollectStats& Q) {
  const StatsArgs& S = CA.S;
  const int lane = threadIdx.x & 63, D = S.D;
  const double n = (double)S.N;
  if (Q.pend_obs) {
    if (lane < D) {
      S.mean[lane] = Q.mean; S.var[lane] = Q.var;
      if (S.obs_acc) { S.obs_acc[lane] += Q.cs; S.obs_acc[D + lane] += Q.cs2; }
    }
    if (lane == 0) { S.count[0] = Q.cnt; if (S.obs_acc) S.obs_acc[2 * D] += n; }
  }
  if (Q.pend_ret && lane == 0) {
    S.ret_mean[0] = Q.ret_mean; S.ret_var[0] = Q.ret_var; S.ret_count[0] = Q.ret_cnt;
    if (S.ret_acc) { S.ret_acc[0] += Q.r1; S.ret_acc[1] += Q.r2; S.ret_acc[2] += n; }
  }
  if (lane == 0) CA.sync[CS_FOLDED_EPOCH] = Q.part_ep;
}

// The merge wave (block 2 n_chunks): same statistics as everybody, written back once every act wave has read the old ones; the
// action sampler's draw counter advances here too (the policy waves read it at their start).  "Has read": in a step launch
// every act wave publishes a word after it did (flag_p / flag_v = the launch's index) -- the merge wave watches those, no
// counter (520 atomics on one word, fired while the step waves poll their flags next to it, cost the hand-off 1.5-4 us);
// fw_collect_close has no such words and few waves: there they count themselves.
__device__ __forceinline__ void collect_merge_wave(const CollectArgs& CA, int n_real, bool advance_rng, bool by_flags) {
  const int lane = threadIdx.x & 63;
  CollectStats Q;
  collect_front_stats(CA, false, Q);
  bool ok = false;
  unsigned int epoch = 0;
  int budget = CA.spin;
  if (by_flags) {
    // The launch's index.  Every other wave takes it from the launch counter of a step workgroup that cannot finish before that
    // wave has published; no step workgroup waits for THIS wave, so a counter could have advanced by the time a late first
    // load of this wave returns.  It therefore takes the index from the first policy wave's publishing word: between step
    // launches that word equals what this wave left in CS_MERGED (both zero after fw_collect_workspace_init), launch indices
    // only grow, so the first value that differs is this launch's.
    const unsigned int last = ld_flag(CA.sync + CS_MERGED);
    epoch = last;
    for (; budget > 0; --budget) {
      epoch = ld_flag(CA.flag_p);
      if (epoch != last) break;
      __builtin_amdgcn_s_sleep(8);
    }
    if (epoch == last && lane == 0) atomicOr(CA.sync + CS_STATUS, (unsigned int)CS_ST_EPOCH);
  }
  for (; budget > 0; --budget) {
    if (by_flags) {
      bool all = true;
      for (int i = lane; i < 2 * CA.n_chunks; i += 64) all = all && ld_flag(i < CA.n_chunks ? CA.flag_p + i : CA.flag_v + (i - CA.n_chunks)) == epoch;
      if (__ballot(!all) == 0ull) { ok = true; break; }
      __builtin_amdgcn_s_sleep(32);
    } else {
      if (ld_flag(CA.sync + CS_READERS) >= (unsigned int)n_real) { ok = true; break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  if (!ok && lane == 0) atomicOr(CA.sync + CS_STATUS, (unsigned int)CS_ST_MERGE);
  if (lane == 0 && CA.sync[CS_INIT] != kCollectInitMagic) atomicOr(CA.sync + CS_STATUS, (unsigned int)CS_ST_NOINIT);   // workspace never initialised
  collect_commit_stats(CA, Q);
  if (lane == 0) {
    if (advance_rng && CA.S.rng) CA.S.rng[1] += 1;
    st_flag(CA.sync + CS_READERS, 0u);
    if (by_flags) st_flag(CA.sync + CS_MERGED, epoch);
  }
}

// One act wave (workgroup of 64 lanes).  Inlined into the collect kernels: CollectArgs is a by-value kernel argument, and
// handing its address to an out-of-line function made the compiler copy the whole struct to scratch in every wave (544 B).
// (tools/check_isa.py therefore tells MFMA accumulator registers from spill slots by the operand ranges of the MFMAs.)
// T = the env's real type: the type of its observation / reward / action buffers.
// CLOSE = the waves of fw_collect_close (one per chunk, value network only): the last observation of a rollout -> last
// values, the finalisation of its last step, and the GAE scan of the wave's own rows -- nobody to hand anything to.
template <typename T, bool CLOSE = false>
__device__ __forceinline__ void collect_act_wave(const CollectArgs& CA, uint32_t epoch, const CloseArgs* GA = nullptr) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const ActArgs& A = CA.A;
  const int aw = (int)blockIdx.x, chunk = CLOSE ? aw : aw >> 1, net = CLOSE ? 1 : aw & 1;
  if (aw == (CLOSE ? 1 : 2) * CA.n_chunks) { collect_merge_wave(CA, (CLOSE ? 1 : 2) * CA.n_chunks, !CLOSE, !CLOSE); return; }
  if (chunk >= CA.n_chunks) return;                                  // padding waves
  const int KO = net == 0 ? 4 : 1;
  const int lane = threadIdx.x;
  const int D = A.D, Dp = (D + 1) & ~1, Dk = D <= 32 ? 32 : 64, ldx = Dk + 1;  // Dp: rows of W1 in the parameter image, Dk: columns of the input tile
  const int row0 = chunk * kCRows;

  long long* tr = CA.trace ? CA.trace + (size_t)blockIdx.x * 8 : nullptr;
  if (tr && lane == 0) tr[0] = collect_now();
  float* p = lds;
  float* X = p;  p += kCRows * ldx;
  float* X2 = p; p += kCRows * ldx;                                  // value wave: terminal observations of the previous step
  float* H1 = p; p += kCRows * kPLdh;
  float* H2 = p; p += kCRows * kPLdh;
  float* out = p; p += kCRows * 4;

  // ---- everything whose address is known at entry leaves first; the statistics lead: the memory counter retires in order,
  // and the inputs are normalised (the statistics' first use) while the 33 KB of weights are still arriving ----
  CollectStats Q;
  collect_front_load(CA, Q);
  const int frow = row0 + (lane & (kCRows - 1));
  const bool fmine = net == 1 && A.prev_reward && lane < kCRows && frow < A.N;
  uint8_t f_term = 0, f_trunc = 0; double f_rew = 0.0;
  if (fmine) {
    f_term = A.prev_term[frow]; f_trunc = A.prev_trunc[frow];
    f_rew = (double)reinterpret_cast<const T*>(A.prev_reward)[frow];
  }
  uint64_t rng_key = 0, rng_ctr = 0;
  if (net == 0 && !A.deterministic && lane < kCRows) { rng_key = A.rng[0]; rng_ctr = A.rng[1]; }
  // The [kCRows][ldx] input tile: a lane keeps ONE column (lane % CW, CW = 32 or 64 columns per pass) and takes every
  // (64 / CW)-th row of it -- its column's mean and reciprocal deviation stay in two registers, a wave-level load is CW
  // consecutive elements of a row, and no index needs a division.
  const int CW = D <= 32 ? 32 : 64, rpp = kWave / CW;                // columns per pass, rows per pass
  const int xcol = lane & (CW - 1), xrow0 = lane / CW;
  constexpr int kXB = 16;                                            // rows a lane takes: kCRows / rpp <= 16 (8 for <= 32 columns)
  const int nxb = kCRows / rpp;
  auto raw_at = [&](const void* base, int row, int d) {
    // (uniform base + 32-bit byte offset: one address register per load instead of 64-bit arithmetic per lane)
    return (double)*reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (unsigned int)((row * D + d) * (int)sizeof(T)));
  };
  double rawv[kXB];
  auto load_batch = [&](const void* base) {
#pragma unroll
    for (int u = 0; u < kXB; ++u) {
      const int row = row0 + xrow0 + u * rpp;
      rawv[u] = (u < nxb && xcol < D && row < A.N) ? raw_at(base, row, xcol) : 0.0;
    }
  };
  load_batch(A.raw);
  // ---- weights of my network: this lane's MFMA operands, all loads in flight together ----
  const int nP0 = ppo_net_params(Dp, 4);
  const int oW1 = net == 0 ? 0 : nP0;
  const int oLs = nP0 + ppo_net_params(Dp, 1);
  const float* __restrict__ params = A.params;
  // Every act wave of an XCD reads the same 33 KB, and the XCD's L2 is cold at the start of a launch: requested by all 65 waves
  // at once the image arrives up to 4 us late on some XCDs (tools/trace_collect.py: the forward pass itself takes 3.4 us
  // everywhere) -- misses to a line that is already on its way are not merged for free.  So the first waves of each XCD fetch
  // it now and the others ask when they are about to need it, for lines the L2 holds by then.
  ActWeights Wt;
  const bool primer = aw < 2 * kCGroups;                             // (workgroups are dealt round-robin to the 8 XCDs: two waves per XCD and network)
  if (primer) act_load_weights(Wt, params, oW1, Dp, Dk, KO);
  float log_std[4];                                                  // (uniform addresses: scalar loads)
#pragma unroll
  for (int k = 0; k < 4; ++k) log_std[k] = params[oLs + k];
  // ---- statistics of THIS step: the old ones (+) the totals of the pending step ----
  collect_front_merge(CA, false, Q);
  // the policy wave's word says "I have read the old statistics" (the merge wave watches it; the fold waves take the launch's
  // index from it) -- the actions themselves need no flag: a step wave sees them replace the NaN it left in their place
  // (the sampler's key and draw counter are among the "old" words: the merge wave advances the counter once every word is up,
  // so they must be in registers before this one is raised)
  asm volatile("" :: "v"(rng_key), "v"(rng_ctr) : "memory");
  if (!CLOSE && net == 0 && lane == 0) st_flag(CA.flag_p + chunk, epoch);
  // ("I have read the old statistics": in a step launch my publishing word says so, the merge wave watches those; in the closing
  // launch a counter at the END of the wave -- in front, every later wait on the memory counter also waited for that atomic's
  // turn at a word 520 waves share: up to 4 us on the XCDs that came last)
  auto announce_read = [&]() { if (CLOSE && lane == 0) (void)__hip_atomic_fetch_add(CA.sync + CS_READERS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  const bool timeout = fmine && f_trunc && !f_term;
  const bool any_timeout = __ballot(timeout) != 0ull;                // wave-uniform: some episode of my rows was truncated
  if (tr && lane == 0) tr[1] = collect_now();                          // statistics of this step known
  // ---- inputs: normalise on load with the statistics of THIS step ----
  // (x - mean) * (1 / sqrt(var + eps)): one division per column and wave instead of one per element -- within an ulp of
  // double of VecNormalize's (x - mean) / sqrt(var + eps), i.e. the same float32 except once in ~1e9 elements
  const int scol = xcol < D ? xcol : 0;                              // (lane d < D holds column d's statistics)
  const double cmean = __shfl(Q.mean, scol, 64), crstd = 1.0 / sqrt(__shfl(Q.var, scol, 64) + (double)A.eps);
  auto build = [&](const void* base, float* dstX, bool copy) {
    if (base != A.raw) load_batch(base);                             // (the observations are already in flight)
#pragma unroll
    for (int u = 0; u < kXB; ++u) {
      if (u >= nxb) break;
      const int s_ = xrow0 + u * rpp, row = row0 + s_;
      float x = 0.f;
      if (xcol < D && row < A.N) {
        x = fminf(fmaxf((float)((rawv[u] - cmean) * crstd), -A.clip), A.clip);
        if (copy && A.obs_copy) *reinterpret_cast<float*>(reinterpret_cast<char*>(A.obs_copy) + (unsigned int)((row * D + xcol) * 4)) = x;
      }
      if (xcol < ldx) dstX[s_ * ldx + xcol] = x;
    }
    // columns CW .. ldx - 1 of the tile (padding when the pass is narrower than the padded width): zero
    for (int e = lane; e < kCRows * (ldx - CW); e += kWave) { const int s_ = e / (ldx - CW), d = CW + e - s_ * (ldx - CW); dstX[s_ * ldx + d] = 0.f; }
  };
  if (tr && lane == 0) tr[6] = collect_now();                          // statistics merged, this lane's column constants known
  build(A.raw, X, net == 0 || CLOSE);
  if (!primer) act_load_weights(Wt, params, oW1, Dp, Dk, KO);
  if (tr && lane == 0) tr[5] = collect_now();                          // observations normalised
  if (net == 1) {
    if (any_timeout) build(A.prev_tobs, X2, false);
    // everything the env step of THIS launch overwrites has been read: let the step waves of my chunk go
    if (!CLOSE) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) st_flag(CA.flag_v + chunk, epoch);
    }
  }
  __syncthreads();
  if (tr && lane == 0) tr[2] = collect_now();                          // inputs in LDS (value wave: flag_v published)

  if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) tr[7] = collect_now(); }      // (tracing: the weight operands have arrived)
  act_forward_wave(Wt, X, H1, H2, out, KO, Dk, ldx);
  if (tr && lane == 0) tr[3] = collect_now();                          // forward done
  float v_mine = 0.f;                                                // value wave: V(row0 + lane)
  if (lane < kCRows) {
    const int row = row0 + lane;
    if (row < A.N) {
      if (net == 1) {
        v_mine = out[lane * 4];
        A.value[row] = v_mine;
      } else {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (!A.deterministic) act_normal4(rng_key, rng_ctr, (uint64_t)(A.env_offset + row), z);
        float lp = 0.f, a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float ls = log_std[k];
          a[k] = out[lane * 4 + k] + z[k] * expf(ls);
          lp += -0.5f * z[k] * z[k] - ls - 0.9189385332046727f;
        }
        reinterpret_cast<float4*>(A.act_raw)[row] = make_float4(a[0], a[1], a[2], a[3]);
        A.logp[row] = lp;
        // fminf(fmaxf(NaN, -1), 1) = -1: the clip below would turn a diverged policy (NaN mean or log_std) into full deflection
        // and the step waves' "a clipped action is never NaN" would still hold -- so the NaN is reported here instead
        if (a[0] != a[0] || a[1] != a[1] || a[2] != a[2] || a[3] != a[3]) atomicOr(CA.sync + CS_STATUS, (unsigned int)CS_ST_NANACT);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = fminf(fmaxf(a[k], -1.0f), 1.0f);
        // the env's action row: write-through, the step waves of other XCDs read it in this same launch
        T* o = reinterpret_cast<T*>(A.act_env) + (size_t)row * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) st_coherent(o + k, (T)a[k]);
      }
    }
  }
  if (net == 0) {
    if (tr && lane == 0) tr[4] = collect_now();
    announce_read();
    return;
  }
  // ---- value wave: finalisation of the previous vec-step ----
  if (A.prev_reward) {
    if (any_timeout) {
      __syncthreads();
      act_forward_wave(Wt, X2, H1, H2, out, 1, Dk, ldx);
    }
    if (fmine) {
      double rn = f_rew;
      if (A.norm_reward) {
        rn *= 1.0 / sqrt(Q.ret_var + (double)A.rew_eps);
        rn = rn > A.clip_reward ? A.clip_reward : (rn < -A.clip_reward ? -A.clip_reward : rn);
      }
      float o = (float)rn;
      if (timeout) o += A.gamma * out[lane * 4];                        // SB3: bootstrap truncated episodes with V(terminal_observation)
      A.rew_out[frow] = o;
      A.start_out[frow] = (f_term || f_trunc) ? 1.0f : 0.0f;
      if (CLOSE) {
        // SB3 RolloutBuffer.compute_returns_and_advantage for env frow (fw_gae's arithmetic): the time loop runs backwards in
        // registers; the last row of the rewards is the one just finalised
        const int N = A.N, Tn = GA->T;
        float next_value = v_mine, next_non_terminal = (f_term || f_trunc) ? 0.0f : 1.0f, last_gae = 0.0f;
        constexpr int kGB = 8;                                            // rows fetched together: the recurrence is a few FMAs per row,
        for (int t1 = Tn - 1; t1 >= 0; t1 -= kGB) {                       // a row's three loads one memory round trip
          float vv[kGB], rr[kGB], ss[kGB];
#pragma unroll
          for (int u = 0; u < kGB; ++u) {
            const int t_ = t1 - u;
            const size_t i = (size_t)(t_ >= 0 ? t_ : 0) * N + frow;
            vv[u] = GA->values[i]; rr[u] = GA->rewards[i]; ss[u] = GA->episode_starts[i];
          }
#pragma unroll
          for (int u = 0; u < kGB; ++u) {
            const int t_ = t1 - u;
            if (t_ < 0) break;
            const size_t i = (size_t)t_ * N + frow;
            const float v = vv[u];
            const float rw = t_ == Tn - 1 ? o : rr[u];
            const float delta = rw + GA->gamma * next_value * next_non_terminal - v;
            last_gae = delta + GA->gamma * GA->lam * next_non_terminal * last_gae;
            GA->adv[i] = last_gae;
            GA->ret[i] = last_gae + v;
            next_value = v;
            next_non_terminal = 1.0f - ss[u];
          }
        }
      }
    }
  }
  if (tr && lane == 0) tr[4] = collect_now();
  announce_read();
}

// Step wave: wait (bounded) until the value waves of the chunks covering rows [env0, env0 + rows) have read what this step
// overwrites (their word = `epoch`) ...
__device__ __forceinline__ void collect_wait_actions(const CollectArgs& CA, uint32_t epoch, int env0, int rows) {
  long long* tr = CA.trace ? CA.trace + (size_t)blockIdx.x * 8 : nullptr;
  if (tr && (threadIdx.x & 63) == 0) tr[1] = collect_now();              // state loaded, wait begins
  const int c0 = env0 / kCRows, c1 = min((env0 + rows - 1) / kCRows, CA.n_chunks - 1);
  const int lane = threadIdx.x & 63;
  const unsigned int* w = CA.flag_v + min(c0 + lane, c1);
  bool ok = false;
  for (int it = 0; it < CA.spin; ++it) {
    if (__ballot(ld_flag(w) != epoch) == 0ull) { ok = true; break; }
    __builtin_amdgcn_s_sleep(4);
  }
  if (!ok && lane == 0) atomicOr(CA.sync + CS_STATUS, (unsigned int)CS_ST_ACTIONS);
}
// ... and until this lane's env has its actions: the policy wave writes them through over the NaN the env's step wave of the
// previous launch left there (a clipped action is never NaN), so the four words announce themselves -- no store wait and no
// flag hop between the policy wave's last store and the step wave's first use.
template <typename T>
__device__ __forceinline__ void collect_load_actions(const CollectArgs& CA, const T* ap, T (&a4)[4]) {
  long long* tr = CA.trace ? CA.trace + (size_t)blockIdx.x * 8 : nullptr;
  const int lane = threadIdx.x & 63;
  bool ok = false;
  for (int it = 0; it < CA.spin; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a4[k] = ld_coherent(ap + k);
    const bool mine = a4[0] == a4[0] && a4[1] == a4[1] && a4[2] == a4[2] && a4[3] == a4[3];
    if (__ballot(!mine) == 0ull) { ok = true; break; }
    __builtin_amdgcn_s_sleep(2);
  }
  if (!ok) {                                                           // (never expected: go on with zeros rather than NaN; the status word voids the rollout)
    if (lane == 0) atomicOr(CA.sync + CS_STATUS, (unsigned int)CS_ST_ACTIONS);
#pragma unroll
    for (int k = 0; k < 4; ++k) a4[k] = a4[k] == a4[k] ? a4[k] : (T)0;
  }
  if (tr && lane == 0) tr[2] = collect_now();                          // actions there
}
template <typename T>
__device__ __forceinline__ void collect_clear_actions(T* ap) {
  const T nan = (T)__builtin_nanf("");
#pragma unroll
  for (int k = 0; k < 4; ++k) ap[k] = nan;
}

// Statistics tail of a step wave: its partial sums, left with plain stores for the next launch (or fw_collect_finish) to fold.
// `tile` = the wave's observation rows in LDS ([rows][ld], final: reset rows included), rew / done = this lane's env (leader
// lanes only), wg = workgroup index among the nblk step workgroups.
template <typename T>
__device__ __forceinline__ void collect_stats_tail(const CollectArgs& CA, uint32_t epoch, const T* tile, int ld, int rows, int wg, int nblk,
                                                   bool is_leader, double rew, bool done, int env, double ret_prev) {
  const StatsArgs& S = CA.S;
  const int lane = threadIdx.x & 63, D = S.D;
  long long* tr = CA.trace ? CA.trace + (size_t)blockIdx.x * 8 : nullptr;
  if (tr && lane == 0) tr[3] = collect_now();                          // step done, tail begins
  double s = 0.0, s2 = 0.0;
  if (lane < D) {
    for (int r = 0; r < rows; ++r) { const double x = (double)tile[r * ld + lane]; s += x; s2 += x * x; }
  }
  double rt = 0.0;
  if (is_leader) {
    if (S.update_ret) { rt = ret_prev * S.gamma + rew; S.returns[env] = done ? 0.0 : rt; }
    else if (done) S.returns[env] = 0.0;
  }
  // leaders' tracker values summed in row order (a fixed order)
  double r1 = 0.0, r2 = 0.0;
  {
    unsigned long long lead = __ballot(is_leader);
    while (lead) {                                   // (ascending lanes)
      const int l = __builtin_ctzll(lead);
      lead &= lead - 1;
      const double v = __shfl(rt, l, 64);
      r1 += v; r2 += v * v;
    }
  }
  const int grp = wg & (kCGroups - 1), mem = wg >> 3, mstride = (nblk + kCGroups - 1) / kCGroups;
  auto p1 = [&](int w) { return CA.part1 + ((size_t)w * kCGroups + grp) * mstride + mem; };
  // write-through: the fold waves at the end of THIS launch read them (and a slot announces itself: it replaces a sentinel)
  if (lane < D) { st_sc1(p1(lane), s); st_sc1(p1(D + lane), s2); }
  if (lane == 0) { st_sc1(p1(2 * D), r1); st_sc1(p1(2 * D + 1), r2); if (wg == 0) CA.sync[CS_PART_EPOCH] = epoch; }
  if (tr && lane == 0) tr[4] = collect_now();
}

// fw_collect_finish: merges the totals the last fw_collect_step left into the caller's statistics (once per rollout; the
// fold itself was done by that launch's fold waves).  One wave.
__global__ __launch_bounds__(64) void fw_collect_finish_kernel(CollectArgs CA) {
  const bool pend = CA.sync[CS_PART_EPOCH] != CA.sync[CS_FOLDED_EPOCH];
  if (!pend) return;
  CollectStats Q;
  collect_front_stats(CA, false, Q);
  collect_commit_stats(CA, Q);
}

// fw_collect_close: grid = n_chunks value waves + the merge wave, 64 lanes, dynamic LDS = collect_act_lds_bytes(D).
template <typename T>
__global__ __launch_bounds__(64) void fw_collect_close_kernel(CollectArgs CA, CloseArgs GA) {
  collect_act_wave<T, true>(CA, 0u, &GA);
}

// fw_collect_workspace_init (after a memset to zero): every partial-sum slot "not there yet", and the mark that says so
__global__ void fw_collect_init_kernel(double* part1, size_t n, unsigned int* sync) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) part1[i] = collect_sentinel();
  if (i == 0) sync[CS_INIT] = kCollectInitMagic;
}

}  // namespace fwsim
