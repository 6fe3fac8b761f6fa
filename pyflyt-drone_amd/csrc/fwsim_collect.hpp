// Rollout-collection kernels: what SB3's `OnPolicyAlgorithm.collect_rollouts` does between two env steps,
// for the reference's MlpPolicy (train/train_Fixedwing_Waypoints_v3.py:293-310) and VecNormalize reward path
// (:260), in two launches instead of ~60 framework ops:
//   fw_policy_act      policy + value forward of all envs, Gaussian sampling, log-prob, action clipping for the
//                      env, and the rollout-buffer writes of the step (obs, action, value, log-prob);
//   fw_rollout_post    VecNormalize reward path (discounted-return tracker, running variance, clip), the
//                      truncation bootstrap r += gamma * V(terminal_obs), episode-start flags.
// Weights are the flat float32 image described in fwsim_ppo.hpp (the same buffer fw_ppo_update trains).
#pragma once
#include "fwsim_ppo.hpp"

namespace fwsim {

struct ActArgs {
  const float* params;        // flat parameter image
  const float* obs;           // [N, D] normalised observations (float32)
  int32_t N, D;
  int32_t nets;               // bit 0: policy block, bit 1: value block
  int32_t deterministic;
  int32_t act_is_f64;         // dtype of act_env
  const uint64_t* rng;        // [2] device: seed, draw counter (advanced by fw_rollout_post)
  int64_t env_offset;         // global env id of row 0 (multi-process sharding keeps streams distinct)
  float* obs_copy;            // [N, D] or null: rollout buffer slot of this step
  float* act_raw;             // [N, 4] sampled action (unclipped: what PPO stores)
  void* act_env;              // [N, 4] clipped to [-1, 1] in the env's dtype (fw_step input)
  float* logp;                // [N]
  float* value;               // [N]
  // value-of-terminal-observation mode (obs == nullptr): rows are normalised on load from the env's raw buffer and
  // a block whose 64 rows hold no truncated-but-not-terminated env returns at once (SB3 only bootstraps those)
  const void* raw; int32_t raw_is_f64;
  const double *mean, *var;
  float clip, eps;
  const uint8_t *terminated, *truncated;
  // fw_collect_act: the value block also finalises the PREVIOUS vec-step for its 64 rows (all null / 0 when there is none):
  // VecNormalize's reward path with the statistics fw_collect_stats left (reward / sqrt(var + eps), clipped), SB3's bootstrap
  // r += gamma V(normalised terminal observation) of the episodes that were truncated but not terminated (a second pass through
  // the value network, only in blocks that hold such a row), and the episode-start flags of the step that begins now
  const void* prev_reward;    // [N] raw reward of the previous fw_step (dtype raw_is_f64)
  const uint8_t *prev_term, *prev_trunc;
  const void* prev_tobs;      // [N, D] raw terminal observations of the previous fw_step
  const double* ret_var;      // [1] running variance of the discounted return
  int32_t norm_reward;
  float clip_reward, rew_eps, gamma;
  float* rew_out;             // [N] rollout-buffer rewards of the previous step
  float* start_out;           // [N] episode starts of the step that begins now
};

// counter-based N(0,1) x 4 for (env, draw): Philox4x32-10 + Box-Muller (float32)
__device__ __forceinline__ void act_normal4(uint64_t seed, uint64_t draw, uint64_t env, float z[4]) {
  uint32_t o[4];
  philox4x32_10((uint32_t)env, (uint32_t)(env >> 32), (uint32_t)draw, (uint32_t)(draw >> 32) ^ 0xAC7C0DEu, (uint32_t)seed, (uint32_t)(seed >> 32), o);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)(o[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);          // (0, 1)
    const float u2 = ((float)(o[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
  }
}

// X[64, Dp] -> tanh -> H1 -> tanh -> H2 -> head: out[64, 4] (KO columns used); all operands in LDS, 4 waves = 2 x 2 tiles of 32 x 32
__device__ __forceinline__ void act_forward(const PpoNetLds& W, const float* X, float* H1, float* H2, float* out, int KO, int Dp, int ldx) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, hh = lane >> 5;
  const int mt = wave >> 1, nt = wave & 1;
  {
    f32x16 c;
    const float bias = W.b1[nt * 32 + r];
#pragma unroll
    for (int v = 0; v < 16; ++v) c[v] = bias;
    c = ppo_mfma_tile(X + mt * 32 * ldx, ldx, 1, W.W1 + nt * 32, kPH, 1, Dp, c);
#pragma unroll
    for (int v = 0; v < 16; ++v) H1[(mt * 32 + ppo_acc_row(v)) * kPLdh + nt * 32 + r] = ppo_tanh(c[v]);
  }
  __syncthreads();
  {
    f32x16 c;
    const float bias = W.b2[nt * 32 + r];
#pragma unroll
    for (int v = 0; v < 16; ++v) c[v] = bias;
    c = ppo_mfma_tile(H1 + mt * 32 * kPLdh, kPLdh, 1, W.W2 + nt * 32, kPLdh, 1, kPH, c);
#pragma unroll
    for (int v = 0; v < 16; ++v) H2[(mt * 32 + ppo_acc_row(v)) * kPLdh + nt * 32 + r] = ppo_tanh(c[v]);
  }
  __syncthreads();
  if (wave < 2) {
    f32x16 c;
    const float bias = r < KO ? W.Wo[kPH * KO + (r < KO ? r : 0)] : 0.f;       // bo follows Wo
#pragma unroll
    for (int v = 0; v < 16; ++v) c[v] = bias;
    const float* a = H2 + (wave * 32 + r) * kPLdh + hh;
    const float* wo = W.Wo + hh * KO + (r < KO ? r : 0);
    c = ppo_mfma_k([&](int k0) { return a[k0]; }, [&](int k0) { return r < KO ? wo[k0 * KO] : 0.f; }, kPH, c);
    if (r < KO) {
#pragma unroll
      for (int v = 0; v < 16; ++v) out[(wave * 32 + ppo_acc_row(v)) * 4 + r] = c[v];
    }
  }
  __syncthreads();
}

// grid = (ceil(N / 64), 2): block (c, net) runs network `net` on rows 64 c .. 64 c + 63; 256 threads.
__global__ __launch_bounds__(kPThreads) void fw_policy_act_kernel(ActArgs A) {
  extern __shared__ __align__(16) float lds[];
  const int net = blockIdx.y;
  if (!((A.nets >> net) & 1)) return;
  const int KO = net == 0 ? 4 : 1;
  const int t = threadIdx.x;
  const int D = A.D, Dp = (D + 1) & ~1, ldx = Dp + 1;
  const int row0 = blockIdx.x * kPChunk;
  if (!A.obs && A.truncated) {
    const int row = row0 + (t & 63);
    const int need = (t < kPChunk && row < A.N && A.truncated[row] && !A.terminated[row]) ? 1 : 0;
    if (!__syncthreads_or(need)) return;
  }

  float* p = lds;
  PpoNetLds W;
  W.W1 = p; p += Dp * kPH; W.b1 = p; p += kPH; W.W2 = p; p += kPH * kPLdh; W.b2 = p; p += kPH; W.Wo = p; p += kPH * 4; W.bo = p; p += 4;
  float* log_std = p; p += 4;
  float* X = p;  p += kPChunk * ldx;
  float* H1 = p; p += kPChunk * kPLdh;
  float* H2 = p; p += kPChunk * kPLdh;
  float* out = p; p += kPChunk * 4;

  // small loads whose results are only needed later leave first, so that their round trips hide behind the weight loads:
  // the column statistics (raw-observation modes) and, for the value block, what it needs to finalise the previous step
  double c_var = 1.0, c_mean = 0.0;
  if (!A.obs && t < D) { c_var = A.var[t]; c_mean = A.mean[t]; }
  const int frow = row0 + (t & 63);
  const bool fmine = net == 1 && A.prev_reward && t < kPChunk && frow < A.N;
  uint8_t f_term = 0, f_trunc = 0; double f_rew = 0.0, f_var = 1.0;
  if (fmine) {
    f_term = A.prev_term[frow]; f_trunc = A.prev_trunc[frow]; f_var = A.ret_var[0];
    f_rew = A.raw_is_f64 ? reinterpret_cast<const double*>(A.prev_reward)[frow] : (double)reinterpret_cast<const float*>(A.prev_reward)[frow];
  }
  // ... and the first batch of observation elements (all of them for D <= 31): their round trip overlaps the weights' too
  constexpr int kXB = 8;
  double rawv[kXB]; float fv[kXB];
  auto load_batch = [&](int e0) {
#pragma unroll
    for (int u = 0; u < kXB; ++u) {
      const int e = e0 + u * kPThreads;
      const int s_ = e / ldx, d = e - s_ * ldx, row = row0 + s_;
      rawv[u] = 0.0; fv[u] = 0.f;
      if (e < kPChunk * ldx && d < D && row < A.N) {
        if (A.obs) fv[u] = A.obs[(size_t)row * D + d];
        else rawv[u] = A.raw_is_f64 ? reinterpret_cast<const double*>(A.raw)[(size_t)row * D + d]
                                    : (double)reinterpret_cast<const float*>(A.raw)[(size_t)row * D + d];
      }
    }
  };
  load_batch(t);
  uint64_t rng_key = 0, rng_ctr = 0;
  if (net == 0 && !A.deterministic && t < kPChunk) { rng_key = A.rng[0]; rng_ctr = A.rng[1]; }
  const int nP0 = ppo_net_params(Dp, 4);
  const int oW1 = net == 0 ? 0 : nP0, ob1 = oW1 + Dp * kPH, oW2 = ob1 + kPH, ob2 = oW2 + kPH * kPH, oWo = ob2 + kPH;
  const int oLs = nP0 + ppo_net_params(Dp, 1);
  const float* __restrict__ params = A.params;
  for (int i = t; i < Dp * kPH; i += kPThreads) W.W1[i] = params[oW1 + i];
  for (int i = t; i < kPH; i += kPThreads) { W.b1[i] = params[ob1 + i]; W.b2[i] = params[ob2 + i]; }
  for (int i = t; i < kPH * kPH; i += kPThreads) W.W2[(i >> 6) * kPLdh + (i & 63)] = params[oW2 + i];
  for (int i = t; i < kPH * KO + KO; i += kPThreads) W.Wo[i] = params[oWo + i];
  if (t < 4) log_std[t] = params[oLs + t];
  // raw-observation modes: sqrt(var + eps) of every column once per block (the division stays per element: bit-identical to
  // fw_normalize_obs); kept in the H2 area, which the forward pass only writes after X has been built
  double* cstd = reinterpret_cast<double*>(H2);
  double* cmean = cstd + 64;
  if (!A.obs) {
    if (t < D) { cstd[t] = sqrt(c_var + (double)A.eps); cmean[t] = c_mean; }
    __syncthreads();
  }
  // observations of the chunk (rows past N are zero), copied to the rollout buffer on the way by the policy block.  The loads
  // of a batch of elements leave together (one memory round trip per batch, not one per element: the fp64 division of the
  // raw modes sits between load and store and keeps the compiler from pipelining the loop itself)
  for (int e0 = t; e0 < kPChunk * ldx; e0 += kXB * kPThreads) {
    if (e0 != t) load_batch(e0);
#pragma unroll
    for (int u = 0; u < kXB; ++u) {
      const int e = e0 + u * kPThreads;
      if (e >= kPChunk * ldx) continue;
      const int s_ = e / ldx, d = e - s_ * ldx, row = row0 + s_;
      float x = 0.f;
      if (d < D && row < A.N) {
        x = A.obs ? fv[u] : fminf(fmaxf((float)((rawv[u] - cmean[d]) / cstd[d]), -A.clip), A.clip);
        if (net == 0 && A.obs_copy) A.obs_copy[(size_t)row * D + d] = x;
      }
      X[e] = x;
    }
  }
  __syncthreads();

  act_forward(W, X, H1, H2, out, KO, Dp, ldx);
  if (t < kPChunk) {
    const int row = row0 + t;
    if (row < A.N) {
      if (net == 1) {
        A.value[row] = out[t * 4];
      } else {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (!A.deterministic) act_normal4(rng_key, rng_ctr, (uint64_t)(A.env_offset + row), z);
        float lp = 0.f, a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float ls = log_std[k];
          a[k] = out[t * 4 + k] + z[k] * expf(ls);
          lp += -0.5f * z[k] * z[k] - ls - 0.9189385332046727f;
        }
        reinterpret_cast<float4*>(A.act_raw)[row] = make_float4(a[0], a[1], a[2], a[3]);
        A.logp[row] = lp;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = fminf(fmaxf(a[k], -1.0f), 1.0f);
        if (A.act_is_f64) {
          double* o = reinterpret_cast<double*>(A.act_env) + (size_t)row * 4;
          o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
        } else {
          reinterpret_cast<float4*>(A.act_env)[row] = make_float4(a[0], a[1], a[2], a[3]);
        }
      }
    }
  }
  // ---- finalisation of the previous vec-step (value block only) ----
  if (net == 1 && A.prev_reward) {
    const int row = frow;
    const bool mine = fmine;
    const bool timeout = mine && f_trunc && !f_term;
    if (__syncthreads_or(timeout ? 1 : 0)) {                      // block-uniform: some episode of my rows was truncated
      for (int e = t; e < kPChunk * ldx; e += kPThreads) {
        const int s_ = e / ldx, d = e - s_ * ldx;
        const int rw = row0 + s_;
        float x = 0.f;
        if (d < D && rw < A.N) {
          const double raw = A.raw_is_f64 ? reinterpret_cast<const double*>(A.prev_tobs)[(size_t)rw * D + d]
                                          : (double)reinterpret_cast<const float*>(A.prev_tobs)[(size_t)rw * D + d];
          x = fminf(fmaxf((float)((raw - A.mean[d]) / sqrt(A.var[d] + (double)A.eps)), -A.clip), A.clip);
        }
        X[e] = x;
      }
      __syncthreads();
      act_forward(W, X, H1, H2, out, 1, Dp, ldx);
    }
    if (mine) {
      double rn = f_rew;
      if (A.norm_reward) {
        rn *= 1.0 / sqrt(f_var + (double)A.rew_eps);
        rn = rn > A.clip_reward ? A.clip_reward : (rn < -A.clip_reward ? -A.clip_reward : rn);
      }
      float o = (float)rn;
      if (timeout) o += A.gamma * out[t * 4];                      // SB3: bootstrap truncated episodes with V(terminal_observation)
      A.rew_out[row] = o;
      A.start_out[row] = (f_term || f_trunc) ? 1.0f : 0.0f;
    }
  }
}

inline size_t act_lds_bytes(int D) {
  const int Dp = (D + 1) & ~1, ldx = Dp + 1;
  return sizeof(float) * ((size_t)Dp * kPH + kPH + kPH * kPLdh + kPH + kPH * 4 + 4 + 4 + (size_t)kPChunk * ldx + 2 * (size_t)kPChunk * kPLdh + kPChunk * 4);
}

struct PostArgs {
  const void* reward;          // [N] env dtype
  int32_t rew_is_f64;
  const uint8_t *terminated, *truncated;
  const float* tvalue;         // [N] V(normalised terminal observation)
  double* returns;             // [N] discounted-return tracker (in/out)
  double *ret_mean, *ret_var, *ret_count;   // running statistics of the tracker (in/out)
  double* ret_acc;             // null, or [3] accumulators (sum, sum of squares, count) of the tracker batches: all-reduced once per rollout by a sharded job
  int32_t N, training, norm_reward;
  double gamma;                // (double: the tracker is float64 in SB3)
  float clip_reward, epsilon;
  float* rew_out;              // [N] normalised (+ bootstrapped) reward of the step
  float* start_out;            // [N] 1 where the next step starts an episode
  uint64_t* rng;               // [2]: the draw counter rng[1] is advanced once per call
};

// one block of 1024 threads: the statistics are a reduction over all envs, and at these sizes a second launch costs more
__global__ __launch_bounds__(1024) void fw_rollout_post_kernel(PostArgs A) {
  __shared__ double red[2][16];
  __shared__ double s_var;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  auto rew = [&](int i) { return A.rew_is_f64 ? reinterpret_cast<const double*>(A.reward)[i] : (double)reinterpret_cast<const float*>(A.reward)[i]; };
  const bool track = A.training && A.norm_reward;
  if (track) {
    // VecNormalize.step_wait: returns = returns * gamma + reward; ret_rms.update(returns) (Chan et al. merge)
    double s1 = 0.0, s2 = 0.0;
    for (int i = t; i < A.N; i += 1024) {
      const double rt = A.returns[i] * A.gamma + rew(i);
      A.returns[i] = rt;
      s1 += rt; s2 += rt * rt;
    }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    if (t == 0) {
      double a = 0.0, b = 0.0;
      for (int w = 0; w < 16; ++w) { a += red[0][w]; b += red[1][w]; }
      const double n = (double)A.N, bm = a / n;
      double bv = b / n - bm * bm; bv = bv > 0.0 ? bv : 0.0;
      const double cnt = A.ret_count[0], mean = A.ret_mean[0], var = A.ret_var[0];
      const double delta = bm - mean, tot = cnt + n;
      const double m2 = var * cnt + bv * n + delta * delta * cnt * n / tot;
      A.ret_mean[0] = mean + delta * n / tot; A.ret_var[0] = m2 / tot; A.ret_count[0] = tot;
      if (A.ret_acc) { A.ret_acc[0] += a; A.ret_acc[1] += b; A.ret_acc[2] += n; }
      s_var = m2 / tot;
    }
    __syncthreads();
  } else {
    if (t == 0) s_var = A.ret_var[0];
    __syncthreads();
  }
  const double inv = 1.0 / sqrt(s_var + (double)A.epsilon);
  for (int i = t; i < A.N; i += 1024) {
    double rn = rew(i);
    if (A.norm_reward) { rn *= inv; rn = rn > A.clip_reward ? A.clip_reward : (rn < -A.clip_reward ? -A.clip_reward : rn); }
    const bool term = A.terminated[i] != 0, trunc = A.truncated[i] != 0;
    float out = (float)rn;
    if (trunc && !term) out += (float)A.gamma * A.tvalue[i];          // SB3: bootstrap truncated episodes with V(terminal_observation)
    A.rew_out[i] = out;
    const bool done = term || trunc;
    A.start_out[i] = done ? 1.0f : 0.0f;
    if (done) A.returns[i] = 0.0;
  }
  if (t == 0 && A.rng) A.rng[1] += 1;
}


// ---- fw_collect_stats: what VecNormalize.step_wait does to its statistics after an env step, in ONE launch ----
//   observations: per-block column sums of obs[N, D] -> Chan merge into (mean, var, count)        [as fw_obs_moments + merge]
//   rewards     : returns = returns * gamma + reward; sums of returns / returns^2 -> Chan merge into the return statistics;
//                 returns = 0 where the episode ended                                               [the statistics half of fw_rollout_post]
// Every block reduces its slice into the caller's workspace; the block that finishes last (one ticket atomic per block) folds the
// partials in a FIXED order -- so the result does not depend on which block that is -- and performs both merges.  The hand-off
// uses write-through (sc1) stores drained before the ticket and sc1 loads after it, NOT agent-scope fences: a release fence at
// agent scope writes back the whole L2 -- full of the state fw_step has just stored -- and made this kernel 15.3 us instead of
// 11.6.  (Measured alternatives: fw_step leaving per-workgroup column sums + a single-workgroup fold of the 512 partials: 15.5 us
// -- one workgroup cannot keep 229 KB of loads in flight; every variant pays 2-3 dependent device-memory round trips behind data
// that fw_step has just written on other XCDs.)  The draw counter of the action sampler is advanced here too.
__device__ __forceinline__ void st_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct StatsArgs {
  const void* obs; int32_t obs_is_f64; int32_t N, D;
  double *mean, *var, *count; int32_t update_obs;
  const void* reward; int32_t rew_is_f64;
  const uint8_t *terminated, *truncated;
  double* returns; double *ret_mean, *ret_var, *ret_count; int32_t update_ret;
  double gamma;
  uint64_t* rng;
  double* part;        // workspace: [nblocks][2 D + 2] partial sums
  unsigned int* ticket; // workspace: finished-block counter (left at 0)
  double *obs_acc, *ret_acc;   // sharded jobs: batch sums accumulated for the per-rollout all-reduce (may be null)
};

template <typename TIN>
__global__ __launch_bounds__(256) void fw_collect_stats_kernel(StatsArgs A) {
  __shared__ double sm1[256], sm2[256];
  __shared__ int s_last;
  const int t = threadIdx.x, D = A.D, N = A.N;
  const int rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
  double* mypart = A.part + (size_t)blockIdx.x * (2 * D + 2);
  const TIN* obs = reinterpret_cast<const TIN*>(A.obs);
  // what the last block will merge into: nobody writes these before the last block's merge, so every block fetches them now and
  // the block that turns out to be last does not pay their round trips after the hand-off
  const int pd = t & 63;
  const bool pq = t < 64 && pd < D;
  const double p_mean = pq ? A.mean[pd] : 0.0, p_var = pq ? A.var[pd] : 0.0, p_cnt = A.count[0];
  const double p_rc = A.ret_count[0], p_rm = A.ret_mean[0], p_rv = A.ret_var[0];
  auto load_rw = [&](int i) {
    return A.rew_is_f64 ? reinterpret_cast<const double*>(A.reward)[i] : (double)reinterpret_cast<const float*>(A.reward)[i];
  };
  const bool f_on = r0 + t < r1;
  const double f_rw = f_on ? load_rw(r0 + t) : 0.0, f_ret = f_on ? A.returns[r0 + t] : 0.0;
  const bool f_done = f_on && (A.terminated[r0 + t] != 0 || A.truncated[r0 + t] != 0);
  if (A.update_obs) {
    const int rpi = 256 / D;
    const bool on = t < rpi * D;
    const int col = on ? t % D : 0, rr = on ? t / D : 0;
    double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0;
    if (on) {
      int r = r0 + rr;
      for (; r + rpi < r1; r += 2 * rpi) {
        const double x0 = (double)obs[(size_t)r * D + col], x1 = (double)obs[(size_t)(r + rpi) * D + col];
        a0 += x0; b0 += x0 * x0; a1 += x1; b1 += x1 * x1;
      }
      if (r < r1) { const double x0 = (double)obs[(size_t)r * D + col]; a0 += x0; b0 += x0 * x0; }
    }
    sm1[t] = a0 + a1; sm2[t] = b0 + b1;
    __syncthreads();
    if (t < D) {
      double s = 0.0, s2 = 0.0;
      for (int k = 0; k < rpi; ++k) { s += sm1[k * D + t]; s2 += sm2[k * D + t]; }
      st_sc1(mypart + t, s); st_sc1(mypart + D + t, s2);
    }
    __syncthreads();
  }
  {
    // discounted-return tracker of my rows (the first row of every thread was fetched at the top)
    double s1 = 0.0, s2 = 0.0;
    for (int i = r0 + t; i < r1; i += 256) {
      const bool first = i == r0 + t;
      const double rw = first ? f_rw : load_rw(i);
      const bool done = first ? f_done : (A.terminated[i] != 0 || A.truncated[i] != 0);
      if (A.update_ret) {
        const double rt = (first ? f_ret : A.returns[i]) * A.gamma + rw;
        s1 += rt; s2 += rt * rt;
        A.returns[i] = done ? 0.0 : rt;
      } else if (done) {
        A.returns[i] = 0.0;
      }
    }
    sm1[t] = s1; sm2[t] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) { sm1[t] += sm1[t + o]; sm2[t] += sm2[t + o]; } __syncthreads(); }
    if (t == 0) { st_sc1(mypart + 2 * D, sm1[0]); st_sc1(mypart + 2 * D + 1, sm2[0]); }
  }
  // Every wave drains its own write-through stores (an explicit s_waitcnt: a workgroup-scope release fence is not required to
  // wait on vmcnt and compiles to nothing here) before the barrier, and the ticket is taken behind the barrier -- so no partial
  // can be overtaken by the ticket that announces it.  tools/check_isa.py asserts the wait is in the ISA of every build.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) s_last = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int nb = gridDim.x;
  const double* part = A.part;
  const size_t ps = (size_t)(2 * D + 2);
  const int q = t >> 6, dl = t & 63;                              // 4 slices of blocks x 64 columns: loads in flight together, fixed fold order
  if (A.update_obs) {
    const double cnt = p_cnt;
    for (int d0 = 0; d0 < D; d0 += 64) {
      const int d = d0 + dl;
      double s = 0.0, s2 = 0.0;
      if (d < D)
        for (int b = q; b < nb; b += 4) { s += ld_sc1(part + b * ps + d); s2 += ld_sc1(part + b * ps + D + d); }
      sm1[t] = s; sm2[t] = s2;
      __syncthreads();
      if (q == 0 && d < D) {
        s = sm1[dl] + sm1[64 + dl] + sm1[128 + dl] + sm1[192 + dl];
        s2 = sm2[dl] + sm2[64 + dl] + sm2[128 + dl] + sm2[192 + dl];
        const double bm = s / N;
        double bv = s2 / N - bm * bm;                             // population variance, as np.var
        bv = bv < 0 ? 0 : bv;
        const double om = d0 == 0 ? p_mean : A.mean[d], ov = d0 == 0 ? p_var : A.var[d];
        const double delta = bm - om, tot = cnt + N;
        const double m2 = ov * cnt + bv * N + delta * delta * cnt * N / tot;
        A.mean[d] = om + delta * N / tot;
        A.var[d] = m2 / tot;
        if (A.obs_acc) { A.obs_acc[d] += s; A.obs_acc[D + d] += s2; }
      }
      __syncthreads();
    }
  }
  // discounted-return statistics: one pair per block, tree in LDS
  sm1[t] = (A.update_ret && t < nb) ? ld_sc1(part + t * ps + 2 * D) : 0.0;
  sm2[t] = (A.update_ret && t < nb) ? ld_sc1(part + t * ps + 2 * D + 1) : 0.0;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) { if (t < o) { sm1[t] += sm1[t + o]; sm2[t] += sm2[t + o]; } __syncthreads(); }
  if (t == 0) {
    if (A.update_obs) { A.count[0] = p_cnt + (double)N; if (A.obs_acc) A.obs_acc[2 * D] += (double)N; }
    if (A.update_ret) {
      const double a = sm1[0], b = sm2[0];
      const double n = (double)N, bm = a / n;
      double bv = b / n - bm * bm; bv = bv > 0.0 ? bv : 0.0;
      const double cntr = p_rc, mean = p_rm, var = p_rv;
      const double delta = bm - mean, tot = cntr + n;
      const double m2 = var * cntr + bv * n + delta * delta * cntr * n / tot;
      A.ret_mean[0] = mean + delta * n / tot; A.ret_var[0] = m2 / tot; A.ret_count[0] = tot;
      if (A.ret_acc) { A.ret_acc[0] += a; A.ret_acc[1] += b; A.ret_acc[2] += n; }
    }
    if (A.rng) A.rng[1] += 1;
    __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
  }
}

}  // namespace fwsim
