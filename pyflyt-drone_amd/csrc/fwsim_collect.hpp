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

// grid = (ceil(N / 64), 2): block (c, net) runs network `net` on rows 64 c .. 64 c + 63; 256 threads.
__global__ __launch_bounds__(kPThreads) void fw_policy_act_kernel(ActArgs A) {
  extern __shared__ __align__(16) float lds[];
  const int net = blockIdx.y;
  if (!((A.nets >> net) & 1)) return;
  const int KO = net == 0 ? 4 : 1;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, hh = lane >> 5;
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

  const int nP0 = ppo_net_params(Dp, 4);
  const int oW1 = net == 0 ? 0 : nP0, ob1 = oW1 + Dp * kPH, oW2 = ob1 + kPH, ob2 = oW2 + kPH * kPH, oWo = ob2 + kPH;
  const int oLs = nP0 + ppo_net_params(Dp, 1);
  const float* __restrict__ params = A.params;
  for (int i = t; i < Dp * kPH; i += kPThreads) W.W1[i] = params[oW1 + i];
  for (int i = t; i < kPH; i += kPThreads) { W.b1[i] = params[ob1 + i]; W.b2[i] = params[ob2 + i]; }
  for (int i = t; i < kPH * kPH; i += kPThreads) W.W2[(i >> 6) * kPLdh + (i & 63)] = params[oW2 + i];
  for (int i = t; i < kPH * KO + KO; i += kPThreads) W.Wo[i] = params[oWo + i];
  if (t < 4) log_std[t] = params[oLs + t];
  // observations of the chunk (rows past N are zero), copied to the rollout buffer on the way by the policy block
  for (int e = t; e < kPChunk * ldx; e += kPThreads) {
    const int s = e / ldx, d = e - s * ldx;
    const int row = row0 + s;
    float x = 0.f;
    if (d < D && row < A.N) {
      if (A.obs) {
        x = A.obs[(size_t)row * D + d];
      } else {
        const double raw = A.raw_is_f64 ? reinterpret_cast<const double*>(A.raw)[(size_t)row * D + d]
                                        : (double)reinterpret_cast<const float*>(A.raw)[(size_t)row * D + d];
        x = fminf(fmaxf((float)((raw - A.mean[d]) / sqrt(A.var[d] + (double)A.eps)), -A.clip), A.clip);
      }
      if (net == 0 && A.obs_copy) A.obs_copy[(size_t)row * D + d] = x;
    }
    X[e] = x;
  }
  __syncthreads();

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
  if (t < kPChunk) {
    const int row = row0 + t;
    if (row < A.N) {
      if (net == 1) {
        A.value[row] = out[t * 4];
      } else {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (!A.deterministic) act_normal4(A.rng[0], A.rng[1], (uint64_t)(A.env_offset + row), z);
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

}  // namespace fwsim
