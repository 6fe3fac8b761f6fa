// fwsim_rollout.hpp -- device kernels of the rollout collector (caller side of the env
// step): GAE scan and the fused VecNormalize observation pass.  HBM-bound, coalesced.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fwsim {

// K3: GAE.  One lane per env; the time loop runs backwards in registers; every access at
// step t is a coalesced row of the [T, N] buffers.  (SB3 RolloutBuffer.compute_returns_and_advantage.)
__global__ __launch_bounds__(256) void fw_gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                                                     const float* __restrict__ episode_starts,
                                                     const float* __restrict__ last_values,
                                                     const float* __restrict__ last_dones, float* __restrict__ adv,
                                                     float* __restrict__ ret, int T, int N, float gamma, float lam) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float next_value = last_values[n];
  float next_non_terminal = 1.0f - last_dones[n];
  float last_gae = 0.0f;
  for (int t = T - 1; t >= 0; --t) {
    const size_t i = (size_t)t * N + n;
    const float v = values[i];
    const float delta = rewards[i] + gamma * next_value * next_non_terminal - v;
    last_gae = delta + gamma * lam * next_non_terminal * last_gae;
    adv[i] = last_gae;
    ret[i] = last_gae + v;
    next_value = v;
    next_non_terminal = 1.0f - episode_starts[i];
  }
}

// Episode bookkeeping of an evaluation (evaluate.ReplayedEvaluation, SB3's evaluate_policy loop): one vec-step's rewards and dones into
// the running accumulators, a finished episode into the next slot of its env -- what the harness did with ~20 framework ops.
// One workgroup: the step counter is read by everybody and advanced once, behind the barrier.
struct EvalTrackArgs {
  const void* reward; int32_t reward_is_f64;
  const uint8_t *terminated, *truncated;
  const int32_t* info; int32_t info_dim;     // may be NULL
  const int64_t* targets;                    // [N] episodes wanted of each env
  int64_t* counts;                           // [N] episodes taken so far
  double* cur_rew; int64_t* cur_len;         // [N] accumulators of the running episodes
  int64_t* step_ctr;                         // [1] vec-steps so far
  double* fin_rew; int64_t *fin_len, *fin_step; int32_t* fin_info;      // [N, E] (fin_info [N, E, info_dim])
  int32_t N, E;
};
__global__ __launch_bounds__(256) void fw_eval_track_kernel(EvalTrackArgs A) {
  const long long step = A.step_ctr[0] + 1;
  for (int i = threadIdx.x; i < A.N; i += (int)blockDim.x) {
    const double r = A.reward_is_f64 ? reinterpret_cast<const double*>(A.reward)[i] : (double)reinterpret_cast<const float*>(A.reward)[i];
    const double cr = A.cur_rew[i] + r;
    const long long cl = A.cur_len[i] + 1;
    const bool done = (A.terminated[i] | A.truncated[i]) != 0;
    const long long c = A.counts[i];
    if (done && c < A.targets[i]) {
      const size_t s = (size_t)i * A.E + (size_t)(c < A.E ? c : A.E - 1);
      A.fin_rew[s] = cr; A.fin_len[s] = cl; A.fin_step[s] = step;
      if (A.info) for (int k = 0; k < A.info_dim; ++k) A.fin_info[s * A.info_dim + k] = A.info[(size_t)i * A.info_dim + k];
      A.counts[i] = c + 1;
    }
    A.cur_rew[i] = done ? 0.0 : cr;
    A.cur_len[i] = done ? 0 : cl;
  }
  __syncthreads();
  if (threadIdx.x == 0) A.step_ctr[0] = step;
}

// K4a: per-column batch moments of obs[N,D] (two-pass-free: shifted sums in double), one
// workgroup per column chunk; K4b merges them into the running statistics (Chan et al.) and
// K4c normalises.  N*D is small (4096 x 28), so the three launches are latency-trivial and
// can all be captured in the rollout hipGraph.
template <typename TIN>
__global__ __launch_bounds__(256) void fw_obs_moments_kernel(const TIN* __restrict__ obs, int N, int D,
                                                             double* __restrict__ part /*[gridDim.x][2][D]*/) {
  // Block b reduces rows [r0, r1).  Thread t handles column t % D of rows r0 + t / D + k * (256 / D): consecutive
  // threads read consecutive addresses; one LDS reduction over the 256 / D row slots at the end.  (D <= 256)
  __shared__ double sm1[256], sm2[256];
  const int t = threadIdx.x;
  const int rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
  const int rpi = 256 / D;
  const bool on = t < rpi * D;
  const int col = on ? t % D : 0, rr = on ? t / D : 0;
  double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0;          // two chains: the loads are independent, the adds are not
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
    part[((size_t)blockIdx.x * 2 + 0) * D + t] = s;
    part[((size_t)blockIdx.x * 2 + 1) * D + t] = s2;
  }
}

// RunningMeanStd.update_from_moments (SB3 common/running_mean_std.py); one block of 256 threads:
// thread (q, d) sums a quarter of the per-block partials of column d (fixed order => reproducible), D <= 64 per pass
// `acc` (may be null): double[2 D + 1] accumulators of the batch sums (sum, sum of squares per column, rows) -- what a
// sharded job all-reduces once per rollout to bring every rank's statistics to the statistics of ALL envs.
__global__ __launch_bounds__(256) void fw_obs_merge_kernel(const double* __restrict__ part, int nblocks, int N, int D,
                                                           double* __restrict__ mean, double* __restrict__ var,
                                                           double* __restrict__ count, double* __restrict__ acc) {
  __shared__ double sm1[256], sm2[256];
  const int t = threadIdx.x, q = t >> 6, dl = t & 63;
  const double cnt = count[0];
  for (int d0 = 0; d0 < D; d0 += 64) {
    const int d = d0 + dl;
    double s = 0, s2 = 0;
    if (d < D)
      for (int b = q; b < nblocks; b += 4) { s += part[((size_t)b * 2 + 0) * D + d]; s2 += part[((size_t)b * 2 + 1) * D + d]; }
    sm1[t] = s; sm2[t] = s2;
    __syncthreads();
    if (q == 0 && d < D) {
      s = sm1[dl] + sm1[64 + dl] + sm1[128 + dl] + sm1[192 + dl];
      s2 = sm2[dl] + sm2[64 + dl] + sm2[128 + dl] + sm2[192 + dl];
      const double bm = s / N;
      double bv = s2 / N - bm * bm;                // population variance, as np.var
      bv = bv < 0 ? 0 : bv;
      const double delta = bm - mean[d];
      const double tot = cnt + N;
      const double new_mean = mean[d] + delta * N / tot;
      const double m2 = var[d] * cnt + bv * N + delta * delta * cnt * N / tot;
      mean[d] = new_mean;
      var[d] = m2 / tot;
      if (acc) { acc[d] += s; acc[D + d] += s2; }
    }
    __syncthreads();
  }
  if (t == 0) { count[0] = cnt + N; if (acc) acc[2 * D] += (double)N; }
}

template <typename TIN>
__global__ __launch_bounds__(256) void fw_obs_normalize_kernel(const TIN* __restrict__ obs, int total, int D,
                                                               const double* __restrict__ mean,
                                                               const double* __restrict__ var, float clip, float eps,
                                                               float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int d = i % D;
  double z = ((double)obs[i] - mean[d]) / sqrt(var[d] + (double)eps);
  float zf = (float)z;
  out[i] = fminf(fmaxf(zf, -clip), clip);
}

}  // namespace fwsim
