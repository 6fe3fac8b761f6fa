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

// K4a: per-column batch moments of obs[N,D] (two-pass-free: shifted sums in double), one
// workgroup per column chunk; K4b merges them into the running statistics (Chan et al.) and
// K4c normalises.  N*D is small (4096 x 28), so the three launches are latency-trivial and
// can all be captured in the rollout hipGraph.
template <typename TIN>
__global__ __launch_bounds__(256) void fw_obs_moments_kernel(const TIN* __restrict__ obs, int N, int D,
                                                             double* __restrict__ part /*[gridDim.x][2][D]*/) {
  // each block reduces rows [r0, r1) for all D columns; thread = (row lane, column)
  extern __shared__ double sm[];       // [256] scratch per pass
  const int rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
  for (int d = 0; d < D; ++d) {
    double s = 0.0, s2 = 0.0;
    for (int r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
      double x = (double)obs[(size_t)r * D + d];
      s += x; s2 += x * x;
    }
    // block reduce (wave shuffles then LDS)
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); s2 += __shfl_down(s2, o, 64); }
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { sm[w] = s; sm[4 + w] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = 0, b = 0;
      for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { a += sm[k]; b += sm[4 + k]; }
      part[((size_t)blockIdx.x * 2 + 0) * D + d] = a;
      part[((size_t)blockIdx.x * 2 + 1) * D + d] = b;
    }
    __syncthreads();
  }
}

// RunningMeanStd.update_from_moments (SB3 common/running_mean_std.py)
__global__ void fw_obs_merge_kernel(const double* __restrict__ part, int nblocks, int N, int D, double* __restrict__ mean,
                                    double* __restrict__ var, double* __restrict__ count) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const double cnt = count[0];
  if (d < D) {
    double s = 0, s2 = 0;
    for (int b = 0; b < nblocks; ++b) { s += part[((size_t)b * 2 + 0) * D + d]; s2 += part[((size_t)b * 2 + 1) * D + d]; }
    const double bm = s / N;
    double bv = s2 / N - bm * bm;                // population variance, as np.var
    bv = bv < 0 ? 0 : bv;
    const double delta = bm - mean[d];
    const double tot = cnt + N;
    const double new_mean = mean[d] + delta * N / tot;
    const double m2 = var[d] * cnt + bv * N + delta * delta * cnt * N / tot;
    mean[d] = new_mean;
    var[d] = m2 / tot;
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) count[0] = cnt + N;
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
