// fw_collect_step: ONE launch per vec-step of the rollout collector (SB3 OnPolicyAlgorithm.collect_rollouts +
// VecNormalize.step_wait around Env.step, train/train_Fixedwing_Waypoints_v3.py:260,293-310).
//
// Round 2 ran a vec-step as three dependent launches, fw_collect_act -> fw_step -> fw_collect_stats: 48.3 us of which 20.5 are
// physics -- each of the two small kernels pays ~8 us of dependent-launch latency (barrier packet, the producer's L2
// write-back, cold L2s on the XCDs that did not write the data) for < 1 us of work.  Here the three are one grid:
//
//   blocks [0, n_act)           "act waves": one wave per (32-row chunk, network).  Same arithmetic as fw_collect_act (raw
//                               observation normalised on load, 64-64 tanh MLP on v_mfma_f32_32x32x2_f32, Philox / Box-Muller
//                               sampling, log-prob, the rollout-buffer rows; the value wave also finalises the PREVIOUS step:
//                               reward normalisation, truncation bootstrap, episode starts).  The policy wave publishes its 32
//                               clipped actions with write-through stores and then a generation word flag_p[chunk] = launch
//                               index; the value wave publishes flag_v[chunk] as soon as it has READ everything the env step
//                               is about to overwrite (observations, rewards, flags, terminal observations).
//   blocks [n_act, n_act+nblk)  the env step waves of fw_step (step_body<..., COLLECT = true>): they load their state, then
//                               wait -- bounded -- for the two words of the chunks their envs sit in, read the actions with
//                               coherent loads, and run the step.  Workgroups are dispatched in block order and an act wave
//                               waits for nobody, so every word a step wave waits for belongs to a wave that is already
//                               running or done: no deadlock whatever the residency.  A wait that runs out (it never should)
//                               raises status[0] and the wave steps with what it finds; tests assert status stays 0.
//   ... their epilogue          the statistics of VecNormalize.step_wait: each step wave reduces its observation tile (still in
//                               LDS) and the discounted-return tracker of its envs to 2 D + 2 partial sums; the LAST wave of
//                               each of 8 groups (group = workgroup index mod 8 = the XCD it runs on) folds its group's
//                               partials in index order, the last of those 8 folds the 8 group sums in group order and does
//                               both Chan merges -- a fixed order, so the statistics do not depend on which wave is last.
//                               Hand-off as in fw_collect_stats: write-through stores drained with s_waitcnt vmcnt(0) before a
//                               relaxed agent-scope ticket, coherent loads after it (no agent-scope fence: it would write back
//                               an L2 full of state).
//   blocks beyond               the shadow / scenario workers of fw_step, unchanged.
#pragma once
#include "fwsim_collect.hpp"

namespace fwsim {

constexpr int kCRows = 32;            // rows (envs) per act wave
constexpr int kCGroups = 8;           // first-level fold groups (workgroup index mod 8)

struct CollectArgs {
  int32_t n_act;                      // act waves at the front of the grid (2 per chunk, padded to a multiple of 8)
  int32_t n_chunks;                   // ceil(N / 32)
  int32_t latch_off;                  // byte offset in dynamic LDS of the step waves' [envs per wave][2] (reward, done) words
  ActArgs A;                          // as fw_collect_act
  StatsArgs S;                        // as fw_collect_stats (S.obs / S.part / S.ticket unused: the tile is read from LDS)
  unsigned int* flag_p;               // [n_chunks] policy wave: actions of launch `epoch` are in act_env
  unsigned int* flag_v;               // [n_chunks] value wave: inputs of launch `epoch` have been read
  double* part1;                      // [nblk][2 D + 2] per step wave
  double* part2;                      // [8][2 D + 2] per group
  unsigned int* tk;                   // [9] tickets: 8 groups + the final fold (left at 0)
  unsigned int* status;               // [1] bit 0: a step wave's wait for its actions ran out
};

__device__ __forceinline__ unsigned int ld_flag(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T ld_coherent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void st_coherent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

inline size_t collect_act_lds_bytes(int D) {
  const int Dp = (D + 1) & ~1, ldx = Dp + 1;
  return sizeof(float) * ((size_t)Dp * kPH + kPH + kPH * kPLdh + kPH + kPH * 4 + 4 + 4 + 2 * (size_t)kCRows * ldx + 2 * (size_t)kCRows * kPLdh + kCRows * 4);
}

// 32 rows through one network, one wave: X[32, Dp] -> tanh -> H1 -> tanh -> H2 -> head: out[32, 4] (KO columns used).
// Tile by tile the same MFMA sequence as act_forward (fwsim_collect.hpp), so a row's numbers are the same bits.
__device__ __forceinline__ void act_forward_wave(const PpoNetLds& W, const float* X, float* H1, float* H2, float* out, int KO, int Dp, int ldx) {
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
#pragma unroll 1
  for (int nt = 0; nt < 2; ++nt) {
    f32x16 c;
    const float bias = W.b1[nt * 32 + r];
#pragma unroll
    for (int v = 0; v < 16; ++v) c[v] = bias;
    c = ppo_mfma_tile(X, ldx, 1, W.W1 + nt * 32, kPH, 1, Dp, c);
#pragma unroll
    for (int v = 0; v < 16; ++v) H1[ppo_acc_row(v) * kPLdh + nt * 32 + r] = ppo_tanh(c[v]);
  }
  __syncthreads();
#pragma unroll 1
  for (int nt = 0; nt < 2; ++nt) {
    f32x16 c;
    const float bias = W.b2[nt * 32 + r];
#pragma unroll
    for (int v = 0; v < 16; ++v) c[v] = bias;
    c = ppo_mfma_tile(H1, kPLdh, 1, W.W2 + nt * 32, kPLdh, 1, kPH, c);
#pragma unroll
    for (int v = 0; v < 16; ++v) H2[ppo_acc_row(v) * kPLdh + nt * 32 + r] = ppo_tanh(c[v]);
  }
  __syncthreads();
  {
    f32x16 c;
    const float bias = r < KO ? W.Wo[kPH * KO + (r < KO ? r : 0)] : 0.f;       // bo follows Wo
#pragma unroll
    for (int v = 0; v < 16; ++v) c[v] = bias;
    const float* a = H2 + r * kPLdh + hh;
    const float* wo = W.Wo + hh * KO + (r < KO ? r : 0);
    c = ppo_mfma_k([&](int k0) { return a[k0]; }, [&](int k0) { return r < KO ? wo[k0 * KO] : 0.f; }, kPH, c);
    if (r < KO) {
#pragma unroll
      for (int v = 0; v < 16; ++v) out[ppo_acc_row(v) * 4 + r] = c[v];
    }
  }
  __syncthreads();
}

// One act wave (workgroup of 64 lanes).  Inlined into the collect kernels: CollectArgs is a by-value kernel argument, and
// handing its address to an out-of-line function made the compiler copy the whole struct to scratch in every wave (544 B).
// (tools/check_isa.py therefore tells MFMA accumulator registers from spill slots by the operand ranges of the MFMAs.)
__device__ __forceinline__ void collect_act_wave(const CollectArgs& CA, uint32_t epoch) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const ActArgs& A = CA.A;
  const int aw = (int)blockIdx.x, chunk = aw >> 1, net = aw & 1;
  if (chunk >= CA.n_chunks) return;                                  // padding waves
  const int KO = net == 0 ? 4 : 1;
  const int lane = threadIdx.x;
  const int D = A.D, Dp = (D + 1) & ~1, ldx = Dp + 1;
  const int row0 = chunk * kCRows;

  float* p = lds;
  PpoNetLds W;
  W.W1 = p; p += Dp * kPH; W.b1 = p; p += kPH; W.W2 = p; p += kPH * kPLdh; W.b2 = p; p += kPH; W.Wo = p; p += kPH * 4; W.bo = p; p += 4;
  float* log_std = p; p += 4;
  float* X = p;  p += kCRows * ldx;
  float* X2 = p; p += kCRows * ldx;                                  // value wave: terminal observations of the previous step
  float* H1 = p; p += kCRows * kPLdh;
  float* H2 = p; p += kCRows * kPLdh;
  float* out = p; p += kCRows * 4;

  // small loads first (their round trips hide behind the weight loads)
  double c_var = 1.0, c_mean = 0.0;
  if (lane < D) { c_var = A.var[lane]; c_mean = A.mean[lane]; }
  const int frow = row0 + (lane & 31);
  const bool fmine = net == 1 && A.prev_reward && lane < kCRows && frow < A.N;
  uint8_t f_term = 0, f_trunc = 0; double f_rew = 0.0, f_var = 1.0;
  if (fmine) {
    f_term = A.prev_term[frow]; f_trunc = A.prev_trunc[frow]; f_var = A.ret_var[0];
    f_rew = A.raw_is_f64 ? reinterpret_cast<const double*>(A.prev_reward)[frow] : (double)reinterpret_cast<const float*>(A.prev_reward)[frow];
  }
  const bool timeout = fmine && f_trunc && !f_term;
  const bool any_timeout = __ballot(timeout) != 0ull;                // wave-uniform: some episode of my rows was truncated
  uint64_t rng_key = 0, rng_ctr = 0;
  if (net == 0 && !A.deterministic && lane < kCRows) { rng_key = A.rng[0]; rng_ctr = A.rng[1]; }
  // raw observations of my rows, in batches of loads that leave together (the fp64 division sits between load and store)
  constexpr int kXB = 8;
  const int nel = kCRows * ldx;
  auto raw_at = [&](const void* base, int row, int d) {
    return A.raw_is_f64 ? reinterpret_cast<const double*>(base)[(size_t)row * D + d] : (double)reinterpret_cast<const float*>(base)[(size_t)row * D + d];
  };
  double rawv[kXB];
  auto load_batch = [&](const void* base, int e0) {
#pragma unroll
    for (int u = 0; u < kXB; ++u) {
      const int e = e0 + u * kWave;
      const int s_ = e / ldx, d = e - s_ * ldx, row = row0 + s_;
      rawv[u] = (e < nel && d < D && row < A.N) ? raw_at(base, row, d) : 0.0;
    }
  };
  load_batch(A.raw, lane);
  // weights of my network -> LDS (float4 global loads for the two matrices)
  const int nP0 = ppo_net_params(Dp, 4);
  const int oW1 = net == 0 ? 0 : nP0, ob1 = oW1 + Dp * kPH, oW2 = ob1 + kPH, ob2 = oW2 + kPH * kPH, oWo = ob2 + kPH;
  const int oLs = nP0 + ppo_net_params(Dp, 1);
  const float* __restrict__ params = A.params;
  {
    const float4* src = reinterpret_cast<const float4*>(params + oW1);
    float4* dst = reinterpret_cast<float4*>(W.W1);
    for (int i = lane; i < Dp * kPH / 4; i += kWave) dst[i] = src[i];
    const float4* s2 = reinterpret_cast<const float4*>(params + oW2);
    for (int i = lane; i < kPH * kPH / 4; i += kWave) {
      const float4 v = s2[i];
      float* q = W.W2 + ((4 * i) >> 6) * kPLdh + ((4 * i) & 63);
      q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
    }
  }
  W.b1[lane] = params[ob1 + lane]; W.b2[lane] = params[ob2 + lane];
  for (int i = lane; i < kPH * KO + KO; i += kWave) W.Wo[i] = params[oWo + i];
  if (lane < 4) log_std[lane] = params[oLs + lane];
  double* cstd = reinterpret_cast<double*>(H2);                       // per-column sqrt(var + eps) and mean (H2 is written after X is built)
  double* cmean = cstd + 64;
  if (lane < D) { cstd[lane] = sqrt(c_var + (double)A.eps); cmean[lane] = c_mean; }
  __syncthreads();
  auto build = [&](const void* base, float* dstX, bool copy) {
    for (int e0 = lane; e0 < nel; e0 += kXB * kWave) {
      if (!(base == A.raw && e0 == lane)) load_batch(base, e0);      // (the first batch of the observations is already in flight)
#pragma unroll
      for (int u = 0; u < kXB; ++u) {
        const int e = e0 + u * kWave;
        if (e >= nel) continue;
        const int s_ = e / ldx, d = e - s_ * ldx, row = row0 + s_;
        float x = 0.f;
        if (d < D && row < A.N) {
          x = fminf(fmaxf((float)((rawv[u] - cmean[d]) / cstd[d]), -A.clip), A.clip);
          if (copy && A.obs_copy) A.obs_copy[(size_t)row * D + d] = x;
        }
        dstX[e] = x;
      }
    }
  };
  build(A.raw, X, net == 0);
  if (net == 1) {
    if (any_timeout) build(A.prev_tobs, X2, false);
    // everything the env step of THIS launch overwrites has been read: let the step waves of my chunk go
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) st_flag(CA.flag_v + chunk, epoch);
  }
  __syncthreads();

  act_forward_wave(W, X, H1, H2, out, KO, Dp, ldx);
  if (lane < kCRows) {
    const int row = row0 + lane;
    if (row < A.N) {
      if (net == 1) {
        A.value[row] = out[lane * 4];
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
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = fminf(fmaxf(a[k], -1.0f), 1.0f);
        // the env's action row: write-through, the step waves of other XCDs read it in this same launch
        if (A.act_is_f64) {
          double* o = reinterpret_cast<double*>(A.act_env) + (size_t)row * 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) st_coherent(o + k, (double)a[k]);
        } else {
          float* o = reinterpret_cast<float*>(A.act_env) + (size_t)row * 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) st_coherent(o + k, a[k]);
        }
      }
    }
  }
  if (net == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the wave's action stores have left ...
    if (lane == 0) st_flag(CA.flag_p + chunk, epoch);                 // ... before the word that announces them
    return;
  }
  // ---- value wave: finalisation of the previous vec-step ----
  if (A.prev_reward) {
    if (any_timeout) {
      __syncthreads();
      act_forward_wave(W, X2, H1, H2, out, 1, Dp, ldx);
    }
    if (fmine) {
      double rn = f_rew;
      if (A.norm_reward) {
        rn *= 1.0 / sqrt(f_var + (double)A.rew_eps);
        rn = rn > A.clip_reward ? A.clip_reward : (rn < -A.clip_reward ? -A.clip_reward : rn);
      }
      float o = (float)rn;
      if (timeout) o += A.gamma * out[lane * 4];                        // SB3: bootstrap truncated episodes with V(terminal_observation)
      A.rew_out[frow] = o;
      A.start_out[frow] = (f_term || f_trunc) ? 1.0f : 0.0f;
    }
  }
}

// Step wave: wait (bounded) until the act waves of the chunks covering rows [env0, env0 + rows) have published for `epoch`.
__device__ __forceinline__ void collect_wait_actions(const CollectArgs& CA, uint32_t epoch, int env0, int rows) {
  const int c0 = env0 / kCRows, c1 = min((env0 + rows - 1) / kCRows, CA.n_chunks - 1);
  const int lane = threadIdx.x & 63;
  const int c = c0 + (lane >> 1);
  const unsigned int* w = ((lane & 1) ? CA.flag_v : CA.flag_p) + (c <= c1 ? c : c1);
  bool ok = false;
  for (int it = 0; it < (1 << 21); ++it) {                             // ~ seconds: far beyond any healthy launch
    const bool mine = (c > c1) || ld_flag(w) == epoch;
    if (__ballot(!mine) == 0ull) { ok = true; break; }
    __builtin_amdgcn_s_sleep(2);
  }
  if (!ok && lane == 0) atomicOr(CA.status, 1u);
}

// Statistics tail of a step wave.  `tile` = the wave's observation rows in LDS ([rows][ld], final: reset rows included),
// `ret_new` / `done` = this lane's env (leader lanes only, `is_leader`), rows = active envs of the wave, wg = workgroup index
// among the nblk step workgroups.
template <typename T>
__device__ __forceinline__ void collect_stats_tail(const CollectArgs& CA, const T* tile, int ld, int rows, int wg, int nblk,
                                                   bool is_leader, int my_row, double rew, bool done, int env) {
  const StatsArgs& S = CA.S;
  const int lane = threadIdx.x & 63, D = S.D, N = S.N, PW = 2 * D + 2;
  // ---- my wave's partial sums ----
  double s = 0.0, s2 = 0.0;
  if (S.update_obs && lane < D) {
    for (int r = 0; r < rows; ++r) { const double x = (double)tile[r * ld + lane]; s += x; s2 += x * x; }
  }
  double rt = 0.0;
  if (is_leader) {
    if (S.update_ret) { rt = S.returns[env] * S.gamma + rew; S.returns[env] = done ? 0.0 : rt; }
    else if (done) S.returns[env] = 0.0;
  }
  // leaders' tracker values summed in row order (a fixed order: the result does not depend on the lane mapping's timing)
  double r1 = 0.0, r2 = 0.0;
  {
    const unsigned long long lead = __ballot(is_leader);
    for (int l = 0; l < 64; ++l) {
      if (!((lead >> l) & 1ull)) continue;
      const double v = __shfl(rt, l, 64);
      r1 += v; r2 += v * v;
    }
  }
  double* mine = CA.part1 + (size_t)wg * PW;
  if (lane < D) { st_sc1(mine + lane, s); st_sc1(mine + D + lane, s2); }
  if (lane == 0) { st_sc1(mine + 2 * D, r1); st_sc1(mine + 2 * D + 1, r2); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- level 1: the last wave of my group folds the group ----
  const int grp = wg & (kCGroups - 1);
  const int members = (nblk - grp + kCGroups - 1) / kCGroups;
  unsigned int t1 = 0;
  if (lane == 0) t1 = __hip_atomic_fetch_add(CA.tk + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t1 = __shfl(t1, 0, 64);
  if ((int)t1 != members - 1) return;
  for (int w0 = 0; w0 < PW; w0 += 64) {
    const int w = w0 + lane;
    double a = 0.0;
    if (w < PW)
      for (int m = 0; m < members; ++m) a += ld_sc1(CA.part1 + (size_t)(grp + m * kCGroups) * PW + w);
    if (w < PW) st_sc1(CA.part2 + (size_t)grp * PW + w, a);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- level 2: the last group folds the groups and merges ----
  const int ngrp = min(nblk, kCGroups);
  unsigned int t2 = 0;
  if (lane == 0) t2 = __hip_atomic_fetch_add(CA.tk + kCGroups, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t2 = __shfl(t2, 0, 64);
  if ((int)t2 != ngrp - 1) return;
  double tot_r1 = 0.0, tot_r2 = 0.0;
  for (int g = 0; g < ngrp; ++g) { tot_r1 += ld_sc1(CA.part2 + (size_t)g * PW + 2 * D); tot_r2 += ld_sc1(CA.part2 + (size_t)g * PW + 2 * D + 1); }
  if (S.update_obs) {
    const double cnt = S.count[0];
    for (int d0 = 0; d0 < D; d0 += 64) {
      const int d = d0 + lane;
      if (d < D) {
        double cs = 0.0, cs2 = 0.0;
        for (int g = 0; g < ngrp; ++g) { cs += ld_sc1(CA.part2 + (size_t)g * PW + d); cs2 += ld_sc1(CA.part2 + (size_t)g * PW + D + d); }
        const double bm = cs / N;
        double bv = cs2 / N - bm * bm;                               // population variance, as np.var
        bv = bv < 0 ? 0 : bv;
        const double om = S.mean[d], ov = S.var[d];
        const double delta = bm - om, tot = cnt + N;
        const double m2 = ov * cnt + bv * N + delta * delta * cnt * N / tot;
        S.mean[d] = om + delta * N / tot;
        S.var[d] = m2 / tot;
        if (S.obs_acc) { S.obs_acc[d] += cs; S.obs_acc[D + d] += cs2; }
      }
    }
    if (lane == 0) { S.count[0] = cnt + (double)N; if (S.obs_acc) S.obs_acc[2 * D] += (double)N; }
  }
  if (lane == 0) {
    if (S.update_ret) {
      const double n = (double)N, bm = tot_r1 / n;
      double bv = tot_r2 / n - bm * bm; bv = bv > 0.0 ? bv : 0.0;
      const double cntr = S.ret_count[0], mean = S.ret_mean[0], var = S.ret_var[0];
      const double delta = bm - mean, tot = cntr + n;
      const double m2 = var * cntr + bv * n + delta * delta * cntr * n / tot;
      S.ret_mean[0] = mean + delta * n / tot; S.ret_var[0] = m2 / tot; S.ret_count[0] = tot;
      if (S.ret_acc) { S.ret_acc[0] += tot_r1; S.ret_acc[1] += tot_r2; S.ret_acc[2] += n; }
    }
    if (S.rng) S.rng[1] += 1;
    for (int g = 0; g <= kCGroups; ++g) st_flag(CA.tk + g, 0u);      // every ticket has been taken: ready for the next launch
  }
}

}  // namespace fwsim
