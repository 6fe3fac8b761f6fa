// fw_collect_step: ONE launch per vec-step of the rollout collector (SB3 OnPolicyAlgorithm.collect_rollouts +
// VecNormalize.step_wait around Env.step, train/train_Fixedwing_Waypoints_v3.py:260,293-310).
//
// Round 2 ran a vec-step as three dependent launches, fw_collect_act -> fw_step -> fw_collect_stats: 48.3 us of which 20.5 are
// physics -- each of the two small kernels pays ~8 us of dependent-launch latency (barrier packet, the producer's L2
// write-back, cold L2s on the XCDs that did not write the data) for < 1 us of work.  Here the three are one grid:
//
//   blocks [0, 2 n_chunks)      "act waves": one wave per (32-row chunk, network).  Same arithmetic as fw_collect_act (raw
//                               observation normalised on load, 64-64 tanh MLP on v_mfma_f32_32x32x2_f32, Philox / Box-Muller
//                               sampling, log-prob, the rollout-buffer rows; the value wave also finalises the PREVIOUS step:
//                               reward normalisation, truncation bootstrap, episode starts).  The policy wave publishes its 32
//                               clipped actions with write-through stores and then a generation word flag_p[chunk] = launch
//                               index; the value wave publishes flag_v[chunk] as soon as it has READ everything the env step
//                               is about to overwrite (observations, rewards, flags, terminal observations).
//   block 2 n_chunks            the "merge wave": writes the updated statistics back for the caller (see below).
//   blocks [n_act, n_act+nblk)  the env step waves of fw_step (step_body<..., COLLECT = true>): they load their state, then
//                               wait -- bounded -- for the two words of the chunks their envs sit in, read the actions with
//                               coherent loads, and run the step.  Workgroups are dispatched in block order and an act wave
//                               waits only for act waves in front of it, so every word a wave waits for belongs to a wave that
//                               is already running or done: no deadlock whatever the residency.  A wait that runs out (it
//                               never should) raises status and the wave goes on with what it finds; tests assert it stays 0.
//   blocks beyond               the shadow / scenario workers of fw_step, unchanged.
//
// The statistics of VecNormalize.step_wait (observation moments, discounted-return tracker) need a reduction over ALL envs
// between the env step and the next policy forward.  A first version folded them in the step waves' tail (last wave of 8
// groups, then the last group): 36 us -- every hop between waves of different XCDs is a write-through store, its
// acknowledgement, a ticket and a coherent load, ~3 us each, six of them in a row (tools/trace_collect.py).  Now a step wave
// only leaves its 2 D + 2 partial sums with plain stores (they become visible at the kernel boundary, for free) and the NEXT
// launch folds them while its weights are in flight anyway: act wave j sums word j over all step waves (one coalesced round
// trip, fixed association) and stores the total over a sentinel the merge wave of the launch before last left -- a total
// announces itself, no flag and no store acknowledgement in between; every act wave reads the 2 D + 2 totals and derives the
// updated statistics itself (same arithmetic, same bits everywhere).  The merge wave does the same and writes them to the
// caller's buffers -- after every act wave has announced that it has read the old ones.  The last step of a rollout is folded by
// fw_collect_finish (one small launch per rollout).
// Measured (tools/trace_collect.py, profiles/r03_collect_step_trace.txt; waypoints, 4096 envs): statistics known 9.4 us into the
// launch, inputs + weights in LDS +4.2, forward +7.7, actions published at 23-24 us, step waves done at 43, launch end 45 us --
// 49.1 us per vec-step with the per-rollout launches, against 48.3 for fw_collect_act -> fw_step -> fw_collect_stats.  A bare
// hand-off between two waves costs 0.36 us inside an XCD and 0.41 us across two (tools/microbench_xcd.hip); in the grid, with
// hundreds of waves polling, each of the three dependent hops (partials -> totals -> inputs; actions -> step waves) measures
// 2-3 us, and one wave per (chunk, network) runs its three GEMM phases in 7.7 us where the four waves of fw_collect_act's
// workgroups need ~4.  (Tried: one completion counter instead of self-announcing totals and a 7 us nap before the step waves
// start polling: 12.1 us to the statistics, 52 us per vec-step -- worse.)  The launch boundaries this design removes cost what
// its in-grid hand-offs cost: it stays an option (PPOConfig.one_launch_collect), off by default.
#pragma once
#include "fwsim_collect.hpp"

namespace fwsim {

constexpr int kCRows = 32;            // rows (envs) per act wave
constexpr int kCGroups = 8;           // partial-sum rows are grouped by workgroup index mod 8 (the XCD a step wave runs on)
enum { CS_PART_EPOCH = 0, CS_FOLDED_EPOCH = 1, CS_READERS = 2, CS_STATUS = 3, CS_CEPOCH = 4 };
// a total that has not been published yet: a quiet NaN no sum can produce
__device__ __forceinline__ double collect_sentinel() { return __longlong_as_double(0x7FF8C0DEC0DE0001ll); }
__device__ __forceinline__ bool collect_is_sentinel(double v) { return __double_as_longlong(v) == 0x7FF8C0DEC0DE0001ll; }

struct CollectArgs {
  int32_t n_act;                      // workgroups in front of the step waves: 2 per chunk + the merge wave, padded to a multiple of 8
  int32_t n_chunks;                   // ceil(N / 32)
  int32_t nblk;                       // step workgroups
  ActArgs A;                          // as fw_collect_act
  StatsArgs S;                        // as fw_collect_stats (S.obs / S.part / S.ticket unused)
  unsigned int* flag_p;               // [n_chunks] policy wave: actions of launch `epoch` are in act_env
  unsigned int* flag_v;               // [n_chunks] value wave: inputs of launch `epoch` have been read
  double* part1;                      // [2 D + 2][8][ceil(nblk / 8)] partial sums of the step waves: word-major, a group's members consecutive
  double* tot;                        // [2][2 D + 2] totals of the pending step, double-buffered by collect-launch parity; unpublished = sentinel
  unsigned int* sync;                 // [8] CS_*: launch index of the partials in part1 / of the last fold, readers counter, status bits, collect-launch counter
  long long* trace;                   // null, or [grid][8] wall-clock stamps (10 ns ticks) per workgroup: tools/trace_collect.py
};

__device__ __forceinline__ unsigned int ld_flag(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long collect_now() { return (long long)__builtin_amdgcn_s_memrealtime(); }
template <typename T> __device__ __forceinline__ T ld_coherent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void st_coherent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

inline size_t collect_act_lds_bytes(int D) {
  const int Dp = (D + 1) & ~1, ldx = Dp + 1;
  return sizeof(float) * ((size_t)Dp * kPH + kPH + kPH * kPLdh + kPH + kPH * 4 + 4 + 4 + 2 * (size_t)kCRows * ldx + 2 * (size_t)kCRows * kPLdh + kCRows * 4);
}

// C0 / C1 (32 x 32 each) += A(32 x K) * B(K x 64): both column tiles of one layer in ONE pass over K -- the A operand is read
// once for the two MFMAs of a k-step and the two accumulators interleave, so one tile's LDS latency hides behind the other's
// MFMA.  Per tile the k order and the operands are those of ppo_mfma_tile: the same bits.
template <int STEPS>
__device__ __forceinline__ void act_mfma2_batch(const float* a, int sak, const float* b, int sbk, int step0, f32x16& c0, f32x16& c1) {
  float av[STEPS], b0[STEPS], b1[STEPS];
#pragma unroll
  for (int i = 0; i < STEPS; ++i) { const int k0 = 2 * (step0 + i); av[i] = a[k0 * sak]; b0[i] = b[k0 * sbk]; b1[i] = b[k0 * sbk + 32]; }
#pragma unroll
  for (int i = 0; i < STEPS; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b0[i], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b1[i], c1, 0, 0, 0);
  }
}
__device__ __forceinline__ void act_mfma2(const float* A, int sam, const float* B, int sbk, int K, f32x16& c0, f32x16& c1) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const float* a = A + r * sam + h;            // A(m, k) = A[m * sam + k]
  const float* b = B + h * sbk + r;            // B(k, n) = B[k * sbk + n]
  const int steps = K >> 1;
  int s = 0;
  for (; s + 8 <= steps; s += 8) act_mfma2_batch<8>(a, 1, b, sbk, s, c0, c1);
  if (s + 4 <= steps) { act_mfma2_batch<4>(a, 1, b, sbk, s, c0, c1); s += 4; }
  if (s + 2 <= steps) { act_mfma2_batch<2>(a, 1, b, sbk, s, c0, c1); s += 2; }
  if (s < steps) act_mfma2_batch<1>(a, 1, b, sbk, s, c0, c1);
}

// 32 rows through one network, one wave: X[32, Dp] -> tanh -> H1 -> tanh -> H2 -> head: out[32, 4] (KO columns used).
__device__ __forceinline__ void act_forward_wave(const PpoNetLds& W, const float* X, float* H1, float* H2, float* out, int KO, int Dp, int ldx) {
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  {
    f32x16 c0, c1;
    const float bias0 = W.b1[r], bias1 = W.b1[32 + r];
#pragma unroll
    for (int v = 0; v < 16; ++v) { c0[v] = bias0; c1[v] = bias1; }
    act_mfma2(X, ldx, W.W1, kPH, Dp, c0, c1);
#pragma unroll
    for (int v = 0; v < 16; ++v) { H1[ppo_acc_row(v) * kPLdh + r] = ppo_tanh(c0[v]); H1[ppo_acc_row(v) * kPLdh + 32 + r] = ppo_tanh(c1[v]); }
  }
  __syncthreads();
  {
    f32x16 c0, c1;
    const float bias0 = W.b2[r], bias1 = W.b2[32 + r];
#pragma unroll
    for (int v = 0; v < 16; ++v) { c0[v] = bias0; c1[v] = bias1; }
    act_mfma2(H1, kPLdh, W.W2, kPLdh, kPH, c0, c1);
#pragma unroll
    for (int v = 0; v < 16; ++v) { H2[ppo_acc_row(v) * kPLdh + r] = ppo_tanh(c0[v]); H2[ppo_acc_row(v) * kPLdh + 32 + r] = ppo_tanh(c1[v]); }
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

// Sum of word w over the step waves' partial rows, by one wave, in a fixed association: lane l adds its slots l, l + 64, ... in
// ascending order, then an xor-butterfly over the lanes.  (Slots of members a group does not have stay at their initial zero.)
__device__ __forceinline__ double collect_fold_word(const double* part1, int w, int slots) {
  const int lane = threadIdx.x & 63;
  const double* row = part1 + (size_t)w * slots;
  double v = 0.0;
  for (int i = lane; i < slots; i += 64) v += row[i];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Running statistics after the pending step: (old statistics) (+) (batch sums), SB3's RunningMeanStd.update_from_moments.
struct CollectMerged { double mean, var; };
__device__ __forceinline__ CollectMerged collect_chan(double om, double ov, double cnt, double cs, double cs2, double n) {
  const double bm = cs / n;
  double bv = cs2 / n - bm * bm;                                     // population variance, as np.var
  bv = bv < 0 ? 0 : bv;
  const double delta = bm - om, tot = cnt + n;
  const double m2 = ov * cnt + bv * n + delta * delta * cnt * n / tot;
  CollectMerged r; r.mean = om + delta * n / tot; r.var = m2 / tot;
  return r;
}

// What every wave in front of the step waves does first: read the statistics the previous launches left, take its share of
// the pending fold, wait for the totals and derive the statistics THIS step is normalised with.  Lane d < D ends with column d's
// (mean, var); `ret_var` is wave-uniform.  Returns false if a wait ran out.
struct CollectStats { double mean, var, cnt, ret_mean, ret_var, ret_cnt; double cs, cs2, r1, r2; bool pend_obs, pend_ret; unsigned int cepoch; };
__device__ __forceinline__ bool collect_front_stats(const CollectArgs& CA, int aw, bool count_me, CollectStats& Q) {
  const StatsArgs& S = CA.S;
  const int lane = threadIdx.x & 63, D = S.D, PW = 2 * D + 2, n_real = 2 * CA.n_chunks;
  const unsigned int part_ep = CA.sync[CS_PART_EPOCH], folded_ep = CA.sync[CS_FOLDED_EPOCH];
  Q.cepoch = CA.sync[CS_CEPOCH];
  // my share of the fold is read whether or not a step is pending (the loads leave with the first round trip of the wave; a
  // wave-uniform branch on `pend` would put them behind the scalar loads above)
  const int slots = kCGroups * ((CA.nblk + kCGroups - 1) / kCGroups);
  const int nf = min(n_real, 64);                                      // fold waves: the first act waves; word w belongs to wave w mod nf
  double fw0 = 0.0, fw1 = 0.0;                                         // (2 D + 2 <= 126 words over up to 64 waves: at most two each)
  if (aw < nf) {
    if (aw < PW) fw0 = collect_fold_word(CA.part1, aw, slots);
    if (aw + nf < PW) fw1 = collect_fold_word(CA.part1, aw + nf, slots);
  }
  Q.mean = 0.0; Q.var = 1.0;
  if (lane < D) { Q.mean = S.mean[lane]; Q.var = S.var[lane]; }
  Q.cnt = S.count[0]; Q.ret_mean = S.ret_mean[0]; Q.ret_var = S.ret_var[0]; Q.ret_cnt = S.ret_count[0];
  Q.cs = Q.cs2 = Q.r1 = Q.r2 = 0.0;
  const bool pend = part_ep != folded_ep;
  Q.pend_obs = pend && S.update_obs; Q.pend_ret = pend && S.update_ret;
  double* tot = CA.tot + (size_t)(Q.cepoch & 1u) * PW;
  if (pend && aw < nf && lane == 0) {
    if (aw < PW) st_sc1(tot + aw, fw0);
    if (aw + nf < PW) st_sc1(tot + aw + nf, fw1);
  }
  // the old statistics are in registers: tell the merge wave (it overwrites them only after every wave in front has said so)
  asm volatile("" :: "v"(Q.mean), "v"(Q.var), "v"(Q.cnt), "v"(Q.ret_mean), "v"(Q.ret_var), "v"(Q.ret_cnt), "s"(part_ep), "s"(folded_ep), "s"(Q.cepoch) : "memory");
  if (count_me && lane == 0) (void)__hip_atomic_fetch_add(CA.sync + CS_READERS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!(Q.pend_obs || Q.pend_ret)) return true;
  // the totals announce themselves: a word is there as soon as it is not the sentinel the merge wave of the launch before
  // last left (no flag, no store acknowledgement in between)
  const int dcol = lane < D ? lane : 0;
  const double* pa = tot + (lane == 63 ? 2 * D : dcol);
  const double* pb = tot + (lane == 63 ? 2 * D + 1 : D + dcol);
  bool ok = false;
  for (int it = 0; it < (1 << 21); ++it) {
    Q.cs = ld_sc1(pa); Q.cs2 = ld_sc1(pb);
    const bool mine = !collect_is_sentinel(Q.cs) && !collect_is_sentinel(Q.cs2);
    if (__ballot(!mine) == 0ull) { ok = true; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  Q.r1 = __shfl(Q.cs, 63, 64); Q.r2 = __shfl(Q.cs2, 63, 64);
  const double n = (double)S.N;
  if (Q.pend_obs) {
    const CollectMerged m = collect_chan(Q.mean, Q.var, Q.cnt, Q.cs, Q.cs2, n);
    if (lane < D) { Q.mean = m.mean; Q.var = m.var; }
    Q.cnt += n;
  }
  if (Q.pend_ret) {
    const CollectMerged m = collect_chan(Q.ret_mean, Q.ret_var, Q.ret_cnt, Q.r1, Q.r2, n);
    Q.ret_mean = m.mean; Q.ret_var = m.var; Q.ret_cnt += n;
  }
  if (!ok && lane == 0) atomicOr(CA.sync + CS_STATUS, 2u);
  return ok;
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
  if (lane == 0) CA.sync[CS_FOLDED_EPOCH] = CA.sync[CS_PART_EPOCH];
}

// The merge wave (block 2 n_chunks): same statistics as everybody, written back once every act wave has read the old ones; the
// action sampler's draw counter advances here too (the policy waves read it at their start).
__device__ __forceinline__ void collect_merge_wave(const CollectArgs& CA) {
  const int lane = threadIdx.x & 63, n_real = 2 * CA.n_chunks, PW = 2 * CA.S.D + 2;
  CollectStats Q;
  (void)collect_front_stats(CA, n_real, false, Q);
  bool ok = false;
  for (int it = 0; it < (1 << 21); ++it) {
    if (ld_flag(CA.sync + CS_READERS) >= (unsigned int)n_real) { ok = true; break; }
    __builtin_amdgcn_s_sleep(8);
  }
  if (!ok && lane == 0) atomicOr(CA.sync + CS_STATUS, 4u);
  collect_commit_stats(CA, Q);
  // the totals buffer of the NEXT collect launch back to "unpublished" (its last readers finished a launch ago)
  double* other = CA.tot + (size_t)((Q.cepoch + 1u) & 1u) * PW;
  for (int w = lane; w < PW; w += 64) other[w] = collect_sentinel();
  if (lane == 0) {
    if (CA.S.rng) CA.S.rng[1] += 1;
    CA.sync[CS_CEPOCH] = Q.cepoch + 1u;
    st_flag(CA.sync + CS_READERS, 0u);
  }
}

// One act wave (workgroup of 64 lanes).  Inlined into the collect kernels: CollectArgs is a by-value kernel argument, and
// handing its address to an out-of-line function made the compiler copy the whole struct to scratch in every wave (544 B).
// (tools/check_isa.py therefore tells MFMA accumulator registers from spill slots by the operand ranges of the MFMAs.)
__device__ __forceinline__ void collect_act_wave(const CollectArgs& CA, uint32_t epoch) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const ActArgs& A = CA.A;
  const int aw = (int)blockIdx.x, chunk = aw >> 1, net = aw & 1;
  if (aw == 2 * CA.n_chunks) { collect_merge_wave(CA); return; }
  if (chunk >= CA.n_chunks) return;                                  // padding waves
  const int KO = net == 0 ? 4 : 1;
  const int lane = threadIdx.x;
  const int D = A.D, Dp = (D + 1) & ~1, ldx = Dp + 1;
  const int row0 = chunk * kCRows;

  long long* tr = CA.trace ? CA.trace + (size_t)blockIdx.x * 8 : nullptr;
  if (tr && lane == 0) tr[0] = collect_now();
  float* p = lds;
  PpoNetLds W;
  W.W1 = p; p += Dp * kPH; W.b1 = p; p += kPH; W.W2 = p; p += kPH * kPLdh; W.b2 = p; p += kPH; W.Wo = p; p += kPH * 4; W.bo = p; p += 4;
  float* log_std = p; p += 4;
  float* X = p;  p += kCRows * ldx;
  float* X2 = p; p += kCRows * ldx;                                  // value wave: terminal observations of the previous step
  float* H1 = p; p += kCRows * kPLdh;
  float* H2 = p; p += kCRows * kPLdh;
  float* out = p; p += kCRows * 4;

  // ---- everything whose address is known at entry leaves first ----
  const int frow = row0 + (lane & 31);
  const bool fmine = net == 1 && A.prev_reward && lane < kCRows && frow < A.N;
  uint8_t f_term = 0, f_trunc = 0; double f_rew = 0.0;
  if (fmine) {
    f_term = A.prev_term[frow]; f_trunc = A.prev_trunc[frow];
    f_rew = A.raw_is_f64 ? reinterpret_cast<const double*>(A.prev_reward)[frow] : (double)reinterpret_cast<const float*>(A.prev_reward)[frow];
  }
  uint64_t rng_key = 0, rng_ctr = 0;
  if (net == 0 && !A.deterministic && lane < kCRows) { rng_key = A.rng[0]; rng_ctr = A.rng[1]; }
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
  // ---- weights of my network: all loads in flight together (registers), LDS writes after the inputs are built ----
  const int nP0 = ppo_net_params(Dp, 4);
  const int oW1 = net == 0 ? 0 : nP0, ob1 = oW1 + Dp * kPH, oW2 = ob1 + kPH, ob2 = oW2 + kPH * kPH, oWo = ob2 + kPH;
  const int oLs = nP0 + ppo_net_params(Dp, 1);
  const float* __restrict__ params = A.params;
  constexpr int kW1V = (64 * kPH / 4 + kWave - 1) / kWave, kW2V = kPH * kPH / 4 / kWave;       // float4 per lane: W1 (<= 16), W2 (16)
  float4 w1v[kW1V], w2v[kW2V];
  const float4* src1 = reinterpret_cast<const float4*>(params + oW1);
  const float4* src2 = reinterpret_cast<const float4*>(params + oW2);
#pragma unroll
  for (int j = 0; j < kW1V; ++j) { const int i = lane + j * kWave; w1v[j] = i < Dp * kPH / 4 ? src1[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
  for (int j = 0; j < kW2V; ++j) w2v[j] = src2[lane + j * kWave];
  const float wb1 = params[ob1 + lane], wb2 = params[ob2 + lane];
  float wov[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) { const int i = lane + j * kWave; wov[j] = i < kPH * KO + KO ? params[oWo + i] : 0.f; }
  const float wls = lane < 4 ? params[oLs + lane] : 0.f;
  // ---- statistics: old ones, my share of the pending fold, the totals ----
  CollectStats Q;
  (void)collect_front_stats(CA, aw, true, Q);
  const bool timeout = fmine && f_trunc && !f_term;
  const bool any_timeout = __ballot(timeout) != 0ull;                // wave-uniform: some episode of my rows was truncated
  if (tr && lane == 0) tr[1] = collect_now();                          // statistics of this step known
  // ---- inputs: normalise on load with the statistics of THIS step ----
  // (x - mean) * (1 / sqrt(var + eps)): one division per column and wave instead of one per element -- within an ulp of
  // double of VecNormalize's (x - mean) / sqrt(var + eps), i.e. the same float32 except once in ~1e9 elements
  double* cstd = reinterpret_cast<double*>(H2);                       // per-column 1 / sqrt(var + eps) and mean (H2 is written after X is built)
  double* cmean = cstd + 64;
  if (lane < D) { cstd[lane] = 1.0 / sqrt(Q.var + (double)A.eps); cmean[lane] = Q.mean; }
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
          x = fminf(fmaxf((float)((rawv[u] - cmean[d]) * cstd[d]), -A.clip), A.clip);
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
  // weights -> LDS
  {
    float4* dst = reinterpret_cast<float4*>(W.W1);
#pragma unroll
    for (int j = 0; j < kW1V; ++j) { const int i = lane + j * kWave; if (i < Dp * kPH / 4) dst[i] = w1v[j]; }
#pragma unroll
    for (int j = 0; j < kW2V; ++j) {
      const int i = lane + j * kWave;
      float* q = W.W2 + ((4 * i) >> 6) * kPLdh + ((4 * i) & 63);
      q[0] = w2v[j].x; q[1] = w2v[j].y; q[2] = w2v[j].z; q[3] = w2v[j].w;
    }
    W.b1[lane] = wb1; W.b2[lane] = wb2;
#pragma unroll
    for (int j = 0; j < 5; ++j) { const int i = lane + j * kWave; if (i < kPH * KO + KO) W.Wo[i] = wov[j]; }
    if (lane < 4) log_std[lane] = wls;
  }
  __syncthreads();
  if (tr && lane == 0) tr[2] = collect_now();                          // inputs and weights in LDS (value wave: flag_v published)

  act_forward_wave(W, X, H1, H2, out, KO, Dp, ldx);
  if (tr && lane == 0) tr[3] = collect_now();                          // forward done
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
    if (tr && lane == 0) tr[4] = collect_now();
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
        rn *= 1.0 / sqrt(Q.ret_var + (double)A.rew_eps);
        rn = rn > A.clip_reward ? A.clip_reward : (rn < -A.clip_reward ? -A.clip_reward : rn);
      }
      float o = (float)rn;
      if (timeout) o += A.gamma * out[lane * 4];                        // SB3: bootstrap truncated episodes with V(terminal_observation)
      A.rew_out[frow] = o;
      A.start_out[frow] = (f_term || f_trunc) ? 1.0f : 0.0f;
    }
  }
  if (tr && lane == 0) tr[4] = collect_now();
}

// Step wave: wait (bounded) until the act waves of the chunks covering rows [env0, env0 + rows) have published for `epoch`.
__device__ __forceinline__ void collect_wait_actions(const CollectArgs& CA, uint32_t epoch, int env0, int rows) {
  long long* tr = CA.trace ? CA.trace + (size_t)blockIdx.x * 8 : nullptr;
  if (tr && (threadIdx.x & 63) == 0) tr[1] = collect_now();              // state loaded, wait begins
  const int c0 = env0 / kCRows, c1 = min((env0 + rows - 1) / kCRows, CA.n_chunks - 1);
  const int lane = threadIdx.x & 63;
  const int c = c0 + (lane >> 1);
  const unsigned int* w = ((lane & 1) ? CA.flag_v : CA.flag_p) + (c <= c1 ? c : c1);
  bool ok = false;
  for (int it = 0; it < (1 << 21); ++it) {                             // ~ seconds: far beyond any healthy launch
    const bool mine = (c > c1) || ld_flag(w) == epoch;
    if (__ballot(!mine) == 0ull) { ok = true; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (!ok && lane == 0) atomicOr(CA.sync + CS_STATUS, 1u);
  if (tr && lane == 0) tr[2] = collect_now();                          // actions there
}

// Statistics tail of a step wave: its partial sums, left with plain stores for the next launch (or fw_collect_finish) to fold.
// `tile` = the wave's observation rows in LDS ([rows][ld], final: reset rows included), rew / done = this lane's env (leader
// lanes only), wg = workgroup index among the nblk step workgroups.
template <typename T>
__device__ __forceinline__ void collect_stats_tail(const CollectArgs& CA, uint32_t epoch, const T* tile, int ld, int rows, int wg, int nblk,
                                                   bool is_leader, double rew, bool done, int env) {
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
    if (S.update_ret) { rt = S.returns[env] * S.gamma + rew; S.returns[env] = done ? 0.0 : rt; }
    else if (done) S.returns[env] = 0.0;
  }
  // leaders' tracker values summed in row order (a fixed order)
  double r1 = 0.0, r2 = 0.0;
  {
    const unsigned long long lead = __ballot(is_leader);
    for (int l = 0; l < 64; ++l) {
      if (!((lead >> l) & 1ull)) continue;
      const double v = __shfl(rt, l, 64);
      r1 += v; r2 += v * v;
    }
  }
  const int grp = wg & (kCGroups - 1), mem = wg >> 3, mstride = (nblk + kCGroups - 1) / kCGroups;
  auto p1 = [&](int w) { return CA.part1 + ((size_t)w * kCGroups + grp) * mstride + mem; };
  if (lane < D) { *p1(lane) = s; *p1(D + lane) = s2; }
  if (lane == 0) { *p1(2 * D) = r1; *p1(2 * D + 1) = r2; if (wg == 0) CA.sync[CS_PART_EPOCH] = epoch; }
  if (tr && lane == 0) tr[4] = collect_now();
}

// fw_collect_finish: folds the partial sums the last fw_collect_step left and merges them into the caller's statistics (once
// per rollout).  One workgroup of 1024 lanes: wave k folds words k, k + 16, ... with collect_fold_word (the association of the
// in-grid fold), wave 0 merges.
__global__ __launch_bounds__(1024) void fw_collect_finish_kernel(CollectArgs CA) {
  __shared__ double s_tot[2 * kMaxObs + 2];
  const StatsArgs& S = CA.S;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, D = S.D, PW = 2 * D + 2;
  const bool pend = CA.sync[CS_PART_EPOCH] != CA.sync[CS_FOLDED_EPOCH];
  if (!pend) return;
  const int slots = kCGroups * ((CA.nblk + kCGroups - 1) / kCGroups);
  for (int w = wave; w < PW; w += 16) {
    const double v = collect_fold_word(CA.part1, w, slots);
    if (lane == 0) s_tot[w] = v;
  }
  __syncthreads();
  if (wave != 0) return;
  CollectStats Q;
  Q.pend_obs = S.update_obs != 0; Q.pend_ret = S.update_ret != 0;
  Q.mean = 0.0; Q.var = 1.0;
  if (lane < D) { Q.mean = S.mean[lane]; Q.var = S.var[lane]; }
  Q.cnt = S.count[0]; Q.ret_mean = S.ret_mean[0]; Q.ret_var = S.ret_var[0]; Q.ret_cnt = S.ret_count[0];
  const int dcol = lane < D ? lane : 0;
  Q.cs = s_tot[dcol]; Q.cs2 = s_tot[D + dcol]; Q.r1 = s_tot[2 * D]; Q.r2 = s_tot[2 * D + 1];
  const double n = (double)S.N;
  if (Q.pend_obs) {
    const CollectMerged m = collect_chan(Q.mean, Q.var, Q.cnt, Q.cs, Q.cs2, n);
    if (lane < D) { Q.mean = m.mean; Q.var = m.var; }
    Q.cnt += n;
  }
  if (Q.pend_ret) {
    const CollectMerged m = collect_chan(Q.ret_mean, Q.ret_var, Q.ret_cnt, Q.r1, Q.r2, n);
    Q.ret_mean = m.mean; Q.ret_var = m.var; Q.ret_cnt += n;
  }
  collect_commit_stats(CA, Q);
}

}  // namespace fwsim
