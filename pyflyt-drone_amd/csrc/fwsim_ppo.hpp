// Fused PPO minibatch updates (SB3 `PPO.train()` inner loop) for the reference's policy: separate
// 64-64 tanh MLPs for pi and V, diagonal Gaussian with a state-independent log_std, Adam.
//   reference: train/train_Fixedwing_Waypoints_v3.py:293-310 (PPO("MlpPolicy", ... batch_size=128,
//   n_epochs=20, clip_range=0.2, ent_coef, vf_coef=0.5, max_grad_norm=0.5)), train/train_objlock.py:255-290.
//
// Why a kernel: with thousands of envs one update is ~10^4 *sequential* minibatch steps of a 12 k-parameter
// network.  As framework ops that is ~30 launches per step and the update, not the simulator, bounds
// end-to-end throughput (4.5 s per 65 536-sample update).  Here up to four workgroups per network walk the whole
// minibatch sequence, each its share of every minibatch: weights stay in LDS for the entire call, activations of a 16- / 32- /
// 64-sample pass live in LDS, every GEMM (forward, dW = A^T G, dX = G W^T) is a tile routine on `v_mfma_f32_32x32x2_f32` /
// `v_mfma_f32_16x16x4_f32` (exact fp32, so parity with the torch path is fp32 rounding), weight-gradient tiles accumulate in
// registers across passes, and clipping + Adam are applied in registers to the elements a thread owns (no second kernel).
// The path is sequential by definition of SGD; the only parallelism is inside a minibatch, which is why it is eight CUs and not
// a grid.  The rows themselves are gathered by a parallel pre-pass (fw_ppo_pack_kernel) so that nothing on the sequential path
// depends on an index.
//
// Flat parameter layout (floats), used for params / exp_avg / exp_avg_sq alike (rollout.py builds it):
//   for net in (pi, vf):  W1[Dp][64]  b1[64]  W2[64][64]  b2[64]  Wo[64][KO]  bo[KO]     (KO = 4 / 1)
//   then log_std[4].       W*[in][out] = torch Linear.weight^T;  Dp = D rounded up to even (zero row).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fwsim {

struct PpoHyper {
  float lr, clip_range, ent_coef, vf_coef, max_grad_norm, beta1, beta2, eps;
  float adv_mean, adv_std;     // used when norm_adv == 2 (statistics over the whole, all-gathered rollout)
  int32_t norm_adv;            // 0 off, 1 per minibatch (SB3), 2 given statistics
  int32_t step0;               // Adam step count before this call
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPH = 64;            // hidden width
constexpr int kPChunk = 64;        // samples per pass through the networks (the 32-sample form of the kernel: template parameter CH)
constexpr int kPMaxSplit = 8;      // blocks per network a minibatch is split over, at most
constexpr int kPLdh = kPH + 1;     // row stride of the hidden activations / of W2 in LDS (odd: conflict-free column reads)
constexpr int kPLdx = 65;          // row stride of the gathered observations in LDS: a CONSTANT (the widest input + 1), so that every operand
                                   // address of the X-sided products is base + immediate.  (Round 3 had Dp + 1: with a run-time stride hipcc
                                   // emitted one address add, one scalar reload and one LDS wait PER MFMA of dW1 -- 5.3 k cycles for 32 MFMAs.)
constexpr int kPThreads = 256;

__host__ __device__ inline int ppo_net_params(int Dp, int KO) { return Dp * kPH + kPH + kPH * kPH + kPH + kPH * KO + KO; }
__host__ __device__ inline int ppo_total_params(int Dp) { return ppo_net_params(Dp, 4) + ppo_net_params(Dp, 1) + 4; }

// LDS image of one network
struct PpoNetLds { float *W1, *b1, *W2, *b2, *Wo, *bo; };
__host__ __device__ inline int ppo_net_lds_floats(int Dp, int KO) { return Dp * kPH + kPH + kPH * kPLdh + kPH + kPH * KO + KO; }

// C(32x32) += A(32xK) * B(Kx32); A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; K even.
// Operand layout of v_mfma_f32_32x32x2_f32: lane l supplies A(l % 32, l / 32) and B(l / 32, l % 32);
// accumulator register v of lane l is C((v / 4) * 8 + (l / 32) * 4 + v % 4, l % 32).
// The operands of STEPS consecutive MFMAs are fetched from LDS first and the MFMAs then issue back to back:
// one exposed LDS latency per batch instead of one per MFMA (the compiler does not pipeline the rolled loop).
template <int STEPS, typename FA, typename FB>
__device__ __forceinline__ f32x16 ppo_mfma_batch(FA&& fa, FB&& fb, int step0, f32x16 acc) {
  float av[STEPS], bv[STEPS];
#pragma unroll
  for (int i = 0; i < STEPS; ++i) { av[i] = fa(2 * (step0 + i)); bv[i] = fb(2 * (step0 + i)); }
#pragma unroll
  for (int i = 0; i < STEPS; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
  return acc;
}
// fa(k0) / fb(k0): this lane's A / B operand of the MFMA covering k0, k0 + 1
template <typename FA, typename FB>
__device__ __forceinline__ f32x16 ppo_mfma_k(FA&& fa, FB&& fb, int K, f32x16 acc) {
  const int steps = K >> 1;
  int s = 0;
  for (; s + 16 <= steps; s += 16) acc = ppo_mfma_batch<16>(fa, fb, s, acc);
  if (s + 8 <= steps) { acc = ppo_mfma_batch<8>(fa, fb, s, acc); s += 8; }
  if (s + 4 <= steps) { acc = ppo_mfma_batch<4>(fa, fb, s, acc); s += 4; }
  if (s + 2 <= steps) { acc = ppo_mfma_batch<2>(fa, fb, s, acc); s += 2; }
  if (s < steps) acc = ppo_mfma_batch<1>(fa, fb, s, acc);
  return acc;
}
__device__ __forceinline__ f32x16 ppo_mfma_tile(const float* A, int sam, int sak, const float* B, int sbk, int sbn, int K, f32x16 acc) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const float* a = A + r * sam + h * sak;
  const float* b = B + h * sbk + r * sbn;
  return ppo_mfma_k([&](int k0) { return a[k0 * sak]; }, [&](int k0) { return b[k0 * sbk]; }, K, acc);
}
__device__ __forceinline__ int ppo_acc_row(int v) { return (v >> 2) * 8 + ((threadIdx.x & 63) >> 5) * 4 + (v & 3); }

// The 32-sample form: a 32 x 64 product is 2 x 4 tiles of 16 x 16 on v_mfma_f32_16x16x4_f32 (the same MAC rate as 32x32x2); wave w
// owns output columns 16 w .. 16 w + 15 for both row tiles (one B operand feeds two MFMAs, two independent accumulator chains hide
// the 40-cycle dependent latency behind the 32-cycle issue).  Operand layout: lane l supplies A(l % 16, l / 16) and B(l / 16, l % 16);
// accumulator register v of lane l is C(4 (l / 16) + v, l % 16).  Which k a lane group supplies is free as long as A and B agree:
// group g = l / 16 takes k = koff(g) + step with koff = 16 g (K = 64) or {0, 16, 8, 24}[g] (K = 32), so that the two groups of a
// half-wave are 16 floats apart and, with the odd row strides of the LDS images, every operand read is bank-conflict free both
// along rows (activations as A, W as B) and along columns (W2^T as B).
// c[rt] += A(rows 16 rt .. 16 rt + 15, K) * B(K, 16 columns), rt < RT (2: 32-sample passes, 1: 16-sample passes -- one accumulator
// chain, the MFMAs then issue every 40 cycles instead of every 32); A(m, k) = A[m * sam + k * sak], B(k, n) = B[k * sbk + n * sbn]; K = 4 STEPS.
template <int STEPS, int RT>
__device__ __forceinline__ void ppo_mfma16_rows(const float* A, int sam, int sak, const float* B, int sbk, int sbn, f32x4 (&c)[RT]) {
  static_assert(STEPS == 8 || STEPS == 16, "K = 32 or 64");
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int koff = STEPS == 16 ? 16 * g : ((g & 1) * 16 + (g >> 1) * 8);
  const float* a = A + i * sam + koff * sak;
  const float* b = B + koff * sbk + i * sbn;
  float av[RT][STEPS], bv[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) av[rt][s] = a[rt * 16 * sam + s * sak];
    bv[s] = b[s * sbk];
  }
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) c[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][s], bv[s], c[rt], 0, 0, 0);
  }
}

// Cross-lane sums on the vector ALU (DPP inside a row of 16 lanes, v_permlane16_swap / v_permlane32_swap across the rows) instead
// of __shfl_xor, which is ds_bpermute_b32: an LDS round trip (~120 cycles) per butterfly step on a path where every step is exposed.
template <int CTRL> __device__ __forceinline__ float ppo_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E;     // quad_perm [1,0,3,2] / [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;               // lane i <-> 7 - i of each 8: "the other quad" for values that are uniform inside a quad
constexpr int kDppRowMirror = 0x140;                // lane i <-> 15 - i of each row: "the other 8" for values that are uniform inside an 8
constexpr int kDppRor4 = 0x124, kDppRor8 = 0x128;   // rotate inside the row of 16
// every lane l gets x[l % 16] + x[l % 16 + 16] + x[l % 16 + 32] + x[l % 16 + 48], summed in the same order everywhere
__device__ __forceinline__ float ppo_sum_rows(float x) {
  typedef unsigned ppo_u2 __attribute__((ext_vector_type(2)));
  const ppo_u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);      // {rows 0 0 2 2, rows 1 1 3 3}
  const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const ppo_u2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);      // {lower half twice, upper half twice}
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float ppo_wave_sum(float x) {
  x += ppo_dpp<kDppXor1>(x); x += ppo_dpp<kDppXor2>(x); x += ppo_dpp<kDppHalfMirror>(x); x += ppo_dpp<kDppRowMirror>(x);
  return ppo_sum_rows(x);
}
// (no barrier in front: for callers whose previous readers of red[0..3] are already behind a later barrier)
__device__ __forceinline__ float ppo_block_sum_nb(float x, float* red) {
  x = ppo_wave_sum(x);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float ppo_block_sum(float x, float* red) {
  // 256 threads -> all threads get the sum (red: 8 floats of LDS)
  x = ppo_wave_sum(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// tanh to ~1e-7 absolute: odd series near 0 (no cancellation), 1 - 2 / (e^{2x} + 1) elsewhere (v_exp_f32 + v_rcp_f32; e^{2x} -> 0 / inf
// give -1 / +1 without a clamp).  The pair form is the same arithmetic on packed fp32 (v_pk_mul / v_pk_fma: two values per
// instruction for everything but the exponential, the reciprocal and the select) -- the epilogues of the update spend ~80 cycles
// per value on this function; collector and update must share it (the ratio of a fresh sample is exactly 1).
typedef float ppo_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ppo_f2 ppo_tanh2(ppo_f2 x) {
  const ppo_f2 x2 = x * x;
  ppo_f2 p = __builtin_elementwise_fma(x2, ppo_f2{0.021869488f, 0.021869488f}, ppo_f2{-0.053968254f, -0.053968254f});
  p = __builtin_elementwise_fma(x2, p, ppo_f2{0.13333334f, 0.13333334f});
  p = __builtin_elementwise_fma(x2, p, ppo_f2{-0.33333334f, -0.33333334f});
  p = __builtin_elementwise_fma(x2, p, ppo_f2{1.0f, 1.0f});
  const ppo_f2 small = x * p;
  const ppo_f2 t = x * 2.8853900817779268f;
  const ppo_f2 e1 = ppo_f2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;      // e^{2x} + 1
  const ppo_f2 rc = {__builtin_amdgcn_rcpf(e1[0]), __builtin_amdgcn_rcpf(e1[1])};
  const ppo_f2 big = __builtin_elementwise_fma(rc, ppo_f2{-2.0f, -2.0f}, ppo_f2{1.0f, 1.0f});
  return ppo_f2{fabsf(x[0]) < 0.25f ? small[0] : big[0], fabsf(x[1]) < 0.25f ? small[1] : big[1]};
}
__device__ __forceinline__ float ppo_tanh(float x) {
  const float x2 = x * x;
  float p = fmaf(x2, 0.021869488f, -0.053968254f);
  p = fmaf(x2, p, 0.13333334f);
  p = fmaf(x2, p, -0.33333334f);
  p = fmaf(x2, p, 1.0f);
  const float small = x * p;
  const float big = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.8853900817779268f) + 1.0f), -2.0f, 1.0f);
  return fabsf(x) < 0.25f ? small : big;
}

// Adam moments are kept in "slot" order: slot = ((net * 3 + kind) * 4 + wave) * 1024 + lane * 16 + v for the
// accumulator tiles (kind 0 = W2, 1 = W1; kind 2 is unused since round 3), then kPTileSlots + q * 256 + thread for the
// per-thread elements (q = 3 net + {0: b1, 1: b2, 2: bo}, q = 6: log_std, q = 7 + net: Wo[thread / 4][thread % 4]).
// Slots nobody owns are padding.
constexpr int kPTileSlots = 2 * 3 * 4 * 64 * 16;
constexpr int kPMomentSlots = kPTileSlots + 9 * kPThreads;
__host__ __device__ inline int ppo_tile_slot(int net, int kind, int wave, int lane) { return (((net * 3 + kind) * 4 + wave) * 64 + lane) * 16; }
// A block's gradient partial in the exchange buffer: [kind W2 | W1][wave][quarter q of the lane's 16 elements][lane][4] for the tiles -- one wave-level
// 16-byte access is 1 KB in a row (the partners read it past their L1: every access is a request to the L2, and in moment-slot order, 64 bytes
// per lane, each request would touch 64 lines for 16 bytes apiece) -- then the five per-thread elements [5][256].
// Per-thread elements behind the tiles: [t < 64][gb1, gb2, gbo, gls] as one float4 per thread of the first wave, then gwo[256].
constexpr int kPGxTile = 2 * 4 * 4 * 64 * 4;
// ... and then the block's share of the UPDATED WEIGHTS in the reduce-scatter form, for the others to fetch ([set][wave][lane][4]; of the
// four regions [parity][net] a block has, the one of parity 0 carries it).  In the block's own region, ~43 KB from the next block's:
// kept as one dense [net][part] array of 8-KB shares (round 5, first version) every block of the call fetched its 7 shares from the
// same 64 KB and the fetch of sixteen blocks took 3.5 k cycles where that of eight takes 1.4 k -- whatever serves those addresses
// (L2 channel, fabric) is selected at a granularity that coarse; the gradient partials, a region apart, showed the same once all
// blocks walked them in the same order.
constexpr int kPWxShare = 3 * kPThreads * 4;      // (three tagged granules per thread in the self-announcing form, two plain float4s otherwise)
constexpr int kPGxSlots = kPGxTile + 2 * kPThreads + kPWxShare;
__host__ __device__ inline int ppo_gx_tile(int kind, int wave, int lane) { return (kind * 4 + wave) * 4 * 256 + lane * 4; }      // + q * 256

// flat parameter index of every moment slot (-1 = padding); host side of the layout above
inline void ppo_moment_map(int D, int32_t* flat_of_slot) {
  const int Dp = (D + 1) & ~1;
  const int tilesW1 = ((Dp + 31) / 32) * 2;
  for (int i = 0; i < kPMomentSlots; ++i) flat_of_slot[i] = -1;
  int off = 0, oLs = 0;
  for (int n = 0; n < 2; ++n) {
    const int KO = n == 0 ? 4 : 1;
    const int oW1 = off, ob1 = oW1 + Dp * kPH, oW2 = ob1 + kPH, ob2 = oW2 + kPH * kPH, oWo = ob2 + kPH, obo = oWo + kPH * KO;
    off = obo + KO;
    for (int wave = 0; wave < 4; ++wave)
      for (int lane = 0; lane < 64; ++lane)
        for (int v = 0; v < 16; ++v) {
          const int r = lane & 31, hh = lane >> 5, mt = wave >> 1, nt = wave & 1;
          const int row = (v >> 2) * 8 + hh * 4 + (v & 3);
          const int i = mt * 32 + row, j = nt * 32 + r;
          flat_of_slot[ppo_tile_slot(n, 0, wave, lane) + v] = oW2 + i * kPH + j;
          if (wave < tilesW1 && i < D) flat_of_slot[ppo_tile_slot(n, 1, wave, lane) + v] = oW1 + i * kPH + j;
        }
    for (int t = 0; t < kPH; ++t) { flat_of_slot[kPTileSlots + (n * 3 + 0) * kPThreads + t] = ob1 + t; flat_of_slot[kPTileSlots + (n * 3 + 1) * kPThreads + t] = ob2 + t; }
    for (int t = 0; t < KO; ++t) flat_of_slot[kPTileSlots + (n * 3 + 2) * kPThreads + t] = obo + t;
    for (int t = 0; t < kPThreads; ++t) if ((t & 3) < KO) flat_of_slot[kPTileSlots + (7 + n) * kPThreads + t] = oWo + (t >> 2) * KO + (t & 3);
    oLs = off;
  }
  for (int t = 0; t < 4; ++t) flat_of_slot[kPTileSlots + 6 * kPThreads + t] = oLs + t;
}

// Pre-pass of fw_ppo_update, off the sequential path: one workgroup per minibatch PACKS the rows the update will walk, in the
// order it will walk them, into one contiguous array -- row = [obs, zero-padded to a multiple of 4 | action 4 | old log-prob,
// advantage (normalised), return, 0] -- so that the sequential kernel fetches a 64-sample chunk as one contiguous block with
// three or four coalesced 16-byte loads per thread and no index, no scattered scalar and no division on its path.  (Round 3
// gathered inside the sequential kernel: per chunk and thread a dependent index load, 7-16 scattered dwords of the observation row
// and three scattered scalars, ~40 wave-level loads of 16 cache lines each through the CU's one address path: 2.2-2.8 k
// cycles per chunk just to issue them.)  It also does what fw_ppo_adv_stats_kernel did: the minibatch's advantage statistics
// (SB3: advantages = (a - a.mean()) / (a.std() + 1e-8) with the unbiased std).
__host__ __device__ inline int ppo_pack_width(int D) { return ((D + 3) & ~3) + 8; }       // floats per packed row (a multiple of 4)
struct PpoPackArgs {
  const float *obs, *act, *old_logp, *adv, *ret;
  const int32_t* perm;
  int32_t B, D, norm_adv;            // norm_adv as PpoHyper
  float adv_mean, adv_std;
  float* out;                        // [n_mb][B][ppo_pack_width(D)]
};
__global__ __launch_bounds__(256) void fw_ppo_pack_kernel(PpoPackArgs P) {
  __shared__ float st[2];
  const int32_t* idx = P.perm + (size_t)blockIdx.x * P.B;
  const int t = threadIdx.x, B = P.B, D = P.D;
  if (t < 64) {                                      // (the arithmetic of round 3's statistics kernel: one wave, strided sums)
    float mean = P.adv_mean, sd = P.adv_std;
    if (P.norm_adv == 1) {
      float s1 = 0.f;
      for (int i = t; i < B; i += 64) s1 += P.adv[idx[i]];
      mean = ppo_wave_sum(s1) / (float)B;
      float s2 = 0.f;
      for (int i = t; i < B; i += 64) { const float d = P.adv[idx[i]] - mean; s2 += d * d; }
      sd = sqrtf(ppo_wave_sum(s2) / (float)(B > 1 ? B - 1 : 1));
    }
    if (t == 0) { st[0] = mean; st[1] = sd; }
  }
  __syncthreads();
  const float mean = st[0], sd = st[1];
  const int Dv4 = (D + 3) >> 2, W4 = Dv4 + 2;      // float4s per row: observation, then action, then scalars
  float4* out = reinterpret_cast<float4*>(P.out) + (size_t)blockIdx.x * B * W4;
  const bool vec = (D & 3) == 0;
  for (int e = t; e < B * W4; e += 256) {
    const int row = e / W4, q = e - row * W4, si = idx[row];
    float4 v;
    if (q < Dv4) {
      const float* o = P.obs + (size_t)si * D + 4 * q;
      if (vec) v = *reinterpret_cast<const float4*>(o);
      else { const int left = D - 4 * q; v = make_float4(o[0], left > 1 ? o[1] : 0.f, left > 2 ? o[2] : 0.f, left > 3 ? o[3] : 0.f); }
    } else if (q == Dv4) {
      v = *reinterpret_cast<const float4*>(P.act + (size_t)si * 4);
    } else {
      float a = P.adv[si];
      if (P.norm_adv != 0) a = (a - mean) / (sd + 1e-8f);
      v = make_float4(P.old_logp[si], a, P.ret[si], 0.f);
    }
    out[e] = v;
  }
}

// The two networks (pi / V) share nothing but the scalar gradient norm that SB3 clips jointly, exchanged once per minibatch
// through 64-bit words (tag | partial sum of squares).  A minibatch is cut over 1, 2 or 4 workgroups per network (ppo_split),
// block q running passes q, q + nsplit, ...; per minibatch they exchange gradients through global memory (release / acquire at
// device scope -- or, on a shared XCD, store wait / loads past the L1 -- double-buffered by minibatch parity).  Two blocks: each
// reads the other's whole partial, both hold the same sum and apply the same Adam step to their own LDS copy of the weights and
// their own register copy of the moments.  Four blocks (RS): reduce-scatter of the tiles, Adam on a quarter, all-gather of the
// new weights (see ppo_net_body).  2, 4 or 8 working blocks, always co-resident; every wait is bounded and a wait that runs out
// ends the call with a status word instead of a result (ppo_wait, include/fwsim.h FW_PPO_ST_*).
// Work split of the 256 threads of a block (4 waves, one per SIMD, fixed for the whole call), 64-sample passes:
//   * every 64 x 64 product (H1, H2, G2, G1, dW2) is 2 x 2 tiles of 32 x 32: wave w owns tile (w >> 1, w & 1) -- 32- / 16-sample
//     passes: H1, H2, G2, G1 are 2 / 1 x 4 tiles of 16 x 16, wave w owns columns 16 w ..; dW2 / dW1 stay 32 x 32 tiles per wave;
//   * dW1 is ceil(Dp / 32) x 2 tiles: all four waves own one (> 32 features), or waves 0-1 own them and every wave computes
//     one over half of the chunk's samples (<= 32 features; the halves meet in LDS once per minibatch);
//   * the 64 x KO head, its loss gradient and dWo run on the vector ALU: thread (sample or hidden unit, quarter / component);
//   * bias gradients are column sums: thread (wave q, lane n) sums rows 16 q .. 16 q + 15 of column n.
// Each lane applies clipping + Adam to the accumulator elements it holds; their moments are loaded once (slot order:
// ppo_moment_map), live in registers (AGPRs) for the whole call and are written back by the first block at the end (RS: every
// block the quarter it owns).
// Exchange words between blocks.  `l2` = they run on one XCD: the word is written through this CU's L1 into the shared L2 (a
// workgroup-scope store) instead of with a device-scope store.  Polls are device-scope loads (past the L1, served by the L2) in
// both cases.  (Rounds 3 / 4 polled with an atomic OR of zero on the shared-L2 path: with up to seven lanes of eight blocks
// polling words of one line, the atomics queued behind each other in the L2 -- 0.87 k -> 0.40 k cycles per flag poll, 11.1 ->
// 10.3 us per minibatch with plain loads.)
__device__ __forceinline__ void ppo_word_store(unsigned long long* p, unsigned long long v, bool l2) {
  if (l2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ppo_word_load(unsigned long long* p, bool l2) {
  (void)l2;
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct PpoArgs {
  float *params, *mom_m, *mom_v;
  const float* packed;               // [n_mb][B][ppo_pack_width(D)] rows in walking order (fw_ppo_pack_kernel)
  int32_t n_mb, B, D;
  PpoHyper H;
  float* loss_acc;                   // [3] += policy, value, entropy loss
  unsigned long long* xch;           // [kPpoWords] exchange words (norm partials [parity][net][part], gradient flags + kPpoWordFlags, XCD ids of
                                     //      the blocks [net][part] + kPpoWordIds, kPpoWordPaths, kPpoWordStatus), zeroed by the host
  float* gx;                         // [2 parities][2 nets][kPMaxSplit parts][kPGxSlots] gradient partials of the blocks of a network
  long long spin;                    // polls a wait for another block may take (kPpoSpin; FWSIM_SPIN_LOG2 shrinks it: tests provoke the timeout)
  int32_t flags;                     // PPO_FLAG_*
};
constexpr long long kPpoSpin = 1ll << 26;
enum { PPO_FLAG_NO_L2_SWAP = 1,      // FWSIM_PPO_NO_L2_SWAP=1: every exchange through device-scope accesses, as if no two blocks shared an XCD
       PPO_FLAG_WRITER_LAST = 2 };   // FWSIM_PPO_WRITER=last: the LAST block of each network writes the result back instead of the first (tests: every
                                     // block of a network must end the call with the same bits)
// xch[kPpoWordPaths]: which exchanges of this call went through a shared L2 -- bit 2 b: block b's gradient swap, bit 2 b + 1: its
// norm exchange (b = 2 part + net).  xch[kPpoWordStatus]: 0, or PPO_ST_* of the waits that ran out: the blocks then leave
// without writing the parameters back (the moments in memory are part-way through the call: the caller must not go on with them).
constexpr int kPpoWordFlags = 4 * kPMaxSplit, kPpoWordIds = 8 * kPMaxSplit, kPpoWordPaths = 10 * kPMaxSplit, kPpoWordStatus = kPpoWordPaths + 1;
constexpr int kPpoWordLoss = kPpoWordStatus + 1;     // [net][part]: the blocks' loss sums (float bits), added up in a fixed order by the block that finishes last
constexpr int kPpoWordDone = kPpoWordLoss + 2 * kPMaxSplit;      // how many blocks have finished
constexpr int kPpoWordArrive = kPpoWordDone + 1;     // how many blocks got through their last minibatch with every wait answered
constexpr int kPpoWordVerdict = kPpoWordArrive + 1;  // written once, by the last arriver: 1 = every block writes its results back, 2 = nobody does
constexpr int kPpoWordFlags2 = 13 * kPMaxSplit;      // flags of the weight all-gather [parity][net][part] (reduce-scatter form)
constexpr int kPpoWords = kPpoWordFlags2 + 4 * kPMaxSplit;
static_assert(kPpoWordIds + 2 * kPMaxSplit <= kPpoWordPaths && kPpoWordVerdict < kPpoWordFlags2, "exchange-word layout");
enum { PPO_ST_IDS = 1, PPO_ST_SWAP = 2, PPO_ST_NORM = 4, PPO_ST_COMMIT = 8 };

// (Measured, round 5: an `s_sleep 1` between two looks -- to take pressure off the lines sixteen blocks spin on -- costs 0.1-0.2 us per
// minibatch: the reaction time of the waiter is what counts.)
// Bounded wait of one thread for a word another block publishes.  `done(word)` ends it; every 256 polls it also looks at the
// status word, so that one block giving up releases the others at once instead of after their own budgets.  Returns false --
// and raises `bit` in the status word -- when the budget ran out or somebody else had given up.
template <typename LOAD, typename DONE>
__device__ __forceinline__ bool ppo_wait(const PpoArgs& A, LOAD&& load, DONE&& done, unsigned long long bit, unsigned long long& w) {
  unsigned long long* status = A.xch + kPpoWordStatus;
  for (long long spins = 0; spins < A.spin; ++spins) {
    w = load();
    if (done(w)) return true;
    if ((spins & 255) == 255 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
  }
  (void)__hip_atomic_fetch_or(status, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// CH = samples per pass through the network (64: 2 x 2 tiles of 32 x 32 per product; 32 / 16: 2 / 1 x 4 tiles of 16 x 16, see ppo_mfma16_rows);
// part / nsplit: this block's place among the blocks of its network (chunk c of a minibatch is run by block c % nsplit).
// NS = 0: the (1, 2 or 4) blocks of a network swap whole partials all to all.  NS = 4 / 8 (then nsplit == NS): the gradient swap is a
// REDUCE-SCATTER -- block q fetches, of every block's partial, only 1 / NS of the tile elements, sums them, takes its share of the
// clipping norm from them and applies Adam to that share alone -- followed by an ALL-GATHER of the updated WEIGHTS.  A block then pulls
// ~one partial + the weights past its L1 per minibatch, whatever NS is, instead of NS - 1 partials, and does 1 / NS of the tile Adam,
// for one more latency round.  What a block owns are the tiles of wave tq = q * 4 / NS (W2's and W1's), and of them
//   four blocks:  everything -- thread (wave w, lane l) owns elements 4 w .. 4 w + 3 of lane l of BOTH tiles (two "sets");
//   eight blocks: half -- quarters 2 (q & 1), 2 (q & 1) + 1 of the lanes' 16 elements; waves 0, 1 take those quarters of the W2 tile, waves
//                 2, 3 the same quarters of the W1 tile (one set per thread).
// Either way a thread's set is 4 consecutive floats of a lane: every exchange access is 16 bytes per lane.  That is what counts:
// tools/microbench_sc1.hip -- the CU's address unit takes 16 cycles per wave-level 8- or 16-byte load (4 for a 4-byte one), all four
// waves in turn, 220 + 64 N cycles for N such loads per wave -- so the cost of an exchange is its INSTRUCTION count, not its lines.
// Eight blocks (round 5): a 128-sample minibatch is 8 x 16 samples -- the chunk pass, which scales with the samples of a block, falls
// from 11.4 k to 7.3 k cycles.
// A pointer every lane holds the same value of, as the compiler can SEE it (two v_readfirstlane).  What it is for: hipcc's divergence
// analysis takes the minibatch counter of fw_ppo_update's loop -- whose exits hang on values read from LDS -- for divergent, and with it
// every address derived from it; a buffer resource built from such a pointer lives in VECTOR registers, and every buffer load through
// it is wrapped in a waterfall loop (readfirstlane x 4, two compares, exec mask, load, branch): ~80 cycles of issue per load, 24 loads
// in a row in the gradient fetch -- half of the 2.9 k cycles that fetch took with eight blocks, where the same fetch in isolation
// takes 0.7 k (tools/microbench_rs.hip).  Round 4's "a CU pulls lines past its L1 at ~20 B / clk" was this.
template <typename P> __device__ __forceinline__ P* ppo_uniform_ptr(P* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (P*)(((unsigned long long)hi << 32) | lo);
}
// (Measured and dropped, round 5: the block-uniform flags read back from LDS -- same_xcd, the give-up flag the loop exits hang on --
// passed through v_readfirstlane so that the control flow on them is scalar: 7.72 -> 7.80 us per (28, 128) minibatch, 7.70 -> 7.83 per
// (56, 64): more values in scalar registers, of which the kernel already spills 290.)
// four floats from a raw buffer, past the L1 (sc1)
__device__ __forceinline__ void ppo_ld_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off, float (&out)[4]) {
  typedef unsigned int ppo_u4 __attribute__((ext_vector_type(4)));
  const ppo_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);
#pragma unroll
  for (int e = 0; e < 4; ++e) out[e] = __uint_as_float(a[e]);
}
// A wave's 32 x 32 gradient tile to the exchange buffer (dst = the tile's base): quarter q of a lane's 16 elements at [q][lane][4]
__device__ __forceinline__ void ppo_store_tile(float* dst, int lane, const f32x16& g) {
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(dst + q * 256 + lane * 4) = make_float4(g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]);
}
// sum of v[0 .. N) as a balanced tree over the index order: the same bits in every block, whoever's own partial sits where
template <int N> __device__ __forceinline__ float ppo_tree_sum(const float (&v)[N]) {
  static_assert(N == 2 || N == 4 || N == 8, "power of two");
  if constexpr (N == 2) return v[0] + v[1];
  else if constexpr (N == 4) return (v[0] + v[1]) + (v[2] + v[3]);
  else return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}
template <int NET, int CH, int NS>
__device__ __forceinline__ void ppo_net_body(const PpoArgs& A, float* lds, const int part, const int nsplit) {
  static_assert(NS == 0 || NS == 4 || NS == 8, "blocks per network in the reduce-scatter form");
  constexpr bool RS = NS != 0;
  constexpr int NSd = RS ? NS : 4;                     // (array extents; the divisor where NS may be 0)
  constexpr int NSETS = NS == 8 ? 1 : 2;               // RS: sets of 4 tile elements a thread owns (see above)
  constexpr bool TAGGED = NS == 4;                     // the weight all-gather as self-announcing granules (see there): pays with four blocks
                                                       // (7.90 -> 7.72 us per (56, 64) minibatch), not with eight, whose rounds are 14 fetches
                                                       // per wave from seven partners that sixteen blocks poll at once (7.88 -> 8.25 us)
  static_assert(CH == 64 || CH == 32 || CH == 16, "chunk size");
  constexpr int RT = CH == 64 ? 1 : CH / 16;     // row tiles of 16 in the 16 x 16 forms
  constexpr int n = NET, KO = NET == 0 ? 4 : 1;
  float* __restrict__ params = A.params;
  float* __restrict__ mom_m = A.mom_m; float* __restrict__ mom_v = A.mom_v;
  const int n_mb = A.n_mb, B = A.B, D = A.D;
  const PpoHyper H = A.H;
  const int t = threadIdx.x, lane = t & 63, r = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // (a scalar: everything decided per wave -- tile ownership above all -- branches, not exec masks)
  const int Dp = (D + 1) & ~1;
  constexpr int ldx = kPLdx;
  // W1 in LDS.  CH = 64: [Dp][64] as in memory.  CH = 32: the 16x16x4 products walk K in steps of 32 or 64 (zero rows behind the last
  // feature) and want the odd row stride (ppo_mfma16_rows).
  constexpr int ldw1 = CH == 64 ? kPH : kPLdh;
  const int K1 = CH == 64 ? Dp : (Dp <= 32 ? 32 : 64);

  // ---- LDS carve-up ----
  float* p = lds;
  PpoNetLds W;
  W.W1 = p; p += K1 * ldw1; W.b1 = p; p += kPH; W.W2 = p; p += kPH * kPLdh; W.b2 = p; p += kPH; W.Wo = p; p += kPH * KO; W.bo = p; p += KO;
  float* log_std = p; p += 4;
  float* X = p;  p += CH * ldx + 64;              // (+64: the padded dW1 tile reads a few floats past the last row)
  float* H1 = p; p += CH * kPLdh;
  float* H2 = p; p += CH * kPLdh > 2112 ? CH * kPLdh : 2112;      // (>= 2048 floats: dW1's split partials pass through it)
  float* gout = p; p += CH * 4;                   // head output, then dL/d(head output) of the chunk
  float* sA = p; p += CH * 4;                     // gathered actions
  float* sS = p; p += CH * 4;                     // per-sample scalars: old_logp, adv (normalised), ret, -
  float* bred = p; p += 2 * 4 * kPH;              // bias-gradient partials [b1 | b2][wave][64]
  float* red = p; p += 8;
  float* sred = p; p += 32;                       // per wave: the four components of dbo and of dlog_std over its samples
  float* sink = p; p += kPThreads;                // one word per thread: where the Adam of a W1 tile "updates" the rows the network does not have
  float* red8 = p; p += 2 * kPMaxSplit;           // (RS) the blocks' shares of the squared gradient norm, [net][part]

  // flat offsets of this net
  const int nP0 = ppo_net_params(Dp, 4);
  const int oW1 = n == 0 ? 0 : nP0, ob1 = oW1 + Dp * kPH, oW2 = ob1 + kPH, ob2 = oW2 + kPH * kPH, oWo = ob2 + kPH;
  const int oLs = nP0 + ppo_net_params(Dp, 1);

  // ---- load the weights once ----
  for (int i = t; i < K1 * ldw1; i += kPThreads) W.W1[i] = 0.f;
  __syncthreads();
  for (int i = t; i < Dp * kPH; i += kPThreads) W.W1[(i >> 6) * ldw1 + (i & 63)] = params[oW1 + i];
  for (int i = t; i < kPH; i += kPThreads) { W.b1[i] = params[ob1 + i]; W.b2[i] = params[ob2 + i]; }
  for (int i = t; i < kPH * kPH; i += kPThreads) W.W2[(i >> 6) * kPLdh + (i & 63)] = params[oW2 + i];
  for (int i = t; i < kPH * KO + KO; i += kPThreads) W.Wo[i] = params[oWo + i];              // Wo and bo are contiguous in both images
  if (t < 4) log_std[t] = params[oLs + t];
  for (int i = t; i < CH * ldx + 64; i += kPThreads) X[i] = 0.f;                                // incl. the pad the dW1 tiles read
  __syncthreads();

  // The blocks of a network swap their gradient partials once per minibatch.  If all of them run on the same XCD
  // they share an L2: the swap then needs no device-scope release / acquire (a write-back and an invalidate of the WHOLE L2,
  // ~12 k cycles per minibatch with the misses that follow) -- the vector L1 writes through, so the producer only waits for
  // its stores and the consumer only reads past its own L1.  The launch puts them there (blocks b and b + 8), but nothing
  // promises that mapping: each block reads the XCD it really runs on and they compare notes once per call; blocks that
  // were split keep the device-scope fences.
  bool same_xcd = false, same_xcd_net = false;      // ... as every other block of my network; as the other network's block of my part
  bool dead = false;                                // a wait for another block ran out (block-uniform): leave, touching nothing more
  {
    if (t == 0) {
      red[7] = 0.f;
      const unsigned my_xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));      // HW_REG_XCC_ID[3:0]
      __hip_atomic_store(A.xch + kPpoWordIds + NET * kPMaxSplit + part, (unsigned long long)(my_xcc + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      auto wait_id = [&](unsigned long long* p_) {
        unsigned long long w = 0;
        if (!ppo_wait(A, [&]() { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
                      [](unsigned long long x) { return x != 0ull; }, (unsigned long long)PPO_ST_IDS, w)) { red[7] = 1.f; return 0u; }
        return (unsigned)w;
      };
      const bool l2ok = !(A.flags & PPO_FLAG_NO_L2_SWAP);
      bool sx = l2ok && nsplit > 1;
      for (int q = 0; q < nsplit; ++q) if (q != part && wait_id(A.xch + kPpoWordIds + NET * kPMaxSplit + q) != my_xcc + 1u) sx = false;
      bool sn = wait_id(A.xch + kPpoWordIds + (1 - NET) * kPMaxSplit + part) == my_xcc + 1u && l2ok;
      if (RS) for (int q = 0; q < nsplit; ++q) if (q != part && wait_id(A.xch + kPpoWordIds + (1 - NET) * kPMaxSplit + q) != my_xcc + 1u) sn = false;      // (the norm gathers all eight words)
      red[5] = sx ? 1.f : 0.f; red[6] = sn ? 1.f : 0.f;
      if (sx || sn) (void)__hip_atomic_fetch_or(A.xch + kPpoWordPaths, (unsigned long long)((sx ? 1u : 0u) | (sn ? 2u : 0u)) << (2 * (2 * part + NET)),
                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    same_xcd = red[5] != 0.f; same_xcd_net = red[6] != 0.f; dead = red[7] != 0.f;
  }
  const int mt = wave >> 1, nt = wave & 1;
  const int tilesW1 = ((Dp + 31) / 32) * 2;       // 2 or 4 tiles of dW1
  const bool hasW1 = wave < tilesW1;              // this wave OWNS a tile of W1 (hand-off, norm, Adam)
  // <= 32 observation features: dW1 is the two tiles (0, 0), (0, 1).  Instead of two waves running 32 MFMAs each while the other
  // two idle, every wave takes one tile over HALF of the chunk's samples (16 MFMAs); waves 2, 3 hand their partial to the
  // owners through LDS once per minibatch (gW1 of a non-owner is scratch).
  const bool splitW1 = tilesW1 == 2;
  const int w1_mt = splitW1 ? 0 : mt, w1_k0 = splitW1 ? (wave >> 1) * (CH / 2) : 0, w1_K = splitW1 ? CH / 2 : CH;
  // head work: thread (sample hs, slice hq of the hidden units; component hq of the action where hq < 4) -- 4 slices of 16 units per
  // sample with 64 samples, 8 slices of 8 with 32; dWo and the elements of Wo: thread (hidden unit us, sample quarter / component uq)
  constexpr int HSL = kPThreads / CH, HPER = kPH / HSL;
  const int hq = t % HSL, hs = t / HSL;
  const int uq = t & 3, us = t >> 2;
  const int l16 = lane & 15, g16 = lane >> 4;     // (CH = 32) column and row group of the 16 x 16 accumulator tiles

  float bc1 = powf(H.beta1, (float)H.step0), bc2 = powf(H.beta2, (float)H.step0);     // beta^t
  float acc_l = 0.f;                              // loss sum (wave 0 lanes, reduced at the end)
  const float invB = 1.0f / (float)B;
#ifdef FW_PPO_PROF
  long long pf_stats = 0, pf_gather = 0, pf_net = 0, pf_adam = 0, pf_xch = 0, pf_red = 0, pf_ho = 0, pf_norm = 0, pf_tile = 0, pf_scal = 0;
  long long pf_a[4] = {0, 0, 0, 0};            // the closing section: thread Adam, wait for the partners' weights, their fetch + LDS writes, closing barrier
  long long pf_g[4] = {0, 0, 0, 0}, pf_n[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_h[6] = {0, 0, 0, 0, 0, 0};      // pf_h: the hand-off, piece by piece      // gather: barrier, commit, prefetch issue, barrier; the phases of the chunk pass
#define PPO_T() ((long long)__builtin_readcyclecounter())
#endif

  // Inputs of a 64-sample chunk: one contiguous block of the packed array (fw_ppo_pack_kernel), 64 x W4 float4s, fetched one
  // chunk ahead -- float4 e of the chunk by thread e % 256 in pass e / 256 (coalesced: a wave-level load is 1 KB in a row) --
  // and scattered to X / sA / sS when the chunk's turn comes.  Where a float4 lands is the same for every chunk: computed once.
  const int cpm = B / CH;
  const int W4 = ((D + 3) >> 2) + 2, npass = (CH * W4 + kPThreads - 1) / kPThreads;      // 64 samples: 9 float4s per row, 3 passes (obs 28); 16, 4 (obs 56); 18, 5 (obs 64)
  typedef float ppo_x4 __attribute__((ext_vector_type(4)));
  constexpr int kPass = 5;                           // 64 rows x (16 + 2) float4s / 256 threads, rounded up (obs 57 .. 64; 28: 3, 56: 4)
  ppo_x4 pre_x[kPass];
  int pre_dst[kPass];                                    // LDS destination (float offset) of pass p's float4; < 0: nothing of mine / not this network's
  {
    const int Dv4 = W4 - 2;
#pragma unroll
    for (int p_ = 0; p_ < kPass; ++p_) {
      const int e = t + p_ * kPThreads, row = e / W4, q = e - row * W4;
      int d = -1;
      if (p_ < npass && e < CH * W4) {
        if (q < Dv4) d = (int)(X - lds) + row * ldx + 4 * q;          // (the row's zero padding lands on columns that hold zero anyway)
        else if (q == Dv4) d = NET == 0 ? (int)(sA - lds) + row * 4 : -1;
        else d = (int)(sS - lds) + row * 4;                           // old log-prob, advantage (normalised), return, -
      }
      pre_dst[p_] = d;
    }
  }
  auto prefetch = [&](int g) {
    const ppo_x4* src = reinterpret_cast<const ppo_x4*>(A.packed) + (size_t)g * (CH * W4) + t;
#pragma unroll
    for (int p_ = 0; p_ < kPass; ++p_) if (p_ < npass && pre_dst[p_] >= 0) pre_x[p_] = src[p_ * kPThreads];
  };
  auto commit = [&]() {
#pragma unroll
    for (int p_ = 0; p_ < kPass; ++p_) {
      if (p_ < npass && pre_dst[p_] >= 0) {
        float* d = lds + pre_dst[p_];
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = pre_x[p_][k];
      }
    }
  };
  int pmb = 0, pci = part;                          // next chunk to prefetch: minibatch, chunk index inside it
  bool gathered = false;                            // the next chunk's inputs are already in X / sA / sS (done inside the hand-off wait)
  // The Adam moments of the elements this lane owns stay in REGISTERS for the whole call (the compiler parks them in AGPRs): 2 x 32
  // tile elements (W2, W1) + 2 x NQ per-thread ones.  Round 3 fetched and stored them every minibatch -- ~100 KB through the CU's
  // 64 B / clk vector-memory path per minibatch, whose store queue the next chunk's gather then had to wait behind (its "issue"
  // took 2.2 k cycles) -- because 472 registers left no room; the leaner gather and dW1 of round 4 did.  Both chunk halves apply
  // the same update to the same initial values; part 0 writes the result back at the end.
  float4 pm[2][4], pv[2][4];
  // RS: what this thread owns -- set s_ = elements set_row .. + 3 of column set_col of W2 (kind 0) or W1 (kind 1), moments at set_slot ..
  float rm[NSETS][4], rv[NSETS][4];
  const int tq = RS ? part * 4 / NSd : 0;           // the wave whose tiles this block reduces
  int set_kind[NSETS], set_q[NSETS], set_slot[NSETS], set_row[NSETS];
  bool set_own[NSETS];
  const int set_col = (tq & 1) * 32 + r;
#pragma unroll
  for (int s_ = 0; s_ < NSETS; ++s_) {
    set_kind[s_] = NS == 8 ? wave >> 1 : s_;
    set_q[s_] = NS == 8 ? 2 * (part & 1) + (wave & 1) : wave;
    set_own[s_] = RS && (set_kind[s_] == 0 || tq < tilesW1);
    set_slot[s_] = ppo_tile_slot(n, set_kind[s_], tq, lane) + 4 * set_q[s_];
    set_row[s_] = (tq >> 1) * 32 + set_q[s_] * 8 + hh * 4;
  }
  if constexpr (RS) {
#pragma unroll
    for (int s_ = 0; s_ < NSETS; ++s_)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        rm[s_][e] = set_own[s_] ? mom_m[set_slot[s_] + e] : 0.f; rv[s_][e] = set_own[s_] ? mom_v[set_slot[s_] + e] : 0.f;
      }
  } else {
    const int s0 = ppo_tile_slot(n, 0, wave, lane), s1 = ppo_tile_slot(n, 1, wave, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pm[0][q] = reinterpret_cast<const float4*>(mom_m + s0)[q]; pv[0][q] = reinterpret_cast<const float4*>(mom_v + s0)[q];
      if (hasW1) { pm[1][q] = reinterpret_cast<const float4*>(mom_m + s1)[q]; pv[1][q] = reinterpret_cast<const float4*>(mom_v + s1)[q]; }
      else { pm[1][q] = make_float4(0.f, 0.f, 0.f, 0.f); pv[1][q] = pm[1][q]; }
    }
  }
  // per-thread elements: b1[t], b2[t] (t < 64), bo[t] (t < KO), Wo[t / 4][t % 4] (t % 4 < KO), log_std[t] (t < 4, pi block)
  constexpr int NQ = NET == 0 ? 5 : 4;
  int sl[NQ];
  float smm[NQ], svv[NQ];
  sl[0] = kPTileSlots + (n * 3 + 0) * kPThreads + t; sl[1] = kPTileSlots + (n * 3 + 1) * kPThreads + t;
  sl[2] = kPTileSlots + (n * 3 + 2) * kPThreads + t; sl[3] = kPTileSlots + (7 + n) * kPThreads + t;
  if (NET == 0) sl[NQ - 1] = kPTileSlots + 6 * kPThreads + t;
#pragma unroll
  for (int q = 0; q < NQ; ++q) { smm[q] = mom_m[sl[q]]; svv[q] = mom_v[sl[q]]; }
  prefetch(pmb * cpm + pci);
  pci += nsplit; if (pci >= cpm) { pmb += 1; pci = part; }

#pragma unroll 1
  for (int mb = 0; mb < n_mb && !dead; ++mb) {
#ifdef FW_PPO_PROF
    const long long pf0 = PPO_T();
#endif
#ifdef FW_PPO_PROF
    pf_stats += PPO_T() - pf0;
#endif
    // (policy head: log_std and 1 / sigma^2 of the action component this lane takes -- they only change with the Adam step)
    const float ls_c = NET == 0 ? log_std[(t % HSL) & 3] : 0.f;
    const float iv_c = NET == 0 ? expf(-2.0f * ls_c) : 0.f;
    // gradient accumulators of this minibatch (registers)
    f32x16 gW2, gW1;
    float gb1p = 0.f, gb2p = 0.f;                                     // column-sum partials of this thread's 16 rows
    float gWoq[KO];                                                   // dWo[hidden unit us][0 .. KO) over sample quarter uq of every chunk
    float gbo_p = 0.f, gls_p = 0.f;                                   // component hq of dbo / dlog_std over this thread's samples
#pragma unroll
    for (int v = 0; v < 16; ++v) { gW2[v] = 0.f; gW1[v] = 0.f; }
#pragma unroll
    for (int k = 0; k < KO; ++k) gWoq[k] = 0.f;

#pragma unroll 1
    for (int c0 = part * CH; c0 < B; c0 += nsplit * CH) {
      // ---- gather the chunk ----
#ifdef FW_PPO_PROF
      const long long pf1 = PPO_T();
#endif
      if (!gathered) {
        __syncthreads();                                                 // the previous chunk's readers of X / sA / sS are done
#ifdef FW_PPO_PROF
        const long long pg1 = PPO_T(); pf_g[0] += pg1 - pf1;
#endif
        commit();
#ifdef FW_PPO_PROF
        const long long pg2 = PPO_T(); pf_g[1] += pg2 - pg1;
#endif
        if (pmb < n_mb) {                                                // the next chunk's loads fly during this chunk's GEMMs
          prefetch(pmb * cpm + pci);
          pci += nsplit; if (pci >= cpm) { pmb += 1; pci = part; }
        }
#ifdef FW_PPO_PROF
        const long long pg3 = PPO_T(); pf_g[2] += pg3 - pg2;
#endif
        __syncthreads();
#ifdef FW_PPO_PROF
        pf_g[3] += PPO_T() - pg3;
#endif
      }
      else if (pmb < n_mb) {                                             // (gathered inside the last hand-off: only the prefetch is left to do)
        prefetch(pmb * cpm + pci);
        pci += nsplit; if (pci >= cpm) { pmb += 1; pci = part; }
      }
      gathered = false;
#ifdef FW_PPO_PROF
      const long long pf2 = PPO_T(); pf_gather += pf2 - pf1;
      long long pn = pf2;
#define PPO_PHASE(i) do { const long long t_ = PPO_T(); pf_n[i] += t_ - pn; pn = t_; } while (0)
#else
#define PPO_PHASE(i) do { } while (0)
#endif
      // ---- forward: H1 = tanh(X W1 + b1), H2 = tanh(H1 W2 + b2) ----
      if constexpr (CH == 64) {
        f32x16 c;
        const float bias = W.b1[nt * 32 + r];
#pragma unroll
        for (int v = 0; v < 16; ++v) c[v] = bias;
        c = ppo_mfma_tile(X + mt * 32 * ldx, ldx, 1, W.W1 + nt * 32, kPH, 1, Dp, c);
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          const ppo_f2 h = ppo_tanh2(ppo_f2{c[v], c[v + 1]});
          H1[(mt * 32 + ppo_acc_row(v)) * kPLdh + nt * 32 + r] = h[0]; H1[(mt * 32 + ppo_acc_row(v + 1)) * kPLdh + nt * 32 + r] = h[1];
        }
      } else {
        const float bias = W.b1[wave * 16 + l16];
        f32x4 c[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) c[rt] = f32x4{bias, bias, bias, bias};
        if (K1 == 32) ppo_mfma16_rows<8, RT>(X, ldx, 1, W.W1 + wave * 16, ldw1, 1, c);
        else ppo_mfma16_rows<16, RT>(X, ldx, 1, W.W1 + wave * 16, ldw1, 1, c);
        float* hp = H1 + (g16 * 4) * kPLdh + wave * 16 + l16;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int v = 0; v < 4; v += 2) {
            const ppo_f2 h = ppo_tanh2(ppo_f2{c[rt][v], c[rt][v + 1]});
            hp[(16 * rt + v) * kPLdh] = h[0]; hp[(16 * rt + v + 1) * kPLdh] = h[1];
          }
      }
      __syncthreads();
      PPO_PHASE(0);
      if constexpr (CH == 64) {
        f32x16 c;
        const float bias = W.b2[nt * 32 + r];
#pragma unroll
        for (int v = 0; v < 16; ++v) c[v] = bias;
        c = ppo_mfma_tile(H1 + mt * 32 * kPLdh, kPLdh, 1, W.W2 + nt * 32, kPLdh, 1, kPH, c);
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          const ppo_f2 h = ppo_tanh2(ppo_f2{c[v], c[v + 1]});
          H2[(mt * 32 + ppo_acc_row(v)) * kPLdh + nt * 32 + r] = h[0]; H2[(mt * 32 + ppo_acc_row(v + 1)) * kPLdh + nt * 32 + r] = h[1];
        }
      } else {
        const float bias = W.b2[wave * 16 + l16];
        f32x4 c[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) c[rt] = f32x4{bias, bias, bias, bias};
        ppo_mfma16_rows<16, RT>(H1, kPLdh, 1, W.W2 + wave * 16, kPLdh, 1, c);
        float* hp = H2 + (g16 * 4) * kPLdh + wave * 16 + l16;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int v = 0; v < 4; v += 2) {
            const ppo_f2 h = ppo_tanh2(ppo_f2{c[rt][v], c[rt][v + 1]});
            hp[(16 * rt + v) * kPLdh] = h[0]; hp[(16 * rt + v + 1) * kPLdh] = h[1];
          }
      }
      __syncthreads();
      PPO_PHASE(1);
      // ---- head and loss gradient, on the vector ALU by all four waves (the 64 x KO head is 6 % of a 32 x 32 MFMA tile):
      // thread (sample hs, slice hq) sums hidden units HPER hq .. HPER hq + HPER - 1 for all KO outputs, the slices are
      // combined by a butterfly inside the group, and every lane of the group then holds the sample's head output.
      // Straight-line code, every LDS operand of the phase fetched up front: written with the bias added inside the choice of the
      // component, log_std / the action / the sample's scalars read where they are used and the three cases of the surrogate's gradient
      // as branches, the phase was five exposed LDS round trips and six branches long -- 1.7 k cycles at 16 samples, the longest phase of
      // the chunk pass (the ISA is in CHANGELOG, round 5) ----
      {
        const int s = hs, hc = hq & 3;
        float hv[HPER];
        float4 w4[HPER];
        const float* h2 = H2 + hs * kPLdh + HPER * hq;
        const float* wo = W.Wo + HPER * hq * KO;
#pragma unroll
        for (int j = 0; j < HPER; ++j) {
          hv[j] = h2[j];
          if (KO == 4) w4[j] = reinterpret_cast<const float4*>(wo)[j];
          else w4[j] = make_float4(wo[j], 0.f, 0.f, 0.f);
        }
        const float bo_c = W.bo[KO == 4 ? hc : 0];
        float act_c = 0.f, old_lp = 0.f, adv_s = 0.f, ret_s = 0.f;
        if (NET == 0) { act_c = sA[s * 4 + hc]; const float2 sv = *reinterpret_cast<const float2*>(sS + s * 4); old_lp = sv.x; adv_s = sv.y; }
        else ret_s = sS[s * 4 + 2];
        float o[KO];
#pragma unroll
        for (int k = 0; k < KO; ++k) o[k] = 0.f;
#pragma unroll
        for (int j = 0; j < HPER; ++j) {
          o[0] += hv[j] * w4[j].x;
          if (KO == 4) { o[KO > 1 ? 1 : 0] += hv[j] * w4[j].y; o[KO > 2 ? 2 : 0] += hv[j] * w4[j].z; o[KO > 3 ? 3 : 0] += hv[j] * w4[j].w; }
        }
#pragma unroll
        for (int k = 0; k < KO; ++k) {
          o[k] += ppo_dpp<kDppXor1>(o[k]); o[k] += ppo_dpp<kDppXor2>(o[k]);
          if (HSL >= 8) o[k] += ppo_dpp<kDppHalfMirror>(o[k]);
          if (HSL == 16) o[k] += ppo_dpp<kDppRowMirror>(o[k]);
        }
        if (NET == 0) {
          // log pi(a|s), ratio, clipped surrogate (SB3 PPO.train): lane hc of the group's first quad takes action component hc
          // (32- / 16-sample forms: the other quads of a group compute along and contribute nothing)
          const bool hl = hq < 4;
          const float mu = (hc == 0 ? o[0] : hc == 1 ? o[KO > 1 ? 1 : 0] : hc == 2 ? o[KO > 2 ? 2 : 0] : o[KO > 3 ? 3 : 0]) + bo_c;
          const float ls = ls_c, iv = iv_c;                   // log_std and 1 / sigma^2 of component hc: per minibatch (see the top of the loop)
          const float z = act_c - mu;
          float logp = -0.5f * z * z * iv - ls - 0.9189385332046727f;
          logp += ppo_dpp<kDppXor1>(logp); logp += ppo_dpp<kDppXor2>(logp);
          const float a = adv_s;
          const float ratio = expf(logp - old_lp);
          const float rc = fminf(fmaxf(ratio, 1.0f - H.clip_range), 1.0f + H.clip_range);
          const float l1 = a * ratio, l2 = a * rc;
          if (hq == 0) acc_l += -fminf(l1, l2);
          // d(-min(l1, l2))/dlogp: through l1 when it is the smaller; on a tie (ratio inside the range: rc == ratio)
          // torch.min halves the gradient between the two branches and the clamp passes its half
          const bool inside = ratio >= 1.0f - H.clip_range && ratio <= 1.0f + H.clip_range;
          const float cf = -a * ratio * invB;
          const bool tie = l1 == l2;
          const float coef = (l1 < l2 || (tie && inside)) ? cf : (tie ? 0.5f * cf : 0.f);
          const float g = coef * z * iv;                                  // dL/dmu_k = dL/dlogp * (a_k - mu_k) / sigma_k^2
          gls_p += hl ? coef * (z * z * iv - 1.0f) : 0.f;                 // dL/dlog_std_k of this thread's samples
          gbo_p += hl ? g : 0.f;
          if (hl) gout[s * 4 + hc] = g;
        } else {
          const float dv = (o[0] + bo_c) - ret_s;
          if (hq == 0) acc_l += dv * dv;
          const float g = H.vf_coef * 2.0f * dv * invB;
          if (hq == 0) { gout[s * 4] = g; gbo_p += g; }
        }
      }
      __syncthreads();
      PPO_PHASE(2);
      // ---- dWo += H2^T gout (before H2 is overwritten): thread (hidden unit us, quarter uq) over a quarter of the chunk's samples ----
      {
        const float* h2 = H2 + (CH / 4) * uq * kPLdh + us;
        const float* go = gout + (CH / 4) * uq * 4;
#pragma unroll
        for (int j = 0; j < CH / 4; ++j) {
          const float h = h2[j * kPLdh];
          if (KO == 4) {
            const float4 g4 = reinterpret_cast<const float4*>(go)[j];
            gWoq[0] += h * g4.x; gWoq[KO > 1 ? 1 : 0] += h * g4.y; gWoq[KO > 2 ? 2 : 0] += h * g4.z; gWoq[KO > 3 ? 3 : 0] += h * g4.w;
          } else {
            gWoq[0] += h * go[j * 4];
          }
        }
      }
      __syncthreads();
      PPO_PHASE(3);
      // ---- G2 = (gout Wo^T) * (1 - H2^2), in place over H2 (K = KO <= 4) ----
      if constexpr (CH != 64) {
        float* hp = H2 + (g16 * 4) * kPLdh + wave * 16 + l16;
        float hv[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int v = 0; v < 4; ++v) hv[rt][v] = hp[(16 * rt + v) * kPLdh];
        const int k = g16 < KO ? g16 : 0;
        const float bv = g16 < KO ? W.Wo[(wave * 16 + l16) * KO + k] : 0.f;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const float av = g16 < KO ? gout[(16 * rt + l16) * 4 + k] : 0.f;
          const f32x4 c = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, z4, 0, 0, 0);
#pragma unroll
          for (int v = 0; v < 4; ++v) hv[rt][v] = c[v] * (1.0f - hv[rt][v] * hv[rt][v]);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int v = 0; v < 4; ++v) hp[(16 * rt + v) * kPLdh] = hv[rt][v];
      } else {
        // (the 16 activations this lane rescales are fetched in one batch ahead of the products: written as read-modify-write
        // per element the compiler keeps every LDS read behind the previous element's write -- 16 exposed LDS round trips)
        float* hp = H2 + (mt * 32 + hh * 4) * kPLdh + nt * 32 + r;
        float hv[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) hv[v] = hp[((v >> 2) * 8 + (v & 3)) * kPLdh];
        f32x16 c;
#pragma unroll
        for (int v = 0; v < 16; ++v) c[v] = 0.f;
#pragma unroll
        for (int k0 = 0; k0 < KO; k0 += 2) {
          const int k = k0 + hh;
          const float av = k < KO ? gout[(mt * 32 + r) * 4 + (k < KO ? k : 0)] : 0.f;
          const float bv = k < KO ? W.Wo[(nt * 32 + r) * KO + (k < KO ? k : 0)] : 0.f;
          c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) hv[v] = c[v] * (1.0f - hv[v] * hv[v]);
#pragma unroll
        for (int v = 0; v < 16; ++v) hp[((v >> 2) * 8 + (v & 3)) * kPLdh] = hv[v];
      }
      __syncthreads();
      PPO_PHASE(4);
      // ---- dW2 += H1^T G2 (rows = input unit), db2 partial ----
      gW2 = ppo_mfma_tile(H1 + mt * 32, 1, kPLdh, H2 + nt * 32, kPLdh, 1, CH, gW2);
      {
        float sgb = 0.f;
#pragma unroll
        for (int s = 0; s < CH / 4; ++s) sgb += H2[(wave * (CH / 4) + s) * kPLdh + lane];
        gb2p += sgb;
      }
      __syncthreads();
      PPO_PHASE(5);
      // ---- G1 = (G2 W2^T) * (1 - H1^2), in place over H1 ----
      if constexpr (CH != 64) {
        float* hp = H1 + (g16 * 4) * kPLdh + wave * 16 + l16;
        float hv[RT][4];
        f32x4 c[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          c[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int v = 0; v < 4; ++v) hv[rt][v] = hp[(16 * rt + v) * kPLdh];
        }
        ppo_mfma16_rows<16, RT>(H2, kPLdh, 1, W.W2 + wave * 16 * kPLdh, 1, kPLdh, c);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int v = 0; v < 4; ++v) hp[(16 * rt + v) * kPLdh] = c[rt][v] * (1.0f - hv[rt][v] * hv[rt][v]);
      } else {
        float* hp = H1 + (mt * 32 + hh * 4) * kPLdh + nt * 32 + r;      // (as for G2: the reads leave ahead of the products)
        float hv[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) hv[v] = hp[((v >> 2) * 8 + (v & 3)) * kPLdh];
        f32x16 c;
#pragma unroll
        for (int v = 0; v < 16; ++v) c[v] = 0.f;
        c = ppo_mfma_tile(H2 + mt * 32 * kPLdh, kPLdh, 1, W.W2 + nt * 32 * kPLdh, 1, kPLdh, kPH, c);
#pragma unroll
        for (int v = 0; v < 16; ++v) hv[v] = c[v] * (1.0f - hv[v] * hv[v]);
#pragma unroll
        for (int v = 0; v < 16; ++v) hp[((v >> 2) * 8 + (v & 3)) * kPLdh] = hv[v];
      }
      __syncthreads();
      PPO_PHASE(6);
      // ---- dW1 += X^T G1 (rows = obs feature, padded to 32 / 64), db1 partial ----
      if (hasW1 || splitW1) gW1 = ppo_mfma_tile(X + w1_k0 * ldx + w1_mt * 32, 1, ldx, H1 + w1_k0 * kPLdh + nt * 32, kPLdh, 1, w1_K, gW1);
      {
        float sgb = 0.f;
#pragma unroll
        for (int s = 0; s < CH / 4; ++s) sgb += H1[(wave * (CH / 4) + s) * kPLdh + lane];
        gb1p += sgb;
      }
      PPO_PHASE(7);
#ifdef FW_PPO_PROF
      pf_net += PPO_T() - pf2;
#endif
    }

#ifdef FW_PPO_PROF
    const long long pf3 = PPO_T();
#endif
    // ---- finish the per-thread gradients: the four row-block partials of the biases; dWo over the quad's sample quarters;
    // dbo / dlog_std component hq over all samples ----
    float my_gwo = 0.f;
#pragma unroll
    for (int k = 0; k < KO; ++k) {
      float g = gWoq[k];
      g += ppo_dpp<kDppXor1>(g); g += ppo_dpp<kDppXor2>(g);
      if (uq == k) my_gwo = g;
    }
    // component (lane % HSL) over the wave's lanes: rotations inside the row, then across the rows
    if (HSL == 4) { gbo_p += ppo_dpp<kDppRor4>(gbo_p); if (NET == 0) gls_p += ppo_dpp<kDppRor4>(gls_p); }
    if (HSL <= 8) { gbo_p += ppo_dpp<kDppRor8>(gbo_p); if (NET == 0) gls_p += ppo_dpp<kDppRor8>(gls_p); }
    gbo_p = ppo_sum_rows(gbo_p);
    if (NET == 0) gls_p = ppo_sum_rows(gls_p);
    // (no barrier in front: the previous readers of bred / sred are behind the barriers of the norm exchange and of the end of the
    // previous minibatch; H2, which carries dW1's split partials, was last read in the G1 phase, a barrier ago -- H1 is still being
    // read by slower waves' dW1)
    bred[wave * kPH + lane] = gb1p; bred[(4 + wave) * kPH + lane] = gb2p;
    if (lane < 4) { sred[wave * 8 + lane] = gbo_p; sred[wave * 8 + 4 + lane] = gls_p; }
    if (splitW1 && wave >= 2) {                    // the second sample half of dW1's two tiles, to its owners (waves 0, 1) through H2's space
      float* hx = H2 + (wave - 2) * 64 + lane;      // [element][wave][lane]: conflict-free dwords (2048 floats; H2 is not 16-byte aligned for KO = 1)
#pragma unroll
      for (int v = 0; v < 16; ++v) hx[v * 128] = gW1[v];
    }
    __syncthreads();
    if (splitW1 && wave < 2) {
      const float* hx = H2 + wave * 64 + lane;
#pragma unroll
      for (int v = 0; v < 16; ++v) gW1[v] += hx[v * 128];
    }
    float gb1 = 0.f, gb2 = 0.f;
    if (t < kPH) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { gb1 += bred[q * kPH + t]; gb2 += bred[(4 + q) * kPH + t]; }
    }
    float my_gbo = 0.f, my_gls = 0.f;
    if (t < 4) {
      if (t < KO) my_gbo = sred[t] + sred[8 + t] + sred[16 + t] + sred[24 + t];
      if (NET == 0) {
        my_gls = sred[4 + t] + sred[12 + t] + sred[20 + t] + sred[28 + t];
        if (part == 0) my_gls -= H.ent_coef;   // entropy bonus: entropy_loss = -mean(sum_k (c + log_std_k)) -> d/dlog_std_k = -ent_coef (once)
      }
    }

#ifdef FW_PPO_PROF
    const long long pfa = PPO_T(); pf_red += pfa - pf3;
#endif
    // ---- swap gradient partials with the other blocks of this network (all to all, or tiles by reduce-scatter), keep the sum ----
    float gq[NSETS][4];                                 // RS: the summed gradient of the tile elements this thread owns
#pragma unroll
    for (int s_ = 0; s_ < NSETS; ++s_)
#pragma unroll
      for (int e = 0; e < 4; ++e) gq[s_][e] = 0.f;
    if (nsplit > 1) {
      // (uniform for the compiler with eight blocks only -- measured on one box, product builds, us per minibatch: (28, 128) 7.79 -> 7.62;
      // with four blocks the burst of fetches it allows costs more elsewhere than it saves: (56, 64) 7.59 -> 7.72, (28, 64) 7.23 -> 7.20;
      // and never for the all-gather, whose readers poll: 7.62 -> 7.71 and 7.72 -> 7.86)
      float* gxb = A.gx + (size_t)((mb & 1) * 2 + NET) * kPMaxSplit * kPGxSlots;
      if constexpr (NS == 8) gxb = ppo_uniform_ptr(gxb);
      float* mine = gxb + (size_t)part * kPGxSlots;
      unsigned long long* fl = A.xch + kPpoWordFlags + ((mb & 1) * 2 + NET) * kPMaxSplit;
      const int g0 = ppo_gx_tile(0, wave, lane), g1 = ppo_gx_tile(1, wave, lane);
      // Plain vector stores, partner loads past the L1, bracketed by a device-scope release (every storing wave, before the barrier
      // and the flag) and acquire (after the flag) unless the blocks share an L2.  (Tried instead of the fences: per-word sc1 atomics
      // -- 27.4 vs 23.9 us; round 3: thread-to-thread self-announcing 8-byte write-through words polled by the receiver, no barrier /
      // flag / fence -- 14.5 k cycles: the memory system serves small device-scope accesses slowly.  Also measured, no gain: the tile
      // partials stored before the bias reductions (the wait moves, 8.7 k -> 7.9 k for the pair of sections), a register copy of the
      // lane's own weights so that Adam needs no LDS read (tile Adam 5.5 k -> 5.1 k, the chunk pass +1 k: 32 more live registers).)
      const int sq4 = kPGxTile + 4 * t, sqo = kPGxTile + kPThreads + t;      // per-thread elements: {gb1, gb2, gbo, gls} of thread t < 64; gwo of thread t
      typedef unsigned int ppo_u4 __attribute__((ext_vector_type(4)));
      typedef float ppo_f4 __attribute__((ext_vector_type(4)));
      float qs[NSETS][NSd][4];                     // RS: my sets in every block's partial, in block order
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gxb, 0, kPMaxSplit * kPGxSlots * (int)sizeof(float), 0x00020000);
      ppo_store_tile(mine + ppo_gx_tile(0, wave, 0), lane, gW2);
      if (hasW1) ppo_store_tile(mine + ppo_gx_tile(1, wave, 0), lane, gW1);
      // (per-thread elements: Wo's by every thread, the biases / log_std -- one float4 -- by threads of the first wave only)
      mine[sqo] = my_gwo;
      if (wave == 0) *reinterpret_cast<float4*>(mine + sq4) = make_float4(gb1, gb2, my_gbo, my_gls);
      // (the builtin, not inline assembly: the compiler's own wait-count bookkeeping must see that the stores are done, or it waits for
      // them one by one between the loads further down -- and with them, in order, for those loads)
      if (same_xcd) __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): my stores are in the L2 the partners read from
      else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");         // every storing wave: write back to where the partners can see it
      __syncthreads();
#ifdef FW_PPO_PROF
      long long ph = PPO_T(); pf_h[0] += ph - pfa;
#define PPO_HO(i) do { const long long t_ = PPO_T(); pf_h[i] += t_ - ph; ph = t_; } while (0)
#else
#define PPO_HO(i) do { } while (0)
#endif
      if (t == 0) {
        // (relaxed: the ordering is the waves' release / acquire -- or, on a shared L2, their store wait / L1 drop -- around the barriers)
        ppo_word_store(fl + part, (unsigned long long)(unsigned)(mb + 1), same_xcd);
      }
      // While the partners' flags are on their way, the inputs of the NEXT minibatch's first chunk go to LDS -- X / sA / sS were last
      // read in this minibatch's chunk pass, and the barrier behind the poll below stands in for the one a chunk's gather ends with.
      // (The loads of the chunk AFTER that one are issued at the start of the next chunk pass, which waits for no memory operation
      // for thousands of cycles: they come from HBM -- every packed row is read once -- and the counter retires in order, so
      // issued here they made every wait of the exchange, whose data the L2 serves in a fraction of that time, a wait for HBM;
      // behind the exchange's fetches but inside a branch, they still did: the compiler's wait counts must hold on both paths.)
      if (mb + 1 < n_mb) { commit(); gathered = true; }
      PPO_HO(1);
      if (t < nsplit && t != part) {               // thread q waits for block q
        unsigned long long w;
        if (!ppo_wait(A, [&]() { return ppo_word_load(fl + t, same_xcd); }, [&](unsigned long long x) { return (unsigned)x == (unsigned)(mb + 1); },
                      (unsigned long long)PPO_ST_SWAP, w)) red[7] = 1.f;       // a partner block is gone: say so and leave instead of hanging
      }
      PPO_HO(2);
      __syncthreads();
      PPO_HO(3);
      if (red[7] != 0.f) { dead = true; break; }
      // The partners' rows are read past this CU's L1 (sc1 loads; the L1 may still hold the lines from two minibatches ago, and
      // `buffer_inv sc0` does not drop them outside tg-split mode: a loop of it and plain loads never saw a word change), served
      // by the L2 the partners' stores sit in -- after a device-scope acquire when the blocks do not share one.  Buffer loads:
      // the builtin takes the cache policy, so the compiler schedules and waits for them itself.  (Round 3 / 4 had them as inline
      // assembly with a hand-placed wait; the compiler, not knowing that they were loads, put waits for older operations between
      // them, which -- the counter retires in order -- serialised the partners: 4.3 k cycles for 39 loads.)
      ppo_f4 ta[3][4], tb[3][4];                   // all-to-all form: partner j of 1 (2 blocks) or 3 (4 blocks) = the other blocks in ascending order
      float vs[5][NSd];                            // per-thread elements {gb1, gb2, gbo, gls, gwo} of every block, in block order (own ones in slot `part`)
      const int np = nsplit - 1;
      if (!same_xcd) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      constexpr int kSc1 = 16;                     // cache-policy bit of the raw buffer loads
      if constexpr (RS) {
#pragma unroll
        for (int s_ = 0; s_ < NSETS; ++s_) {
          const int o = (ppo_gx_tile(set_kind[s_], tq, lane) + set_q[s_] * 256) * (int)sizeof(float);
          if (set_own[s_]) {                       // (ONE branch around all the loads of a set, never one per load: see the note below)
#pragma unroll
            for (int b_ = 0; b_ < NS; ++b_) ppo_ld_sc1(rs, (b_ ^ part) * kPGxSlots * (int)sizeof(float) + o, qs[s_][b_]);      // (own partial included: it comes back from the L2 my stores went to)
          } else {
#pragma unroll
            for (int b_ = 0; b_ < NS; ++b_)
#pragma unroll
              for (int e = 0; e < 4; ++e) qs[s_][b_][e] = 0.f;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int base = (j < part ? j : j + 1) * kPGxSlots * (int)sizeof(float);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (j < np) {
              const ppo_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (g0 + q * 256) * 4, 0, kSc1);
              ta[j][q] = ppo_f4{__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]), __uint_as_float(a[3])};
              if (hasW1) {
                const ppo_u4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (g1 + q * 256) * 4, 0, kSc1);
                tb[j][q] = ppo_f4{__uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[2]), __uint_as_float(b[3])};
              } else tb[j][q] = ppo_f4{0.f, 0.f, 0.f, 0.f};
            } else { ta[j][q] = ppo_f4{0.f, 0.f, 0.f, 0.f}; tb[j][q] = ta[j][q]; }
          }
        }
      }
      // per-thread elements, all to all in both forms: Wo's word of every thread and, for the first wave, one float4 {gb1, gb2, gbo, gls}
      // per block.  No load sits in a branch of its own (own slot and slots past nsplit included: they are inside the buffer and merely
      // not used) and the choice is a select afterwards: a load inside a branch on `part` makes the compiler wait for it at the join,
      // one L2 round trip after the other (measured: loads 3.0 k -> 4.3 k cycles with four blocks, 6.7 k with eight).
      ppo_u4 va[NSd];
      unsigned vo[NSd];
#pragma unroll
      for (int b_ = 0; b_ < NSd; ++b_) vo[b_] = __builtin_amdgcn_raw_buffer_load_b32(rs, (b_ ^ part) * kPGxSlots * (int)sizeof(float) + sqo * 4, 0, kSc1);      // ALL the loads first ...
      if (wave == 0) {                             // (one branch, the last loads of the section: the wait at its join is the wait for everything anyway)
#pragma unroll
        for (int b_ = 0; b_ < NSd; ++b_) va[b_] = __builtin_amdgcn_raw_buffer_load_b128(rs, (b_ ^ part) * kPGxSlots * (int)sizeof(float) + sq4 * 4, 0, kSc1);
      } else {
#pragma unroll
        for (int b_ = 0; b_ < NSd; ++b_) va[b_] = ppo_u4{0u, 0u, 0u, 0u};
      }
      // (... the selects behind them: with a select next to its load the compiler emitted load, wait, select per block -- eight round
      // trips in a row, most of the 4.5 k cycles this section took)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b_ = 0; b_ < NSd; ++b_) {           // slot b_ = block b_ ^ part (slot 0: my own, from registers; blocks past nsplit: nothing)
        const bool other = b_ != 0 && (b_ ^ part) < nsplit;
        vs[0][b_] = b_ == 0 ? gb1 : other ? __uint_as_float(va[b_][0]) : 0.f;
        vs[1][b_] = b_ == 0 ? gb2 : other ? __uint_as_float(va[b_][1]) : 0.f;
        vs[2][b_] = b_ == 0 ? my_gbo : other ? __uint_as_float(va[b_][2]) : 0.f;
        vs[3][b_] = b_ == 0 ? my_gls : other ? __uint_as_float(va[b_][3]) : 0.f;
        vs[4][b_] = b_ == 0 ? my_gwo : other ? __uint_as_float(vo[b_]) : 0.f;
      }
      PPO_HO(4);
      // WHICH block's partial a slot holds: slot i = block i ^ part.  At every step of the fetch the blocks of a network are then at
      // DIFFERENT partials (round 5: all of them walking 0, 1, 2 .. together queued at whatever part of the L2 holds that partial -- the
      // fetch of eight blocks took twice that of four for 1.3 x the instructions).  The sum does not notice: an XOR of the leaf index
      // swaps the two children of some nodes of a balanced tree, and fp addition commutes.
      // Every block must end up with the same bits: the partials p0 .. are summed as a balanced tree over the block order everywhere --
      // (p0 + p1) + (p2 + p3) with four, ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)) with eight (own partial in registers, the
      // others as loaded; fp addition commutes).  Two blocks: own + partner.
      auto sum4 = [&](auto own, auto t0, auto t1, auto t2) {
        if (np == 1) return own + t0;
        return part < 2 ? (own + t0) + (t1 + t2) : (t0 + t1) + (own + t2);
      };
      auto tree = [&](const float (&v)[NSd]) {
        if constexpr (RS) return ppo_tree_sum<NSd>(v);
        else return np == 1 ? v[0] + v[1] : (v[0] + v[1]) + (v[2] + v[3]);      // (slots past nsplit hold zero)
      };
      if constexpr (RS) {
#pragma unroll
        for (int s_ = 0; s_ < NSETS; ++s_)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float c[NSd];
#pragma unroll
            for (int b_ = 0; b_ < NSd; ++b_) c[b_] = qs[s_][b_][e];
            gq[s_][e] = ppo_tree_sum<NSd>(c);
          }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const ppo_f4 o2 = {gW2[4 * q], gW2[4 * q + 1], gW2[4 * q + 2], gW2[4 * q + 3]};
          const ppo_f4 r2 = sum4(o2, ta[0][q], ta[1][q], ta[2][q]);
          gW2[4 * q] = r2[0]; gW2[4 * q + 1] = r2[1]; gW2[4 * q + 2] = r2[2]; gW2[4 * q + 3] = r2[3];
          if (hasW1) {
            const ppo_f4 o1 = {gW1[4 * q], gW1[4 * q + 1], gW1[4 * q + 2], gW1[4 * q + 3]};
            const ppo_f4 r1 = sum4(o1, tb[0][q], tb[1][q], tb[2][q]);
            gW1[4 * q] = r1[0]; gW1[4 * q + 1] = r1[1]; gW1[4 * q + 2] = r1[2]; gW1[4 * q + 3] = r1[3];
          }
        }
      }
      gb1 = tree(vs[0]); gb2 = tree(vs[1]); my_gbo = tree(vs[2]); my_gls = tree(vs[3]); my_gwo = tree(vs[4]);
      PPO_HO(5);
    }

#ifdef FW_PPO_PROF
    const long long pfb = PPO_T(); pf_ho += pfb - pfa;
#endif
    // ---- global gradient norm: own elements, then the other network's partial ----
    // thread 0: an early look at the other network's word of this minibatch -- a device-scope load (past the L1, served by the L2
    // the two blocks share), issued now and first used behind this block's own norm: the block that arrives second (the policy
    // block, as a rule) finds the word there and never polls.  A stale or missing word only means the poll below runs.
    unsigned long long spec_w = 0ull;
    if (t == 0) spec_w = __hip_atomic_load(A.xch + ((mb & 1) * 2 + (1 - NET)) * kPMaxSplit + part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float ss = 0.f;
    if constexpr (RS) {                                   // my share of the tiles; the per-thread elements (every block holds their sums) count once
#pragma unroll
      for (int s_ = 0; s_ < NSETS; ++s_)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (set_own[s_] && (set_kind[s_] == 0 || set_row[s_] + e < D)) ss += gq[s_][e] * gq[s_][e];
    } else {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        ss += gW2[v] * gW2[v];
        if (hasW1 && mt * 32 + ppo_acc_row(v) < D) ss += gW1[v] * gW1[v];
      }
    }
    if (!RS || part == 0) {
      if (t < kPH) ss += gb1 * gb1 + gb2 * gb2;
      if (t < KO) ss += my_gbo * my_gbo;
      if (uq < KO) ss += my_gwo * my_gwo;
      if (NET == 0 && t < 4) ss += my_gls * my_gls;
    }
    const float ss_mine = ppo_block_sum_nb(ss, red);      // (red[0..3] were last read before the previous minibatch's closing barrier)
    float ss_other = 0.f;
#ifdef FW_PPO_PROF
    const long long pfx = PPO_T(); pf_norm += pfx - pfb;
#endif
    if constexpr (RS) {
      // all 2 NS shares, summed in word order ([net][part]) by everybody: lane i of the first wave fetches word i
      unsigned long long* words = A.xch + (mb & 1) * 2 * kPMaxSplit;
      const int me = NET * kPMaxSplit + part;
      if (t == 0) ppo_word_store(words + me, ((unsigned long long)(unsigned)(mb + 1) << 32) | (unsigned long long)__float_as_uint(ss_mine), same_xcd_net);
      if (t < 2 * kPMaxSplit && (t & (kPMaxSplit - 1)) < NS) {
        unsigned long long w = (unsigned long long)__float_as_uint(ss_mine);
        if (t != me && !ppo_wait(A, [&]() { return ppo_word_load(words + t, same_xcd_net); }, [&](unsigned long long x) { return (unsigned)(x >> 32) == (unsigned)(mb + 1); },
                                 (unsigned long long)PPO_ST_NORM, w)) red[7] = 1.f;     // a block is gone
        red8[t] = __uint_as_float((unsigned)w);
      }
      __syncthreads();
      if (red[7] != 0.f) { dead = true; break; }
      float tot = red8[0];
#pragma unroll
      for (int i = 1; i < 2 * kPMaxSplit; ++i) if ((i & (kPMaxSplit - 1)) < NS) tot += red8[i];
      ss_other = tot;                                      // (the total; ss_mine is not added again below)
      (void)spec_w;
    } else {
      // one 64-bit word per block and minibatch parity: (minibatch + 1) << 32 | float bits
      unsigned long long* mine = A.xch + ((mb & 1) * 2 + NET) * kPMaxSplit + part;
      unsigned long long* other = A.xch + ((mb & 1) * 2 + (1 - NET)) * kPMaxSplit + part;
      if (t == 0) {
        ppo_word_store(mine, ((unsigned long long)(unsigned)(mb + 1) << 32) | (unsigned long long)__float_as_uint(ss_mine), same_xcd_net);
        unsigned long long w = spec_w;
        if ((unsigned)(w >> 32) != (unsigned)(mb + 1) && !ppo_wait(A, [&]() { return ppo_word_load(other, same_xcd_net); }, [&](unsigned long long x) { return (unsigned)(x >> 32) == (unsigned)(mb + 1); },
                      (unsigned long long)PPO_ST_NORM, w)) red[7] = 1.f;     // the other network's block is gone
        red[4] = __uint_as_float((unsigned)w);
      }
      __syncthreads();
      if (red[7] != 0.f) { dead = true; break; }
      ss_other = red[4];
    }
#ifdef FW_PPO_PROF
    const long long pfc = PPO_T(); pf_xch += pfc - pfx;
#endif
    const float total_norm = sqrtf(RS ? ss_other : ss_mine + ss_other);
    const float clipc = fminf(H.max_grad_norm / (total_norm + 1e-6f), 1.0f);

    // ---- Adam on the elements each lane holds (moments: registers, see above) ----
    bc1 *= H.beta1; bc2 *= H.beta2;
    const float c1 = H.lr / (1.0f - bc1), sc2 = 1.0f / sqrtf(1.0f - bc2);      // step size, 1 / sqrt(bias correction 2)
    auto adam_tile = [&](const f32x16& g, float4 (&m4)[4], float4 (&v4)[4], auto&& lds_of /* v -> weight in LDS (rows the network does not have: the lane's sink word) */) {
      // the 16 weights are read in one batch, updated in registers and written in one batch (element by element the compiler
      // keeps each read behind the previous write: 32 exposed LDS round trips per minibatch, ~3 k cycles)
      float wv[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) wv[v] = *lds_of(v);
      // two elements per instruction where the ISA has a packed form (v_pk_mul_f32 / v_pk_fma_f32: twice the fp32 rate); the
      // square root and the reciprocal have none
      typedef float ppo_f2 __attribute__((ext_vector_type(2)));
      const ppo_f2 b1 = {H.beta1, H.beta1}, ob1 = {1.0f - H.beta1, 1.0f - H.beta1}, b2 = {H.beta2, H.beta2}, ob2 = {1.0f - H.beta2, 1.0f - H.beta2};
      float upd[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* mm = reinterpret_cast<float*>(&m4[q]);
        float* vv = reinterpret_cast<float*>(&v4[q]);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const ppo_f2 gg = ppo_f2{g[q * 4 + e], g[q * 4 + e + 1]} * clipc;
          const ppo_f2 mn = b1 * ppo_f2{mm[e], mm[e + 1]} + ob1 * gg;
          const ppo_f2 vn = b2 * ppo_f2{vv[e], vv[e + 1]} + ob2 * (gg * gg);
          mm[e] = mn[0]; mm[e + 1] = mn[1]; vv[e] = vn[0]; vv[e + 1] = vn[1];
          const ppo_f2 den = ppo_f2{__builtin_amdgcn_sqrtf(vn[0]), __builtin_amdgcn_sqrtf(vn[1])} * sc2 + H.eps;
          const ppo_f2 u = (mn * c1) * ppo_f2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
          upd[q * 4 + e] = u[0]; upd[q * 4 + e + 1] = u[1];
        }
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) *lds_of(v) = wv[v] - upd[v];
    };
    float* const wxn = A.gx + (size_t)NET * kPMaxSplit * kPGxSlots + (kPGxTile + 2 * kPThreads);      // (RS) this network's weight shares: part p's at + p * kPGxSlots, [set][wave * 64 + lane][4]
    unsigned long long* const fl2 = A.xch + kPpoWordFlags2 + ((mb & 1) * 2 + NET) * kPMaxSplit;
    if constexpr (RS) {
      // Adam on the share this block owns: 4 elements per set; the new weights go to this block's LDS image and to the exchange
      // buffer, from where the other blocks fetch them below (they hold the same old weights, so the copies stay the same bits)
      float* wp[NSETS][4];
      float wv[NSETS][4];
#pragma unroll
      for (int s_ = 0; s_ < NSETS; ++s_)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = set_row[s_] + e;
          wp[s_][e] = !set_own[s_] ? sink + t : set_kind[s_] == 0 ? W.W2 + row * kPLdh + set_col : row < D ? W.W1 + row * ldw1 + set_col : sink + t;
          wv[s_][e] = *wp[s_][e];
        }
#pragma unroll
      for (int s_ = 0; s_ < NSETS; ++s_) {
        if (set_own[s_]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gg = gq[s_][e] * clipc;
            const float mn = H.beta1 * rm[s_][e] + (1.0f - H.beta1) * gg, vn = H.beta2 * rv[s_][e] + (1.0f - H.beta2) * (gg * gg);
            rm[s_][e] = mn; rv[s_][e] = vn;
            wv[s_][e] -= (mn * c1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vn) * sc2 + H.eps);
          }
        }
      }
#pragma unroll
      for (int s_ = 0; s_ < NSETS; ++s_) {
#pragma unroll
        for (int e = 0; e < 4; ++e) *wp[s_][e] = wv[s_][e];
        if (!(TAGGED && same_xcd) && set_own[s_]) *reinterpret_cast<float4*>(wxn + (size_t)part * kPGxSlots + (s_ * kPThreads + t) * 4) = make_float4(wv[s_][0], wv[s_][1], wv[s_][2], wv[s_][3]);
      }
      if (TAGGED && same_xcd) {
        // SELF-ANNOUNCING shares (blocks on one XCD): a thread's new weights leave as 16-byte granules of three floats and a TAG (the
        // minibatch number) -- [granule][thread][4], 1 KB in a row per wave -- and the readers poll the granules themselves.  No wait
        // for the stores, no barrier, no flag, no poll of a flag: one hop between CUs instead of two (the flag's took 0.65 k cycles of
        // waiting behind 0.5 k of draining).  A 16-byte aligned store is one request to the L2: a reader sees all of a granule or
        // none of it; granules nobody needs (W1 shares nobody owns) are written all the same, so that every tag can be waited for.
        float* gp = wxn + (size_t)part * kPGxSlots + t * 4;
        const float tagf = __uint_as_float((unsigned)(mb + 1));
        if constexpr (NSETS == 2) {
          *reinterpret_cast<float4*>(gp) = make_float4(wv[0][0], wv[0][1], wv[0][2], tagf);
          *reinterpret_cast<float4*>(gp + kPThreads * 4) = make_float4(wv[0][3], wv[NSETS - 1][0], wv[NSETS - 1][1], tagf);
          *reinterpret_cast<float4*>(gp + 2 * kPThreads * 4) = make_float4(wv[NSETS - 1][2], wv[NSETS - 1][3], 0.f, tagf);
        } else {
          *reinterpret_cast<float4*>(gp) = make_float4(wv[0][0], wv[0][1], wv[0][2], tagf);
          *reinterpret_cast<float4*>(gp + kPThreads * 4) = make_float4(wv[0][3], 0.f, 0.f, tagf);
        }
      } else {
        if (same_xcd) __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0), as for the partials
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (t == 0) ppo_word_store(fl2 + part, (unsigned long long)(unsigned)(mb + 1), same_xcd);
      }
    } else {
    adam_tile(gW2, pm[0], pv[0], [&](int v) { return W.W2 + (mt * 32 + ppo_acc_row(v)) * kPLdh + nt * 32 + r; });
    if (hasW1) adam_tile(gW1, pm[1], pv[1],
                         [&](int v) { const int i = mt * 32 + ppo_acc_row(v); return i < D ? W.W1 + i * ldw1 + nt * 32 + r : sink + t; });
    }
#ifdef FW_PPO_PROF
    const long long pfd = PPO_T(); pf_tile += pfd - pfc;
#endif
    {
      // per-thread elements (moments fetched above); the owner test only gates the LDS write
      float gs[NQ], *ws[NQ];
      gs[0] = gb1; ws[0] = t < kPH ? W.b1 + t : nullptr;
      gs[1] = gb2; ws[1] = t < kPH ? W.b2 + t : nullptr;
      gs[2] = my_gbo; ws[2] = t < KO ? W.bo + t : nullptr;
      gs[3] = my_gwo; ws[3] = uq < KO ? W.Wo + us * KO + uq : nullptr;
      if (NET == 0) { gs[NQ - 1] = my_gls; ws[NQ - 1] = t < 4 ? log_std + t : nullptr; }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float gg = gs[q] * clipc;
        smm[q] = H.beta1 * smm[q] + (1.0f - H.beta1) * gg; svv[q] = H.beta2 * svv[q] + (1.0f - H.beta2) * gg * gg;
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) if (ws[q]) *ws[q] -= c1 * smm[q] * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(svv[q]) * sc2 + H.eps);
    }
#ifdef FW_PPO_PROF
    const long long pa1 = PPO_T(); pf_a[0] += pa1 - pfd;
    long long pa3 = pa1;
#endif
    if constexpr (RS) {
      // ---- all-gather of the updated weights: the other NS - 1 shares, from the blocks that own them ----
      float wg[NS - 1][NSETS][4];
      const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(wxn, 0, kPMaxSplit * kPGxSlots * (int)sizeof(float), 0x00020000);
      if (TAGGED && same_xcd) {
        // every wave polls the granules it needs itself (no block-wide wait): fetch all of them, look at the tags, again if one is old
        constexpr int NG = NSETS == 2 ? 3 : 2;
        const unsigned want = (unsigned)(mb + 1);
        float gr[NS - 1][NG][4];
        bool ok = false;
        for (long long rounds = 0; rounds < A.spin && !ok; ++rounds) {
          asm volatile("" ::: "memory");           // (the fetches of a round are not those of the last one: nothing may be hoisted out of the loop)
#pragma unroll
          for (int j = 0; j < NS - 1; ++j) {
            const int pj = (j + 1) ^ part;         // (every block at a different partner at every step, as in the reduce-scatter)
#pragma unroll
            for (int g = 0; g < NG; ++g) ppo_ld_sc1(rsw, (int)((pj * kPGxSlots + (g * kPThreads + t) * 4) * sizeof(float)), gr[j][g]);
          }
          bool mine = true;
#pragma unroll
          for (int j = 0; j < NS - 1; ++j)
#pragma unroll
            for (int g = 0; g < NG; ++g) mine = mine && __float_as_uint(gr[j][g][3]) == want;
          ok = __builtin_amdgcn_ballot_w64(!mine) == 0ull;
          if (!ok && (rounds & 255) == 255 && __hip_atomic_load(A.xch + kPpoWordStatus, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
        }
        if (!ok) {                                 // a partner is gone (or the budget was one round: tests)
          if (lane == 0) { (void)__hip_atomic_fetch_or(A.xch + kPpoWordStatus, (unsigned long long)PPO_ST_SWAP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); red[7] = 1.f; }
        }
#pragma unroll
        for (int j = 0; j < NS - 1; ++j) {
          if constexpr (NSETS == 2) {
            wg[j][0][0] = gr[j][0][0]; wg[j][0][1] = gr[j][0][1]; wg[j][0][2] = gr[j][0][2]; wg[j][0][3] = gr[j][1][0];
            wg[j][NSETS - 1][0] = gr[j][1][1]; wg[j][NSETS - 1][1] = gr[j][1][2]; wg[j][NSETS - 1][2] = gr[j][NG - 1][0]; wg[j][NSETS - 1][3] = gr[j][NG - 1][1];
          } else {
            wg[j][0][0] = gr[j][0][0]; wg[j][0][1] = gr[j][0][1]; wg[j][0][2] = gr[j][0][2]; wg[j][0][3] = gr[j][1][0];
          }
        }
      } else {
        if (t < nsplit && t != part) {
          unsigned long long w;
          if (!ppo_wait(A, [&]() { return ppo_word_load(fl2 + t, same_xcd); }, [&](unsigned long long x) { return (unsigned)x == (unsigned)(mb + 1); },
                        (unsigned long long)PPO_ST_SWAP, w)) red[7] = 1.f;
        }
        __syncthreads();
        if (red[7] != 0.f) { dead = true; break; }
        if (!same_xcd) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // (every load unconditional -- a share of W1 nobody owns is read and not used: loads inside branches wait one by one at the joins)
#pragma unroll
        for (int j = 0; j < NS - 1; ++j) {
          const int pj = (j + 1) ^ part;
#pragma unroll
          for (int s_ = 0; s_ < NSETS; ++s_) ppo_ld_sc1(rsw, (int)((pj * kPGxSlots + (s_ * kPThreads + t) * 4) * sizeof(float)), wg[j][s_]);
        }
      }
      // partner thread (w, l) of block pj wrote what its sets are: the same wave and lane as mine, block pj's tiles and quarters.
      // Branch-free: every element is stored, to its place or -- rows W1 does not have, W1 tiles nobody owns -- to this thread's sink
      // word.  (Written as `if (kind == 0) W2[..] = x; else if (row < D) W1[..] = x;` the eight-block form, whose `kind` is a run-time
      // scalar, compiled into a branch, an exec mask and two spilled-SGPR reloads PER ELEMENT: 1.5 k cycles for 28 LDS writes.)
#pragma unroll
      for (int j = 0; j < NS - 1; ++j) {
        const int pj = (j + 1) ^ part;
        const int tj = pj * 4 / NS, col = (tj & 1) * 32 + r;
#pragma unroll
        for (int s_ = 0; s_ < NSETS; ++s_) {
          const int kind = NS == 8 ? wave >> 1 : s_, qj = NS == 8 ? 2 * (pj & 1) + (wave & 1) : wave;
          const int row = (tj >> 1) * 32 + qj * 8 + hh * 4;
          const int ld = kind == 0 ? kPLdh : ldw1;
          float* const base = (kind == 0 ? W.W2 : W.W1) + row * ld + col;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = kind == 0 || (tj < tilesW1 && row + e < D);
            float* const d = ok ? base + e * ld : sink + t;
            *d = wg[j][s_][e];
          }
        }
      }
    }
#ifdef FW_PPO_PROF
    if constexpr (RS) { pa3 = PPO_T(); pf_a[2] += pa3 - pa1 - 0; }      // (from the end of the thread Adam: wait + fetch + LDS writes)
#endif
    __syncthreads();
#ifdef FW_PPO_PROF
    { const long long pfe = PPO_T(); pf_adam += pfe - pf3; pf_scal += pfe - pfd; pf_a[3] += pfe - pa3; }
#endif
    if (red[7] != 0.f) { dead = true; break; }     // (a wave's poll of the partners' weight granules ran out)
  }

  // ---- the verdict: results are written back only when EVERY block of the call got through its last minibatch with every wait
  // answered.  A block that gave up (`dead`) does not arrive, so the count never fills and nobody writes; the block that arrives
  // last reads the status word once more (a block may have raised it and still finished its own waits) and publishes the verdict
  // the others wait for.  (Round 4 decided per block: one whose own waits had all succeeded wrote back beside partners that had
  // given up -- mixed buffers under a non-zero status.  A verdict wait that itself runs out raises PPO_ST_COMMIT: status != 0 then
  // means "some blocks may have written": the caller restores the buffers from its own copies, see include/fwsim.h.)
  if (t == 0) {
    bool commit = false;
    if (!dead) {
      const int total = 2 * nsplit;
      const unsigned long long before = __hip_atomic_fetch_add(A.xch + kPpoWordArrive, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (before == (unsigned long long)(total - 1)) {
        commit = __hip_atomic_load(A.xch + kPpoWordStatus, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0ull;
        __hip_atomic_store(A.xch + kPpoWordVerdict, commit ? 1ull : 2ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        unsigned long long w = 0ull;
        if (ppo_wait(A, [&]() { return __hip_atomic_load(A.xch + kPpoWordVerdict, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); },
                     [](unsigned long long x) { return x != 0ull; }, (unsigned long long)PPO_ST_COMMIT, w)) commit = w == 1ull;
      }
    }
    red[6] = commit ? 1.f : 0.f;
  }
  __syncthreads();
  if (red[6] == 0.f) return;
  if constexpr (RS) {                               // the tile moments: every block its share
#pragma unroll
    for (int s_ = 0; s_ < NSETS; ++s_)
      if (set_own[s_]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { mom_m[set_slot[s_] + e] = rm[s_][e]; mom_v[set_slot[s_] + e] = rv[s_][e]; }
      }
  }
  if (part == ((A.flags & PPO_FLAG_WRITER_LAST) ? nsplit - 1 : 0)) {
    {
      const int s0 = ppo_tile_slot(n, 0, wave, lane), s1 = ppo_tile_slot(n, 1, wave, lane);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (RS) break;
        reinterpret_cast<float4*>(mom_m + s0)[q] = pm[0][q]; reinterpret_cast<float4*>(mom_v + s0)[q] = pv[0][q];
        if (hasW1) { reinterpret_cast<float4*>(mom_m + s1)[q] = pm[1][q]; reinterpret_cast<float4*>(mom_v + s1)[q] = pv[1][q]; }
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) { mom_m[sl[q]] = smm[q]; mom_v[sl[q]] = svv[q]; }
    }
    for (int i = t; i < Dp * kPH; i += kPThreads) params[oW1 + i] = W.W1[(i >> 6) * ldw1 + (i & 63)];
    for (int i = t; i < kPH; i += kPThreads) { params[ob1 + i] = W.b1[i]; params[ob2 + i] = W.b2[i]; }
    for (int i = t; i < kPH * kPH; i += kPThreads) params[oW2 + i] = W.W2[(i >> 6) * kPLdh + (i & 63)];
    for (int i = t; i < kPH * KO + KO; i += kPThreads) params[oWo + i] = W.Wo[i];
    if (NET == 0 && t < 4) params[oLs + t] = log_std[t];
  }
  const float lsum = ppo_block_sum(acc_l, red);
  if (t == 0 && A.loss_acc) {
    // The losses: every block leaves its sum in its own word; the block that finishes last adds them up in a fixed order (a float
    // atomicAdd per block would make the logged value depend on which of four blocks came first).
    __hip_atomic_store(A.xch + kPpoWordLoss + NET * kPMaxSplit + part, (unsigned long long)__float_as_uint(lsum * invB), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long before = __hip_atomic_fetch_add(A.xch + kPpoWordDone, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before == (unsigned long long)(2 * nsplit - 1)) {
      for (int net = 0; net < 2; ++net) {
        float l = 0.f;
        for (int q = 0; q < nsplit; ++q)
          l += __uint_as_float((unsigned)__hip_atomic_load(A.xch + kPpoWordLoss + net * kPMaxSplit + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        atomicAdd(&A.loss_acc[net], l);                               // sums over minibatches of the per-minibatch means
      }
    }
    if (NET == 0 && part == 0) {
      float ent = 0.f;
      for (int k = 0; k < 4; ++k) ent += 1.4189385332046727f + log_std[k];
      atomicAdd(&A.loss_acc[2], -ent * (float)n_mb);                  // entropy loss at the final log_std (it is state-independent)
    }
#ifdef FW_PPO_PROF
    float* pr = A.loss_acc + 3 + NET * 4;
    pr[0] = (float)pf_xch / n_mb; pr[1] = (float)pf_gather / n_mb; pr[2] = (float)pf_net / n_mb; pr[3] = (float)pf_adam / n_mb;
    if (NET == 0 && part == 0) {      // the finish section of the policy block, piece by piece
      float* px = A.loss_acc + 11;
      px[0] = (float)pf_red / n_mb; px[1] = (float)pf_ho / n_mb; px[2] = (float)pf_norm / n_mb; px[3] = (float)pf_tile / n_mb; px[4] = (float)pf_scal / n_mb;
      for (int i = 0; i < 4; ++i) A.loss_acc[16 + i] = (float)pf_h[i] / n_mb;      // (the profiling tool hands a 32-float buffer)
      A.loss_acc[28] = (float)pf_h[4] / n_mb; A.loss_acc[29] = (float)pf_h[5] / n_mb; (void)pf_g;
      for (int i = 0; i < 8; ++i) A.loss_acc[20 + i] = (float)pf_n[i] / n_mb;
      for (int i = 0; i < 4; ++i) A.loss_acc[30 + i] = (float)pf_a[i] / n_mb;      // (the profiling tool hands a 40-float buffer)
    }
#endif
  }
}

// 256 threads per block, dynamic LDS = ppo_lds_bytes().  Every 8th block of the grid works (the others leave at once): where
// workgroups are dealt round-robin to the 8 XCDs that puts all working blocks on ONE XCD, whose L2 then carries their
// exchanges (checked at run time, see ppo_net_body).  Working block i = blockIdx / 8 runs network i & 1 (0: policy, 1: value)
// as part i >> 1 of gridDim / 16 blocks per network (1, 2, 4 or 8).
template <int CH, int NS>
__global__ __launch_bounds__(kPThreads) void fw_ppo_update_kernel(PpoArgs A) {
  extern __shared__ __align__(16) float lds[];
  if (blockIdx.x & 7) return;
  const int i = (int)blockIdx.x >> 3;
  const int part = i >> 1, nsplit = NS ? NS : (int)gridDim.x >> 4;
  if ((i & 1) == 0) ppo_net_body<0, CH, NS>(A, lds, part, nsplit); else ppo_net_body<1, CH, NS>(A, lds, part, nsplit);
}

// How a minibatch of B samples is cut: samples per pass (64, 32 or 16) and blocks per network.  The path is sequential, so the
// smaller the share of a block the better -- down to 16 samples (one 16 x 16 row tile per wave) and up to eight blocks: from four
// on they reduce-scatter, so the exchange volume per block does not grow with their number.  `max_blocks` = 4: the cuts of round 4
// (FWSIM_PPO_RS=0 needs them: the all-to-all swap is written for up to four blocks).
struct PpoSplit { int ch, nsplit; };
inline PpoSplit ppo_split(int B, int max_blocks = kPMaxSplit) {
  auto cut = [&](int ch) { PpoSplit s; s.ch = ch; const int c = B / ch; s.nsplit = (c >= 8 && max_blocks >= 8) ? 8 : c >= 4 ? 4 : c >= 2 ? 2 : 1; return s; };
  // passes of the busiest block x what a pass costs (64 samples : 32 : 16 ~ 10 : 6 : 4), + 1 where the blocks of a network are two:
  // they swap whole gradients, four and eight swap shares
  auto cost = [&](PpoSplit s) { return ((B / s.ch + s.nsplit - 1) / s.nsplit) * (s.ch == 64 ? 10 : s.ch == 32 ? 6 : 4) + (s.nsplit == 2 ? 1 : 0); };
  PpoSplit best = cut(16);
  if (B % 32 == 0 && cost(cut(32)) <= cost(best)) best = cut(32);
  if (B % 64 == 0 && cost(cut(64)) <= cost(best)) best = cut(64);
  return best;
}

inline size_t ppo_lds_bytes(int D) {
  (void)D;                                          // (sized for the larger of the two forms: W1 as 64 rows of 65)
  const int ldx = kPLdx;
  size_t f = (size_t)(kPH * kPLdh + kPH + kPH * kPLdh + kPH + kPH * 4 + 4) + 4 + (size_t)kPChunk * ldx + 64 + 2 * (size_t)kPChunk * kPLdh +
             3 * (size_t)kPChunk * 4 + 8 * kPH + 8 + 32 + kPThreads + 2 * kPMaxSplit;
  return f * sizeof(float);
}

}  // namespace fwsim
