// Dev tool: the reduce-scatter fetch of fw_ppo_update in isolation.  NB workgroups on one XCD (every 8th block of the grid); per round
// every block rewrites its own 32-KB "partial" (8 x 4 KB shares), all meet at a counter, then every block fetches ITS share of every
// partial (8 wave-level 16-byte sc1 loads per wave) and stamps the cycle counter from first issue to last data.  Variants: the order in
// which the partials are walked (all blocks 0, 1, 2 ... together, or block p at partial i ^ p), the distance between partials.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/microbench_rs tools/microbench_rs.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int NB, bool XOR_ORDER, int STRIDE_KB>
__global__ __launch_bounds__(256) void k(float* buf, unsigned long long* ctr, long long* out, int reps) {
  if (blockIdx.x & 7) return;
  const int part = blockIdx.x >> 3, t = threadIdx.x;
  if (part >= NB) return;
  float* mine = buf + (size_t)part * STRIDE_KB * 256;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 64 << 20, 0x00020000);
  long long tot = 0; unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(mine + i * 1024 + t * 4) = make_float4((float)r, (float)i, (float)t, 1.f);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (t == 0) {
      __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)NB * (r + 1)) {}
    }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    u4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int src = XOR_ORDER ? ((i ^ part) & 7) : i;                  // (8 partials: with 16 blocks two "networks" of 8 share them pairwise)
      const int base = (NB > 8 && part >= 8) ? 8 * STRIDE_KB * 1024 : 0;  // second network: its own eight partials
      v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + src * STRIDE_KB * 1024 + (part & 7) * 4096 + t * 16, 0, 16);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    tot += __builtin_readcyclecounter() - t0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i][0] ^ v[i][3];
  }
  if (t == 0) { out[part * 2] = tot / reps; out[part * 2 + 1] = acc; }
}

template <int NB, bool X, int S> void run(const char* name, float* buf, unsigned long long* ctr, long long* out) {
  hipMemset(ctr, 0, 64);
  hipLaunchKernelGGL((k<NB, X, S>), dim3(NB * 8), dim3(256), 0, 0, buf, ctr, out, 400);
  hipDeviceSynchronize();
  std::vector<long long> h(2 * NB);
  hipMemcpy(h.data(), out, sizeof(long long) * 2 * NB, hipMemcpyDeviceToHost);
  long long mx = 0, mn = 1ll << 60;
  for (int b = 0; b < NB; ++b) { mx = std::max(mx, h[2 * b]); mn = std::min(mn, h[2 * b]); }
  printf("%-60s %2d blocks: %5lld .. %5lld cycles from first issue to last data\n", name, NB, mn, mx);
}

int main() {
  float* buf; unsigned long long* ctr; long long* out;
  hipMalloc(&buf, 64 << 20); hipMalloc(&ctr, 64); hipMalloc(&out, 4096);
  hipMemset(buf, 0, 64 << 20);
  run<4, false, 34>("same order, partials 34 KB apart", buf, ctr, out);
  run<4, true, 34>("XOR order", buf, ctr, out);
  run<8, false, 34>("same order, partials 34 KB apart", buf, ctr, out);
  run<8, true, 34>("XOR order", buf, ctr, out);
  run<16, false, 34>("same order (two networks of eight)", buf, ctr, out);
  run<16, true, 34>("XOR order (two networks of eight)", buf, ctr, out);
  run<8, true, 43>("XOR order, 43 KB apart", buf, ctr, out);
  run<8, true, 64>("XOR order, 64 KB apart", buf, ctr, out);
  run<8, true, 260>("XOR order, 260 KB apart", buf, ctr, out);
  run<16, true, 260>("XOR order, 260 KB apart (two networks)", buf, ctr, out);
  run<8, false, 260>("same order, 260 KB apart", buf, ctr, out);
  return 0;
}
