// Dev tool: what does a batch of sc1 (past-the-L1) buffer loads of data ANOTHER CU of the same XCD has just written cost one
// workgroup of 256 threads -- by instruction count, width and batching?  fw_ppo_update's exchanges are exactly this shape.
// Pairs of workgroups (b, b + 8: one XCD where workgroups are dealt round-robin) ping-pong: the writer rewrites a buffer with plain
// stores, drains (vmcnt(0)), raises a flag; the reader polls the flag, then issues N sc1 loads per thread and stamps the cycle
// counter from first issue to last data; roles swap every round.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/microbench_sc1 tools/microbench_sc1.hip && tools/_build/microbench_sc1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

template <int N, int W /* dwords per lane: 1, 2, 4 */, int STRIDE_KB /* distance between the N loads of a thread */, int BATCH /* loads between two waits */, int DELAY /* s_sleep units between flag and loads */>
__global__ __launch_bounds__(256) void k(float* buf, unsigned long long* flags, long long* out, int reps) {
  if (blockIdx.x & 7) return;                       // every 8th block works: one XCD
  const int i8 = blockIdx.x >> 3, pair = i8 >> 1, side = i8 & 1;
  const int t = threadIdx.x;
  float* mine = buf + (size_t)pair * (1 << 20);
  unsigned long long* fl = flags + pair * 16;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 1 << 22, 0x00020000);
  long long tot = 0, tot_issue = 0;
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    if ((r & 1) == side) {                          // writer
      for (int i = 0; i < N; ++i)
        for (int w = 0; w < W; ++w) mine[i * STRIDE_KB * 256 + t * W + w] = (float)(r + i + w);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      if (t == 0) {
        __hip_atomic_store(fl, (unsigned long long)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(fl + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)(r + 1)) {}
      }
      __syncthreads();
    } else {                                        // reader
      if (t == 0) while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)(r + 1)) {}
      __syncthreads();
      for (int d = 0; d < DELAY; ++d) __builtin_amdgcn_s_sleep(16);
      const long long t0 = __builtin_readcyclecounter();
      unsigned v[N][4];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int off = (i * STRIDE_KB * 256 + t * W) * 4;
        if constexpr (W == 4) { const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16); v[i][0] = a[0]; v[i][1] = a[1]; v[i][2] = a[2]; v[i][3] = a[3]; }
        else if constexpr (W == 2) { const u2 a = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 16); v[i][0] = a[0]; v[i][1] = a[1]; v[i][2] = 0; v[i][3] = 0; }
        else { v[i][0] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 16); v[i][1] = v[i][2] = v[i][3] = 0; }
        if ((i + 1) % BATCH == 0) __builtin_amdgcn_s_waitcnt(0x0F70);
      }
      const long long t1 = __builtin_readcyclecounter();      // (issue alone: the TA takes a wave-level 8- / 16-byte load in 16 cycles, a 4-byte one in 4)
      __builtin_amdgcn_s_waitcnt(0x0F70);                   // every load has returned
      __syncthreads();
      tot += __builtin_readcyclecounter() - t0; tot_issue += t1 - t0;
#pragma unroll
      for (int i = 0; i < N; ++i) acc += v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
      if (t == 0) __hip_atomic_store(fl + 8, (unsigned long long)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  if (t == 0) { out[i8 * 2] = tot / (reps / 2); out[i8 * 2 + 1] = tot_issue / (reps / 2) + (acc == 12345u); }
}

template <int N, int W, int S, int BATCH = 64, int DELAY = 0> void run(const char* name, float* buf, unsigned long long* flags, long long* out, int pairs) {
  hipMemset(flags, 0, 4096);
  hipLaunchKernelGGL((k<N, W, S, BATCH, DELAY>), dim3(pairs * 16), dim3(256), 0, 0, buf, flags, out, 400);
  hipDeviceSynchronize();
  std::vector<long long> h(4 * pairs);
  hipMemcpy(h.data(), out, sizeof(long long) * 4 * pairs, hipMemcpyDeviceToHost);
  long long mx = 0, mn = 1ll << 60, is = 0;
  for (int b = 0; b < 2 * pairs; ++b) { mx = std::max(mx, h[2 * b]); mn = std::min(mn, h[2 * b]); is = std::max(is, h[2 * b + 1]); }
  const int lines = N * 4 * ((64 * W * 4 + 127) / 128);
  printf("%-40s pairs %d: %6lld .. %6lld cycles to the last data, %5lld to issue (%2d load instr / wave, %4d lines / block -> %5.1f cycles per instr, %4.1f per line)\n", name, pairs, mn, mx, is, N, lines,
         (double)mx / N, (double)mx / lines);
}

int main() {
  float* buf; unsigned long long* flags; long long* out;
  hipMalloc(&buf, (size_t)16 << 22); hipMalloc(&out, 4096); hipMalloc(&flags, 4096);
  hipMemset(buf, 0, (size_t)16 << 22);
  for (int pairs : {1, 8}) {
    run<1, 4, 1>("1 x b128", buf, flags, out, pairs);
    run<4, 4, 1>("4 x b128", buf, flags, out, pairs);
    run<8, 4, 1>("8 x b128", buf, flags, out, pairs);
    run<16, 4, 1>("16 x b128", buf, flags, out, pairs);
    run<32, 4, 1>("32 x b128", buf, flags, out, pairs);
    run<8, 2, 1>("8 x b64", buf, flags, out, pairs);
    run<16, 2, 1>("16 x b64", buf, flags, out, pairs);
    run<32, 2, 1>("32 x b64", buf, flags, out, pairs);
    run<16, 1, 1>("16 x b32", buf, flags, out, pairs);
    run<32, 1, 1>("32 x b32", buf, flags, out, pairs);
    run<8, 4, 34>("8 x b128, 34 KB apart", buf, flags, out, pairs);
    run<16, 2, 34>("16 x b64, 34 KB apart", buf, flags, out, pairs);
    run<16, 4, 1, 8>("16 x b128 in batches of 8", buf, flags, out, pairs);
    run<32, 2, 1, 8>("32 x b64 in batches of 8", buf, flags, out, pairs);
    run<16, 4, 1, 64, 4>("16 x b128, ~1 k cycles after the flag", buf, flags, out, pairs);
    run<16, 4, 1, 64, 16>("16 x b128, ~4 k cycles after the flag", buf, flags, out, pairs);
    run<32, 2, 1, 64, 16>("32 x b64, ~4 k cycles after the flag", buf, flags, out, pairs);
  }
  return 0;
}
