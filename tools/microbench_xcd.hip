// Dev tool: what a hand-off between two workgroups costs on MI355X, by where the two run and how the words travel.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/microbench_xcd tools/microbench_xcd.hip && tools/_build/microbench_xcd
// 1. which XCD (hardware register XCC_ID) workgroup b of a launch lands on -- the round-robin (b mod 8) every XCD-aware map in
//    this package assumes for speed;
// 2. ping-pong of one 64-bit word between workgroup 0 and workgroup k (k = 8: same XCD, k = 1: the next XCD), with
//      "agent"     relaxed agent-scope atomics (sc1: coherent across XCDs -- what the in-grid hand-offs of round 3 use),
//      "workgroup" relaxed workgroup-scope atomics (sc0: bypass the CU's L1, coherent in the XCD's L2 only);
//    the workgroup-scope pair across two XCDs is expected NOT to see each other (bounded spin -> "no hand-off").
// Timed with s_memrealtime (100 MHz): ns per one-way hop = round trip / 2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

__device__ __forceinline__ unsigned xcc_id() {
  // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, width 4)
  return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
}

__global__ void where(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

template <int SCOPE>
__device__ __forceinline__ unsigned long long ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE>
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }

// workgroup 0 writes a[i] = i and waits for b[i] == i; workgroup `peer` waits for a[i] == i and writes b[i] = i.
template <int SCOPE>
__global__ void pingpong(unsigned long long* a, unsigned long long* b, int peer, int iters, long long* out, unsigned* xcc) {
  if (threadIdx.x != 0) return;
  const int me = blockIdx.x;
  if (me != 0 && me != peer) return;
  xcc[me == 0 ? 0 : 1] = xcc_id();
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  int ok = 1;
  for (int i = 1; i <= iters && ok; ++i) {
    if (me == 0) {
      st<SCOPE>(a, (unsigned long long)i);
      long long spins = 0;
      while (ld<SCOPE>(b) != (unsigned long long)i) if (++spins > 2000000) { ok = 0; break; }
    } else {
      long long spins = 0;
      while (ld<SCOPE>(a) != (unsigned long long)i) if (++spins > 2000000) { ok = 0; break; }
      if (ok) st<SCOPE>(b, (unsigned long long)i);
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
  if (me == 0) { out[0] = t1 - t0; out[1] = ok; }
}

// a payload of `words` 64-bit words behind the flag: producer stores payload, drains, flags; consumer sees the flag, loads payload
template <int SCOPE>
__global__ void handoff(unsigned long long* flag, unsigned long long* back, unsigned long long* data, int words, int peer, int iters, long long* out) {
  const int me = blockIdx.x, t = threadIdx.x;
  if (me != 0 && me != peer) return;
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  unsigned long long acc = 0;
  for (int i = 1; i <= iters; ++i) {
    if (me == 0) {
      for (int w = t; w < words; w += 64) st<SCOPE>(data + w, (unsigned long long)i * 1000 + w);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) { st<SCOPE>(flag, (unsigned long long)i); long long s = 0; while (ld<SCOPE>(back) != (unsigned long long)i) if (++s > 2000000) break; }
      __syncthreads();
    } else {
      if (t == 0) { long long s = 0; while (ld<SCOPE>(flag) != (unsigned long long)i) if (++s > 2000000) break; }
      __syncthreads();
      for (int w = t; w < words; w += 64) acc += ld<SCOPE>(data + w) - ((unsigned long long)i * 1000 + w);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) st<SCOPE>(back, (unsigned long long)i);
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
  if (me == 0 && t == 0) out[0] = t1 - t0;
  if (me == peer && t == 0) out[2] = (long long)acc;          // 0 if every payload word arrived intact
}

// 3. A reader that HAS a line in its vector L1 and a writer on another CU of the same XCD (workgroups 0 and 8) that changes it
//    with a PLAIN store (then s_waitcnt vmcnt(0)): which kind of re-read sees the change?
//      mode 0: plain loads                       (expected: never -- the L1 copy is served)
//      mode 1: `buffer_inv sc0` + plain loads    (the workgroup-scope invalidate)
//      mode 2: `buffer_inv sc1` + plain loads    (the agent-scope invalidate)
//      mode 3: loads with sc1 (device scope)     (past the L1, served by the shared L2)
//    The reader first reads the word (old value, now in its L1), raises a flag (agent scope); the writer then stores the new
//    value; the reader re-reads in the chosen way for a bounded number of rounds.
__global__ void visible(unsigned* data, unsigned* flag, int mode, int peer, long long* out) {
  if (threadIdx.x != 0) return;
  const int me = blockIdx.x;
  if (me != 0 && me != peer) return;
  if (me == 0) {                                    // reader
    unsigned v;                                     // a plain load: the line is in my L1 now
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(data) : "memory");
    __hip_atomic_store(flag, 1u + v * 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    long long seen = -1;
    for (int it = 0; it < 200000; ++it) {
      unsigned x;
      if (mode == 0) { asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(data) : "memory"); }
      else if (mode == 1) { asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(data) : "memory"); }
      else if (mode == 2) { asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(data) : "memory"); }
      else { asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(data) : "memory"); }
      if (x == 0xABCDu) { seen = (long long)__builtin_amdgcn_s_memrealtime() - t0; break; }
    }
    out[0] = seen;
  } else {                                          // writer
    long long spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) if (++spins > 2000000) break;
    *data = 0xABCDu;                                // plain store: through my L1 into the XCD's L2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const bool only_visible = argc > 1 && std::string(argv[1]) == "visible";      // section 3 alone (the others take minutes)
  long long* out; hipMalloc(&out, 64);
  if (!only_visible) {
  unsigned* dx; hipMalloc(&dx, 64 * sizeof(unsigned));
  hipLaunchKernelGGL(where, dim3(32), dim3(64), 0, 0, dx); hipDeviceSynchronize();
  std::vector<unsigned> hx(32); hipMemcpy(hx.data(), dx, 32 * sizeof(unsigned), hipMemcpyDeviceToHost);
  printf("XCC_ID of workgroups 0..31:");
  bool rr = true;
  for (int b = 0; b < 32; ++b) { printf(" %u", hx[b]); rr = rr && (hx[b] == hx[b % 8]); }
  printf("\n  -> workgroup b and b + 8 share an XCD: %s\n", rr ? "yes" : "NO");
  unsigned long long* w; hipMalloc(&w, 4096 * 8);
  unsigned* xc; hipMalloc(&xc, 16);
  const int iters = 2000;
  for (int peer : {8, 1}) {
    for (int scope = 0; scope < 2; ++scope) {
      hipMemset(w, 0, 4096 * 8); hipMemset(out, 0, 64);
      if (scope == 0) hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_AGENT>, dim3(16), dim3(64), 0, 0, w, w + 64, peer, iters, out, xc);
      else hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(16), dim3(64), 0, 0, w, w + 64, peer, iters, out, xc);
      hipDeviceSynchronize();
      long long h[2]; unsigned x[2];
      hipMemcpy(h, out, 16, hipMemcpyDeviceToHost); hipMemcpy(x, xc, 8, hipMemcpyDeviceToHost);
      printf("ping-pong workgroup 0 (XCD %u) <-> %d (XCD %u), %s-scope words: ", x[0], peer, x[1], scope == 0 ? "agent" : "workgroup");
      if (h[1]) printf("%.0f ns per one-way hop\n", h[0] * 10.0 / iters / 2);
      else printf("no hand-off (spin ran out)\n");
    }
  }
  for (int peer : {8, 1}) {
    for (int scope = 0; scope < 2; ++scope) {
      if (peer == 1 && scope == 1) continue;                    // (not coherent across XCDs: skipped)
      for (int words : {64, 512, 3072}) {
        hipMemset(w, 0, 4096 * 8); hipMemset(out, 0, 64);
        if (scope == 0) hipLaunchKernelGGL(handoff<__HIP_MEMORY_SCOPE_AGENT>, dim3(16), dim3(64), 0, 0, w, w + 8, w + 64, words, peer, 500, out);
        else hipLaunchKernelGGL(handoff<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(16), dim3(64), 0, 0, w, w + 8, w + 64, words, peer, 500, out);
        hipDeviceSynchronize();
        long long h[3]; hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
        printf("payload hand-off 0 -> %d, %s scope, %4d x 8 B: %.0f ns per round (store, drain, flag, load, ack)%s\n", peer, scope == 0 ? "agent" : "workgroup",
               words, h[0] * 10.0 / 500, h[2] == 0 ? "" : "  PAYLOAD MISMATCH");
      }
    }
  }
  }
  {
    unsigned* d; hipMalloc(&d, 4096);
    const char* names[4] = {"plain loads", "buffer_inv sc0 + plain loads", "buffer_inv sc1 + plain loads", "sc1 loads"};
    for (int peer : {8, 1}) {
      for (int mode = 0; mode < 4; ++mode) {
        hipMemset(d, 0, 4096); hipMemset(out, 0, 64);
        hipLaunchKernelGGL(visible, dim3(16), dim3(64), 0, 0, d, d + 64, mode, peer, out);
        hipDeviceSynchronize();
        long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        printf("plain store by workgroup %d, reader workgroup 0 with the line in its L1, %s: ", peer, names[mode]);
        if (h >= 0) printf("seen after %.0f ns\n", h * 10.0); else printf("NEVER seen (200000 rounds)\n");
      }
    }
  }
  return 0;
}
