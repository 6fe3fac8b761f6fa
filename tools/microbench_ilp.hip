// Dev tool: how much ILP one wave needs to saturate fp64 issue on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void chains(double* out, int iters) {
  double x[CH];
  for (int c = 0; c < CH; ++c) x[c] = 0.3 + 1e-3 * threadIdx.x + 0.01 * c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = fma(x[c], 0.9999999, 1e-9);
  }
  double a = 0; for (int c = 0; c < CH; ++c) a += x[c];
  out[blockIdx.x * 64 + threadIdx.x] = a;
}
template <int CH> void run(double* d, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 20000;
  hipLaunchKernelGGL(chains<CH>, dim3(blocks), dim3(64), 0, 0, d, 10); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(chains<CH>, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("chains=%d blocks=%d: %.2f ns per fma-level (%.2f ns per fma)\n", CH, blocks, ms * 1e6 / iters, ms * 1e6 / iters / CH);
}
int main() {
  double* d; hipMalloc(&d, 8 * 64 * 4096);
  for (int blocks : {1, 1024, 2048}) { run<1>(d, blocks); run<2>(d, blocks); run<4>(d, blocks); run<8>(d, blocks); }
  return 0;
}
