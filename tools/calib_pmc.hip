// Dev tool: calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters for the access widths the step
// kernel uses (8-byte-per-lane SoA loads/stores), on a known byte count.  MI355X_MICROARCH.md: FETCH_SIZE is
// only calibrated for 16 B/lane streaming; other widths must be calibrated before an absolute is trusted.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/calib_pmc tools/calib_pmc.hip
//   rocprofv3 --pmc FETCH_SIZE  --kernel-trace -d out -- tools/_build/calib_pmc
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_read8(const double* __restrict__ a, double* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  double s = 0;
  for (; i < n; i += stride) s += a[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void calib_write8(double* __restrict__ a, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) a[i] = (double)i;
}
__global__ void calib_read16(const double2* __restrict__ a, double* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  double s = 0;
  for (; i < n; i += stride) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 12345.678) out[0] = s;
}
int main() {
  const size_t n = (size_t)1 << 27;   // 1 GiB of doubles: far past L2 and the 256 MiB Infinity Cache
  double *a, *o;
  hipMalloc(&a, n * sizeof(double)); hipMalloc(&o, 8);
  hipMemset(a, 0, n * sizeof(double));
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL(calib_read8, dim3(4096), dim3(256), 0, 0, a, o, n);
    hipLaunchKernelGGL(calib_read16, dim3(4096), dim3(256), 0, 0, (const double2*)a, o, n / 2);
    hipLaunchKernelGGL(calib_write8, dim3(4096), dim3(256), 0, 0, a, n);
  }
  hipDeviceSynchronize();
  printf("calib: %zu bytes per kernel\n", n * sizeof(double));
  return 0;
}
