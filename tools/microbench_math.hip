// Dev tool (not product): single-wave latency and full-chip throughput of the fp64 math
// building blocks used by the fixed-wing tick, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_math tools/microbench_math.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = fma(fma(-d, x, 1.0), x, x);
  x = fma(fma(-d, x, 1.0), x, x);
  return x;
}
__device__ __forceinline__ double fast_div(double n, double d) {
  double x = fast_rcp(d);
  double q = n * x;
  return fma(fma(-d, q, n), x, q);
}
__device__ __forceinline__ double fast_sqrt(double a) {        // rsq + 2 NR (Goldschmidt-ish)
  double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  return (a == 0.0) ? 0.0 : g;
}
// bounded-range sincos: |x| <~ 1e4, Cody-Waite with fma + fdlibm kernels
__device__ __forceinline__ void my_sincos(double x, double* s, double* c) {
  const double INV_PIO2 = 6.36619772367581382433e-01;
  const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
  double k = rint(x * INV_PIO2);
  double r = fma(-k, PIO2_HI, x);
  r = fma(-k, PIO2_LO, r);
  double z = r * r;
  // sin kernel
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  double sn = fma(z * r, ps, r);
  // cos kernel
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  double cs = fma(z * z, pc, fma(z, -0.5, 1.0));
  int q = (int)k;
  double s0 = (q & 1) ? cs : sn, c0 = (q & 1) ? sn : cs;
  *s = (q & 2) ? -s0 : s0;
  *c = ((q + 1) & 2) ? -c0 : c0;
}

enum { OP_FMA, OP_DIV, OP_FDIV, OP_SQRT, OP_FSQRT, OP_SINCOS, OP_MYSINCOS, OP_ATAN2, OP_ASIN, OP_LOG, OP_SIN_F32, OP_COUNT };
const char* names[] = {"fma", "div(ieee)", "div(rcp+NR)", "sqrt(ieee)", "sqrt(rsq+NR)", "sincos(ocml)", "sincos(custom)", "atan2(ocml)", "asin(ocml)", "log(ocml)", "sincosf(ocml f32)"};

template <int OP>
__global__ void bench(double* out, int iters, double seed) {
  double x = seed + 1e-3 * threadIdx.x, acc = 0.0;
  for (int i = 0; i < iters; ++i) {
    double y;
    if (OP == OP_FMA) y = fma(x, 1.0000001, 1e-9);
    else if (OP == OP_DIV) y = 1.0 / (x + 1.5);
    else if (OP == OP_FDIV) y = fast_div(1.0, x + 1.5);
    else if (OP == OP_SQRT) y = sqrt(x + 2.0);
    else if (OP == OP_FSQRT) y = fast_sqrt(x + 2.0);
    else if (OP == OP_SINCOS) { double s, c; sincos(x, &s, &c); y = s + 0.5 * c; }
    else if (OP == OP_MYSINCOS) { double s, c; my_sincos(x, &s, &c); y = s + 0.5 * c; }
    else if (OP == OP_ATAN2) y = atan2(x, 0.7 + acc * 1e-30);
    else if (OP == OP_ASIN) y = asin(0.9 * x / (1.0 + fabs(x)));
    else if (OP == OP_LOG) y = log(x + 2.0);
    else { float s, c; sincosf((float)x, &s, &c); y = s + 0.5f * c; }
    acc += y;
    x = y * 0.999 + 0.1;      // dependent chain
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + x;
}

template <int OP> float run(int blocks, int iters, double* d) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(64), 0, 0, d, 10, 0.3);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.3);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

__global__ void accuracy(double* err) {
  double worst = 0;
  for (int i = 0; i < 20000; ++i) {
    double x = -8.0 + 16.0 * (i + threadIdx.x / 64.0) / 20000.0;
    double s, c, s2, c2; sincos(x, &s, &c); my_sincos(x, &s2, &c2);
    worst = fmax(worst, fmax(fabs(s - s2), fabs(c - c2)));
    double a = 0.3 + x * x, b = 1.7 + fabs(x);
    worst = fmax(worst, fabs(fast_div(a, b) - a / b) / (a / b));
    worst = fmax(worst, fabs(fast_sqrt(a) - sqrt(a)) / sqrt(a));
  }
  err[threadIdx.x] = worst;
}

int main() {
  double* d; CHECK(hipMalloc(&d, sizeof(double) * 64 * 4096));
  const int iters = 2000;
  printf("%-20s %14s %18s\n", "op (dependent chain)", "1 wave: ns/op", "2048 waves: ns/op");
#define ROW(OP) { float l = run<OP>(1, iters, d), t = run<OP>(2048, iters, d); printf("%-20s %14.1f %18.1f\n", names[OP], l * 1e6 / iters, t * 1e6 / iters); }
  ROW(OP_FMA) ROW(OP_DIV) ROW(OP_FDIV) ROW(OP_SQRT) ROW(OP_FSQRT) ROW(OP_SINCOS) ROW(OP_MYSINCOS) ROW(OP_ATAN2) ROW(OP_ASIN) ROW(OP_LOG) ROW(OP_SIN_F32)
  hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(64); CHECK(hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost));
  double w = 0; for (double v : h) w = fmax(w, v);
  printf("custom sincos/div/sqrt worst abs/rel error vs ocml on [-8,8]: %.3e\n", w);
  return 0;
}
