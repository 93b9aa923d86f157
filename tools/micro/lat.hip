// dependent-chain latency of fp64 ops on one wave (one wave per CU): cycles per op via s_memtime
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int OP> __global__ void k(double *out, unsigned long long *cyc, double a, double b) {
  double x = a + threadIdx.x * 1e-9, y = b;
  unsigned long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    if (OP == 0) x = x + y;
    if (OP == 1) x = x * y;
    if (OP == 2) x = __fma_rn(x, y, y);
    if (OP == 3) x = y / x;
    if (OP == 4) x = sqrt(x + 1.5);
    if (OP == 5) { double t = fmax(fabs(x), fabs(y)); double d1 = x / t, d2 = y / t; x = t * sqrt(d1 * d1 + d2 * d2); }
    if (OP == 6) { int lo = __builtin_amdgcn_readlane(__double2loint(x), i & 31); int hi = __builtin_amdgcn_readlane(__double2hiint(x), i & 31); x = y + __hiloint2double(hi, lo); }
  }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[OP] = t1 - t0;
}
int main() {
  double *out; unsigned long long *cyc, h[8];
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64);
  k<0><<<1, 64>>>(out, cyc, 1.0, 1e-3); k<1><<<1, 64>>>(out, cyc, 1.0, 1.0000001); k<2><<<1, 64>>>(out, cyc, 1.0, 0.5);
  k<3><<<1, 64>>>(out, cyc, 1.3, 1.7); k<4><<<1, 64>>>(out, cyc, 1.3, 1.7); k<5><<<1, 64>>>(out, cyc, 0.3, 0.4); k<6><<<1, 64>>>(out, cyc, 0.3, 0.4);
  hipDeviceSynchronize(); hipMemcpy(h, cyc, 56, hipMemcpyDeviceToHost);
  const char *nm[] = {"add", "mul", "fma", "div", "sqrt(+add)", "givens_norm", "readlane+add"};
  for (int i = 0; i < 7; ++i) printf("%-14s %.1f cycles/op\n", nm[i], (double)h[i] / N);
  return 0;
}
