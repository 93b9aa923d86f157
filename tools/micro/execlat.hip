// does a dependent fp64 chain run faster when only a few lanes are enabled?  (cycles per op via s_memtime; one wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int LANES> __global__ void k(double *out, unsigned long long *cyc, double a, double b, int slot) {
  double x = a + threadIdx.x * 1e-9, y = b;
  unsigned long long t0 = clock64();
  if ((int)threadIdx.x < LANES) {
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = __fma_rn(x, y, y);
  }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[slot] = t1 - t0;
}
// two independent chains in the same wave on disjoint lanes cannot overlap (one instruction stream); two chains interleaved can
template <int LANES> __global__ void k2(double *out, unsigned long long *cyc, double a, double b, int slot) {
  double x = a + threadIdx.x * 1e-9, y = b, z = a * 0.5;
  unsigned long long t0 = clock64();
  if ((int)threadIdx.x < LANES) {
#pragma unroll 16
    for (int i = 0; i < N; ++i) { x = __fma_rn(x, y, y); z = __fma_rn(z, y, y); }
  }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = x + z;
  if (threadIdx.x == 0) cyc[slot] = t1 - t0;
}
int main() {
  double *out; unsigned long long *cyc, h[16];
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 128);
  k<64><<<1, 64>>>(out, cyc, 1.0, 0.5, 0); k<32><<<1, 64>>>(out, cyc, 1.0, 0.5, 1); k<16><<<1, 64>>>(out, cyc, 1.0, 0.5, 2);
  k<1><<<1, 64>>>(out, cyc, 1.0, 0.5, 3);
  k2<64><<<1, 64>>>(out, cyc, 1.0, 0.5, 4); k2<16><<<1, 64>>>(out, cyc, 1.0, 0.5, 5); k2<1><<<1, 64>>>(out, cyc, 1.0, 0.5, 6);
  hipDeviceSynchronize(); hipMemcpy(h, cyc, 56, hipMemcpyDeviceToHost);
  const char *nm[] = {"fma chain, 64 lanes", "fma chain, 32 lanes", "fma chain, 16 lanes", "fma chain, 1 lane", "2 chains, 64 lanes", "2 chains, 16 lanes", "2 chains, 1 lane"};
  for (int i = 0; i < 7; ++i) printf("%-22s %.2f cycles per loop step\n", nm[i], (double)h[i] / N);
  return 0;
}
