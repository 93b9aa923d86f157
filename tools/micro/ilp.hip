// Can ONE wave overlap independent fp64 VALU work with a dependent chain?  Cycles per instruction (clock64; the s_memtime
// counter ticks at 100 MHz on this part, so the ratio between the variants is what matters) for
//   dep1  : one dependent chain of fp64 adds
//   ind4  : four independent chains interleaved (4x the instructions)
//   ind4f : the same with fp32 adds
// and the same three with TWO waves on the SIMD (blockDim 512 = 8 waves per CU = 2 per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 8192
template <typename T, int CH> __global__ void k(T *out, unsigned long long *cyc, T a, T b, int slot) {
  T x0 = a + threadIdx.x * (T)1e-6, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y = b;
  unsigned long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    x0 = x0 + y;
    if (CH > 1) { x1 = x1 + y; x2 = x2 + y; x3 = x3 + y; }
  }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
}
int main() {
  double *out; float *outf; unsigned long long *cyc, h[8];
  hipMalloc(&out, 512 * 8); hipMalloc(&outf, 512 * 4); hipMalloc(&cyc, 64);
  k<double, 1><<<1, 64>>>(out, cyc, 1.0, 1e-3, 0);
  k<double, 4><<<1, 64>>>(out, cyc, 1.0, 1e-3, 1);
  k<float, 4><<<1, 64>>>(outf, cyc, 1.0f, 1e-3f, 2);
  k<double, 1><<<1, 512>>>(out, cyc, 1.0, 1e-3, 3);
  k<double, 4><<<1, 512>>>(out, cyc, 1.0, 1e-3, 4);
  k<float, 4><<<1, 512>>>(outf, cyc, 1.0f, 1e-3f, 5);
  hipDeviceSynchronize(); hipMemcpy(h, cyc, 48, hipMemcpyDeviceToHost);
  const char *nm[] = {"1 wave/SIMD  f64 1 chain ", "1 wave/SIMD  f64 4 chains", "1 wave/SIMD  f32 4 chains", "2 waves/SIMD f64 1 chain ",
                      "2 waves/SIMD f64 4 chains", "2 waves/SIMD f32 4 chains"};
  const int ops[] = {1, 4, 4, 1, 4, 4};
  for (int i = 0; i < 6; ++i) printf("%s: %.2f ticks per loop trip, %.2f per instruction\n", nm[i], (double)h[i] / N, (double)h[i] / N / ops[i]);
  return 0;
}
