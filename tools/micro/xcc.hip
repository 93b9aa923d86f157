#include <hip/hip_runtime.h>
__global__ void k(int *out) {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
  int n = 4096; int *d; hipMalloc(&d, n * 4);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d);
  int *h = (int *)malloc(n * 4); hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
  int cnt[16] = {0}, mism = 0;
  for (int i = 0; i < n; i++) { cnt[h[i] & 15]++; if ((h[i] & 7) != (i & 7)) mism++; }
  for (int i = 0; i < 16; i++) printf("xcc %d: %d\n", i, cnt[i]);
  printf("first 32:"); for (int i = 0; i < 32; i++) printf(" %d", h[i]); printf("\nmismatch vs blockIdx%%8: %d\n", mism);
  return 0;
}
