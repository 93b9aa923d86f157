// What does a phase hand-over inside a multi-wave workgroup cost?  (docs/HISTORY.md 7's estimate for the N = 32 "Z in LDS, several waves
// per gait" layout rested on an unmeasured s_barrier + LDS hand-over; this measures it at that layout's residency.)
//
// One workgroup = one "gait" = W waves (1, 2, 4), 52 KB of LDS each so that exactly THREE workgroups share a CU (Z 41.5 KB + the
// rest), every CU of the chip busy (grid = 3 x CUs).  Per "phase":
//   wave 0 runs a dependent chain of CH fp64 adds (the stand-in for a serial chain of the solver), stores its result to LDS;
//   hand-over; every wave reads it and folds it into its own lane-parallel work (PW independent adds per lane).
// Hand-over variants:
//   none   : W = 1 only -- the same instructions with no synchronisation at all (the baseline the shipped one-wave layout pays)
//   barrier: __syncthreads() (s_waitcnt + s_barrier) after the store, and a second one after the parallel part (the chain's next
//            link must not overwrite what a slow wave has not read yet: two barriers per phase is what ql_solve would need)
//   flag   : wave 0 bumps a sequence number in LDS after the store; the others spin on it (ds_read + s_sleep 0); they report back
//            through a per-wave arrival counter wave 0 spins on -- hand-over without s_barrier
// Time per phase = kernel time / (phases); the cost of a hand-over = that minus the W = 1 / none figure.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/barrier tools/micro/barrier.hip && tools/micro/barrier
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

template <int W, int MODE, int CH, int PW>   // MODE 0 none, 1 barrier, 2 flag
__global__ __launch_bounds__(64 * W) void k(double *out, int phases, double a, double b) {
  extern __shared__ double lds[];
  volatile double *slot = lds;                 // the handed-over value
  volatile int *seq = reinterpret_cast<volatile int *>(lds + 8);    // flag mode: producer's sequence number
  volatile int *arr = reinterpret_cast<volatile int *>(lds + 16);   // flag mode: arrival counters, one per wave
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < 64) { lds[threadIdx.x] = 0.0; }
  __syncthreads();
  double chain = a + 1e-9 * blockIdx.x, acc[4] = {b, b + 1, b + 2, b + 3};
  for (int p = 1; p <= phases; ++p) {
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < CH; ++i) chain = chain + b;            // dependent: 8-cycle adds back to back
      if (lane == 0) *slot = chain;
      if (MODE == 2) { __builtin_amdgcn_s_waitcnt(0xc07f); if (lane == 0) *seq = p; }
    }
    if (MODE == 1) __syncthreads();
    if (MODE == 2 && wave != 0) { while (*seq < p) __builtin_amdgcn_s_sleep(0); }
    const double v = *slot;
#pragma unroll
    for (int i = 0; i < PW; ++i) acc[i & 3] = acc[i & 3] + v;    // lane-parallel part: four independent chains
    if (MODE == 1) __syncthreads();
    if (MODE == 2) {
      if (wave != 0) { if (lane == 0) arr[wave] = p; }
      else for (int w = 1; w < W; ++w) while (arr[w] < p) __builtin_amdgcn_s_sleep(0);
    }
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = chain + acc[0] + acc[1] + acc[2] + acc[3];
}

template <int W, int MODE, int CH, int PW>
double run(int grid, int phases, double *out) {
  const size_t lds = 52 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<W, MODE, CH, PW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<W, MODE, CH, PW><<<grid, 64 * W, lds>>>(out, 64, 1.0, 1e-3);      // warm
  CHECK(hipEventRecord(e0));
  k<W, MODE, CH, PW><<<grid, 64 * W, lds>>>(out, phases, 1.0, 1e-3);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return 1e6 * ms / phases;                     // ns per phase
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, grid = 3 * cus, phases = 20000;
  const double ghz = prop.clockRate * 1e-6;
  double *out;
  CHECK(hipMalloc(&out, (size_t)grid * 256 * 8));
  printf("# %s, %d CUs, %.2f GHz; grid = %d workgroups (3 per CU by LDS), %d phases; ns per phase (cycles at the shader clock)\n",
         prop.name, cus, ghz, grid, phases);
  printf("# a phase = a dependent chain of CH fp64 adds on wave 0 + hand-over + PW adds per lane on every wave\n");
#define ROW(W, MODE, CH, PW, name) do { const double ns = run<W, MODE, CH, PW>(grid, phases, out); \
    printf("W=%d %-8s CH=%-3d PW=%-3d : %8.1f ns  (%7.0f cycles)\n", W, name, CH, PW, ns, ns * ghz); } while (0)
  ROW(1, 0, 16, 16, "none");
  ROW(1, 1, 16, 16, "barrier");
  ROW(2, 1, 16, 16, "barrier");
  ROW(4, 1, 16, 16, "barrier");
  ROW(2, 2, 16, 16, "flag");
  ROW(4, 2, 16, 16, "flag");
  ROW(1, 0, 64, 16, "none");
  ROW(4, 1, 64, 16, "barrier");
  ROW(4, 2, 64, 16, "flag");
  ROW(1, 0, 0, 0, "none");
  ROW(1, 1, 0, 0, "barrier");
  ROW(2, 1, 0, 0, "barrier");
  ROW(4, 1, 0, 0, "barrier");
  ROW(2, 2, 0, 0, "flag");
  ROW(4, 2, 0, 0, "flag");
  return 0;
}
