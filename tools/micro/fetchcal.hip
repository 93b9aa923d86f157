// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the N = 32 element-view solver
// (MI355X_MICROARCH.md calibrates FETCH_SIZE only for 16-B-per-lane streams: "x 2"; everything else must be measured on a
// known byte count).  Every kernel moves exactly BYTES bytes of a buffer larger than the Infinity Cache, once:
//   wg_cal_rd4 / rd8 / rd16     coalesced streaming reads, 4 / 8 / 16 bytes per lane
//   wg_cal_rd8_cols             the Z^T a walk: a wave owns a 72 x 73 slot of doubles, lane i reads column i (and i + 64) eight
//                               consecutive doubles at a time -- the lanes of one load instruction are 584 B apart
//   wg_cal_rd8_rows             the sweep's walk of the same slot: lane i reads Z(i, c), c descending (512-B rows)
//   wg_cal_wr8 / wr16           coalesced streaming stores
//   wg_cal_wr8_rows             the sweep's stores
// Build:  hipcc --offload-arch=gfx950 -O3 -o tools/micro/fetchcal tools/micro/fetchcal.hip
// Run:    rocprofv3 --pmc FETCH_SIZE -d out -- tools/micro/fetchcal     (and a second pass with --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr size_t kBytes = 2ull << 30;        // 2 GiB: eight times the Infinity Cache
constexpr int kN = 72, kLd = 73, kSlot = kN * kLd;   // doubles per slot (the element view's Z)

template <class T>
__global__ void wg_cal_rd(const T *__restrict__ p, size_t n, double *sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (; i < n; i += stride) {
    T v = p[i];
    acc += *reinterpret_cast<const float *>(&v);
  }
  if (acc == 123.456) sink[0] = acc;
}
__global__ __launch_bounds__(64) void wg_cal_rd8_cols(const double *__restrict__ z, int slots, double *sink) {
  const int lane = threadIdx.x;
  double a0 = 0.0, a1 = 0.0;
  for (int s = blockIdx.x; s < slots; s += gridDim.x) {
    const double *zs = z + (size_t)s * kSlot;
    const double *z0 = zs + lane * kLd, *z1 = zs + (lane + 64 < kN ? lane + 64 : lane) * kLd;
    for (int j = 0; j + 8 <= kN; j += 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { a0 += z0[j + e]; a1 += z1[j + e]; }
    }
  }
  if (a0 + a1 == 123.456) sink[0] = a0;
}
__global__ __launch_bounds__(64) void wg_cal_rd8_rows(const double *__restrict__ z, int slots, double *sink) {
  const int lane = threadIdx.x;
  double a0 = 0.0, a1 = 0.0;
  for (int s = blockIdx.x; s < slots; s += gridDim.x) {
    const double *zs = z + (size_t)s * kSlot;
    const int i1 = lane + 64 < kN ? lane + 64 : lane;
    for (int c = kN - 1; c >= 0; --c) { a0 += zs[lane + c * kLd]; a1 += zs[i1 + c * kLd]; }
  }
  if (a0 + a1 == 123.456) sink[0] = a0;
}
template <class T>
__global__ void wg_cal_wr(T *__restrict__ p, size_t n, T v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
__global__ __launch_bounds__(64) void wg_cal_wr8_rows(double *__restrict__ z, int slots, double v) {
  const int lane = threadIdx.x;
  for (int s = blockIdx.x; s < slots; s += gridDim.x) {
    double *zs = z + (size_t)s * kSlot;
    const int i1 = lane + 64 < kN ? lane + 64 : lane;
    for (int c = kN - 1; c >= 0; --c) { zs[lane + c * kLd] = v; zs[i1 + c * kLd] = v; }
  }
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

int main() {
  void *buf; double *sink;
  CK(hipMalloc(&buf, kBytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 0, kBytes));
  const int grid = 256 * 16;
  const int slots = (int)(kBytes / (kSlot * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](const char *name, double bytes, auto &&launch) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-18s %14.0f bytes  %8.3f ms  %7.1f GB/s\n", name, bytes, ms, bytes / ms / 1e6);
  };
  for (int rep = 0; rep < 2; ++rep) {
    timed("wg_cal_rd<float>", (double)kBytes, [&] { hipLaunchKernelGGL(wg_cal_rd<float>, dim3(grid), dim3(256), 0, 0, (const float *)buf, kBytes / 4, sink); });
    timed("wg_cal_rd<double>", (double)kBytes, [&] { hipLaunchKernelGGL(wg_cal_rd<double>, dim3(grid), dim3(256), 0, 0, (const double *)buf, kBytes / 8, sink); });
    timed("wg_cal_rd<double2>", (double)kBytes, [&] { hipLaunchKernelGGL(wg_cal_rd<double2>, dim3(grid), dim3(256), 0, 0, (const double2 *)buf, kBytes / 16, sink); });
    // the column walk reads 72 of the 73 doubles of each of the 72 columns; surplus lanes (8..63 second column) re-read their first
    timed("wg_cal_rd8_cols", (double)slots * kN * kN * 8, [&] { hipLaunchKernelGGL(wg_cal_rd8_cols, dim3(2048), dim3(64), 0, 0, (const double *)buf, slots, sink); });
    timed("wg_cal_rd8_rows", (double)slots * kN * kN * 8, [&] { hipLaunchKernelGGL(wg_cal_rd8_rows, dim3(2048), dim3(64), 0, 0, (const double *)buf, slots, sink); });
    timed("wg_cal_wr<double>", (double)kBytes, [&] { hipLaunchKernelGGL(wg_cal_wr<double>, dim3(grid), dim3(256), 0, 0, (double *)buf, kBytes / 8, 1.0); });
    timed("wg_cal_wr<double2>", (double)kBytes, [&] { hipLaunchKernelGGL(wg_cal_wr<double2>, dim3(grid), dim3(256), 0, 0, (double2 *)buf, kBytes / 16, make_double2(1.0, 2.0)); });
    timed("wg_cal_wr8_rows", (double)slots * kN * kN * 8, [&] { hipLaunchKernelGGL(wg_cal_wr8_rows, dim3(2048), dim3(64), 0, 0, (double *)buf, slots, 3.0); });
  }
  CK(hipDeviceSynchronize());
  printf("slots %d  slot bytes %d  useful bytes per slot %d\n", slots, kSlot * 8, kN * kN * 8);
  return 0;
}
