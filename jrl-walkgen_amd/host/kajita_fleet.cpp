// kajita_fleet.cpp -- the Kajita stage-1 fleet path from plain C++ through the C ABI (no Python, no PyTorch): B step sequences
// in device memory -> wg_zmpdisc_batch_dev (ZMP reference queues, time-major) -> wg_preview_run_batch_dev (cart-table CoM),
// nothing leaves the device in between.  Gait 0 walks TestKajita2003's StraightWalking sequence and is checked against
// the host-pointer entry points on the same input (bit for bit); the others vary step length and heading.
//
//   kajita_fleet [--batch B] [--steps S]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/wg_mpc.h"

#define CHECK_HIP(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "FAILED: %s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
#define CHECK_WG(e) do { int r_ = (e); if (r_ != WG_OK) { fprintf(stderr, "FAILED: %s: %s\n", #e, wg_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
  int B = 4096, S = 16;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--batch") && i + 1 < argc) B = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) S = atoi(argv[++i]);
  }
  if (B < 1 || S < 2 || S > WG_ZMPDISC_MAX_STEPS) { fprintf(stderr, "FAILED: need B >= 1, 2 <= steps <= %d\n", WG_ZMPDISC_MAX_STEPS); return 1; }
  CHECK_WG(wg_init(0));
  wg_zmpdisc_model_t zm;
  wg_zmpdisc_defaults(&zm);                                   // 5 ms, 1.6 s preview, 0.78 / 0.02 s supports, 0.07 m step height
  // preview gains for the same sampling period / window (PreviewControl::ComputeOptimalWeights)
  const int nl = (int)(zm.preview_time / zm.T);
  double Kg[4];
  std::vector<double> F(nl);
  CHECK_WG(wg_riccati_gains(zm.T, 0.8078, 1.0, 1e-6, nl, WG_RICCATI_WITHOUT_INITIALPOS, Kg, F.data()));
  wg_preview_gains_t pg = {zm.T, 0.8078, Kg[0], {Kg[1], Kg[2], Kg[3]}, nl, 0};
  CHECK_WG(wg_preview_configure(&pg, F.data()));

  std::vector<wg_rel_step_t> steps((size_t)B * S);
  std::vector<int> n_steps(B, S);
  std::vector<double> feet((size_t)B * 6);
  for (int g = 0; g < B; ++g) {
    std::mt19937_64 rng(20100 + g);
    std::uniform_real_distribution<double> len(0.1, 0.25), turn(-5.0, 5.0);
    double side = (g & 1) ? 1.0 : -1.0;
    for (int i = 0; i < S; ++i) {
      wg_rel_step_t &s = steps[(size_t)g * S + i];
      memset(&s, 0, sizeof s);
      const bool ends = i == 0 || i == S - 1;
      s.sx = ends ? 0.0 : (g == 0 ? 0.2 : len(rng));
      s.sy = side * (i == 0 ? 0.105 : 0.21);
      s.theta = (ends || g == 0) ? 0.0 : turn(rng);
      s.ss_time = zm.t_single; s.ds_time = zm.t_double; s.step_type = 1;
      side = -side;
    }
    const double f[6] = {0.0094903, 0.095, 0.0, 0.0094903, -0.095, 0.0};
    memcpy(&feet[(size_t)g * 6], f, sizeof f);
  }
  const int L = wg_zmpdisc_length(&zm, steps.data(), S);
  if (L < nl) { fprintf(stderr, "FAILED: sequence of %d samples\n", L); return 1; }
  const int Lrun = L - nl + 1;

  wg_rel_step_t *d_steps; int *d_ns, *d_len; double *d_feet, *d_zx, *d_zy, *d_state, *d_com;
  CHECK_HIP(hipMalloc((void **)&d_steps, sizeof(wg_rel_step_t) * steps.size()));
  CHECK_HIP(hipMalloc((void **)&d_ns, sizeof(int) * B));
  CHECK_HIP(hipMalloc((void **)&d_len, sizeof(int) * B));
  CHECK_HIP(hipMalloc((void **)&d_feet, sizeof(double) * feet.size()));
  CHECK_HIP(hipMalloc((void **)&d_zx, sizeof(double) * (size_t)L * B));
  CHECK_HIP(hipMalloc((void **)&d_zy, sizeof(double) * (size_t)L * B));
  CHECK_HIP(hipMalloc((void **)&d_state, sizeof(double) * 8 * B));
  CHECK_HIP(hipMalloc((void **)&d_com, sizeof(double) * (size_t)Lrun * 6 * B));
  CHECK_HIP(hipMemcpy(d_steps, steps.data(), sizeof(wg_rel_step_t) * steps.size(), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_ns, n_steps.data(), sizeof(int) * B, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_feet, feet.data(), sizeof(double) * feet.size(), hipMemcpyHostToDevice));
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  double sec = 0.0;
  for (int rep = 0; rep < 2; ++rep) {                          // the second pass is the timed one
    CHECK_HIP(hipMemsetAsync(d_state, 0, sizeof(double) * 8 * B, st));
    CHECK_HIP(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    CHECK_WG(wg_zmpdisc_batch_dev(&zm, B, S, d_steps, d_ns, d_feet, L, d_zx, d_zy, d_len, st));
    CHECK_WG(wg_preview_run_batch_dev(B, Lrun, d_zx, d_zy, d_state, d_com, nullptr, 1, st));
    CHECK_HIP(hipStreamSynchronize(st));
    sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  // gait 0 against the host-pointer entry points
  std::vector<double> zmp((size_t)L * 2), com_h((size_t)Lrun * 6), state_h(8, 0.0), zx(L), zy(L);
  int len0 = 0;
  CHECK_WG(wg_zmpdisc_batch(&zm, 1, S, steps.data(), n_steps.data(), feet.data(), L, zmp.data(), nullptr, nullptr, nullptr, nullptr,
                            nullptr, nullptr, &len0));
  for (int l = 0; l < L; ++l) { zx[l] = zmp[2 * l]; zy[l] = zmp[2 * l + 1]; }
  CHECK_WG(wg_preview_run_batch(1, Lrun, zx.data(), zy.data(), state_h.data(), com_h.data(), nullptr, 1));
  std::vector<double> com_d((size_t)Lrun * 6 * B);
  CHECK_HIP(hipMemcpy(com_d.data(), d_com, sizeof(double) * com_d.size(), hipMemcpyDeviceToHost));
  for (int l = 0; l < Lrun; ++l)
    for (int c = 0; c < 6; ++c)
      if (com_d[((size_t)l * 6 + c) * B] != com_h[(size_t)l * 6 + c]) { fprintf(stderr, "FAILED: device chain differs from the host entry points at step %d\n", l); return 1; }
  double far = 0.0;
  for (int g = 0; g < B; ++g) { const double x = com_d[((size_t)(Lrun - 1) * 6) * B + g]; far = x > far ? x : far; }
  printf("kajita_fleet: %d walks of %d steps (%d samples each) in %.2f ms = %.0f walks/s; gait 0 ends at x = %.4f m (%d samples), "
         "farthest %.2f m; device chain == host entry points\n", B, S, L, sec * 1e3, B / sec, com_h[(size_t)(Lrun - 1) * 6], len0, far);
  wg_shutdown();
  return 0;
}
