// fleet_bench.cpp -- the fleet path from plain C++ through the C ABI (no Python, no PyTorch): B gaits resident in device
// memory, velocity references redrawn every 50 ticks, ticks advanced with wg_mpc_run_batch_dev (device-side work queue)
// or, with --per-tick, one wg_mpc_tick_batch_dev launch per tick.  Prints MPC ticks/s.  Same workload as bench.py
// (std::mt19937_64 seeded 20100 + gait index; bench.py draws the same distribution through numpy).
//
//   fleet_bench [--batch B] [--ticks K] [--per-tick]
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
  int B = 4096, K = 200;
  bool per_tick = false;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--batch") && i + 1 < argc) B = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--ticks") && i + 1 < argc) K = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--per-tick")) per_tick = true;
  }
  const int REDRAW = 50, W = 50;
  CHECK_WG(wg_init(0));
  wg_model_t model;
  wg_model_defaults(&model);
  CHECK_WG(wg_mpc_configure(&model));

  std::vector<wg_gait_state_t> host(B);
  const double com0[3] = {0.0316055, 0.0, 0.7116911}, lf[3] = {0.0, 0.09, 0.0}, rf[3] = {0.0, -0.09, 0.0};
  for (int g = 0; g < B; ++g) { wg_gait_init(&model, &host[g], com0, lf, rf); host[g].nb_steps_left = 2; }
  const int n_seg = (W + K + REDRAW - 1) / REDRAW;
  std::vector<double> vref((size_t)n_seg * B * 3);
  for (int g = 0; g < B; ++g) {
    std::mt19937_64 rng(20100 + g);
    std::uniform_real_distribution<double> ux(-0.1, 0.3), uy(-0.1, 0.1), uw(-0.2, 0.2);
    for (int s = 0; s < n_seg; ++s) {
      double *v = &vref[((size_t)s * B + g) * 3];
      v[0] = ux(rng); v[1] = uy(rng); v[2] = uw(rng);
    }
  }
  wg_gait_state_t *d_states = nullptr; double *d_vref = nullptr; int *d_diag = nullptr;
  CHECK_HIP(hipMalloc((void **)&d_states, sizeof(wg_gait_state_t) * B));
  CHECK_HIP(hipMalloc((void **)&d_vref, sizeof(double) * vref.size()));
  CHECK_HIP(hipMalloc((void **)&d_diag, sizeof(int) * 6 * (size_t)B * REDRAW));
  CHECK_HIP(hipMemcpy(d_states, host.data(), sizeof(wg_gait_state_t) * B, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_vref, vref.data(), sizeof(double) * vref.size(), hipMemcpyHostToDevice));
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));

  auto advance = [&](int t0, int t1) -> int {           // ticks [t0, t1)
    for (int t = t0; t < t1;) {
      if (t % REDRAW == 0) CHECK_WG(wg_mpc_set_velref_dev(B, d_states, d_vref + (size_t)(t / REDRAW) * B * 3, st));
      const int adv = t == 0 ? 1 : (t == 1 ? 19 : 20);
      int n = 1;
      if (t >= 2 && !per_tick) { n = (t / REDRAW + 1) * REDRAW - t; if (t + n > t1) n = t1 - t; }
      if (n == 1) CHECK_WG(wg_mpc_tick_batch_dev(B, d_states, nullptr, d_diag, adv, nullptr, 0, nullptr, st));
      else CHECK_WG(wg_mpc_run_batch_dev(B, d_states, n, adv, nullptr, d_diag, st));
      t += n;
    }
    return 0;
  };
  if (advance(0, W)) return 1;
  CHECK_HIP(hipStreamSynchronize(st));
  const auto t0 = std::chrono::steady_clock::now();
  if (advance(W, W + K)) return 1;
  CHECK_HIP(hipStreamSynchronize(st));
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  CHECK_HIP(hipMemcpy(host.data(), d_states, sizeof(wg_gait_state_t) * B, hipMemcpyDeviceToHost));
  long long ticks = 0; int running = 0;
  for (int g = 0; g < B; ++g) { ticks += host[g].tick_count; running += host[g].running; }
  printf("fleet_bench: %d gaits x %d ticks in %.3f s = %.0f MPC ticks/s (%s); %lld ticks done in all, %d gaits still walking\n", B, K,
         sec, (double)B * K / sec, per_tick ? "one launch per tick" : "multi-tick launches", ticks, running);
  (void)hipFree(d_states); (void)hipFree(d_vref); (void)hipFree(d_diag);
  wg_shutdown();
  return 0;
}
