// latency_b1.cpp -- what ONE robot pays per MPC tick (BASELINE configs[1]: batch = 1), split into its parts.
// The reference's tick (ZMPVelocityReferencedQP::OnLine, ZMPVelocityReferencedQP.cpp:346-452) runs once per 0.1 s per robot
// and takes ~61 us on one host core; this program times the same tick through the C ABI on the GPU:
//   host     wg_mpc_tick_batch(B = 1): host pointers in and out (what the facade calls)
//   split    the same steps issued by hand on own device buffers -- copy in, launch, kernel (HIP events), synchronise,
//            copy out -- each timed on the host clock
//   pinned   wg_mpc_tick_pinned (when the library exports it): state and outputs in host-mapped memory, no copies
// Output: one JSON line (medians over --ticks ticks of a walking gait).
//   latency_b1 [--ticks K]
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/wg_mpc.h"

#define CHECK_HIP(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "FAILED: %s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
#define CHECK_WG(e) do { int r_ = (e); if (r_ != WG_OK) { fprintf(stderr, "FAILED: %s: %s\n", #e, wg_last_error()); return 1; } } while (0)

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }
static double p90(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[(v.size() * 9) / 10]; }

int main(int argc, char **argv) {
  int K = 200;
  for (int i = 1; i < argc; ++i)
    if (!strcmp(argv[i], "--ticks") && i + 1 < argc) K = atoi(argv[++i]);
  CHECK_WG(wg_init(0));
  wg_model_t model;
  wg_model_defaults(&model);
  CHECK_WG(wg_mpc_configure(&model));
  const double com0[3] = {0.0316055, 0.0, 0.7116911}, lf[3] = {0.0, 0.09, 0.0}, rf[3] = {0.0, -0.09, 0.0};
  auto fresh = [&](wg_gait_state_t *s) {
    wg_gait_init(&model, s, com0, lf, rf);
    s->nb_steps_left = 2;
    s->vref[0] = 0.2; s->vref[1] = 0.0; s->vref[2] = 0.05;
  };
  static wg_tick_out_t out;
  int diag[6];

  // ---- (1) the host-pointer entry point, as the facade calls it ----
  wg_gait_state_t s;
  fresh(&s);
  CHECK_WG(wg_mpc_tick_batch(1, &s, &out, diag, 1, nullptr, 0, nullptr));
  CHECK_WG(wg_mpc_tick_batch(1, &s, &out, diag, 19, nullptr, 0, nullptr));
  std::vector<double> t_host;
  long iters = 0;
  for (int k = 0; k < K; ++k) {
    const auto a = clk::now();
    CHECK_WG(wg_mpc_tick_batch(1, &s, &out, diag, 20, nullptr, 0, nullptr));
    t_host.push_back(us(a, clk::now()));
    iters += diag[1];
    if (diag[0] != 0) { fprintf(stderr, "FAILED: QP failed at tick %d (ifail %d)\n", k, diag[0]); return 1; }
  }
  const wg_gait_state_t ref_end = s;

  // ---- (2) the same steps by hand ----
  wg_gait_state_t *d_s; wg_tick_out_t *d_o; int *d_d;
  CHECK_HIP(hipMalloc(&d_s, sizeof *d_s)); CHECK_HIP(hipMalloc(&d_o, sizeof *d_o)); CHECK_HIP(hipMalloc(&d_d, 24));
  hipEvent_t e0, e1; CHECK_HIP(hipEventCreate(&e0)); CHECK_HIP(hipEventCreate(&e1));
  fresh(&s);
  std::vector<double> t_in, t_launch, t_sync, t_out, t_kernel, t_all;
  for (int k = -2; k < K; ++k) {
    const int adv = k == -2 ? 1 : (k == -1 ? 19 : 20);
    const auto a = clk::now();
    CHECK_HIP(hipMemcpy(d_s, &s, sizeof s, hipMemcpyHostToDevice));
    const auto b = clk::now();
    CHECK_HIP(hipEventRecord(e0, nullptr));
    CHECK_WG(wg_mpc_tick_batch_dev(1, d_s, d_o, d_d, adv, nullptr, 0, nullptr, nullptr));
    CHECK_HIP(hipEventRecord(e1, nullptr));
    const auto c = clk::now();
    CHECK_HIP(hipDeviceSynchronize());
    const auto d = clk::now();
    CHECK_HIP(hipMemcpy(&s, d_s, sizeof s, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&out, d_o, sizeof out, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(diag, d_d, 24, hipMemcpyDeviceToHost));
    const auto e = clk::now();
    float ms = 0.f; CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (k >= 0) {
      t_in.push_back(us(a, b)); t_launch.push_back(us(b, c)); t_sync.push_back(us(c, d)); t_out.push_back(us(d, e));
      t_kernel.push_back(ms * 1e3); t_all.push_back(us(a, e));
    }
  }
  const bool same = memcmp(&s, &ref_end, sizeof s) == 0;

  // ---- (3) host-mapped state and outputs, when the library has the entry point ----
  typedef int (*pinned_fn)(wg_gait_state_t *, wg_tick_out_t *, int *, int);
  typedef int (*alloc_fn)(void **, size_t);
  pinned_fn tick_pinned = reinterpret_cast<pinned_fn>(dlsym(RTLD_DEFAULT, "wg_mpc_tick_pinned"));
  alloc_fn host_alloc = reinterpret_cast<alloc_fn>(dlsym(RTLD_DEFAULT, "wg_host_alloc"));
  std::vector<double> t_pin;
  bool same_pin = true;
  if (tick_pinned && host_alloc) {
    void *mem = nullptr;
    CHECK_WG(host_alloc(&mem, sizeof(wg_gait_state_t) + sizeof(wg_tick_out_t) + 64));
    wg_gait_state_t *ps = static_cast<wg_gait_state_t *>(mem);
    wg_tick_out_t *po = reinterpret_cast<wg_tick_out_t *>(ps + 1);
    int *pd = reinterpret_cast<int *>(po + 1);
    fresh(ps);
    CHECK_WG(tick_pinned(ps, po, pd, 1));
    CHECK_WG(tick_pinned(ps, po, pd, 19));
    for (int k = 0; k < K; ++k) {
      const auto a = clk::now();
      CHECK_WG(tick_pinned(ps, po, pd, 20));
      t_pin.push_back(us(a, clk::now()));
    }
    same_pin = memcmp(ps, &ref_end, sizeof *ps) == 0;
  }

  printf("{\"workload\": \"Herdt2010 N=16 fp64, batch=1 (one robot), %d walking ticks\", \"mean_ql_iterations\": %.1f, "
         "\"host_pointer_call_us\": {\"median\": %.1f, \"p90\": %.1f}, "
         "\"split_us\": {\"copy_in\": %.1f, \"launch\": %.1f, \"wait_for_kernel\": %.1f, \"copy_out\": %.1f, \"kernel_hip_events\": %.1f, \"sum\": %.1f}, "
         "\"split_matches_host_call\": %s",
         K, (double)iters / K, median(t_host), p90(t_host), median(t_in), median(t_launch), median(t_sync), median(t_out),
         median(t_kernel), median(t_all), same ? "true" : "false");
  if (!t_pin.empty())
    printf(", \"host_mapped_call_us\": {\"median\": %.1f, \"p90\": %.1f, \"same_bytes\": %s}", median(t_pin), p90(t_pin),
           same_pin ? "true" : "false");
  printf("}\n");
  return (same && same_pin) ? 0 : 2;
}
