// fleet_bench.cpp -- the fleet path from plain C++ through the C ABI (no Python, no PyTorch): B gaits per GPU resident in
// device memory, velocity references redrawn every 50 ticks and staged on the device, ticks advanced with wg_mpc_run_sched_dev (device-side work
// queue) or, with --per-tick, one wg_mpc_tick_batch_dev launch per tick.  Same workload as bench.py (std::mt19937_64
// seeded 20100 + GLOBAL gait index; bench.py draws the same distribution through numpy).
//
// Multi-GPU (BASELINE north_star: "host C++ ... a single RCCL broadcast of the robot model over xGMI"):
// one process per GPU.  `--ranks N` makes this program start N copies of itself -- before it has made any GPU call -- one
// per device; a launcher that sets RANK / WORLD_SIZE / LOCAL_RANK (torchrun, mpirun wrappers) works the same way.  Rank 0
// owns the robot model (wg_model_t) and broadcasts it ONCE with ncclBroadcast (RCCL); every rank then configures its own
// context from the received block, takes the contiguous shard wg_shard_range() gives it (weak scaling: --batch gaits per
// GPU) and never talks to the others again on the data path.  Timing: all-reduce barrier + device synchronise on both
// sides of the timed region, MAX over ranks; rank 0 prints one JSON line with the whole-job rate.
//
//   fleet_bench [--batch B] [--ticks K] [--warmup W] [--per-tick] [--ranks N] [--json]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wg_mpc.h"
#include "wg_rendezvous.hpp"

#define CHECK_HIP(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "FAILED: %s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
#define CHECK_WG(e) do { int r_ = (e); if (r_ != WG_OK) { fprintf(stderr, "FAILED: %s: %s\n", #e, wg_last_error()); return 1; } } while (0)
#define CHECK_NCCL(e) do { ncclResult_t r_ = (e); if (r_ != ncclSuccess) { fprintf(stderr, "FAILED: %s: %s\n", #e, ncclGetErrorString(r_)); return 1; } } while (0)

static int env_int(const char *k, int dflt) { const char *e = getenv(k); return e ? atoi(e) : dflt; }

// rank 0 publishes the RCCL unique id through a file named after the job; the other ranks take it only if it IS this job's
// (host/wg_rendezvous.hpp: the name comes from the launcher's environment, a file whose writer is no longer alive is a previous
// job's and is ignored)
static int exchange_id(const std::string &path, int rank, int world, ncclUniqueId *id) {
  if (rank == 0) {
    CHECK_NCCL(ncclGetUniqueId(id));
    return wg_rdv::publish(path, id, sizeof *id, world);
  }
  return wg_rdv::fetch(path, id, sizeof *id, world);
}

int main(int argc, char **argv) {
  int B = 4096, K = 200, W = 50, ranks = 0;
  bool per_tick = false, json = false;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--batch") && i + 1 < argc) B = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--ticks") && i + 1 < argc) K = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) W = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--ranks") && i + 1 < argc) ranks = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--per-tick")) per_tick = true;
    else if (!strcmp(argv[i], "--json")) json = true;
  }
  const int REDRAW = 50;

  // ---- launcher: N copies of this program, one per GPU, started before this process touches the GPU ----
  if (ranks > 0 && !getenv("WORLD_SIZE")) {
    char idfile[64];
    snprintf(idfile, sizeof idfile, "/tmp/wg_fleet_%d.id", (int)getpid());
    std::vector<pid_t> kids;
    for (int r = 0; r < ranks; ++r) {
      const pid_t pid = fork();
      if (pid < 0) { perror("fork"); return 1; }
      if (pid == 0) {
        setenv("RANK", std::to_string(r).c_str(), 1);
        setenv("LOCAL_RANK", std::to_string(r).c_str(), 1);
        setenv("WORLD_SIZE", std::to_string(ranks).c_str(), 1);
        setenv("WG_NCCL_ID_FILE", idfile, 1);
        execv("/proc/self/exe", argv);                   // nothing has initialised the GPU in this process
        perror("execv");
        _exit(127);
      }
      kids.push_back(pid);
    }
    // a rank that fails (no such device ...) would leave the others waiting in ncclCommInitRank for ever: the first
    // failure ends the job
    int bad = 0;
    for (size_t left = kids.size(); left > 0; --left) {
      int st = 0;
      const pid_t done = waitpid(-1, &st, 0);
      if (done < 0) { bad = 1; break; }
      if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
        if (!bad) {
          fprintf(stderr, "FAILED: a rank exited with an error; stopping the others\n");
          for (pid_t k : kids) if (k != done) kill(k, SIGKILL);
        }
        bad = 1;
      }
    }
    unlink(idfile);
    return bad;
  }

  const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0), local = env_int("LOCAL_RANK", rank);
  const bool use_rccl = getenv("WORLD_SIZE") != nullptr;       // also with one rank: the same code path as with eight
  wg_ctx_t *ctx = nullptr;
  CHECK_WG(wg_ctx_create(local, &ctx));                         // one process per GPU: device = LOCAL_RANK
  CHECK_HIP(hipSetDevice(local));
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));

  // ---- the one collective: the constant model block from rank 0 ----
  wg_model_t model;
  memset(&model, 0, sizeof model);
  if (rank == 0) wg_model_defaults(&model);
  ncclComm_t comm = nullptr;
  double *d_red = nullptr;                                      // one double for the barriers / the max over ranks
  if (use_rccl) {
    const std::string idpath = wg_rdv::id_path();            // WG_NCCL_ID_FILE, or named after port and the launcher's job identity
    ncclUniqueId id;
    if (exchange_id(idpath, rank, world, &id)) return 1;
    CHECK_NCCL(ncclCommInitRank(&comm, world, id, rank));
    wg_model_t *d_model = nullptr;
    CHECK_HIP(hipMalloc((void **)&d_model, sizeof model));
    CHECK_HIP(hipMalloc((void **)&d_red, sizeof(double)));
    CHECK_HIP(hipMemcpy(d_model, &model, sizeof model, hipMemcpyHostToDevice));
    CHECK_NCCL(ncclBroadcast(d_model, d_model, sizeof model, ncclChar, 0, comm, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(&model, d_model, sizeof model, hipMemcpyDeviceToHost));
    (void)hipFree(d_model);
    // the broadcast needed every rank inside the communicator: all of them have read the id, the file has served
    if (rank == 0) unlink(idpath.c_str());
  }
  if (model.N <= 0) { fprintf(stderr, "FAILED: rank %d did not receive the model\n", rank); return 1; }
  CHECK_WG(wg_mpc_configure_ctx(ctx, &model));                  // tables rebuilt locally from the 208-byte block

  // ---- this rank's shard: global gaits [lo, hi) ----
  long long lo = 0, hi = 0;
  CHECK_WG(wg_shard_range((long long)B * world, rank, world, &lo, &hi));
  const int Bl = (int)(hi - lo);
  std::vector<wg_gait_state_t> host(Bl);
  const double com0[3] = {0.0316055, 0.0, 0.7116911}, lf[3] = {0.0, 0.09, 0.0}, rf[3] = {0.0, -0.09, 0.0};
  for (int g = 0; g < Bl; ++g) { wg_gait_init(&model, &host[g], com0, lf, rf); host[g].nb_steps_left = 2; }
  const int n_seg = (W + K + REDRAW - 1) / REDRAW;
  std::vector<double> vref((size_t)n_seg * Bl * 3);
  for (int g = 0; g < Bl; ++g) {
    std::mt19937_64 rng(20100 + (unsigned long long)(lo + g));
    std::uniform_real_distribution<double> ux(-0.1, 0.3), uy(-0.1, 0.1), uw(-0.2, 0.2);
    for (int s = 0; s < n_seg; ++s) {
      double *v = &vref[((size_t)s * Bl + g) * 3];
      v[0] = ux(rng); v[1] = uy(rng); v[2] = uw(rng);
    }
  }
  wg_gait_state_t *d_states = nullptr; double *d_vref = nullptr; int *d_diag = nullptr;
  CHECK_HIP(hipMalloc((void **)&d_states, sizeof(wg_gait_state_t) * Bl));
  CHECK_HIP(hipMalloc((void **)&d_vref, sizeof(double) * vref.size()));
  CHECK_HIP(hipMalloc((void **)&d_diag, sizeof(int) * 6 * (size_t)Bl * (size_t)(W > K ? W : K)));   // tick-major, one launch
  CHECK_HIP(hipMemcpy(d_states, host.data(), sizeof(wg_gait_state_t) * Bl, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_vref, vref.data(), sizeof(double) * vref.size(), hipMemcpyHostToDevice));

  auto advance = [&](int t0, int t1) -> int {           // ticks [t0, t1)
    for (int t = t0; t < t1;) {
      const int adv = t == 0 ? 1 : (t == 1 ? 19 : 20);
      const double *refs = d_vref + (size_t)(t / REDRAW) * Bl * 3;
      // a launch that starts where the references change runs to t1 with the references of all its stretches staged
      // (wg_mpc_run_sched_dev); any other ends where they change next
      const bool staged = t >= 2 && !per_tick && t % REDRAW == 0 && t1 - t > 1;
      if (t % REDRAW == 0 && !staged) CHECK_WG(wg_mpc_set_velref_dev_ctx(ctx, Bl, d_states, refs, st));
      int n = 1;
      if (staged) n = t1 - t;
      else if (t >= 2 && !per_tick) { n = (t / REDRAW + 1) * REDRAW - t; if (t + n > t1) n = t1 - t; }
      if (staged) CHECK_WG(wg_mpc_run_sched_dev_ctx(ctx, Bl, d_states, n, adv, refs, REDRAW, nullptr, d_diag, st));
      else if (n == 1) CHECK_WG(wg_mpc_tick_batch_dev_ctx(ctx, Bl, d_states, nullptr, d_diag, adv, nullptr, 0, nullptr, st));
      else CHECK_WG(wg_mpc_run_batch_dev_ctx(ctx, Bl, d_states, n, adv, nullptr, d_diag, st));
      t += n;
    }
    return 0;
  };
  // barrier = an all-reduce every rank must enter, then a device synchronise
  auto barrier = [&](double *value, ncclRedOp_t op) -> int {
    if (use_rccl) {
      CHECK_HIP(hipMemcpyAsync(d_red, value, sizeof(double), hipMemcpyHostToDevice, st));
      CHECK_NCCL(ncclAllReduce(d_red, d_red, 1, ncclDouble, op, comm, st));
      CHECK_HIP(hipMemcpyAsync(value, d_red, sizeof(double), hipMemcpyDeviceToHost, st));
    }
    CHECK_HIP(hipStreamSynchronize(st));
    return 0;
  };
  if (advance(0, W)) return 1;
  double one = 1.0;
  if (barrier(&one, ncclSum)) return 1;
  const auto t0 = std::chrono::steady_clock::now();
  if (advance(W, W + K)) return 1;
  one = 1.0;
  if (barrier(&one, ncclSum)) return 1;
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (barrier(&sec, ncclMax)) return 1;                         // the slowest rank's time

  CHECK_HIP(hipMemcpy(host.data(), d_states, sizeof(wg_gait_state_t) * Bl, hipMemcpyDeviceToHost));
  long long ticks = 0; int running = 0;
  for (int g = 0; g < Bl; ++g) { ticks += host[g].tick_count; running += host[g].running; }
  double tk = (double)ticks;
  if (barrier(&tk, ncclSum)) return 1;
  const double total_gaits = (double)B * world;
  if (rank == 0) {
    if (json || use_rccl)
      printf("{\"metric\": \"QP-MPC ticks/sec (batch=4096, N=16)\", \"value\": %.1f, \"unit\": \"ticks/s\", \"n_gpus\": %d, "
             "\"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.6f, \"higher_is_better\": true, \"scaling\": \"weak\", "
             "\"dtype\": \"f64\", \"data\": \"synthetic\", \"config\": {\"workload\": \"Herdt2010 N=16 fp64, %d gaits per GPU\", "
             "\"host\": \"C++ (fleet_bench) over the C ABI\", \"collective\": \"%s\", \"launch\": \"%s\"}, "
             "\"ticks_done_in_all\": %.0f}\n",
             total_gaits * K / sec, world, K, W, 1e3 * sec / K, B,
             use_rccl ? "one ncclBroadcast of wg_model_t (RCCL), all-reduce barriers around the timed region" : "none (single process)",
             per_tick ? "one launch per tick" : "multi-tick launches", tk);
    printf("fleet_bench: %d rank(s) x %d gaits x %d ticks in %.3f s = %.0f MPC ticks/s (%s); %.0f ticks done in all, %d gaits of rank 0 still walking\n",
           world, B, K, sec, total_gaits * K / sec, per_tick ? "one launch per tick" : "multi-tick launches", tk, running);
  }
  (void)hipFree(d_states); (void)hipFree(d_vref); (void)hipFree(d_diag);
  if (d_red) (void)hipFree(d_red);
  if (comm) (void)ncclCommDestroy(comm);
  wg_ctx_destroy(ctx);
  return 0;
}
