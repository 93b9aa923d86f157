// test_rendezvous.cpp -- the file rendezvous of fleet_bench (host/wg_rendezvous.hpp) without RCCL or a GPU:
//   * a well-formed file a previous "job" left under the very name this job uses is not consumed (its writer is gone), the blob
//     rank 0 publishes afterwards is -- also by a reader that was already polling;
//   * ranks started through per-rank wrappers that do NOT exec (every rank a different parent) agree on the name and meet;
//   * a file of another world size, a torn file and a foreign file are not taken; names come from the launcher's environment.
//   test_rendezvous <dir>      exit code 0 = every check passed
#include <sys/stat.h>
#include <sys/wait.h>
#include <fcntl.h>

#include <cstdint>

#include "wg_rendezvous.hpp"

struct Blob { unsigned char b[128]; };                          // sizeof(ncclUniqueId)

static int fail(const char *what) { fprintf(stderr, "FAILED: %s\n", what); return 1; }

static int wait_ok(pid_t pid) {
  int st = 0;
  if (waitpid(pid, &st, 0) != pid || !WIFEXITED(st)) return -1;
  return WEXITSTATUS(st);
}

int main(int argc, char **argv) {
  if (argc < 2) return fail("usage: test_rendezvous <dir>");
  const std::string dir = argv[1];
  Blob stale, fresh, got;
  memset(&stale, 0xAA, sizeof stale);
  for (size_t i = 0; i < sizeof fresh.b; ++i) fresh.b[i] = (unsigned char)(i * 7 + 3);
  const int world = 4;

  // ---- /proc gives this process a start time; a pid that does not exist has none
  if (!wg_rdv::process_start_ticks((long)getpid())) return fail("own start time");
  if (wg_rdv::process_start_ticks(0x3fffffffL)) return fail("start time of a pid that cannot exist");

  // ---- names: the launcher's environment decides, never the parent -- unless the launcher says nothing at all
  unsetenv("WG_NCCL_ID_FILE"); unsetenv("TORCHELASTIC_RUN_ID"); unsetenv("PMIX_NAMESPACE"); unsetenv("OMPI_MCA_orte_hnp_uri");
  unsetenv("SLURM_JOB_ID"); unsetenv("SLURM_STEP_ID");
  setenv("MASTER_PORT", "29511", 1);
  if (wg_rdv::id_path() != "/tmp/wg_fleet_29511_ppid" + std::to_string((long)getppid()) + ".id") return fail(wg_rdv::id_path().c_str());
  setenv("SLURM_JOB_ID", "77", 1); setenv("SLURM_STEP_ID", "3", 1);
  if (wg_rdv::id_path() != "/tmp/wg_fleet_29511_slurm77_3.id") return fail(wg_rdv::id_path().c_str());
  setenv("OMPI_MCA_orte_hnp_uri", "123.0;tcp://10.0.0.1:5", 1);
  if (wg_rdv::id_path().find("_ompi123_0_tcp___10_0_0_1_5.id") == std::string::npos) return fail(wg_rdv::id_path().c_str());
  setenv("TORCHELASTIC_RUN_ID", "job/42", 1);
  if (wg_rdv::id_path() != "/tmp/wg_fleet_29511_runjob_42.id") return fail(wg_rdv::id_path().c_str());
  setenv("WG_NCCL_ID_FILE", "/tmp/x.id", 1);
  if (wg_rdv::id_path() != "/tmp/x.id") return fail("WG_NCCL_ID_FILE");
  unsetenv("WG_NCCL_ID_FILE"); unsetenv("OMPI_MCA_orte_hnp_uri"); unsetenv("SLURM_JOB_ID"); unsetenv("SLURM_STEP_ID");

  const std::string path = dir + "/wg_fleet_29511_runjob_42.id";

  // ---- a previous job's file under this job's name: well-formed, right world size, seconds old -- but its writer has exited
  {
    const pid_t w = fork();
    if (w < 0) return fail("fork");
    if (w == 0) _exit(wg_rdv::publish(path, &stale, sizeof stale, world));
    if (wait_ok(w) != 0) return fail("publish (stale)");
  }
  memset(&got, 0, sizeof got);
  if (wg_rdv::fetch(path, &got, sizeof got, world, 0.3) == 0) return fail("a dead writer's file was consumed");

  // ---- four ranks behind per-rank wrappers that do not exec: every rank has a different parent.  The readers start BEFORE
  //      rank 0 publishes (the stale file is lying there): each must come back with the fresh blob
  pid_t wrappers[4];
  for (int r = 0; r < world; ++r) {
    wrappers[r] = fork();
    if (wrappers[r] < 0) return fail("fork");
    if (wrappers[r] == 0) {                                      // the wrapper: stays alive as the rank's parent
      const pid_t rk = fork();
      if (rk < 0) _exit(9);
      if (rk == 0) {                                             // the rank
        // what a rank does: the name from its own environment (WG_NCCL_ID_FILE stands in for /tmp: the test writes under <dir>)
        if (wg_rdv::id_path() != "/tmp/wg_fleet_29511_runjob_42.id") _exit(4);
        if (r == 0) {
          std::this_thread::sleep_for(std::chrono::milliseconds(300));
          if (wg_rdv::publish(path, &fresh, sizeof fresh, world)) _exit(5);
          std::this_thread::sleep_for(std::chrono::milliseconds(1500));   // rank 0 lives on while the others read (it runs the job)
          _exit(0);
        }
        Blob b;
        memset(&b, 0, sizeof b);
        if (wg_rdv::fetch(path, &b, sizeof b, world, 10.0)) _exit(2);
        _exit(memcmp(&b, &fresh, sizeof b) == 0 ? 0 : 3);
      }
      _exit(wait_ok(rk));
    }
  }
  for (int r = 0; r < world; ++r) {
    const int rc = wait_ok(wrappers[r]);
    if (rc != 0) {
      fprintf(stderr, "FAILED: rank %d ended with status %d (2 = timed out, 3 = took the wrong blob, 4 = other name, 5 = publish)\n", r, rc);
      return 1;
    }
  }
  // rank 0 has exited by now: its file is a finished job's and must not serve a later one
  if (wg_rdv::fetch(path, &got, sizeof got, world, 0.2) == 0) return fail("a finished job's file was consumed");

  // ---- a live writer, but a job of another size (two jobs that were given one name): not taken
  if (wg_rdv::publish(path, &fresh, sizeof fresh, world + 4)) return fail("publish (other world)");
  if (wg_rdv::fetch(path, &got, sizeof got, world, 0.2) == 0) return fail("another world size was consumed");
  if (wg_rdv::fetch(path, &got, sizeof got, world + 4, 1.0) != 0 || memcmp(&got, &fresh, sizeof got)) return fail("own file not readable");
  // ---- a torn / foreign file is not taken either
  {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return fail("fopen");
    fwrite("garbage", 7, 1, f);
    fclose(f);
    if (wg_rdv::fetch(path, &got, sizeof got, world, 0.2) == 0) return fail("a torn file was consumed");
  }
  unlink(path.c_str());
  printf("rendezvous ok\n");
  return 0;
}
