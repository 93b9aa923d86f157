// test_rendezvous.cpp -- the file rendezvous of fleet_bench (host/wg_rendezvous.hpp) without RCCL or a GPU: a file a previous
// "job" left under the very name this job uses is not consumed; the blob rank 0 publishes afterwards is.
//   test_rendezvous <dir>      exit code 0 = every check passed
#include <sys/stat.h>
#include <sys/wait.h>
#include <fcntl.h>

#include <cstdint>

#include "wg_rendezvous.hpp"

struct Blob { unsigned char b[128]; };                          // sizeof(ncclUniqueId)

static int fail(const char *what) { fprintf(stderr, "FAILED: %s\n", what); return 1; }

int main(int argc, char **argv) {
  if (argc < 2) return fail("usage: test_rendezvous <dir>");
  const std::string path = std::string(argv[1]) + "/wg_fleet_29511_norun_" + std::to_string((long)getppid()) + ".id";
  Blob stale, fresh, got;
  memset(&stale, 0xAA, sizeof stale);
  for (size_t i = 0; i < sizeof fresh.b; ++i) fresh.b[i] = (unsigned char)(i * 7 + 3);

  // ---- the launcher's start time is readable and in the past
  const double nb = wg_rdv::job_not_before();
  if (!(nb > 0.0 && nb < wg_rdv::now_epoch())) return fail("launcher start time");
  if (wg_rdv::process_start_epoch((long)getpid()) < nb) return fail("this process started before its parent");

  // ---- a previous job's file under this job's name: well-formed, but an hour old
  if (wg_rdv::publish(path, &stale, sizeof stale)) return fail("publish (stale)");
  {
    struct timespec ts[2];
    ts[0].tv_sec = ts[1].tv_sec = (time_t)(wg_rdv::now_epoch() - 3600.0);
    ts[0].tv_nsec = ts[1].tv_nsec = 0;
    if (utimensat(AT_FDCWD, path.c_str(), ts, 0) != 0) return fail("utimensat");
  }
  // alone, it is never taken: the reader times out
  memset(&got, 0, sizeof got);
  if (wg_rdv::fetch(path, &got, sizeof got, nb, 0.3) == 0) return fail("a stale file was consumed");

  // ---- a reader that starts BEFORE rank 0 publishes: it must come back with the fresh blob, not the stale one
  const pid_t pid = fork();
  if (pid < 0) return fail("fork");
  if (pid == 0) {
    Blob r;
    memset(&r, 0, sizeof r);
    // the child's launcher is this process: the stale file predates it as well
    if (wg_rdv::fetch(path, &r, sizeof r, wg_rdv::job_not_before(), 10.0)) _exit(2);
    _exit(memcmp(&r, &fresh, sizeof r) == 0 ? 0 : 3);
  }
  std::this_thread::sleep_for(std::chrono::milliseconds(300));
  if (wg_rdv::publish(path, &fresh, sizeof fresh)) return fail("publish (fresh)");
  int st = 0;
  if (waitpid(pid, &st, 0) != pid || !WIFEXITED(st) || WEXITSTATUS(st) != 0) {
    fprintf(stderr, "FAILED: the reader ended with status %d (2 = timed out, 3 = took the wrong blob)\n", WIFEXITED(st) ? WEXITSTATUS(st) : -1);
    return 1;
  }
  // ---- a torn / foreign file of the right age is not taken either
  {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return fail("fopen");
    fwrite("garbage", 7, 1, f);
    fclose(f);
    if (wg_rdv::fetch(path, &got, sizeof got, nb, 0.2) == 0) return fail("a torn file was consumed");
  }
  // ---- names: the launcher's pid and the run id are part of it, WG_NCCL_ID_FILE overrides
  unsetenv("WG_NCCL_ID_FILE");
  setenv("MASTER_PORT", "29511", 1);
  setenv("TORCHELASTIC_RUN_ID", "job/42", 1);
  const std::string n1 = wg_rdv::id_path();
  if (n1.find("_29511_job_42_" + std::to_string((long)getppid()) + ".id") == std::string::npos) return fail(n1.c_str());
  setenv("WG_NCCL_ID_FILE", "/tmp/x.id", 1);
  if (wg_rdv::id_path() != "/tmp/x.id") return fail("WG_NCCL_ID_FILE");
  unlink(path.c_str());
  printf("rendezvous ok\n");
  return 0;
}
