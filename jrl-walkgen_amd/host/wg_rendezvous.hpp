// wg_rendezvous.hpp -- how the ranks of ONE job on one node find the blob rank 0 publishes (fleet_bench: the ncclUniqueId), through
// a file, without ever taking the blob a previous job left behind.
//
//   * the file's NAME carries the job's identity as the launcher hands it to every rank in the environment -- never something that
//     depends on how a rank was started: WG_NCCL_ID_FILE when the launcher names the file (fleet_bench --ranks N does), otherwise
//     /tmp/wg_fleet_<MASTER_PORT>_<job>.id with <job> = TORCHELASTIC_RUN_ID, PMIX_NAMESPACE / OMPI_MCA_orte_hnp_uri (mpirun),
//     SLURM_JOB_ID.SLURM_STEP_ID, in that order; only when the launcher set none of them, the parent's pid ("ppid<n>": ranks forked
//     by one process).  A per-rank wrapper that does not exec (mpirun ./wrap.sh prog, torchrun --no-python bash -c ..., numactl
//     scripts) gives every rank a different parent: names built from getppid() made such ranks wait for a file rank 0 never wrote;
//   * FRESHNESS does not come from clocks or parents either: rank 0 writes, in front of the blob, its own pid, its start time in
//     clock ticks since boot (/proc/<pid>/stat, field 22) and the job's WORLD_SIZE; a reader takes the file only while a process
//     with that pid AND that start time is alive and the world size is its own.  What a crashed or finished job left behind names a
//     dead writer (a recycled pid has another start time) and is passed over until this job's rank 0 has replaced it -- whatever
//     the age of the launcher (ranks started from one long-lived shell) and however long ago the file was written;
//   * rank 0 removes whatever lies at the name before it writes (under another name, then renamed: never a torn blob), and
//     removes its own file once every rank has it (fleet_bench: after the first collective).
// Same node only (pids are compared through /proc).  Plain C++ / POSIX, no GPU: tests/test_rendezvous.py runs
// host/test_rendezvous.cpp on it.
#pragma once
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

namespace wg_rdv {

static const char kMagic[8] = {'W', 'G', 'I', 'D', '0', '0', '3', '\0'};
struct Header {                      // in front of the blob
  char magic[8];
  long long writer_pid;
  unsigned long long writer_start_ticks;   // /proc/<pid>/stat field 22: clock ticks since boot at which the writer started
  int world;
  int pad_;
};

inline double now_epoch() {
  return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
}

// start of process `pid` in clock ticks since boot (/proc/<pid>/stat field 22); 0 when there is no such process
inline unsigned long long process_start_ticks(long pid) {
  char path[64];
  snprintf(path, sizeof path, "/proc/%ld/stat", pid);
  FILE *f = fopen(path, "r");
  if (!f) return 0;
  char buf[2048];
  const size_t n = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  buf[n] = 0;
  const char *p = strrchr(buf, ')');                         // the command name may contain spaces and parentheses
  if (!p) return 0;
  int field = 2;
  for (++p; *p; ++p) {
    if (*p != ' ') continue;
    if (++field == 22) return strtoull(p + 1, nullptr, 10);
  }
  return 0;
}

inline std::string sanitized(const char *s) {
  std::string o;
  for (; s && *s && o.size() < 48; ++s) o += (isalnum((unsigned char)*s) || *s == '-' || *s == '_') ? *s : '_';
  return o;
}

// the job's identity as the launcher's environment carries it (the same in every rank, however the rank was started)
inline std::string job_identity() {
  if (const char *e = getenv("TORCHELASTIC_RUN_ID")) return "run" + sanitized(e);
  if (const char *e = getenv("PMIX_NAMESPACE")) return "pmix" + sanitized(e);
  if (const char *e = getenv("OMPI_MCA_orte_hnp_uri")) return "ompi" + sanitized(e);
  if (const char *e = getenv("SLURM_JOB_ID")) {
    const char *st = getenv("SLURM_STEP_ID");
    return "slurm" + sanitized(e) + "_" + sanitized(st ? st : "0");
  }
  return "ppid" + std::to_string((long)getppid());           // no launcher identity at all: ranks forked by one process
}

inline std::string id_path() {
  if (const char *e = getenv("WG_NCCL_ID_FILE")) return e;
  const char *port = getenv("MASTER_PORT");
  return "/tmp/wg_fleet_" + sanitized(port ? port : "29511") + "_" + job_identity() + ".id";
}

// rank 0
inline int publish(const std::string &path, const void *blob, size_t len, int world) {
  unlink(path.c_str());                                      // whatever a previous job left under this name
  Header h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, kMagic, sizeof kMagic);
  h.writer_pid = (long long)getpid();
  h.writer_start_ticks = process_start_ticks((long)getpid());
  h.world = world;
  if (!h.writer_start_ticks) { fprintf(stderr, "FAILED: /proc does not show this process's start time\n"); return 1; }
  const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) { fprintf(stderr, "FAILED: cannot write %s\n", tmp.c_str()); return 1; }
  const bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(blob, len, 1, f) == 1;
  if (fclose(f) != 0 || !ok) { fprintf(stderr, "FAILED: cannot write %s\n", tmp.c_str()); unlink(tmp.c_str()); return 1; }
  if (rename(tmp.c_str(), path.c_str()) != 0) { fprintf(stderr, "FAILED: rename %s\n", path.c_str()); unlink(tmp.c_str()); return 1; }
  return 0;
}

// the other ranks: poll until a file of THIS job is there -- whole, with the magic, written for this world size by a process that
// is alive right now (same pid, same start time)
inline int fetch(const std::string &path, void *blob, size_t len, int world, double timeout_s = 60.0) {
  const double give_up = now_epoch() + timeout_s;
  bool told = false;
  for (;;) {
    struct stat sb;
    if (stat(path.c_str(), &sb) == 0 && (size_t)sb.st_size == sizeof(Header) + len) {
      if (FILE *f = fopen(path.c_str(), "rb")) {
        Header h;
        const bool whole = fread(&h, sizeof h, 1, f) == 1 && !memcmp(h.magic, kMagic, sizeof kMagic) && fread(blob, len, 1, f) == 1;
        fclose(f);
        if (whole) {
          const bool alive = h.writer_start_ticks != 0 && process_start_ticks((long)h.writer_pid) == h.writer_start_ticks;
          if (alive && h.world == world) return 0;
          if (!told) {
            if (!alive) fprintf(stderr, "note: %s was written by a process that is gone (pid %lld): a previous job's, waiting for rank 0\n", path.c_str(), h.writer_pid);
            else fprintf(stderr, "note: %s belongs to a job of %d ranks, this one has %d: waiting for rank 0\n", path.c_str(), h.world, world);
            told = true;
          }
        }
      }
    }
    if (now_epoch() > give_up) break;
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  fprintf(stderr, "FAILED: no rendezvous file of this job at %s within %.0f s\n", path.c_str(), timeout_s);
  return 1;
}

}  // namespace wg_rdv
