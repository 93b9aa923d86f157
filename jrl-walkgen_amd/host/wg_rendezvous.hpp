// wg_rendezvous.hpp -- how the ranks of ONE job on one node find the blob rank 0 publishes (fleet_bench: the ncclUniqueId), through
// a file, without ever taking the blob a previous job left behind.
//
//   * the file's name carries the job: WG_NCCL_ID_FILE when the launcher names it (fleet_bench --ranks N does: its own pid),
//     otherwise /tmp/wg_fleet_<MASTER_PORT>_<TORCHELASTIC_RUN_ID>_<pid of the launcher>.id -- every rank of a job is a child of the
//     same launcher process (torch.distributed.run's agent, mpirun, a shell), two jobs alive at once have different ones;
//   * a file that is older than the launcher itself cannot be this job's (a recycled pid, a crashed job's leftovers): readers
//     ignore it and keep polling -- "older" by the file's modification time against the launcher's start time from /proc, so
//     the test does not depend on how far apart the launcher starts its ranks;
//   * rank 0 removes whatever lies at the name before it writes (under another name, then renamed: never a torn blob), and
//     removes its own file once every rank has it (fleet_bench: after the first collective).
// Plain C++ / POSIX, no GPU: tests/test_rendezvous.py runs host/test_rendezvous.cpp on it.
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

static const char kMagic[8] = {'W', 'G', 'I', 'D', '0', '0', '2', '\0'};

inline double now_epoch() {
  return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
}

// start of process `pid` in seconds since the epoch (/proc/<pid>/stat field 22 in clock ticks since boot + /proc/stat btime);
// 0 when /proc does not say
inline double process_start_epoch(long pid) {
  char path[64];
  snprintf(path, sizeof path, "/proc/%ld/stat", pid);
  FILE *f = fopen(path, "r");
  if (!f) return 0.0;
  char buf[2048];
  const size_t n = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  buf[n] = 0;
  const char *p = strrchr(buf, ')');                         // the command name may contain spaces and parentheses
  if (!p) return 0.0;
  unsigned long long ticks = 0;
  int field = 2;
  for (++p; *p; ++p) {
    if (*p != ' ') continue;
    if (++field == 22) { ticks = strtoull(p + 1, nullptr, 10); break; }
  }
  if (field != 22) return 0.0;
  long long btime = 0;
  if (FILE *s = fopen("/proc/stat", "r")) {
    char line[256];
    while (fgets(line, sizeof line, s))
      if (!strncmp(line, "btime ", 6)) { btime = atoll(line + 6); break; }
    fclose(s);
  }
  if (btime <= 0) return 0.0;
  const long hz = sysconf(_SC_CLK_TCK) > 0 ? sysconf(_SC_CLK_TCK) : 100;
  return (double)btime + (double)ticks / (double)hz;
}

// a file of this job is not older than this (btime has one-second resolution: two seconds of slack)
inline double job_not_before() {
  const double s = process_start_epoch((long)getppid());
  return s > 0.0 ? s - 2.0 : 0.0;
}

inline std::string sanitized(const char *s) {
  std::string o;
  for (; s && *s && o.size() < 48; ++s) o += (isalnum((unsigned char)*s) || *s == '-' || *s == '_') ? *s : '_';
  return o;
}

inline std::string id_path() {
  if (const char *e = getenv("WG_NCCL_ID_FILE")) return e;
  const char *port = getenv("MASTER_PORT");
  const char *run = getenv("TORCHELASTIC_RUN_ID");
  return "/tmp/wg_fleet_" + sanitized(port ? port : "29511") + "_" + sanitized(run ? run : "norun") + "_" + std::to_string((long)getppid()) + ".id";
}

// rank 0
inline int publish(const std::string &path, const void *blob, size_t len) {
  unlink(path.c_str());                                      // whatever a previous job left under this name
  const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) { fprintf(stderr, "FAILED: cannot write %s\n", tmp.c_str()); return 1; }
  const bool ok = fwrite(kMagic, sizeof kMagic, 1, f) == 1 && fwrite(blob, len, 1, f) == 1;
  if (fclose(f) != 0 || !ok) { fprintf(stderr, "FAILED: cannot write %s\n", tmp.c_str()); unlink(tmp.c_str()); return 1; }
  if (rename(tmp.c_str(), path.c_str()) != 0) { fprintf(stderr, "FAILED: rename %s\n", path.c_str()); unlink(tmp.c_str()); return 1; }
  return 0;
}

// the other ranks: poll until a file of THIS job is there (modified at or after not_before, whole, with the magic)
inline int fetch(const std::string &path, void *blob, size_t len, double not_before, double timeout_s = 60.0) {
  const double give_up = now_epoch() + timeout_s;
  bool told = false;
  for (;;) {
    struct stat sb;
    if (stat(path.c_str(), &sb) == 0) {
      const double mtime = (double)sb.st_mtim.tv_sec + 1e-9 * (double)sb.st_mtim.tv_nsec;
      if (mtime < not_before) {
        if (!told) { fprintf(stderr, "note: %s is older than this job's launcher: not ours, waiting for rank 0\n", path.c_str()); told = true; }
      } else if ((size_t)sb.st_size == sizeof kMagic + len) {
        if (FILE *f = fopen(path.c_str(), "rb")) {
          char magic[sizeof kMagic];
          const bool ok = fread(magic, sizeof magic, 1, f) == 1 && !memcmp(magic, kMagic, sizeof kMagic) && fread(blob, len, 1, f) == 1;
          fclose(f);
          if (ok) return 0;
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
