// pcd_guard.cpp - libpcd_guard.so: the resident-set watchdog of this
// repository's scripts (include/pcd_guard.h).  Deliberately its own tiny
// library with NO OpenMP and no other dependency: it is loaded first thing in
// a process, and loading an OpenMP runtime that early (before the caller set
// its OMP_* environment, before torch loaded its own) changes where every
// later OpenMP team runs.
//
// Build: g++ -O2 -std=c++17 -shared -fPIC -pthread pcd_guard.cpp -o libpcd_guard.so
#include "../../include/pcd_guard.h"

#include <fcntl.h>
#include <pthread.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>

// ------------------------------------------------------------- watchdog
static std::atomic<int64_t> g_wd_limit{0}, g_wd_peak{0};
static std::atomic<int> g_wd_started{0};
static int g_wd_interval_ms = 50, g_wd_exit = 97;

static int64_t statm_rss_bytes() {
  int fd = open("/proc/self/statm", O_RDONLY);
  if (fd < 0) return 0;
  char buf[128];
  ssize_t n = read(fd, buf, sizeof buf - 1);
  close(fd);
  if (n <= 0) return 0;
  buf[n] = 0;
  long long size = 0, res = 0;
  if (sscanf(buf, "%lld %lld", &size, &res) != 2) return 0;
  return (int64_t)res * (int64_t)sysconf(_SC_PAGESIZE);
}

static void* watchdog_main(void*) {
  for (;;) {
    const int64_t r = statm_rss_bytes();
    int64_t pk = g_wd_peak.load(std::memory_order_relaxed);
    while (r > pk && !g_wd_peak.compare_exchange_weak(pk, r)) {}
    const int64_t lim = g_wd_limit.load(std::memory_order_relaxed);
    if (lim > 0 && r > lim) {
      char msg[256];
      int len = snprintf(msg, sizeof msg,
                         "\nRSS watchdog (libpcd_guard): resident set %.1f GB exceeds the limit of "
                         "%.1f GB - ending this process with status %d before the host runs out of "
                         "memory\n", r / 1e9, lim / 1e9, g_wd_exit);
      if (len > 0) { ssize_t w = write(2, msg, (size_t)len); (void)w; }
      _exit(g_wd_exit);
    }
    usleep((useconds_t)g_wd_interval_ms * 1000);
  }
  return nullptr;
}

extern "C" {

int pcdg_watchdog_start(int64_t limit_bytes, int interval_ms, int exit_code) {
  if (limit_bytes <= 0 || interval_ms <= 0 || exit_code <= 0 || exit_code > 255)
    return 1;
  int64_t cur = g_wd_limit.load();
  if (cur == 0 || limit_bytes < cur) g_wd_limit.store(limit_bytes);
  if (g_wd_started.exchange(1)) return 0;
  g_wd_interval_ms = interval_ms;
  g_wd_exit = exit_code;
  pthread_t th;
  pthread_attr_t at;
  pthread_attr_init(&at);
  pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
  const int rc = pthread_create(&th, &at, watchdog_main, nullptr);
  pthread_attr_destroy(&at);
  if (rc) { g_wd_started.store(0); return 2; }
  return 0;
}
int64_t pcdg_watchdog_peak(void) {
  const int64_t r = statm_rss_bytes();
  int64_t pk = g_wd_peak.load();
  return r > pk ? r : pk;
}
int64_t pcdg_watchdog_limit(void) { return g_wd_limit.load(); }

}  // extern "C"
