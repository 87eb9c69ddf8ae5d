/* abort_bt.c -- LD_PRELOAD helper of tools/debug/rccl_exit_abort.py: on SIGABRT / SIGSEGV print
 * the C backtrace and the HIP / RCCL / comgr libraries mapped into the process, then let the
 * signal take its course.  Diagnostics only.
 *
 * Async-signal-safe since round 4.  glibc raises SIGABRT for `double free or corruption` from
 * INSIDE free(), with the arena lock held: the round-3 handler (fopen / fgets, and backtrace()'s
 * first call, which dlopens libgcc_s and allocates) could block on that very lock - a child that
 * aborted then sat there until the driver script's timeout, which is what the script's old remark
 * "a torch step after these hung under the preload" recorded without a log.  Now: backtrace() is
 * warmed up in the constructor, /proc/self/maps is read with open/read into a static buffer, and
 * alarm(20) ends a handler that blocks anyway (SIGALRM's default action kills the process). */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static char g_buf[1 << 16];

static int wanted(const char* line) {
  static const char* const k[] = {"rccl", "amdhip", "hiprtc", "comgr", "hsa-runtime", "rocm_smi", "roctx",
                                  "rocprofiler", "libmhx", "libstdc++", "libtorch", 0};
  if (!strstr(line, "r-xp")) return 0;
  for (int i = 0; k[i]; ++i)
    if (strstr(line, k[i])) return 1;
  return 0;
}

static void on_signal(int sig) {
  void* frames[64];
  const char* head = sig == SIGABRT ? "\n== SIGABRT backtrace ==\n" : "\n== SIGSEGV backtrace ==\n";
  signal(sig, SIG_DFL); /* a fault in here takes the default course */
  alarm(20);
  (void)!write(2, head, strlen(head));
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  int fd = open("/proc/self/maps", O_RDONLY);
  if (fd >= 0) {
    const char* m = "== mapped (r-x) HIP / RCCL / compiler libraries ==\n";
    (void)!write(2, m, strlen(m));
    size_t have = 0;
    for (;;) {
      ssize_t r = read(fd, g_buf + have, sizeof g_buf - 1 - have);
      if (r <= 0) break;
      have += (size_t)r;
      g_buf[have] = 0;
      char* p = g_buf;
      for (char* nl; (nl = strchr(p, '\n')); p = nl + 1) {
        *nl = 0;
        if (wanted(p)) {
          (void)!write(2, p, strlen(p));
          (void)!write(2, "\n", 1);
        }
      }
      have = strlen(p);
      memmove(g_buf, p, have);
    }
    close(fd);
  }
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  void* warm[4];
  (void)backtrace(warm, 4); /* loads libgcc_s now, not inside the handler */
  signal(SIGABRT, on_signal);
  signal(SIGSEGV, on_signal);
}
