/* abort_bt.c -- LD_PRELOAD helper of tools/debug/rccl_exit_abort.py: on SIGABRT / SIGSEGV print
 * the C backtrace and the HIP / RCCL / comgr libraries mapped into the process, then let the
 * signal take its course.  Diagnostics only. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_signal(int sig) {
  void* frames[64];
  const char* head = sig == SIGABRT ? "\n== SIGABRT backtrace ==\n" : "\n== SIGSEGV backtrace ==\n";
  (void)!write(2, head, strlen(head));
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  FILE* f = fopen("/proc/self/maps", "r");
  if (f) {
    char line[1024];
    const char* m = "== mapped (r-x) HIP / RCCL / compiler libraries ==\n";
    (void)!write(2, m, strlen(m));
    while (fgets(line, sizeof line, f))
      if (strstr(line, "r-xp") &&
          (strstr(line, "rccl") || strstr(line, "amdhip") || strstr(line, "hiprtc") ||
           strstr(line, "comgr") || strstr(line, "hsa-runtime") || strstr(line, "rocm_smi") ||
           strstr(line, "roctx") || strstr(line, "rocprofiler") || strstr(line, "libmhx") ||
           strstr(line, "libstdc++") || strstr(line, "libtorch")))
        (void)!write(2, line, strlen(line));
    fclose(f);
  }
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  signal(SIGABRT, on_signal);
  signal(SIGSEGV, on_signal);
}
