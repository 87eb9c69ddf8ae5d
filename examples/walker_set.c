/* walker_set.c -- a list of walkers mapped in one image (mcmc-fitting.lisp:1029-1033,
 * nv-specific.lisp:58-66), from plain C through the mhx_group_* entry points of include/mhx.h:
 * ONE host process, one engine per device, contiguous global chain ranges, every device's
 * launch enqueued before any is waited for; with MHX_ADAPT_POOLED the 200-iteration tick's
 * all-reduce goes through RCCL on the engines' own streams.
 *
 *   walker_set [n_devices [chains]]      devices 0 .. n_devices-1 (default 1), 256 walkers
 *
 * The problem: two Gaussian peaks on a linear background, 8 parameters, 4000 points, weighted
 * normal likelihood, bounds prior - what the reference would write as
 *   (walker-create :function (lambda (x &key b0 b1 a1 mu1 w1 a2 mu2 w2 &allow-other-keys) ...)
 *                  :data data :params params :data-error sigmas
 *                  :log-prior (prior-bounds-let ((:b0 .25 .75) ...) bounds-total))
 * once per walker.
 *
 *   gcc -I include examples/walker_set.c -L lisp-mcmc_amd -lmhx -Wl,-rpath,$PWD/lisp-mcmc_amd -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mhx.h"

#define TRY(call)                                                          \
  do {                                                                     \
    int rc_ = (call);                                                      \
    if (rc_ != MHX_OK) {                                                   \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mhx_last_error());     \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static double lcg(unsigned long long* s) { /* uniform (0,1): data and starts only */
  *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((double)(*s >> 11) + 0.5) / 9007199254740992.0;
}
static double gauss(unsigned long long* s) {
  return sqrt(-2.0 * log(lcg(s))) * cos(6.283185307179586 * lcg(s));
}

int main(int argc, char** argv) {
  enum { N = 4000, D = 8 };
  const int n_dev = argc > 1 ? atoi(argv[1]) : 1;
  const long chains = argc > 2 ? atol(argv[2]) : 256;
  const double star[D] = {0.5, 0.3, 1.0, 0.3, 0.05, 0.7, 0.7, 0.08}; /* b0 b1 a1 mu1 w1 a2 mu2 w2 */
  static double x[N], y[N], sigma[N];
  unsigned long long s = 12345;
  for (int i = 0; i < N; ++i) {
    x[i] = (double)i / (N - 1);
    sigma[i] = 0.05 + 0.1 * lcg(&s);
    const double t1 = (x[i] - star[3]) / star[4], t2 = (x[i] - star[6]) / star[7];
    y[i] = star[0] + star[1] * x[i] + star[2] * exp(-t1 * t1) + star[5] * exp(-t2 * t2) +
           sigma[i] * gauss(&s);
  }
  int32_t devices[64], idx[D], shape[2] = {2, 2};
  double lo[D], hi[D];
  if (n_dev < 1 || n_dev > 64 || chains < n_dev) return 2;
  for (int i = 0; i < n_dev; ++i) devices[i] = i;
  for (int j = 0; j < D; ++j) {
    idx[j] = j;
    lo[j] = 0.5 * star[j];
    hi[j] = 1.5 * star[j];
  }
  mhx_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_chains = chains; /* in all; device i gets its contiguous share (mhx_group_partition) */
  cfg.n_params = D;
  cfg.n_functions = 1;
  cfg.seed = 7;
  cfg.adapt_mode = n_dev > 1 ? MHX_ADAPT_POOLED : MHX_ADAPT_FAITHFUL;
  mhx_group* g = NULL;
  TRY(mhx_group_create(&cfg, devices, n_dev, &g));
  TRY(mhx_group_set_function(g, 0, MHX_MODEL_GAUSS_PEAKS, shape, 2, idx, D));
  TRY(mhx_group_set_dataset(g, 0, x, y, sigma, N, MHX_LIK_NORMAL));
  TRY(mhx_group_set_bounds(g, 0, idx, lo, hi, D));
  double* th0 = malloc(sizeof(double) * (size_t)chains * D);
  for (long c = 0; c < chains; ++c)
    for (int j = 0; j < D; ++j) th0[c * D + j] = star[j] * (1.0 + 0.01 * gauss(&s));
  TRY(mhx_group_init_chains(g, th0, 0));
  mhx_run_opts o;
  mhx_run_opts_default(&o);
  o.n = 6000;           /* (walker-adaptive-steps w 6000) for every walker of the set */
  o.temperature = 10.0;
  o.auto_mode = 1;
  TRY(mhx_group_adaptive_steps_full(g, &o));
  double* best = malloc(sizeof(double) * (size_t)chains * D);
  double* best_lp = malloc(sizeof(double) * (size_t)chains);
  int64_t* age = malloc(sizeof(int64_t) * (size_t)chains);
  TRY(mhx_group_get_state(g, NULL, NULL, best, best_lp, NULL, age));
  uint64_t steps = 0, launches = 0;
  TRY(mhx_group_get_counters(g, &steps, &launches));
  double mean[D] = {0};
  int ok = 1;
  for (long c = 0; c < chains; ++c)
    for (int j = 0; j < D; ++j) mean[j] += best[c * D + j] / (double)chains;
  printf("%d device(s), %ld walkers, %llu chain-steps in %llu launches\n", mhx_group_size(g),
         chains, (unsigned long long)steps, (unsigned long long)launches);
  for (int i = 0; i < mhx_group_size(g); ++i) {
    int64_t first = 0, count = 0;
    TRY(mhx_group_chain_range(g, i, &first, &count));
    printf("  device %d walks chains %lld .. %lld (%s)\n", (int)devices[i], (long long)first,
           (long long)(first + count - 1), mhx_kernel_name(mhx_group_engine(g, i)));
  }
  printf("mean most-likely parameters (generating values):\n");
  for (int j = 0; j < D; ++j) {
    printf("  %8.5f (%g)\n", mean[j], star[j]);
    ok = ok && fabs(mean[j] / star[j] - 1.0) < 0.05;
  }
  for (long c = 0; c < chains; ++c) ok = ok && isfinite(best_lp[c]) && age[c] > 2000;
  mhx_group_destroy(g);
  free(th0);
  free(best);
  free(best_lp);
  free(age);
  return ok ? 0 : 3;
}
