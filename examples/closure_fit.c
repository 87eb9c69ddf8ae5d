/* closure_fit.c -- "pass a lambda" (mcmc-fitting.lisp:1134-1137) from a host that is not Python:
 * the BODY of
 *
 *   (lambda (x &key b0 b1 a1 mu1 w1 a2 mu2 w2 &allow-other-keys)
 *     (+ b0 (* b1 x) (* a1 (exp (- (expt (/ (- x mu1) w1) 2))))
 *                    (* a2 (exp (- (expt (/ (- x mu2) w2) 2))))))
 *
 * as the C-syntax text a shim's form walker produces (lisp-mcmc_amd/lisp/expr.lisp), handed to
 * mhx_set_function_expr with the plist's keys in ANOTHER order than the enumerated model's.
 * libmhx recognises background + Gaussian peaks below the ABI (csrc/mhx_expr.cpp) and runs the
 * kernel of BASELINE's config 2; with recognition off the same text is compiled as written.
 * Prints what mhx_expr_classify says, both kernels' names and log-posteriors, and walks.
 *
 *   gcc -I include examples/closure_fit.c -L lisp-mcmc_amd -lmhx -Wl,-rpath,$PWD/lisp-mcmc_amd -lm
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

#define N 20000
static double xs[N], ys[N], sg[N];

static double model(const double* t, double x) { /* plist order: see names[] */
  const double u1 = (x - t[3]) / t[1], u2 = (x - t[6]) / t[7];
  return t[4] + t[0] * x + t[2] * exp(-u1 * u1) + t[5] * exp(-u2 * u2);
}

int main(void) {
  /* the plist of the caller: (:b1 .3 :w1 .05 :a1 1 :mu1 .3 :b0 .5 :a2 .7 :mu2 .7 :w2 .08) */
  const char* names[8] = {"b1", "w1", "a1", "mu1", "b0", "a2", "mu2", "w2"};
  const double truth[8] = {0.3, 0.05, 1.0, 0.3, 0.5, 0.7, 0.7, 0.08};
  const int32_t index[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  const char* text =
      "(b0 + (b1 * x) + (a1 * exp((-ipow(((x - mu1) / w1), 2)))) + (a2 * exp((-ipow(((x - mu2) / w2), 2)))))";
  unsigned long long s = 88172645463325252ull; /* xorshift: the data only have to be noisy */
  for (int i = 0; i < N; ++i) {
    xs[i] = (double)i / (N - 1);
    double g = 0.0;
    for (int k = 0; k < 12; ++k) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      g += (double)(s >> 11) * (1.0 / 9007199254740992.0);
    }
    sg[i] = 0.1;
    ys[i] = model(truth, xs[i]) + 0.1 * (g - 6.0);
  }
  int32_t m = -1, shape[2], order[8], n_order = 0;
  TRY(mhx_expr_classify(text, names, 8, &m, shape, order, &n_order));
  printf("mhx_expr_classify: model %d shape {%d, %d}, the model's parameters in order:", m, shape[0],
         shape[1]);
  for (int j = 0; j < n_order; ++j) printf(" %s", names[order[j]]);
  printf("\n");
  if (m != MHX_MODEL_GAUSS_PEAKS || shape[0] != 2 || shape[1] != 2 || n_order != 8) return 2;

  double lp[2] = {0, 0};
  char kernel[2][256];
  for (int as_written = 0; as_written < 2; ++as_written) {
    mhx_config cfg;
    mhx_engine* e = NULL;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_chains = 4096;
    cfg.n_params = 8;
    cfg.n_functions = 1;
    cfg.seed = 7;
    TRY(mhx_create(&cfg, &e));
    if (as_written) TRY(mhx_set_expr_recognition(e, 0));
    TRY(mhx_set_function_expr(e, 0, text, names, index, 8));
    TRY(mhx_set_dataset(e, 0, xs, ys, sg, N, MHX_LIK_NORMAL));
    TRY(mhx_set_bounds(e, 0, NULL, NULL, NULL, 0));
    const char* kn = mhx_kernel_name(e);
    if (!kn) {
      fprintf(stderr, "mhx_kernel_name: %s\n", mhx_last_error());
      return 1;
    }
    snprintf(kernel[as_written], sizeof kernel[0], "%s", kn);
    TRY(mhx_logpost(e, truth, 1, &lp[as_written], NULL));
    printf("%-24s kernel %-44s log-posterior at the generating parameters %.12f\n",
           as_written ? "compiled as written:" : "recognised (default):", kernel[as_written], lp[as_written]);
    if (!as_written) {
      TRY(mhx_init_chains(e, truth, 1));
      TRY(mhx_adaptive_steps(e, 2000));
      static double best[4096 * 8], best_lp[4096];
      TRY(mhx_get_state(e, NULL, NULL, best, best_lp, NULL, NULL));
      int top = 0;
      for (int c = 1; c < 4096; ++c)
        if (best_lp[c] > best_lp[top]) top = c;
      printf("after (walker-adaptive-steps w 2000) of 4096 walkers: most likely");
      for (int j = 0; j < 8; ++j) printf(" %s %.4f", names[j], best[top * 8 + j]);
      printf("\n");
      for (int j = 0; j < 8; ++j)
        if (fabs(best[top * 8 + j] / truth[j] - 1.0) > 0.05) return 3;
    }
    mhx_destroy(e);
  }
  if (!strstr(kernel[0], "gauss22_normal") || !strstr(kernel[1], "rtc[expr")) return 4;
  /* the two evaluate the same function: a few ulp per point apart, far inside 1e-12 sum|term| */
  if (fabs(lp[0] - lp[1]) > 1e-9 * fabs(lp[0])) return 5;
  return 0;
}
