/* line_fit.c -- the reference's basic example (mcmc-fitting.lisp:1186) straight through the
 * C ABI of include/mhx.h, from plain C:
 *
 *   (mfit:mcmc-fit :function (lambda (x &key m b &allow-other-keys) (+ b (* m x)))
 *                  :data '((-4 -1 2 5 10) (0 2 5 9 13)) :params '(:b -1 :m 2) :data-error 0.2)
 *
 *   gcc -I include examples/line_fit.c -L lisp-mcmc_amd -lmhx -Wl,-rpath,$PWD/lisp-mcmc_amd -lm
 */
#include <math.h>
#include <stdio.h>
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

int main(void) {
  const double x[5] = {-4, -1, 2, 5, 10}, y[5] = {0, 2, 5, 9, 13};
  const double sigma[5] = {0.2, 0.2, 0.2, 0.2, 0.2};
  const double theta0[2] = {-1.0, 2.0}; /* plist order: :b :m */
  const int32_t idx[2] = {0, 1};        /* POLY local order c0 c1 = b m */
  mhx_config cfg;
  mhx_engine* e = NULL;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_chains = 1;
  cfg.n_params = 2;
  cfg.n_functions = 1;
  cfg.seed = 1;
  TRY(mhx_create(&cfg, &e));
  TRY(mhx_set_function(e, 0, MHX_MODEL_POLY, NULL, 0, idx, 2));
  TRY(mhx_set_dataset(e, 0, x, y, sigma, 5, MHX_LIK_NORMAL));
  TRY(mhx_set_bounds(e, 0, NULL, NULL, NULL, 0)); /* log-prior-flat */
  TRY(mhx_init_chains(e, theta0, 1));
  double th[2], lp, best[2], best_lp;
  int64_t length, age;
  TRY(mhx_get_state(e, th, &lp, best, &best_lp, &length, &age));
  printf("first step: prob %.16g (closed form -1821.5475031038527), length %lld\n", lp,
         (long long)length);
  if (fabs(lp - -1821.5475031038527) > 1e-9) return 2;
  TRY(mhx_adaptive_steps(e, 5000)); /* (walker-adaptive-steps w 5000) */
  TRY(mhx_get_state(e, th, &lp, best, &best_lp, &length, &age));
  double acc;
  TRY(mhx_get_acceptance(e, 1000, &acc));
  printf("after %lld steps: most likely b = %.4f m = %.4f (least squares 3.4778 0.9676), "
         "prob %.4f, acceptance(1000) %.3f\n",
         (long long)(age - 1), best[0], best[1], best_lp, acc);
  mhx_destroy(e);
  return (fabs(best[0] - 3.4778) < 0.2 && fabs(best[1] - 0.9676) < 0.05) ? 0 : 3;
}
