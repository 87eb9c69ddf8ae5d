/*
 * mhx_oracle.h -- CPU restatement of the walker-adaptive-steps path of afranson/Lisp-MCMC.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (lisp-mcmc_amd/, include/) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker.
 *
 * Written from the reference's source TEXT (the reference is Common Lisp and no Lisp
 * implementation exists in the build container or on the GPU box), every
 * function citing the lines it follows (M: = mcmc-fitting.lisp).  Pinned by the
 * reference's own known-answer comments (covariance M:745, Cholesky factor M:749-751)
 * and by closed-form values of its usage examples (tests/golden/).  The random stream is
 * NOT pinned by the reference (unseeded cl:random + alexandria:gaussian-random,
 * M:687, M:1092): parity of random draws is "parity unpinned"; the oracle and the
 * engine share one documented Philox4x32-10 + Box-Muller specification instead.
 *
 * Arithmetic is IEEE binary64 with NO contraction (compile with -ffp-contract=off), sums
 * strictly left to right like (reduce #'+ ...) (M:400), libm exp/log/pow/cos/sqrt as the
 * SBCL runtime calls them.
 */
#ifndef MHX_ORACLE_H
#define MHX_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status of (walker-get :get :l-matrix) under the handler-case of M:891-894 */
enum {
  ORC_L_OK = 0,
  ORC_L_CAUGHT = 1,  /* type-error (no forward steps), division-by-zero or overflow */
  ORC_L_INVALID = 2, /* 0/0 -> floating-point-invalid-operation: NOT handled, M:891-894 */
  ORC_L_EMPTY = 3    /* one forward step -> no displacement -> 0x0 matrix (M:936 keeps L) */
};

enum { ORC_RUNNING = 0, ORC_DONE = 1, ORC_FP_TRAP = 2, ORC_STOPPED = 3 };

typedef struct orc_problem orc_problem;
typedef struct orc_walker orc_walker;

/* ---- primitives (M:372-383) ------------------------------------------- */
double orc_log_normal(double x, double mu, double sigma);
double orc_log_factorial(double k, int in_double); /* single-float sum unless in_double */
double orc_log_poisson(double lambda, double k, int in_double);
double orc_bound_penalty(double p, double lo, double hi); /* one p-bound of M:358-360 */

/* ---- models (formula spec in include/mhx.h) ------------------------------ */
double orc_model_eval(int model, const int32_t* shape, const double* p, int np, double x);

/* ---- proposal linear algebra (M:583-643, M:679-700) ----------------------- */
/* population covariance of m vectors v[m][d] -> cov[d][d]; returns ORC_L_* */
int orc_lplist_covariance(const double* v, int m, int d, double* cov);
int orc_cholesky(const double* cov, int d, double* L);
void orc_covariant_sample(const double* theta, const double* L, const double* z, int d,
                          double* out);

/* ---- the shared random-stream specification ------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_det_log(double x);     /* fdlibm-style log, pure IEEE mul/add/div        */
double orc_det_cos2pi(double t);  /* cos(2 pi t), t in [0,1)                         */
double orc_rng_normal(uint64_t seed, uint64_t chain, uint64_t draw, uint32_t slot);
double orc_rng_uniform(uint64_t seed, uint64_t chain, uint64_t draw);

/* ---- temperature schedule (M:875-878) ------------------------------------- */
/* writes temp_steps = max(n, 10*steps_to_settle) entries; returns temp_steps */
int64_t orc_temperature_schedule(int64_t n, int d, double temperature, double* out,
                                 int64_t cap);

/* ---- problem = functions + datasets + priors (walker-create inputs) -------- */
orc_problem* orc_problem_create(int d, int K);
void orc_problem_destroy(orc_problem* p);
int orc_problem_set_function(orc_problem* p, int k, int model, const int32_t* shape,
                             int n_shape, const int32_t* idx, int n_idx);
int orc_problem_set_dataset(orc_problem* p, int k, const double* x, const double* y,
                            const double* sigma, size_t n, int lik);
int orc_problem_set_bounds(orc_problem* p, int k, const int32_t* idx, const double* lo,
                           const double* hi, int n);
void orc_problem_set_logfact_double(orc_problem* p, int flag);
/* walker-make-step's prob (M:1067-1070); parts[0] = sum ll, parts[1] = sum lp */
double orc_logpost(const orc_problem* p, const double* theta, double* parts);
/* MIRROR of the GPU kernel's arithmetic (GAUSS_PEAKS + normal likelihood problems only, NaN
 * otherwise): bit-identical to the device, see mhx_oracle.c */
/* mirror mode only: 0 = restate the kernel with MHX_NO_RECURRENCE=1 (direct exp at every point) */
void orc_mirror_set_recurrence(int on);
/* mirror mode only: 0 = restate the kernel with MHX_NO_WINDOW_GRIDS=1 (the recurrence only on
 * datasets whose x are ONE grid; default 1: window by window, csrc/mhx_engine.cpp) */
void orc_mirror_set_window_grids(int on);
double orc_logpost_mirror(const orc_problem* p, const double* theta, double* parts);
/* sum_i |term_i| over all likelihood points: the scale of the stated tolerance */
double orc_logpost_abs_terms(const orc_problem* p, const double* theta);

/* ---- walker (M:462-581, M:1067-1163) -------------------------------------- */
orc_walker* orc_walker_create(const orc_problem* p, const double* theta0);
/* mirror != 0: posterior and log u in the kernel's arithmetic (bit-comparable runs) */
orc_walker* orc_walker_create2(const orc_problem* p, const double* theta0, int mirror);
void orc_walker_destroy(orc_walker* w);
/* walker-take-step (M:1072-1095) with the caller's randomness; returns 1 accepted,
 * 0 rejected, -1 the reference would have trapped (walker unchanged) */
int orc_walker_take_step_injected(orc_walker* w, const double* L, const double* z, double u,
                                  double T);
/* walker-modify M:566-578: 0 :burn-walks n, 1 :keep-walks n, 2 :reset, 3 :reset-to-most-likely */
int orc_walker_modify(orc_walker* w, int action, int64_t n);
int64_t orc_walker_length(const orc_walker* w);
int64_t orc_walker_age(const orc_walker* w);
void orc_walker_last(const orc_walker* w, double* theta, double* prob);
void orc_walker_best(const orc_walker* w, double* theta, double* prob);
/* newest-first, like (walker-get :get :steps :take take); returns count */
int orc_walker_trace(const orc_walker* w, int take, double* prob, double* theta);
/* (walker-get :get :acceptance :take take) = num/den (M:506-508) */
void orc_walker_acceptance(const orc_walker* w, int take, int64_t* num, int64_t* den);
int orc_walker_forward_count(const orc_walker* w, int take);
int orc_walker_l_matrix(const orc_walker* w, int take, double* L, int* n_forward);

/* ---- controller (M:862-947) ------------------------------------------------ */
typedef struct orc_run_opts {
  int64_t n;
  double temperature;
  int32_t auto_mode; /* 1 = :prob-settle */
  int64_t max_walker_length;
  const double* l_matrix;
} orc_run_opts;
int orc_walker_adaptive_begin(orc_walker* w, const orc_run_opts* o, uint64_t seed,
                              uint64_t chain_id);
/* up to max_iters iterations of the do loop; returns ORC_* status */
int orc_walker_adaptive_advance(orc_walker* w, int64_t max_iters);
int orc_walker_status(const orc_walker* w);
int64_t orc_walker_loop_index(const orc_walker* w);
double orc_walker_temperature(const orc_walker* w);
void orc_walker_current_l(const orc_walker* w, double* L);
void orc_walker_request_stop(orc_walker* w);
/* walker-many-steps (M:849-853) with Philox randomness */
int orc_walker_many_steps(orc_walker* w, int64_t n, const double* L, uint64_t seed,
                          uint64_t chain_id);

#ifdef __cplusplus
}
#endif
#endif
