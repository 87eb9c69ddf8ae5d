/*
 * mhx.h -- C ABI of libmhx, the MI355X-native batched Metropolis-Hastings engine.
 *
 * This is the drop-in boundary for ONE path of afranson/Lisp-MCMC: everything
 * `walker-adaptive-steps` does per step, batched over many independent walkers
 * ("chains").  The reference has no FFI of its own (it is 100 % Common Lisp); the
 * boundary is the exported Lisp surface listed below, and each entry point here
 * names the reference function(s) whose work it takes over.  Citations are
 * file:line under the reference checkout, `M:` = mcmc-fitting.lisp.
 *
 *   walker-create              M:1132-1163   -> mhx_create + mhx_set_function /
 *                                               mhx_set_dataset / mhx_set_bounds +
 *                                               mhx_init_chains
 *   walker-make-step           M:1067-1070   -> mhx_logpost
 *   walker-take-step           M:1072-1095   -> mhx_take_step (Philox z,u), mhx_step_injected
 *                                               (caller's z,u) and the body of
 *                                               mhx_adaptive_advance
 *   walker-adaptive-steps-full M:862-942     -> mhx_adaptive_begin / _advance / _steps_full
 *   walker-adaptive-steps      M:946-947     -> mhx_adaptive_steps
 *   walker-many-steps          M:849-853     -> mhx_many_steps
 *   walker-get                 M:487-543     -> mhx_get_state / _acceptance / _lmatrix /
 *                                               _trace / _proposal_factor
 *   walker-modify              M:547-580     -> mhx_walker_modify (+ mhx_set_history)
 *   create-log-liklihood-function M:402-416  -> mhx_set_likelihood_expr
 *   prior-bounds-let           M:346-369     -> mhx_set_bounds (+ mhx_set_prior_expr)
 *   mfit-walker-estop          M:860-861     -> mhx_request_stop
 *   a list of walkers mapped in one image (M:1029-1033, nv-specific.lisp:58-66)
 *                                            -> mhx_group_* : one host process, several GPUs
 *
 * Conventions: every function returns MHX_OK (0) or a negative MHX_E* code and never
 * throws; the message of the last failure on the calling thread is available from
 * mhx_last_error().  All arrays are caller-allocated, caller-owned, dense row-major
 * IEEE binary64 / int32 HOST buffers; the engine owns its device memory and copies
 * in/out synchronously.  One handle is not thread-safe (the reference is single
 * threaded); distinct handles are independent.  No torch / C++ types appear here.
 */
#ifndef MHX_H
#define MHX_H

#ifndef __HIPCC_RTC__
#include <stddef.h>
#include <stdint.h>
#else /* hiprtc (run-time compiled expression kernels): no libc headers, built-in types */
using __hip_internal::int32_t;
using __hip_internal::int64_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
using __hip_internal::uint8_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define MHX_VERSION 200 /* 0.2.0 */

/* ---- limits ------------------------------------------------------------ */
#define MHX_MAX_PARAMS 63    /* d: length of the shared parameter vector (one lane of the
                                chain's wavefront per parameter, lane 63 draws the accept
                                uniform)                                              */
#define MHX_MAX_FUNCTIONS 16 /* K: functions / datasets of one (global) fit         */
#define MHX_MAX_FN_PARAMS 32 /* parameters one function gathers from the vector      */
#define MHX_MAX_BOUNDS 64    /* bounds in one prior-bounds-let block                */

/* ---- status codes ------------------------------------------------------ */
enum {
  MHX_OK = 0,
  MHX_EINVAL = -1,   /* bad argument (message says which)                          */
  MHX_ENOMEM = -2,   /* host or device allocation failed                           */
  MHX_EDEVICE = -3,  /* a HIP call failed / no usable gfx950 device                */
  MHX_ESTATE = -4,   /* call out of order (e.g. stepping before mhx_init_chains)   */
  MHX_EUNSUPPORTED = -5,
  MHX_ECOMM = -6     /* collective hook failed                                     */
};

/* ---- model designators ----------------------------------------------------
 * The reference's :function is a Lisp closure (lambda (x &key ... &allow-other-keys))
 * (M:1134-1137) that cannot cross to the GPU; the boundary takes an enumerated device
 * model plus an index map into the shared parameter vector (global fits share
 * parameters through one plist, README "Global Parameter Fitting").  Local parameter j
 * of function k is theta[param_index[j]].  `shape` carries the model's integer shape.
 *
 *  POLY          p = n_index;            f = c0 + c1 x + ... (Horner)
 *  GAUSS_PEAKS   shape = {nbg, npk};     f = bg(x) + sum_p A_p exp(-((x-mu_p)/w_p)^2)
 *                local order: bg_0..bg_{nbg-1}, then (A, mu, w) per peak
 *  LORENTZ_PEAKS shape = {nbg, npk};     f = bg(x) + sum_p A_p / (1 + ((x-mu_p)/w_p)^2)
 *  LORDER_MIXED  6 params scale, linewidth, x0, mix, bg0, bg1 (test.lisp:16-17 names)
 *                u=(x-x0)/linewidth; f = scale*(cos(mix)*(-2u) + sin(mix)*(1-u^2))/(1+u^2)^2
 *                                        + bg0 + bg1*x
 *  EXP_DECAY     3 params A, tau, c;     f = A exp(-x/tau) + c
 *  SINUSOID      4 params A, omega, phi, c;  f = A sin(omega x + phi) + c
 *  PVOIGT2       11 params A, b0, b1, mu1, w1, eta1, mu2, w2, eta2, rho, c2
 *                pv(x;mu,w,eta) = eta/(1+u^2) + (1-eta) exp(-u^2), u=(x-mu)/w
 *                f = b0 + b1 x + c2 x^2 + A (pv1 + rho pv2)
 */
enum {
  MHX_MODEL_POLY = 0,
  MHX_MODEL_GAUSS_PEAKS = 1,
  MHX_MODEL_LORENTZ_PEAKS = 2,
  MHX_MODEL_LORDER_MIXED = 3,
  MHX_MODEL_EXP_DECAY = 4,
  MHX_MODEL_SINUSOID = 5,
  MHX_MODEL_PVOIGT2 = 6,
  MHX_MODEL_EXPR = 7, /* set by mhx_set_function_expr, never passed to mhx_set_function */
  MHX_MODEL__COUNT = 8
};

/* ---- likelihood kinds (what the reference's :log-liklihood closure computes) */
enum {
  MHX_LIK_NORMAL = 0,        /* log-liklihood-normal (+ README weighted form) M:393-400 */
  MHX_LIK_NORMAL_CUTOFF = 1, /* log-liklihood-normal-cutoff, each term >= -5000 M:419-427 */
  MHX_LIK_POISSON = 2,       /* log-poisson over points, M:379-383 via M:402-416      */
  MHX_LIK_EXPR = 3           /* create-log-liklihood-function M:402-416: the per-point term
                                is the expression given to mhx_set_likelihood_expr       */
};

/* ---- adaptation modes --------------------------------------------------- */
enum {
  MHX_ADAPT_FAITHFUL = 0, /* per-walker rule of M:888-942, no collective              */
  MHX_ADAPT_POOLED = 1    /* extension: forward-step displacement statistics pooled
                             over all chains (and ranks) every 200 steps              */
};

/* ---- per-chain status (mhx_get_chain_status) ----------------------------- */
enum {
  MHX_CHAIN_RUNNING = 0,
  MHX_CHAIN_DONE = 1,          /* loop index reached n (M:904)                        */
  MHX_CHAIN_FP_TRAP = 2,       /* reference would have signalled an unhandled float
                                  trap (invalid/overflow outside the handler-case of
                                  M:891-894): the walker is frozen where it stood     */
  MHX_CHAIN_STOPPED = 3        /* mfit-walker-estop seen                              */
};

typedef struct mhx_engine mhx_engine;

/* Engine-wide configuration.  Zero-initialise, then set fields; 0 means "default". */
typedef struct mhx_config {
  int64_t n_chains;        /* walkers on THIS engine (one engine per GPU/rank)        */
  int32_t n_params;        /* d                                                        */
  int32_t n_functions;     /* K (1 for an ordinary fit)                                */
  int32_t device;          /* HIP device ordinal                                        */
  int32_t adapt_mode;      /* MHX_ADAPT_*                                               */
  uint64_t seed;           /* Philox key                                                */
  int64_t chain_offset;    /* global id of local chain 0 (multi-GPU sharding): the
                              Philox counter uses global ids, so results do not
                              depend on how chains are partitioned                      */
  int32_t history_capacity;/* steps of (prob, theta) kept per chain (ring).  0 ->
                              1024 = enough for every window the controller reads
                              (acceptance 1000, settle 10*max(50,d)); the reference
                              keeps everything (M:549) - set >= n to do the same       */
  int32_t poisson_logfact_double; /* 0: log-factorial summed in single floats as
                              M:379-380 does; 1: lgamma in binary64                    */
} mhx_config;

/* Options of one walker-adaptive-steps-full call (M:862).  Defaults of the Lisp
 * lambda list are applied by mhx_run_opts_default(). */
typedef struct mhx_run_opts {
  int64_t n;               /* :n, default 100000 (walker-adaptive-steps passes 30000)  */
  double temperature;      /* :temperature, default 1d3 (walker-adaptive-steps: 10)    */
  int32_t auto_mode;       /* :auto  0 = nil, 1 = :prob-settle (:slope-settle is
                              outside the path)                             */
  int64_t max_walker_length; /* :max-walker-length, 0 = nil                            */
  const double* l_matrix;  /* :l-matrix, d*d row-major, NULL = nil                     */
  int32_t l_matrix_per_chain; /* 1: l_matrix holds n_chains matrices                   */
} mhx_run_opts;

/* Collective hook for MHX_ADAPT_POOLED on several ranks: sum `n` doubles in place
 * over all ranks.  `buf` is a DEVICE pointer when device_buffer != 0 (RCCL path),
 * else a host pointer.  Return 0 on success. */
typedef int (*mhx_allreduce_fn)(void* ctx, double* buf, size_t n, int device_buffer);

/* ---- lifecycle ----------------------------------------------------------- */
int mhx_version(void);
/* Which sources this binary was built from: "csrc:<16 hex digits>", the leading digits of the
 * SHA-256 over the library's sources in a fixed order (csrc/Makefile: SRC_ID).  bench.py and
 * tools/profile_summary.py store it next to every measurement, so that an instruction count
 * taken from a committed rocprof summary is only ever combined with the binary that produced it. */
const char* mhx_build_id(void);
const char* mhx_last_error(void);
int mhx_device_count(int* count);
int mhx_create(const mhx_config* cfg, mhx_engine** out);
void mhx_destroy(mhx_engine* e);

/* ---- problem definition (walker-create, M:1132-1163) ---------------------- */
/* Function k: device model + gather map (param_index[j] in [0,d) ).               */
int mhx_set_function(mhx_engine* e, int k, int model_id, const int32_t* shape, int n_shape,
                     const int32_t* param_index, int n_index);
/* Dataset k in the layout clean-data/clean-data-error produce (M:774-825): x, y and a
 * per-point sigma (sigma == NULL -> 1.0 everywhere, the (or data-error 1) of M:1144).
 * The engine copies.  For MHX_LIK_POISSON y holds the counts k_i and sigma is ignored. */
int mhx_set_dataset(mhx_engine* e, int k, const double* x, const double* y,
                    const double* sigma, size_t n, int likelihood);
/* The same with a VECTOR-VALUED x: "multiple or linked independent variables" - the reference
 * hands each element of the x list to the function as it is (M:400), and a closure reads its
 * components with (elt x 0), (elt x 1) (M:1136-1137).  xcols[j][i] = component j of point i,
 * n_cols 1 or 2.  Component 0 is the x of every enumerated model and of the windows' ranges;
 * an expression function names the components xcol0 (= x) and xcol1.  Not with
 * MHX_LIK_NORMAL_CUTOFF (MHX_EUNSUPPORTED). */
int mhx_set_dataset_cols(mhx_engine* e, int k, const double* const* xcols, int n_cols,
                         const double* y, const double* sigma, size_t n, int likelihood);
/* prior-bounds-let block of function k (M:346-369): idx[i] < 0 means "key absent from
 * the plist" (getf default 0d0, M:353).  n == 0 -> log-prior-flat (M:340-343). */
int mhx_set_bounds(mhx_engine* e, int k, const int32_t* idx, const double* lo,
                   const double* hi, int n);
/* Function k given as an EXPRESSION: what a host shim makes of the body
 * of (lambda (x &key a b &allow-other-keys) <body>) (M:1134-1137).  `expr` is a C-syntax
 * arithmetic expression over `x` (with a vector-valued x, mhx_set_dataset_cols: `xcol0`, `xcol1`), the
 * identifiers in param_names (local parameter j =
 * theta[param_index[j]]), numeric literals, + - * / ?: < <= > >= == != && || !, and the
 * functions exp log sqrt sin cos tan atan tanh abs pow min max.  It is compiled for gfx950
 * with hiprtc into the same fused kernels when the problem is finalised (first
 * mhx_init_chains / mhx_logpost), and evaluated without contraction.
 * A body that IS one of the enumerated models - a polynomial background c0 + c1 x + ... plus
 * Gaussian peaks a * exp(-ipow((x - mu) / w, 2)) or Lorentzian peaks a / (1 + ipow((x - mu) / w,
 * 2)) over distinct parameters (pow(u, 2.0), u * u and -1 * S are understood; csrc/mhx_expr.cpp) -
 * is recognised here, below the ABI, and runs as MHX_MODEL_POLY / _GAUSS_PEAKS / _LORENTZ_PEAKS
 * with the gather map permuted into that model's order: the same function through the peak
 * kernels' fused arithmetic (a few ulp per point from the text's own rounding, inside the path's
 * tolerance), with their per-window peak skipping and uniform-grid recurrence - the lambda a
 * Lisp host hands to walker-create gets the kernels of BASELINE's config 2 without knowing
 * them.  A function whose likelihood is MHX_LIK_EXPR always stays an expression. */
int mhx_set_function_expr(mhx_engine* e, int k, const char* expr, const char* const* param_names,
                          const int32_t* param_index, int n_index);
/* on = 0: every expression of this engine is compiled exactly as written (default: on). */
int mhx_set_expr_recognition(mhx_engine* e, int on);
/* What mhx_set_function_expr makes of `expr` - needs no engine and no device (hosts' logs, tests):
 * *model = the MHX_MODEL_* that serves it (MHX_MODEL_EXPR: compiled as written), shape[2] its
 * {nbg, npk}, order[j] = which of param_names is local parameter j of that model (*n_order
 * entries, at most n_names; 0 for MHX_MODEL_EXPR).  shape, order, n_order may be NULL. */
int mhx_expr_classify(const char* expr, const char* const* param_names, int n_names,
                      int32_t* model, int32_t* shape, int32_t* order, int32_t* n_order);
/* Body of function k's prior-bounds-let prior (M:366-369) as an expression over
 * `bounds_total` (the sum of the block set by mhx_set_bounds) and the identifiers in `names`
 * (names[i] = theta[index[i]]), e.g. NV's "bounds_total + (mu1 > mu2 ? -1e9 : 0.0)". */
int mhx_set_prior_expr(mhx_engine* e, int k, const char* expr, const char* const* names,
                       const int32_t* index, int n);
/* Per-point log-likelihood of function k as an expression: the closure handed to
 * create-log-liklihood-function (M:402-416), (lambda (y model error) <body>), with `y` the
 * measured value, `model` the model's prediction at that x and `error` the point's sigma as
 * its docstring defines them (the reference's code passes the WHOLE stddev list as the third
 * argument, M:415, so only bodies that ignore `error` ever ran there).  The log-likelihood is
 * the plain sum of the terms over the points.  Dataset k must have been set with
 * MHX_LIK_EXPR (x, y, sigma are kept as given) and function k with mhx_set_function_expr. */
int mhx_set_likelihood_expr(mhx_engine* e, int k, const char* expr);
/* First step of every chain (M:1148-1150): theta0 is [n_chains][d], or [d] when
 * broadcast != 0.  Resets history, age, length, most-likely step. */
int mhx_init_chains(mhx_engine* e, const double* theta0, int broadcast);

/* ---- pure evaluation / injected-randomness parity hooks -------------------- */
/* walker-make-step's prob for n arbitrary parameter vectors theta[n][d] (M:1067-1070).
 * parts (optional, [n][2]) receives the likelihood sum and the prior sum. */
int mhx_logpost(mhx_engine* e, const double* theta, size_t n, double* out, double* parts);
/* One walker-take-step per chain (M:1072-1095) with the caller's randomness:
 * L  [d][d] (per_chain_l == 0) or [n_chains][d][d];  z [n_chains][d] standard normals
 * (what alexandria:gaussian-random would have returned, M:687);  u [n_chains] the
 * (random 1.0d0) of M:1092;  T [n_chains] temperatures.  accepted_out (optional)
 * receives 1 where the proposal was taken. */
int mhx_step_injected(mhx_engine* e, const double* L, int per_chain_l, const double* z,
                      const double* u, const double* T, uint8_t* accepted_out);

/* ---- the controller (walker-adaptive-steps-full, M:862-942) ---------------- */
void mhx_run_opts_default(mhx_run_opts* o);
/* Everything before the do loop: schedule, steps-to-settle, initial L (M:866-901). */
int mhx_adaptive_begin(mhx_engine* e, const mhx_run_opts* o);
/* Run up to max_iters iterations of the do loop (M:902-942) for every chain that is
 * still running; *n_running (optional) receives how many chains have not finished.
 * (Asking for n_running lets the engine look at the chain states: between launches it deals
 * the chains still walking evenly over the GPU's workgroups - a chain's results do not depend
 * on where it runs; MHX_NO_COMPACT=1 keeps every chain in its first place.) */
int mhx_adaptive_advance(mhx_engine* e, int64_t max_iters, int64_t* n_running);
/* begin + advance until every chain is done or mhx_request_stop was called. */
int mhx_adaptive_steps_full(mhx_engine* e, const mhx_run_opts* o);
/* (walker-adaptive-steps w n): n, :temperature 10, :auto :prob-settle (M:946-947). */
int mhx_adaptive_steps(mhx_engine* e, int64_t n);
/* walker-many-steps (M:849-853): n steps with a constant L, temperature 1.  The nil default
 * of M:851, diag(1e-2 * median-params), is formed by the host shims (mhx_get_trace); pass L. */
int mhx_many_steps(mhx_engine* e, int64_t n, const double* L, int per_chain_l);
/* (walker-take-step w :l-matrix L :temperature T) for every chain (M:1072-1095): ONE step with
 * the device's own randomness (Philox, the next draw of each chain).  The nil default of
 * M:1074, diag(1e-2 * most-likely-params of the newest 1000 steps), is the shims' to form. */
int mhx_take_step(mhx_engine* e, const double* L, int per_chain_l, double temperature);
int mhx_request_stop(mhx_engine* e);

/* Multi-rank pooled adaptation through a caller-supplied sum (MPI, a test stub ...): installs the
 * all-reduce used every adaptation tick.  The native path is RCCL: mhx_comm_init_rank (one
 * process per GPU) or mhx_group_create (one process, several GPUs) below.  A hook installed
 * AFTER mhx_comm_init_rank replaces that communicator (it is destroyed): every rank must then
 * exchange through its hook. */
int mhx_set_allreduce(mhx_engine* e, mhx_allreduce_fn fn, void* ctx, int wants_device_buffer);

/* ---- native RCCL (librccl.so is loaded on first use; MHX_ECOMM when it is absent) ----------
 * One process per GPU: rank 0 calls mhx_comm_get_unique_id and hands the 128 bytes to the other
 * ranks by whatever channel the host has (a file, MPI, torch.distributed ...); then EVERY rank
 * calls mhx_comm_init_rank on its engine (collective: ncclCommInitRank).  From then on the
 * pooled tick of MHX_ADAPT_POOLED is k_pool_stats -> k_pool_reduce -> ncclAllReduce(1+d+d*d
 * doubles, sum) -> k_pool_factor, all on the engine's stream with no host synchronisation. */
int mhx_comm_get_unique_id(uint8_t id[128]);
int mhx_comm_init_rank(mhx_engine* e, const uint8_t id[128], int rank, int n_ranks);

/* ---- one host process, several GPUs ---------------------------------------------------------
 * The reference runs many walkers as a list mapped in ONE Lisp image (M:1029-1033); a group is
 * that list spread over GPUs: cfg->n_chains walkers in all, contiguous global id ranges per
 * device (mhx_group_partition; Philox counters use global ids, so the walks do not depend on the
 * number of devices), one engine + one HIP stream per device, datasets replicated.  Every
 * group call enqueues its launches on ALL devices before it waits for any.  With
 * MHX_ADAPT_POOLED and more than one device the communicators come from ncclCommInitAll and the
 * tick's all-reduce is issued for all devices inside ncclGroupStart/End.  (Engines of a group
 * that name the SAME device - a rehearsal on one GPU - sum their statistics through the host.)
 * cfg->device is ignored; cfg->chain_offset is the global id of the group's first chain.
 * mhx_group_engine(g, i) exposes engine i to every per-engine entry point above (read-backs of
 * one walker, mhx_logpost ...); problem definition goes through the mhx_group_set_* twins. */
typedef struct mhx_group mhx_group;
int mhx_group_partition(int64_t n_chains, int n_parts, int part, int64_t* first, int64_t* count);
int mhx_group_create(const mhx_config* cfg, const int32_t* devices, int n_devices,
                     mhx_group** out);
void mhx_group_destroy(mhx_group* g);
int mhx_group_size(const mhx_group* g);
mhx_engine* mhx_group_engine(mhx_group* g, int i);
int mhx_group_chain_range(const mhx_group* g, int i, int64_t* first, int64_t* count);
int mhx_group_set_function(mhx_group* g, int k, int model_id, const int32_t* shape, int n_shape,
                           const int32_t* param_index, int n_index);
int mhx_group_set_dataset(mhx_group* g, int k, const double* x, const double* y,
                          const double* sigma, size_t n, int likelihood);
int mhx_group_set_dataset_cols(mhx_group* g, int k, const double* const* xcols, int n_cols,
                               const double* y, const double* sigma, size_t n, int likelihood);
int mhx_group_set_bounds(mhx_group* g, int k, const int32_t* idx, const double* lo,
                         const double* hi, int n);
int mhx_group_set_function_expr(mhx_group* g, int k, const char* expr,
                                const char* const* param_names, const int32_t* param_index,
                                int n_index);
int mhx_group_set_expr_recognition(mhx_group* g, int on);
int mhx_group_set_prior_expr(mhx_group* g, int k, const char* expr, const char* const* names,
                             const int32_t* index, int n);
int mhx_group_set_likelihood_expr(mhx_group* g, int k, const char* expr);
/* theta0: [cfg->n_chains][d] in global chain order, or [d] when broadcast != 0 */
int mhx_group_init_chains(mhx_group* g, const double* theta0, int broadcast);
int mhx_group_adaptive_begin(mhx_group* g, const mhx_run_opts* o);
int mhx_group_adaptive_advance(mhx_group* g, int64_t max_iters, int64_t* n_running);
int mhx_group_adaptive_steps_full(mhx_group* g, const mhx_run_opts* o);
int mhx_group_request_stop(mhx_group* g);
/* gathered in global chain order; any pointer may be NULL */
int mhx_group_get_state(mhx_group* g, double* theta, double* logpost, double* best_theta,
                        double* best_logpost, int64_t* length, int64_t* age);
int mhx_group_get_counters(mhx_group* g, uint64_t* chain_steps, uint64_t* kernel_launches);

/* ---- read-back (walker-get, M:487-543) ------------------------------------ */
/* Any pointer may be NULL.  theta/best_theta [n_chains][d]; others [n_chains]. */
int mhx_get_state(mhx_engine* e, double* theta, double* logpost, double* best_theta,
                  double* best_logpost, int64_t* length, int64_t* age);
/* The same for ONE chain (what the accessors of one walker need: walker-last-step,
 * walker-most-likely-step, walker-length, walker-age): d doubles instead of n_chains * d. */
int mhx_get_chain(mhx_engine* e, int64_t chain, double* theta, double* logpost,
                  double* best_theta, double* best_logpost, int64_t* length, int64_t* age);
int mhx_get_chain_status(mhx_engine* e, int32_t* status, int64_t* loop_index);
int mhx_get_lmatrix(mhx_engine* e, double* L /* [n_chains][d][d] */);
int mhx_get_temperature(mhx_engine* e, double* T /* [n_chains] */);
/* (walker-get w :get :acceptance :take take) for every chain, as a double. */
int mhx_get_acceptance(mhx_engine* e, int take, double* out);
/* Newest-first steps of one chain, as (walker-get w :get :steps :take take):
 * prob[take], theta[take][d]; *n_out = steps actually available. */
int mhx_get_trace(mhx_engine* e, int64_t chain, int take, double* prob, double* theta,
                  int* n_out);
/* (walker-get w :get :l-matrix :take take) of one chain recomputed on the host from the
 * device trace: status 0 ok, 1 caught error (type-error/div0/overflow -> fallback in
 * M:891-894), 2 uncaught invalid-operation.  n_forward = (length :forward-steps). */
int mhx_get_proposal_factor(mhx_engine* e, int64_t chain, int take, double* L_out,
                            int* status, int* n_forward);

/* Restore a saved walk (walker-load, sketched in the comments M:987-1001): prob[n], theta[n][d]
 * NEWEST FIRST, as walker-save would have written them.  Sets the ring (newest
 * min(n, history_capacity) steps), last-step, length, age = n and the most-likely step. */
int mhx_set_history(mhx_engine* e, int64_t chain, const double* prob, const double* theta, int n);

/* walker-modify's list surgery (M:566-578) for every chain: :burn-walks n drops the n oldest
 * steps, :keep-walks n keeps the n newest, :reset makes the walk its oldest retained step,
 * :reset-to-most-likely the most likely step (both also move last-step); n is ignored by the
 * resets.  (:add-step happens inside walker-take-step on the device; :delete = mhx_destroy.) */
enum {
  MHX_MODIFY_BURN_WALKS = 0,
  MHX_MODIFY_KEEP_WALKS = 1,
  MHX_MODIFY_RESET = 2,
  MHX_MODIFY_RESET_TO_MOST_LIKELY = 3
};
int mhx_walker_modify(mhx_engine* e, int action, int64_t n);

/* MHX_ADAPT_POOLED read-back: stats [1+d+d*d] = (n, sum delta, sum delta delta^T) pooled over
 * chains (and ranks) at the last 200-iteration tick; L_pool [d][d] = (2.38^2/d) chol(cov);
 * valid = 1 when that factor is in use; refreshes = ticks performed. */
int mhx_get_pooled(mhx_engine* e, double* stats, double* L_pool, int32_t* valid,
                   uint64_t* refreshes);

/* Total chain-steps taken by this engine since creation (all chains). */
int mhx_get_counters(mhx_engine* e, uint64_t* chain_steps, uint64_t* kernel_launches);

/* Which kernels serve the current problem, for logs and benchmarks: "w16/gauss22_normal" (an
 * ahead-of-time specialisation of the 16-chains-per-workgroup family), "w8/generic", or
 * "w16/rtc[PeaksModel<2, 3, false>:normal]" (compiled at run time, one entry per function).
 * Finalises the problem like mhx_init_chains does; NULL (and mhx_last_error) if that fails.
 * The string lives until the problem is changed or the engine destroyed. */
const char* mhx_kernel_name(mhx_engine* e);

/* Timing of the step kernel on the engine's own stream (HIP events): average
 * milliseconds per launch and launches since the last reset. */
int mhx_kernel_timing(mhx_engine* e, int reset, double* avg_ms, uint64_t* launches,
                      double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* MHX_H */
