/*
 * mhx_oracle.c -- CPU restatement of the reference's walker-adaptive-steps path.
 * TEST INFRASTRUCTURE ONLY (see mhx_oracle.h).  M: = /root/reference/mcmc-fitting.lisp.
 */
#include "mhx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mhx.h" /* enum values of the public interface only */

/* ======================================================================== */
/* primitives                                                                */
/* ======================================================================== */

/* M:372-377  (+ (* -1/2 (log (* 2 pi))) (* -1 (log sigma)) (* -1/2 (expt (/ (- x mu) sigma) 2d0)))
 * (+ a b c) associates left; (expt q 2d0) with a double exponent goes to libm pow. */
double orc_log_normal(double x, double mu, double sigma) {
  double a = -0.5 * log(2.0 * M_PI);
  double b = -1.0 * log(sigma);
  double q = (x - mu) / sigma;
  double c = -0.5 * pow(q, 2.0);
  return (a + b) + c;
}

/* M:379-380 (reduce (lambda (x y) (+ x (log y))) (up-to n)) over the INTEGERS 0..n
 * (M:256-261): (log m) of an integer is a single-float, so the running sum is single. */
double orc_log_factorial(double k, int in_double) {
  long n = (long)k;
  if (in_double) return lgamma((double)n + 1.0);
  if (n <= 0) return 0.0; /* (reduce f '(0)) = 0 */
  float acc = 0.0f;
  for (long m = 1; m <= n; ++m) acc = acc + (float)log((double)m);
  return (double)acc;
}

/* M:382-383 (- (* k (log lambda)) lambda (log-factorial k)) */
double orc_log_poisson(double lambda, double k, int in_double) {
  return ((k * log(lambda)) - lambda) - orc_log_factorial(k, in_double);
}

/* M:358-360 one <key>-bound of prior-bounds-let: strict < at both ends, exp(x)-1 literal */
double orc_bound_penalty(double p, double lo, double hi) {
  if (lo < p && p < hi) return 0.0;
  double a = fabs(p - hi), b = fabs(p - lo);
  double m = a < b ? a : b; /* (min a b) */
  return -1e10 * (exp(m * 1e-5) - 1.0);
}

/* ======================================================================== */
/* models: the device-model formulas of include/mhx.h, literal, no fusing     */
/* ======================================================================== */
static double horner(const double* c, int n, double x) {
  if (n <= 0) return 0.0;
  double acc = c[n - 1];
  for (int j = n - 2; j >= 0; --j) acc = acc * x + c[j];
  return acc;
}

double orc_model_eval(int model, const int32_t* shape, const double* p, int np, double x) {
  switch (model) {
    case MHX_MODEL_POLY:
      return horner(p, np, x);
    case MHX_MODEL_GAUSS_PEAKS:
    case MHX_MODEL_LORENTZ_PEAKS: {
      int nbg = shape[0], npk = shape[1];
      double f = horner(p, nbg, x);
      for (int k = 0; k < npk; ++k) {
        const double* q = p + nbg + 3 * k;
        double t = (x - q[1]) / q[2];
        if (model == MHX_MODEL_GAUSS_PEAKS)
          f = f + q[0] * exp(-(t * t));
        else
          f = f + q[0] / (1.0 + t * t);
      }
      return f;
    }
    case MHX_MODEL_LORDER_MIXED: {
      double u = (x - p[2]) / p[1];
      double q = 1.0 + u * u;
      double num = cos(p[3]) * (-2.0 * u) + sin(p[3]) * (1.0 - u * u);
      return ((p[0] * num) / (q * q) + p[4]) + p[5] * x;
    }
    case MHX_MODEL_EXP_DECAY:
      return p[0] * exp(-(x / p[1])) + p[2];
    case MHX_MODEL_SINUSOID:
      return p[0] * sin(p[1] * x + p[2]) + p[3];
    case MHX_MODEL_PVOIGT2: {
      double u1 = (x - p[3]) / p[4], u2 = (x - p[6]) / p[7];
      double s1 = u1 * u1, s2 = u2 * u2;
      double pv1 = p[5] / (1.0 + s1) + (1.0 - p[5]) * exp(-s1);
      double pv2 = p[8] / (1.0 + s2) + (1.0 - p[8]) * exp(-s2);
      return ((p[1] + p[2] * x) + p[10] * (x * x)) + p[0] * (pv1 + p[9] * pv2);
    }
    default:
      return NAN;
  }
}

/* ======================================================================== */
/* proposal linear algebra                                                   */
/* ======================================================================== */

/* Float-trap bookkeeping: SBCL runs with :overflow :invalid :divide-by-zero enabled.
 * An operation on finite inputs that yields inf is an overflow, one that yields NaN an
 * invalid operation.  The handler-case of M:891-894 catches overflow / div-by-zero /
 * type-error, NOT invalid. */
static int trap_of(double r) {
  /* with finite inputs an inf is an overflow and a NaN can only follow an earlier overflow
   * (inf - inf): either way floating-point-overflow is signalled first and is caught.  The
   * only invalid operation on this path is the explicit 0/0 of M:597. */
  return isfinite(r) ? ORC_L_OK : ORC_L_CAUGHT;
}

/* M:614-643 population covariance; avg via (reduce #'+ x) then / n; the /n sits inside
 * the accumulation of each entry (M:643). */
int orc_lplist_covariance(const double* v, int m, int d, double* cov) {
  if (m <= 0) return ORC_L_EMPTY;
  double* an = (double*)malloc(sizeof(double) * (size_t)m * (size_t)d);
  int st = ORC_L_OK;
  for (int i = 0; i < d && st == ORC_L_OK; ++i) {
    double s = v[i];
    for (int k = 1; k < m; ++k) s = s + v[(size_t)k * d + i];
    double avg = s / (double)m;
    if ((st = trap_of(avg)) != ORC_L_OK) break;
    for (int k = 0; k < m; ++k) an[(size_t)i * m + k] = v[(size_t)k * d + i] - avg;
  }
  for (int i = 0; i < d && st == ORC_L_OK; ++i)
    for (int j = 0; j < d && st == ORC_L_OK; ++j) {
      double mini = 0.0;
      for (int k = 0; k < m; ++k) {
        mini = mini + (an[(size_t)i * m + k] * an[(size_t)j * m + k]) / (double)m;
      }
      st = trap_of(mini);
      cov[i * d + j] = mini;
    }
  free(an);
  return st;
}

/* M:583-598 Cholesky-Banachiewicz, diagonal sqrt(max 0 .), upper triangle left 0 */
int orc_cholesky(const double* cov, int d, double* L) {
  for (int i = 0; i < d * d; ++i) L[i] = 0.0;
  for (int i = 0; i < d; ++i)
    for (int k = 0; k <= i; ++k) {
      double tmp = 0.0;
      for (int j = 0; j < k; ++j) tmp = tmp + L[i * d + j] * L[k * d + j];
      int st = trap_of(tmp);
      if (st != ORC_L_OK) return st;
      if (i == k) {
        double a = cov[i * d + k] - tmp;
        L[i * d + k] = sqrt(a > 0.0 ? a : 0.0); /* (max 0d0 a) */
      } else {
        double num = cov[i * d + k] - tmp, den = L[k * d + k];
        if (den == 0.0) return num == 0.0 ? ORC_L_INVALID : ORC_L_CAUGHT;
        double q = num / den;
        if ((st = trap_of(q)) != ORC_L_OK) return st;
        L[i * d + k] = q;
      }
    }
  return ORC_L_OK;
}

/* M:679-700 L.z + theta: full d x d product, mini-sum from 0d0, multiply then add */
void orc_covariant_sample(const double* theta, const double* L, const double* z, int d,
                          double* out) {
  for (int i = 0; i < d; ++i) {
    double mini = 0.0;
    for (int j = 0; j < d; ++j) mini = mini + L[i * d + j] * z[j];
    out[i] = mini;
  }
  for (int i = 0; i < d; ++i) out[i] = out[i] + theta[i];
}

/* ======================================================================== */
/* shared random-stream specification (parity unpinned by the reference)      */
/* ======================================================================== */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static uint64_t to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* log(x) for positive normal x, the fdlibm e_log.c recipe on its general path:
 * x = 2^k (1+f), s = f/(2+f), log(1+f) = f - hfsq + s (hfsq + R(s^2)). */
double orc_det_log(double x) {
  static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                      Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                      Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                      Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                      Lg7 = 1.479819860511658591e-01;
  uint64_t b = to_bits(x);
  int32_t hx = (int32_t)(b >> 32);
  uint32_t lx = (uint32_t)b;
  int32_t k = (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int32_t i = (hx + 0x95f64) & 0x100000;
  uint64_t nb = ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | lx;
  k += (i >> 20);
  double f = from_bits(nb) - 1.0;
  double s = f / (2.0 + f);
  double dk = (double)k;
  double z = s * s;
  double w = z * z;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  double R = t2 + t1;
  double hfsq = (0.5 * f) * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

static double ksin(double x) {
  static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                      S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                      S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x + v * (S1 + z * r);
}
static double kcos(double x) {
  static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                      C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                      C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  return 1.0 - (0.5 * z - z * r);
}

/* cos(2 pi t), t in [0,1): quadrant q = trunc(4t + 1/2), remainder exact, then the
 * fdlibm kernels on y = r * pi/2, |y| <= pi/4. */
double orc_det_cos2pi(double t) {
  double t4 = 4.0 * t;
  int q = (int)(t4 + 0.5);
  double r = t4 - (double)q;
  double y = r * 1.57079632679489661923;
  switch (q & 3) {
    case 0: return kcos(y);
    case 1: return -ksin(y);
    case 2: return -kcos(y);
    default: return ksin(y);
  }
}

static void rng_block(uint64_t seed, uint64_t chain, uint64_t draw, uint32_t slot,
                      uint32_t r[4]) {
  uint32_t ctr[4] = {(uint32_t)chain, slot, (uint32_t)draw, (uint32_t)(draw >> 32)};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  orc_philox4x32_10(ctr, key, r);
}
static uint64_t bits53(uint32_t a, uint32_t b) {
  return ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6);
}

/* standard normal number `slot` of proposal `draw` of global chain `chain`:
 * Box-Muller cosine branch, u1 in (0,1], u2 in [0,1). */
double orc_rng_normal(uint64_t seed, uint64_t chain, uint64_t draw, uint32_t slot) {
  uint32_t r[4];
  rng_block(seed, chain, draw, slot, r);
  double u1 = (double)(bits53(r[0], r[1]) + 1) * 0x1p-53;
  double u2 = (double)bits53(r[2], r[3]) * 0x1p-53;
  double rad = sqrt(-2.0 * orc_det_log(u1));
  return rad * orc_det_cos2pi(u2);
}

/* the (random 1.0d0) of M:1092, in (0,1] so (log u) never traps */
double orc_rng_uniform(uint64_t seed, uint64_t chain, uint64_t draw) {
  uint32_t r[4];
  rng_block(seed, chain, draw, 0xFFFFFFFFu, r);
  return (double)(bits53(r[0], r[1]) + 1) * 0x1p-53;
}

/* ======================================================================== */
/* temperature schedule M:873-878                                            */
/* ======================================================================== */
static int64_t steps_to_settle_of(int d) { return 10 * (int64_t)(d > 50 ? d : 50); }

int64_t orc_temperature_schedule(int64_t n, int d, double temperature, double* out,
                                 int64_t cap) {
  int64_t sts = steps_to_settle_of(d);
  int64_t ts = n > 10 * sts ? n : 10 * sts;
  /* (* x pi (+ 1 (* 2 (floor temp-steps 5000))) (/ (* 2 temp-steps))) left to right; the
   * last factor is a rational converted to double when it meets the double product */
  double kfac = (double)(1 + 2 * (ts / 5000));
  double inv = 1.0 / (double)(2 * ts);
  for (int64_t x = 0; x < ts && x < cap; ++x) {
    double arg = (((double)x * M_PI) * kfac) * inv;
    double v = cos(arg) * temperature;
    out[x] = v > 1.0 ? v : 1.0; /* (max 1 v) */
  }
  return ts;
}

/* ======================================================================== */
/* problem                                                                    */
/* ======================================================================== */
typedef struct {
  int model, n_shape, n_idx;
  int32_t shape[4];
  int32_t idx[MHX_MAX_FN_PARAMS];
  int lik;
  size_t n;
  double *x, *y, *sigma;
  int n_bounds;
  int32_t bidx[MHX_MAX_BOUNDS];
  double blo[MHX_MAX_BOUNDS], bhi[MHX_MAX_BOUNDS];
} orc_fn;

struct orc_problem {
  int d, K;
  int logfact_double;
  orc_fn* fn;
};

orc_problem* orc_problem_create(int d, int K) {
  if (d < 1 || d > MHX_MAX_PARAMS || K < 1 || K > MHX_MAX_FUNCTIONS) return NULL;
  orc_problem* p = (orc_problem*)calloc(1, sizeof(*p));
  p->d = d;
  p->K = K;
  p->fn = (orc_fn*)calloc((size_t)K, sizeof(orc_fn));
  return p;
}
void orc_problem_destroy(orc_problem* p) {
  if (!p) return;
  for (int k = 0; k < p->K; ++k) {
    free(p->fn[k].x);
    free(p->fn[k].y);
    free(p->fn[k].sigma);
  }
  free(p->fn);
  free(p);
}
int orc_problem_set_function(orc_problem* p, int k, int model, const int32_t* shape,
                             int n_shape, const int32_t* idx, int n_idx) {
  if (k < 0 || k >= p->K || n_idx < 0 || n_idx > MHX_MAX_FN_PARAMS || n_shape > 4) return -1;
  orc_fn* f = &p->fn[k];
  f->model = model;
  f->n_shape = n_shape;
  f->n_idx = n_idx;
  for (int i = 0; i < 4; ++i) f->shape[i] = i < n_shape ? shape[i] : 0;
  for (int i = 0; i < n_idx; ++i) {
    if (idx[i] < 0 || idx[i] >= p->d) return -1;
    f->idx[i] = idx[i];
  }
  return 0;
}
int orc_problem_set_dataset(orc_problem* p, int k, const double* x, const double* y,
                            const double* sigma, size_t n, int lik) {
  if (k < 0 || k >= p->K) return -1;
  orc_fn* f = &p->fn[k];
  free(f->x); free(f->y); free(f->sigma);
  f->n = n;
  f->lik = lik;
  f->x = (double*)malloc(sizeof(double) * (n ? n : 1));
  f->y = (double*)malloc(sizeof(double) * (n ? n : 1));
  f->sigma = (double*)malloc(sizeof(double) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) {
    f->x[i] = x[i];
    f->y[i] = y[i];
    f->sigma[i] = sigma ? sigma[i] : 1.0; /* (if data-error data-error 1), M:1144 */
  }
  return 0;
}
int orc_problem_set_bounds(orc_problem* p, int k, const int32_t* idx, const double* lo,
                           const double* hi, int n) {
  if (k < 0 || k >= p->K || n < 0 || n > MHX_MAX_BOUNDS) return -1;
  orc_fn* f = &p->fn[k];
  f->n_bounds = n;
  for (int i = 0; i < n; ++i) {
    f->bidx[i] = idx[i];
    f->blo[i] = lo[i];
    f->bhi[i] = hi[i];
  }
  return 0;
}
void orc_problem_set_logfact_double(orc_problem* p, int flag) { p->logfact_double = flag; }

/* one likelihood term of point i of function k */
static double lik_term(const orc_problem* p, const orc_fn* f, const double* local, size_t i) {
  double m = orc_model_eval(f->model, f->shape, local, f->n_idx, f->x[i]);
  switch (f->lik) {
    case MHX_LIK_NORMAL: /* M:400 */
      return orc_log_normal(f->y[i], m, f->sigma[i]);
    case MHX_LIK_NORMAL_CUTOFF: { /* M:426 (max -5000d0 term) */
      double t = orc_log_normal(f->y[i], m, f->sigma[i]);
      return t > -5000.0 ? t : -5000.0;
    }
    case MHX_LIK_POISSON: /* M:402-416 with (lambda (y model err) (log-poisson model y)) */
      return orc_log_poisson(m, f->y[i], p->logfact_double);
    default:
      return NAN;
  }
}

static void gather(const orc_fn* f, const double* theta, double* local) {
  for (int j = 0; j < f->n_idx; ++j) local[j] = theta[f->idx[j]];
}

/* (reduce #'+ (mapcar ...)): t0, then ((t0+t1)+t2)...; empty list -> 0 */
static double loglik_fn(const orc_problem* p, const orc_fn* f, const double* theta) {
  double local[MHX_MAX_FN_PARAMS];
  gather(f, theta, local);
  if (f->n == 0) return 0.0;
  double acc = lik_term(p, f, local, 0);
  for (size_t i = 1; i < f->n; ++i) acc = acc + lik_term(p, f, local, i);
  return acc;
}

/* prior-bounds-let body = bounds-total, M:366-368: (+ b1 b2 ...) */
static double logprior_fn(const orc_problem* p, const orc_fn* f, const double* theta) {
  (void)p;
  if (f->n_bounds == 0) return 0.0; /* log-prior-flat M:340-343 */
  double acc = 0.0;
  for (int i = 0; i < f->n_bounds; ++i) {
    double v = f->bidx[i] >= 0 ? theta[f->bidx[i]] : 0.0; /* (getf params key 0d0) */
    double b = orc_bound_penalty(v, f->blo[i], f->bhi[i]);
    acc = i == 0 ? b : acc + b;
  }
  return acc;
}

/* M:1068-1069 (+ (reduce #'+ lls) (reduce #'+ lps)) */
double orc_logpost(const orc_problem* p, const double* theta, double* parts) {
  double ll = loglik_fn(p, &p->fn[0], theta);
  for (int k = 1; k < p->K; ++k) ll = ll + loglik_fn(p, &p->fn[k], theta);
  double lp = logprior_fn(p, &p->fn[0], theta);
  for (int k = 1; k < p->K; ++k) lp = lp + logprior_fn(p, &p->fn[k], theta);
  if (parts) {
    parts[0] = ll;
    parts[1] = lp;
  }
  return ll + lp;
}

/* ======================================================================== */
/* MIRROR mode: the same posterior in the GPU kernel's own arithmetic          */
/* ======================================================================== */
/* The faithful functions above follow the reference (serial sums, libm, divisions).  The
 * mirror follows lisp-mcmc_amd/csrc for problems made of GAUSS_PEAKS functions with the
 * normal likelihood (BASELINE config 2's kernel): 1/sigma and y/sigma formed once, exp as
 * 2^(-t^2) with the exponent built inside two fmas, a 256-entry table and a cubic, lane-strided
 * accumulation (lane l of 64 takes points l, l+64, ...; two accumulators alternate) and the
 * xor butterfly.  fma() here is the C99 correctly rounded one, so the result is BIT-IDENTICAL
 * to the device's.  It exists to turn "within tolerance" into "equal" in the tests; the
 * tolerance between mirror and faithful is checked on the CPU. */
/* csrc/mhx_device.hpp: mexp2_negsq.  s = -t^2 = k + j/256 + r from two fmas (the low dword of
 * kd = fma(-t, t, 1.5 2^44) is 256 k + j), {Th_j, rho_j} from the table the device stages in LDS
 * (the SAME generated file, tools/gen_exp2_table.py), 2^r - 1 = r q(r). */
static const double mir_exp2_tab[256][2] = {
#include "../lisp-mcmc_amd/csrc/mhx_exp2_table.inc"
};
#define MIR_FAST_T 2890.0 /* kFastT */
static double mir_exp2_negsq(double t) {
  const double MAGIC = 0x1.8p44;
  const double q3 = 0x1.3b2ab83eadfb0p-7, q2 = 0x1.c6b0902b5a0abp-5, q1 = 0x1.ebfbdff82c585p-3,
               q0 = 0x1.62e42fefa39d9p-1;
  double kd = fma(-t, t, MAGIC);
  double kf = kd - MAGIC;
  double r = fma(-t, t, -kf);
  int32_t lo = (int32_t)(uint32_t)to_bits(kd);
  const double th = mir_exp2_tab[lo & 255][0], rho = mir_exp2_tab[lo & 255][1];
  double a = fma(r, q3, q2);
  a = fma(r, a, q1);
  a = fma(r, a, q0);
  double ee = fma(r, a, rho);
  return ldexp(fma(th, ee, th), (int)(lo >> 8)); /* arithmetic shift, as v_ashrrev_i32 */
}

/* csrc/mhx_device.hpp: mexp2_head + mexp2_negsq_tail = 2^s for a plain exponent s (the seeds
 * r_0 of the uniform-grid recurrence) */
static double mir_exp2_plain(double sx) {
  const double MAGIC = 0x1.8p44;
  const double q3 = 0x1.3b2ab83eadfb0p-7, q2 = 0x1.c6b0902b5a0abp-5, q1 = 0x1.ebfbdff82c585p-3,
               q0 = 0x1.62e42fefa39d9p-1;
  double kd = sx + MAGIC;
  double kf = kd - MAGIC;
  double r = sx - kf;
  int32_t lo = (int32_t)(uint32_t)to_bits(kd);
  const double th = mir_exp2_tab[lo & 255][0], rho = mir_exp2_tab[lo & 255][1];
  double a = fma(r, q3, q2);
  a = fma(r, a, q1);
  a = fma(r, a, q0);
  double ee = fma(r, a, rho);
  return ldexp(fma(th, ee, th), (int)(lo >> 8));
}

/* csrc/mhx_device.hpp: mexp2 (degree-11 polynomial form; the recurrence's q = 2^(-2 D^2)) */
static double mir_mexp2(double s) {
  const double MAGIC = 0x1.8p52;
  const double kd = s + MAGIC;
  const double kf = kd - MAGIC;
  const double f = s - kf;
  double p = 0x1.e9d3fe3952179p-32;
  p = fma(p, f, 0x1.e6063f7217bc6p-28);
  p = fma(p, f, 0x1.b524fae627834p-24);
  p = fma(p, f, 0x1.62bfd47773353p-20);
  p = fma(p, f, 0x1.ffcbfc670dcd4p-17);
  p = fma(p, f, 0x1.430913096fd9fp-13);
  p = fma(p, f, 0x1.5d87fe78a5276p-10);
  p = fma(p, f, 0x1.3b2ab6fba1ddap-7);
  p = fma(p, f, 0x1.c6b08d704a0c2p-5);
  p = fma(p, f, 0x1.ebfbdff82c598p-3);
  p = fma(p, f, 0x1.62e42fefa39efp-1);
  p = fma(p, f, 1.0);
  if (kf != kf) return kf;
  return ldexp(p, (int)kf);
}

/* csrc/mhx_device.hpp: tlog_rate, the table-driven log of the Poisson term (Tang's method, the
 * SAME generated table; the device forms z with one ldexp and -k with one subtraction - the same
 * numbers as below).  Everything that is not a positive normal number is a NaN there and here. */
static const double mir_log_tab[128][2] = {
#include "../lisp-mcmc_amd/csrc/mhx_log_table.inc"
};
static double mir_tlog(double x, int* plain) {
  const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
  const double A1 = 0x1.5555555555555p-2, A3 = 0x1.999999999999ap-3, A4 = -0x1.5555555555555p-3;
  const uint64_t b = to_bits(x);
  const uint32_t hx = (uint32_t)(b >> 32), lx = (uint32_t)b;
  const uint32_t th = hx - 0x3FE60000u;
  const int i = (int)((th >> 13) & 127u);
  const int k = (int32_t)th >> 20;
  const uint32_t zh = hx - (th & 0xFFF00000u);
  const double z = from_bits(((uint64_t)zh << 32) | lx);
  const double invc = mir_log_tab[i][0], logc = mir_log_tab[i][1];
  const double r = fma(z, invc, -1.0);
  const double kd = (double)k;
  const double w = fma(kd, Ln2hi, logc);
  const double r2 = r * r;
  const double p1 = fma(r, A4, A3);
  const double p3 = fma(r, -0.25, fma(r2, p1, A1));
  const double a = fma(kd, Ln2lo, r);
  const double q = fma(r, p3, -0.5);
  const double res = w + fma(r2, q, a);
  /* tlog_rate(): positive and normal, else NaN (v_cmp_class_f64 on the device) */
  const int ordinary = (uint32_t)(hx - 0x00100000u) < (uint32_t)(0x7ff00000u - 0x00100000u);
  (void)plain;
  return ordinary ? res : NAN;
}

/* csrc/mhx_engine.cpp, mhx_set_dataset: x is a uniform grid x_0 + i h to 8 ulp of max |x| ->
 * 64 h (the step between two successive points of one lane), else 0 */
static int mir_no_recurrence = 0; /* orc_mirror_set_recurrence(0): MHX_NO_RECURRENCE=1's twin */
void orc_mirror_set_recurrence(int on) { mir_no_recurrence = !on; }
/* PeaksModel::kSeedSteps: the recurrence is re-seeded every 32 points of a lane (at the start
 * of every 2048-point window of the dataset, in either kernel family) */
#define MIR_SEED_STEPS 32
static double mir_grid_H(const orc_fn* f) {
  const size_t n = f->n;
  if (mir_no_recurrence || n < 2 || !isfinite(f->x[0]) || !isfinite(f->x[n - 1])) return 0.0;
  const double h = (f->x[n - 1] - f->x[0]) / (double)(n - 1);
  const double a0 = fabs(f->x[0]), a1 = fabs(f->x[n - 1]);
  const double tol = 8.0 * 0x1p-52 * (a0 > a1 ? a0 : a1);
  if (h == 0.0 || !isfinite(h)) return 0.0;
  for (size_t i = 0; i < n; ++i)
    if (!(fabs(f->x[i] - (f->x[0] + (double)i * h)) <= tol)) return 0.0;
  return 64.0 * h;
}

/* csrc/mhx_engine.cpp, finalize_problem ("Per-window grids"): on a dataset that is not ONE grid,
 * window w (2048 points) gets H_w = 64 h when its data points are x_first + i h to 8 ulp of the
 * window's max |x| - tried with the previous grid window's h first, then with its own
 * (x_last - x_first) / (points - 1) - and 0 otherwise.  *h_prev carries the last accepted h. */
static int mir_no_window_grids = 0; /* orc_mirror_set_window_grids(0): MHX_NO_WINDOW_GRIDS=1's twin */
void orc_mirror_set_window_grids(int on) { mir_no_window_grids = !on; }
static int mir_window_fits(const double* xw, size_t cnt, double h, double tol) {
  if (h == 0.0 || !isfinite(h)) return 0;
  for (size_t i = 0; i < cnt; ++i)
    if (!(fabs(xw[i] - (xw[0] + (double)i * h)) <= tol)) return 0;
  return 1;
}
static double mir_window_H(const orc_fn* f, size_t base, size_t win, double* h_prev) {
  if (mir_no_recurrence || mir_no_window_grids || f->n < 2) return 0.0;
  const size_t cnt = f->n - base < win ? f->n - base : win;
  if (cnt < 2) return 0.0;
  const double* xw = f->x + base;
  if (!isfinite(xw[0]) || !isfinite(xw[cnt - 1])) return 0.0;
  const double a0 = fabs(xw[0]), a1 = fabs(xw[cnt - 1]);
  const double tol = 8.0 * 0x1p-52 * (a0 > a1 ? a0 : a1);
  double h = *h_prev;
  if (!mir_window_fits(xw, cnt, h, tol)) h = (xw[cnt - 1] - xw[0]) / (double)(cnt - 1);
  if (!mir_window_fits(xw, cnt, h, tol)) return 0.0;
  *h_prev = h;
  return 64.0 * h;
}

/* csrc/mhx_device.hpp: mexp2_negsq_safe = mexp2(max(-(t t), -1100)), the guarded form a chain
 * uses for a function whose parameters put |t| beyond kFastT somewhere in the data range */
static double mir_exp2_negsq_safe(double t) {
  const double MAGIC = 0x1.8p52;
  double s = -(t * t);
  s = s < -1100.0 ? -1100.0 : s;
  const double kd = s + MAGIC;
  const double kf = kd - MAGIC;
  const double f = s - kf;
  double p = 0x1.e9d3fe3952179p-32;
  p = fma(p, f, 0x1.e6063f7217bc6p-28);
  p = fma(p, f, 0x1.b524fae627834p-24);
  p = fma(p, f, 0x1.62bfd47773353p-20);
  p = fma(p, f, 0x1.ffcbfc670dcd4p-17);
  p = fma(p, f, 0x1.430913096fd9fp-13);
  p = fma(p, f, 0x1.5d87fe78a5276p-10);
  p = fma(p, f, 0x1.3b2ab6fba1ddap-7);
  p = fma(p, f, 0x1.c6b08d704a0c2p-5);
  p = fma(p, f, 0x1.ebfbdff82c598p-3);
  p = fma(p, f, 0x1.62e42fefa39efp-1);
  p = fma(p, f, 1.0);
  if (kf != kf) return kf; /* NaN in, NaN out (the device's v_cvt_i32_f64 of a NaN is 0) */
  return ldexp(p, (int)kf);
}

static double mir_dexp(double x) {
  /* csrc/mhx_device.hpp, dexp: beyond the range of a double the answer is decided up front */
  if (!(x < 0x1.62e42fefa39efp+9)) return x != x ? x : INFINITY;
  if (x < -745.2) return 0.0;
  const double MAGIC = 0x1.8p52;
  double kd = fma(x, 1.4426950408889634074, MAGIC);
  double kf = kd - MAGIC;
  double r = fma(-kf, 6.93147180369123816490e-01, x);
  r = fma(-kf, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)kf);
}

static double mir_bound_penalty(double p, double lo, double hi) {
  if (lo < p && p < hi) return 0.0;
  double a = fabs(p - hi), b = fabs(p - lo);
  double m = a < b ? a : b;
  return -1e10 * (mir_dexp(m * 1e-5) - 1.0);
}

#define MIR_LANES 64

/* returns NaN with *supported = 0 when the problem is outside the mirrored kernel */
static double mir_loglik_fn(const orc_fn* f, const double* theta, int* supported,
                            int logfact_double) {
  double local[MHX_MAX_FN_PARAMS];
  gather(f, theta, local);
  if (f->model != MHX_MODEL_GAUSS_PEAKS ||
      (f->lik != MHX_LIK_NORMAL && f->lik != MHX_LIK_POISSON && f->lik != MHX_LIK_NORMAL_CUTOFF)) {
    *supported = 0;
    return NAN;
  }
  const int poisson = f->lik == MHX_LIK_POISSON;
  const int cutoff = f->lik == MHX_LIK_NORMAL_CUTOFF; /* log-liklihood-normal-cutoff M:419-427 */
  const int nbg = f->shape[0], npk = f->shape[1];
  const double ksl2e = 1.2011224087864497594; /* sqrt(log2 e) */
  double iw[MHX_MAX_FN_PARAMS], cc[MHX_MAX_FN_PARAMS], A[MHX_MAX_FN_PARAMS];
  double xmin = f->n ? f->x[0] : 0.0, xmax = xmin;
  for (size_t i = 1; i < f->n; ++i) {
    if (f->x[i] < xmin) xmin = f->x[i];
    if (f->x[i] > xmax) xmax = f->x[i];
  }
  int fast = 1;
  /* PeaksModel::kHasSkip: such models choose the table exp or the guarded form per 2048-point
   * WINDOW (tile_mask): a peak with |t| >= 40 over a whole window is exactly zero there and is
   * left out (all amplitudes finite), and the window is fast when every other peak has
   * |t| < kFastT at both of its ends; the others decide once for the whole dataset */
  const int has_skip = nbg >= 1 && nbg <= 2 && npk <= 30;
  int afin = 1;
  for (int k = 0; k < npk; ++k) afin = afin && isfinite(local[nbg + 3 * k]);
  for (int k = 0; k < npk; ++k) {
    A[k] = local[nbg + 3 * k];
    iw[k] = ksl2e / local[nbg + 3 * k + 2];
    cc[k] = -local[nbg + 3 * k + 1] * iw[k];
    /* the kernel's fast path needs |t| < kFastT over the data range for EVERY peak of the
     * function (PeaksModel::prepare); otherwise the chain evaluates all of them in the guarded
     * form (the choice is the chain's own: csrc/mhx_kernels.hpp, sweep) */
    double ta = fabs(fma(xmin, iw[k], cc[k])), tb = fabs(fma(xmax, iw[k], cc[k]));
    if (!(ta < MIR_FAST_T) || !(tb < MIR_FAST_T)) fast = 0;
  }
  /* the uniform-grid recurrence (PeaksModel, "Gaussians on a uniformly spaced x grid"): per
   * peak D = 64 h iw, r(t) = 2^(-2 t D - D^2), q = 2^(-2 D^2); a peak goes by it this step when
   * the function is on the fast path, the data on a grid and 32 |D| <= 1.  Tile-level skipping
   * is exact for these values as well, so - as for the direct form - it is not restated. */
  const double gH = mir_grid_H(f);
  double rm2d[MHX_MAX_FN_PARAMS], rnd2[MHX_MAX_FN_PARAMS], rq[MHX_MAX_FN_PARAMS];
  int rec[MHX_MAX_FN_PARAMS], s16[MHX_MAX_FN_PARAMS], s8[MHX_MAX_FN_PARAMS];
  int bgrec = 0;
  double bgH = 0.0;
  /* PeaksModel::prepare (H = the dataset's grid_H) and PeaksModel::regrid (H = the window's own,
   * on datasets that are not one grid: sweep(), "Per-window grids"): the same operations */
#define MIR_SET_GRID(H)                                                                         \
  do {                                                                                          \
    for (int k = 0; k < npk; ++k) {                                                             \
      const double dl = (H) * iw[k];                                                            \
      rm2d[k] = -2.0 * dl;                                                                      \
      rnd2[k] = -(dl * dl);                                                                     \
      rq[k] = mir_mexp2(2.0 * rnd2[k]);                                                         \
      /* three classes by width: S |D| <= 1 for the seeding period S = 32, 16 or 8 points of a  \
       * lane; a peak of a shorter class is re-seeded inside the window too */                  \
      const int base = (fast || has_skip) && (H) != 0.0;                                        \
      const int ok32 = base && (fabs(dl) * (double)MIR_SEED_STEPS <= 1.0);                      \
      const int multi = npk <= 2; /* PeaksModel::kMultiSeed: models of at most two peaks */     \
      const int ok16 = multi ? base && (fabs(dl) * (double)(MIR_SEED_STEPS / 2) <= 1.0) : ok32; \
      const int ok8 = multi ? base && (fabs(dl) * (double)(MIR_SEED_STEPS / 4) <= 1.0) : ok32;  \
      rec[k] = ok8;                                                                             \
      s16[k] = ok8 && !ok32;                                                                    \
      s8[k] = ok8 && !ok16;                                                                     \
    }                                                                                           \
    /* ... and when EVERY peak goes by the recurrence, a constant or linear background does too \
     * (Prep::bgrec): b(x + 64 h) = b(x) + 64 h b1, re-seeded with the peaks */                 \
    bgrec = nbg >= 1 && nbg <= 2 && npk <= 2; /* (at most two peaks) */                         \
    for (int k = 0; k < npk; ++k) bgrec = bgrec && rec[k];                                      \
    bgH = nbg == 2 ? local[1] * (H) : 0.0;                                                      \
  } while (0)
  MIR_SET_GRID(gH);
  /* per-window grids: only models that choose per window (kHasSkip) on datasets that are not one
   * grid; a window that is not a grid takes the direct form */
  const int wgrid = has_skip && gH == 0.0;
  double h_prev = 0.0, h_cur = 0.0;
  double acc0[MIR_LANES] = {0}, acc1[MIR_LANES] = {0};
  long double csum = 0.0L;
  const double half_log_2pi = -0.5 * log(2.0 * M_PI);
  if (!poisson) {
    for (size_t i = 0; i < f->n; ++i)
      csum += (long double)(half_log_2pi + (-1.0 * log(f->sigma[i])));
  } else {
    /* mhx_set_dataset: -sum_i log k_i!, each factorial the single-float running sum of M:379-380
     * (or lgamma in binary64), added up in a long double */
    for (size_t i = 0; i < f->n; ++i)
      csum -= (long double)orc_log_factorial(f->y[i], logfact_double);
  }
  int plain_log = 1;
  /* lane l of the chain's wave takes the points l, l + 64, ... ; 32 successive points of a lane
   * (2048 of the wave) are one seeding period of the recurrence */
  const size_t mir_tile = (size_t)MIR_LANES * MIR_SEED_STEPS;
  for (size_t base = 0; base < f->n; base += mir_tile) {
    int fastw = fast, far[MHX_MAX_FN_PARAMS] = {0};
    if (has_skip) {
      /* the window's x range as mhx_engine.cpp (finalize_problem) forms it: pads repeat the last
       * point; a non-finite x opens the range to (-inf, inf) */
      double xlo = INFINITY, xhi = -INFINITY;
      int okx = 1;
      for (size_t i = base; i < base + mir_tile; ++i) {
        const double xv = f->x[i < f->n ? i : f->n - 1];
        okx = okx && isfinite(xv);
        if (xv < xlo) xlo = xv;
        if (xv > xhi) xhi = xv;
      }
      if (!okx) {
        xlo = -INFINITY;
        xhi = INFINITY;
      }
      fastw = 1;
      for (int k = 0; k < npk; ++k) {
        const double tl = fma(xlo, iw[k], cc[k]), th = fma(xhi, iw[k], cc[k]);
        const int same_side = (tl > 0.0 && th > 0.0) || (tl < 0.0 && th < 0.0);
        const double al = fabs(tl), ah = fabs(th);
        const double tmin = al < ah ? al : ah;
        far[k] = afin && same_side && (tmin >= 40.0);
        const int in = (al < MIR_FAST_T) && (ah < MIR_FAST_T);
        if (!(far[k] || in)) fastw = 0;
      }
    }
    /* sweep(), "Per-window grids": the window's own H; the constants follow it where it changes,
     * and a window that is not a grid evaluates every peak directly (the constants of the last
     * grid stay in place for the next window on it) */
    int onw = 1;
    if (wgrid) {
      const double hw = mir_window_H(f, base, mir_tile, &h_prev);
      onw = hw != 0.0;
      if (onw && to_bits(hw) != to_bits(h_cur)) {
        MIR_SET_GRID(hw);
        h_cur = hw;
      }
    }
    const int bgrec_w = onw && bgrec && fastw;
    for (size_t lane = 0; lane < MIR_LANES; ++lane) {
      if (base + lane >= f->n) break;
      double g[MHX_MAX_FN_PARAMS], r[MHX_MAX_FN_PARAMS], bgv = 0.0;
      const double x0 = f->x[base + lane];
      if (bgrec_w) {
        bgv = local[nbg - 1];
        for (int j = nbg - 2; j >= 0; --j) bgv = fma(bgv, x0, local[j]);
      }
      for (int k = 0; k < npk; ++k)
        if (onw && rec[k] && fastw && !far[k]) {
          const double ts = fma(x0, iw[k], cc[k]);
          g[k] = mir_exp2_negsq(ts);
          r[k] = mir_exp2_plain(fma(rm2d[k], ts, rnd2[k]));
        }
      for (size_t kk = 0; kk < mir_tile / MIR_LANES; ++kk) {
        const size_t i = base + kk * MIR_LANES + lane;
        if (i >= f->n) break;
        const double s = f->sigma[i];
        const double w = 1.0 / s, yw = f->y[i] * w, x = f->x[i];
        /* sweep(): the narrower classes are re-seeded from the direct formulas, at the lane's
         * actual x, every 16 (every 8) points of the lane inside the window */
        if (kk > 0 && kk % (MIR_SEED_STEPS / 4) == 0)
          for (int k = 0; k < npk; ++k) {
            const int due = kk % (MIR_SEED_STEPS / 2) == 0 ? s16[k] : s8[k];
            if (due && onw && rec[k] && fastw && !far[k]) {
              const double ts = fma(x, iw[k], cc[k]);
              g[k] = mir_exp2_negsq(ts);
              r[k] = mir_exp2_plain(fma(rm2d[k], ts, rnd2[k]));
            }
          }
        double m = nbg > 0 ? local[nbg - 1] : 0.0;
        if (bgrec_w) {
          m = bgv;
          if (nbg > 1) bgv = bgv + bgH;
        } else {
          for (int j = nbg - 2; j >= 0; --j) m = fma(m, x, local[j]);
        }
        for (int k = 0; k < npk; ++k) {
          if (fastw && far[k]) continue; /* exactly zero over this window: left out */
          if (fastw && onw && rec[k]) {
            m = fma(A[k], g[k], m);
            g[k] = g[k] * r[k];
            r[k] = r[k] * rq[k];
          } else {
            const double t = fma(x, iw[k], cc[k]);
            m = fma(A[k], fastw ? mir_exp2_negsq(t) : mir_exp2_negsq_safe(t), m);
          }
        }
        if (poisson) { /* (- (* k (log lambda)) lambda ...) M:383; acc = acc + term */
          const double tt = fma(f->y[i], mir_tlog(m, &plain_log), -m);
          if (kk & 1)
            acc1[lane] = acc1[lane] + tt;
          else
            acc0[lane] = acc0[lane] + tt;
          continue;
        }
        const double rr = fma(-m, w, yw);
        if (cutoff) {
          /* sweep<>, MHX_LIK_NORMAL_CUTOFF: the point's own constant c_i = -1/2 log 2 pi - log
           * sigma_i (mhx_set_dataset) inside the term, (max -5000d0 term) M:426, plain adds */
          const double ci = half_log_2pi + (-1.0 * log(s));
          const double tt = fma(-0.5 * rr, rr, ci);
          const double tc = tt > -5000.0 ? tt : -5000.0;
          if (kk & 1)
            acc1[lane] = acc1[lane] + tc;
          else
            acc0[lane] = acc0[lane] + tc;
          continue;
        }
        if (kk & 1)
          acc1[lane] = fma(rr, rr, acc1[lane]);
        else
          acc0[lane] = fma(rr, rr, acc0[lane]);
      }
    }
  }
  double v[MIR_LANES], nv[MIR_LANES];
  for (int l = 0; l < MIR_LANES; ++l) v[l] = acc0[l] + acc1[l];
  for (int m = 32; m >= 1; m >>= 1) {
    for (int l = 0; l < MIR_LANES; ++l) nv[l] = v[l] + v[l ^ m];
    memcpy(v, nv, sizeof v);
  }
  if (poisson) {
    if (!plain_log) { /* some rate went through the device's mlog(): not restated */
      *supported = 0;
      return NAN;
    }
    return v[0] + (double)csum;
  }
  if (cutoff) return v[0]; /* finish_lik<MHX_LIK_NORMAL_CUTOFF>: the constants sit in the terms */
  return fma(-0.5, v[0], (double)csum);
}

static double mir_logprior_fn(const orc_fn* f, const double* theta) {
  if (f->n_bounds == 0) return 0.0;
  double acc = 0.0;
  for (int i = 0; i < f->n_bounds; ++i) {
    double v = f->bidx[i] >= 0 ? theta[f->bidx[i]] : 0.0;
    double b = mir_bound_penalty(v, f->blo[i], f->bhi[i]);
    acc = i == 0 ? b : acc + b;
  }
  return acc;
}

/* the kernel's value of walker-make-step's prob; NaN when the problem is not mirrored */
double orc_logpost_mirror(const orc_problem* p, const double* theta, double* parts) {
  int ok = 1;
  double ll = 0.0, lp = 0.0;
  for (int k = 0; k < p->K && ok; ++k) {
    double v = mir_loglik_fn(&p->fn[k], theta, &ok, p->logfact_double);
    ll = k == 0 ? v : ll + v;
    double q = mir_logprior_fn(&p->fn[k], theta);
    lp = k == 0 ? q : lp + q;
  }
  if (!ok) return NAN;
  if (parts) {
    parts[0] = ll;
    parts[1] = lp;
  }
  return ll + lp;
}

double orc_logpost_abs_terms(const orc_problem* p, const double* theta) {
  double s = 0.0, local[MHX_MAX_FN_PARAMS];
  for (int k = 0; k < p->K; ++k) {
    const orc_fn* f = &p->fn[k];
    gather(f, theta, local);
    for (size_t i = 0; i < f->n; ++i) s += fabs(lik_term(p, f, local, i));
  }
  return s;
}

/* ======================================================================== */
/* walker                                                                     */
/* ======================================================================== */
struct orc_walker {
  const orc_problem* p;
  int d;
  /* the walk, OLDEST first (the Lisp list is newest first); `length` newest entries are
   * what (walker-walk w) holds after :keep-walks (M:568-569) */
  double* prob;
  double* theta;
  size_t n_hist, cap;
  int64_t length, age;
  double best_prob;
  double best_theta[MHX_MAX_PARAMS];
  /* controller state, M:866-879 */
  int64_t n, i, sts, temp_steps, reset_index, mwl;
  int auto_mode, shutting_down, status, estop, has_mwl;
  int mirror; /* evaluate the posterior and log u in the kernel's arithmetic (mirror mode) */
  double temperature;
  double* temps;
  double L[MHX_MAX_PARAMS * MHX_MAX_PARAMS];
  uint64_t seed, chain_id, draw;
};

static void hist_push(orc_walker* w, double prob, const double* theta) {
  if (w->n_hist == w->cap) {
    w->cap = w->cap ? w->cap * 2 : 1024;
    w->prob = (double*)realloc(w->prob, sizeof(double) * w->cap);
    w->theta = (double*)realloc(w->theta, sizeof(double) * w->cap * (size_t)w->d);
  }
  w->prob[w->n_hist] = prob;
  memcpy(w->theta + w->n_hist * (size_t)w->d, theta, sizeof(double) * (size_t)w->d);
  w->n_hist++;
}

/* M:1132-1163: first step's prob, walk = (first-step), length 1, age 1 */
static double walker_logpost(const orc_walker* w, const double* theta) {
  return w->mirror ? orc_logpost_mirror(w->p, theta, NULL) : orc_logpost(w->p, theta, NULL);
}

orc_walker* orc_walker_create2(const orc_problem* p, const double* theta0, int mirror) {
  orc_walker* w = (orc_walker*)calloc(1, sizeof(*w));
  w->p = p;
  w->d = p->d;
  w->mirror = mirror;
  double pr = walker_logpost(w, theta0);
  hist_push(w, pr, theta0);
  w->length = 1;
  w->age = 1;
  w->best_prob = pr;
  memcpy(w->best_theta, theta0, sizeof(double) * (size_t)w->d);
  w->status = ORC_DONE;
  return w;
}
orc_walker* orc_walker_create(const orc_problem* p, const double* theta0) {
  return orc_walker_create2(p, theta0, 0);
}
void orc_walker_destroy(orc_walker* w) {
  if (!w) return;
  free(w->prob);
  free(w->theta);
  free(w->temps);
  free(w);
}

int64_t orc_walker_length(const orc_walker* w) { return w->length; }
int64_t orc_walker_age(const orc_walker* w) { return w->age; }
void orc_walker_last(const orc_walker* w, double* theta, double* prob) {
  size_t l = w->n_hist - 1;
  if (theta) memcpy(theta, w->theta + l * (size_t)w->d, sizeof(double) * (size_t)w->d);
  if (prob) *prob = w->prob[l];
}
void orc_walker_best(const orc_walker* w, double* theta, double* prob) {
  if (theta) memcpy(theta, w->best_theta, sizeof(double) * (size_t)w->d);
  if (prob) *prob = w->best_prob;
}

/* (walker-get :get :steps :take take): newest `min(length,take)` steps, M:490 */
static int window(const orc_walker* w, int take) {
  int64_t t = take > 0 && take < w->length ? take : w->length;
  return (int)t;
}
int orc_walker_trace(const orc_walker* w, int take, double* prob, double* theta) {
  int t = window(w, take);
  for (int s = 0; s < t; ++s) {
    size_t src = w->n_hist - 1 - (size_t)s;
    if (prob) prob[s] = w->prob[src];
    if (theta)
      memcpy(theta + (size_t)s * w->d, w->theta + src * (size_t)w->d,
             sizeof(double) * (size_t)w->d);
  }
  return t;
}

/* M:506-508 with remove-consecutive-duplicates (M:220-223, eql on doubles = same bits) */
void orc_walker_acceptance(const orc_walker* w, int take, int64_t* num, int64_t* den) {
  int t = window(w, take);
  int64_t runs = 0;
  for (int s = 0; s < t; ++s) {
    size_t a = w->n_hist - 1 - (size_t)s;
    if (s == t - 1 || to_bits(w->prob[a]) != to_bits(w->prob[a - 1])) runs++;
  }
  *num = runs;
  *den = t;
}

/* M:497-502: newest-first; keep step a when its prob is strictly above the next-older
 * step's; the oldest step of the window is never kept. Returns indices into history. */
static int forward_steps(const orc_walker* w, int take, size_t* out) {
  int t = window(w, take), n = 0;
  for (int s = 0; s + 1 < t; ++s) {
    size_t a = w->n_hist - 1 - (size_t)s;
    if (!(w->prob[a] <= w->prob[a - 1])) {
      if (out) out[n] = a;
      n++;
    }
  }
  return n;
}
int orc_walker_forward_count(const orc_walker* w, int take) {
  return forward_steps(w, take, NULL);
}

/* M:543 (lplist-to-l-matrix (diff-lplist forward-steps)); diff = older - newer (M:267-273) */
int orc_walker_l_matrix(const orc_walker* w, int take, double* L, int* n_forward) {
  int t = window(w, take), d = w->d;
  size_t* idx = (size_t*)malloc(sizeof(size_t) * (size_t)(t > 0 ? t : 1));
  int nf = forward_steps(w, take, idx);
  if (n_forward) *n_forward = nf;
  int st;
  if (nf == 0) {
    st = ORC_L_CAUGHT; /* (elt nil 0) -> index error, a type-error: "TE in Cholesky" */
  } else if (nf == 1) {
    st = ORC_L_EMPTY;
  } else {
    int m = nf - 1;
    double* v = (double*)malloc(sizeof(double) * (size_t)m * (size_t)d);
    for (int k = 0; k < m; ++k)
      for (int j = 0; j < d; ++j)
        v[(size_t)k * d + j] = w->theta[idx[k + 1] * (size_t)d + j] - w->theta[idx[k] * (size_t)d + j];
    double* cov = (double*)malloc(sizeof(double) * (size_t)d * (size_t)d);
    st = orc_lplist_covariance(v, m, d, cov);
    if (st == ORC_L_OK) st = orc_cholesky(cov, d, L);
    free(cov);
    free(v);
  }
  free(idx);
  return st;
}

/* M:549-555 */
static void add_step(orc_walker* w, double prob, const double* theta) {
  hist_push(w, prob, theta);
  w->length++;
  w->age++;
  if (prob > w->best_prob) {
    w->best_prob = prob;
    memcpy(w->best_theta, theta, sizeof(double) * (size_t)w->d);
  }
}

/* walker-modify M:566-578: 0 :burn-walks n, 1 :keep-walks n, 2 :reset, 3 :reset-to-most-likely.
 * Returns -1 where subseq would signal (bounding index outside the walk). */
int orc_walker_modify(orc_walker* w, int action, int64_t n) {
  size_t d = (size_t)w->d;
  if (action == 0) { /* (subseq walk 0 (- length burn-number)): the NEWEST length-n steps stay */
    if (n < 0 || n > w->length) return -1;
    w->length -= n;
  } else if (action == 1) { /* (subseq walk 0 keep-number) */
    if (n < 0 || n > w->length) return -1;
    w->length = n;
  } else if (action == 2 || action == 3) {
    double pr, th[MHX_MAX_PARAMS];
    if (action == 2) { /* (last (walker-walk w)): the OLDEST step of a newest-first list */
      size_t src = w->n_hist - (size_t)w->length;
      pr = w->prob[src];
      memcpy(th, w->theta + src * d, sizeof(double) * d);
    } else { /* (list (walker-most-likely-step w)) */
      pr = w->best_prob;
      memcpy(th, w->best_theta, sizeof(double) * d);
    }
    w->n_hist = 0;
    hist_push(w, pr, th); /* last-step <- (car walk) */
    w->length = 1;
  } else {
    return -1;
  }
  return 0;
}

/* M:1072-1095 */
int orc_walker_take_step_injected(orc_walker* w, const double* L, const double* z, double u,
                                  double T) {
  int d = w->d;
  size_t l = w->n_hist - 1;
  double prob0 = w->prob[l];
  double prev[MHX_MAX_PARAMS], next[MHX_MAX_PARAMS];
  memcpy(prev, w->theta + l * (size_t)d, sizeof(double) * (size_t)d);
  orc_covariant_sample(prev, L, z, d, next);
  double prob1 = walker_logpost(w, next);
  if (!isfinite(prob1)) return -1; /* a trap (or a complex/type error) in the reference */
  int acc = (prob1 > prob0) || ((prob1 - prob0) / T > (w->mirror ? orc_det_log(u) : log(u)));
  if (acc)
    add_step(w, prob1, next);
  else
    add_step(w, prob0, prev);
  return acc;
}

/* ---- comparisons of a rational acceptance with the single-float literals ---- */
static int acc_lt(int64_t num, int64_t den, float f) { return (double)num < (double)f * (double)den; }
static int acc_gt(int64_t num, int64_t den, float f) { return (double)num > (double)f * (double)den; }

static void diag_of(const double* v, int d, double* L) {
  for (int i = 0; i < d * d; ++i) L[i] = 0.0;
  for (int i = 0; i < d; ++i) L[i * d + i] = v[i]; /* M:710-727 */
}
static void scale_array(double s, double* L, int d) { /* M:605-611, in place */
  for (int i = 0; i < d * d; ++i) L[i] = s * L[i];
}

/* get-optimal-mcmc-l-matrix :covariance M:888-894.  Returns ORC_L_*; on OK/CAUGHT w->L is
 * the new factor (CAUGHT: the CURRENT L rescaled in place, as scale-array mutates);
 * EMPTY leaves L alone (dimension test of M:936 fails). */
static int get_optimal(orc_walker* w) {
  int d = w->d;
  double f = (2.38 * 2.38) / (double)d; /* (/ (expt 2.38d0 2) num-params) */
  double tmp[MHX_MAX_PARAMS * MHX_MAX_PARAMS];
  int st = orc_walker_l_matrix(w, (int)w->sts, tmp, NULL);
  if (st == ORC_L_OK) {
    scale_array(f, tmp, d);
    memcpy(w->L, tmp, sizeof(double) * (size_t)d * (size_t)d);
  } else if (st == ORC_L_CAUGHT) {
    scale_array(f, w->L, d);
  }
  return st;
}

/* stable-probs-p M:880-885 on the newest-first list of (take steps-to-settle) probs */
static int stable_probs(const orc_walker* w) {
  int t = window(w, (int)w->sts);
  if (t < 200) return 0; /* (subseq probs 0 200) would signal; unreachable at i > 1000 */
  double early = -INFINITY, late = -INFINITY, mn = INFINITY;
  for (int s = 0; s < t; ++s) {
    double v = w->prob[w->n_hist - 1 - (size_t)s];
    if (s < 200 && v > early) early = v;
    if (s >= t - 200 && v > late) late = v;
    if (v < mn) mn = v;
  }
  double spread = early - mn;
  return fabs(early - late) < 0.5 && 4.0 < spread && spread < 9.0;
}

int orc_walker_adaptive_begin(orc_walker* w, const orc_run_opts* o, uint64_t seed,
                              uint64_t chain_id) {
  int d = w->d;
  w->estop = 0; /* (setf mfit-walker-estop nil) M:865 */
  w->n = o->n;
  w->reset_index = 10000;
  w->has_mwl = o->max_walker_length > 0;
  w->mwl = o->max_walker_length / 2;
  w->sts = steps_to_settle_of(d);
  w->shutting_down = 0;
  w->temp_steps = w->n > 10 * w->sts ? w->n : 10 * w->sts;
  w->temperature = o->temperature;
  w->auto_mode = o->auto_mode;
  free(w->temps);
  w->temps = (double*)malloc(sizeof(double) * (size_t)w->temp_steps);
  orc_temperature_schedule(w->n, d, o->temperature, w->temps, w->temp_steps);
  w->seed = seed;
  w->chain_id = chain_id;
  w->status = ORC_RUNNING;
  if (o->l_matrix) {
    memcpy(w->L, o->l_matrix, sizeof(double) * (size_t)d * (size_t)d);
  } else { /* M:896-901 */
    int64_t num, den;
    orc_walker_acceptance(w, 100, &num, &den);
    diag_of(w->best_theta, d, w->L);
    if (!(w->length < w->sts || acc_lt(num, den, 0.1f))) {
      int st = get_optimal(w);
      if (st == ORC_L_INVALID || st == ORC_L_EMPTY) w->status = ORC_FP_TRAP;
    }
  }
  w->i = 1;
  return w->status;
}

static int64_t floor_mod(int64_t a, int64_t b) {
  int64_t r = a % b;
  return r < 0 ? r + b : r;
}

int orc_walker_adaptive_advance(orc_walker* w, int64_t max_iters) {
  int d = w->d;
  int64_t tail = w->sts > 2000 ? w->sts : 2000; /* (max 2000 steps-to-settle) */
  for (int64_t it = 0; it < max_iters && w->status == ORC_RUNNING; ++it) {
    if (w->i >= w->n) { w->status = ORC_DONE; break; }
    if (w->estop) { w->status = ORC_STOPPED; break; }
    /* M:905-917 */
    int shut = !w->shutting_down && (w->n - w->i) < tail;
    if (!shut && w->auto_mode && !w->shutting_down && floor_mod(w->i, 1000) == 0 &&
        w->i > 2 * w->sts) {
      int64_t num, den;
      orc_walker_acceptance(w, 1000, &num, &den);
      if (acc_gt(num, den, 0.2f) && acc_lt(num, den, 0.5f) && stable_probs(w)) shut = 1;
    }
    if (shut) {
      w->temperature = 1.0;
      w->shutting_down = 1;
      w->i = w->n - tail;
    }
    /* M:918 */
    double z[MHX_MAX_PARAMS];
    for (int j = 0; j < d; ++j) z[j] = orc_rng_normal(w->seed, w->chain_id, w->draw, (uint32_t)j);
    double u = orc_rng_uniform(w->seed, w->chain_id, w->draw);
    w->draw++;
    if (orc_walker_take_step_injected(w, w->L, z, u, w->temperature) < 0) {
      w->status = ORC_FP_TRAP;
      break;
    }
    /* M:920-921 */
    if (!w->shutting_down && w->i < w->temp_steps) w->temperature = w->temps[w->i];
    /* M:923-927 */
    if (w->has_mwl && w->i == w->reset_index) {
      if (w->length > w->mwl) {
        w->length = w->mwl; /* :keep-walks M:568-569 */
        w->reset_index += w->mwl + 1;
      } else {
        w->reset_index += 1 + (w->mwl - w->length);
      }
    }
    /* M:929-942 */
    if (w->i > 0) {
      int m200 = floor_mod(w->i, 200) == 0;
      int64_t num = 0, den = 1;
      if (m200) orc_walker_acceptance(w, 200, &num, &den);
      if ((m200 && acc_lt(num, den, 0.2f)) || (m200 && acc_gt(num, den, 0.4f)) ||
          (!w->shutting_down && floor_mod(w->i, 2 * w->sts) == 0)) {
        orc_walker_acceptance(w, 200, &num, &den);
        if (acc_gt(num, den, 0.2f) && acc_lt(num, den, 0.4f)) {
          if (get_optimal(w) == ORC_L_INVALID) {
            w->status = ORC_FP_TRAP;
            break;
          }
        } else if (acc_lt(num, den, 0.2f)) {
          scale_array(0.1, w->L, d);
        } else if (acc_gt(num, den, 0.4f)) {
          scale_array(1.9, w->L, d);
        }
      }
    }
    w->i++;
  }
  if (w->status == ORC_RUNNING && w->i >= w->n) w->status = ORC_DONE;
  return w->status;
}

int orc_walker_status(const orc_walker* w) { return w->status; }
int64_t orc_walker_loop_index(const orc_walker* w) { return w->i; }
double orc_walker_temperature(const orc_walker* w) { return w->temperature; }
void orc_walker_current_l(const orc_walker* w, double* L) {
  memcpy(L, w->L, sizeof(double) * (size_t)w->d * (size_t)w->d);
}
void orc_walker_request_stop(orc_walker* w) { w->estop = 1; }

/* M:849-853 with a caller L (the nil-L default of M:851 needs :median-params, out of path) */
int orc_walker_many_steps(orc_walker* w, int64_t n, const double* L, uint64_t seed,
                          uint64_t chain_id) {
  int d = w->d;
  w->seed = seed;
  w->chain_id = chain_id;
  for (int64_t it = 0; it < n; ++it) {
    double z[MHX_MAX_PARAMS];
    for (int j = 0; j < d; ++j) z[j] = orc_rng_normal(seed, chain_id, w->draw, (uint32_t)j);
    double u = orc_rng_uniform(seed, chain_id, w->draw);
    w->draw++;
    if (orc_walker_take_step_injected(w, L, z, u, 1.0) < 0) return ORC_FP_TRAP;
  }
  return ORC_DONE;
}
