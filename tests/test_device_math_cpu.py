"""CPU checks of device math that has no oracle twin: the table-driven logarithm of the Poisson
sweep (tlog, csrc/mhx_device.hpp).  The table is parsed out of the header, its two defining
properties are verified with mpmath, and a C restatement of the algorithm (same operations, C99
fma) is measured against the 80-bit logl."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "lisp-mcmc_amd", "csrc", "mhx_device.hpp")
EXP2_INC = os.path.join(ROOT, "lisp-mcmc_amd", "csrc", "mhx_exp2_table.inc")


def table_text():
    return open(os.path.join(ROOT, "lisp-mcmc_amd", "csrc", "mhx_log_table.inc")).read()


def test_log_table_properties():
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 200
    rows = re.findall(r"\{(-?0x[0-9a-f.]+p[+-]\d+), (-?0x[0-9a-f.]+p[+-]\d+)\}", table_text())
    assert len(rows) == 128
    G = mp.mpf(2) ** -43
    for i, (a, b) in enumerate(rows):
        invc, logc = float.fromhex(a), float.fromhex(b)
        assert mp.mpf(logc) / G == mp.nint(mp.mpf(logc) / G), i      # k ln2_hi + log c is exact
        assert abs(mp.log(1 / mp.mpf(invc)) - mp.mpf(logc)) < mp.mpf(2) ** -67, i
        # c_i sits inside its subinterval of [0.6875, 1.375)
        lo = 0.6875 + i * 2.0 ** -8 if i < 80 else 1.0 + (i - 80) * 2.0 ** -7
        hi = lo + (2.0 ** -8 if i < 80 else 2.0 ** -7)
        assert lo < 1.0 / invc < hi, i


C_SRC = r'''
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static const double T[128][2] = {
%s
};
static double tlog(double x) {
  const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
  const double A1 = 0x1.5555555555555p-2, A3 = 0x1.999999999999ap-3, A4 = -0x1.5555555555555p-3;
  uint64_t b; memcpy(&b, &x, 8);
  uint32_t hx = (uint32_t)(b >> 32), lx = (uint32_t)b;
  uint32_t th = hx - 0x3FE60000u;
  int i = (int)((th >> 13) & 127u);
  int k = (int32_t)th >> 20;
  uint32_t zh = hx - (th & 0xFFF00000u);
  uint64_t zb = ((uint64_t)zh << 32) | lx; double z; memcpy(&z, &zb, 8);
  double invc = T[i][0], logc = T[i][1];
  double r = fma(z, invc, -1.0), kd = (double)k;
  double w = fma(kd, Ln2hi, logc);
  double r2 = r * r;
  double p1 = fma(r, A4, A3);
  double p3 = fma(r, -0.25, fma(r2, p1, A1));
  double a = fma(kd, Ln2lo, r);
  double q = fma(r, p3, -0.5);
  return w + fma(r2, q, a);
}
int main(void) {
  double maxu = 0, maxa = 0; srand48(3);
  for (long n = 0; n < 4000000; ++n) {
    double x; int m = n %% 4;
    if (m == 0) x = exp((drand48() * 2 - 1) * 700);
    else if (m == 1) x = drand48() * 200 + 1e-3;
    else if (m == 2) x = 0.5 + drand48() * 1.5;
    else x = ldexp(0.6875 + drand48() * 0.6875, (int)(drand48() * 40) - 20);
    long double ref = logl((long double)x);
    if (fabs(x - 1.0) < 0.0625) {  /* tlog() sends these through mlog(); tlog_rate() does not: */
      double a = fabs((double)((long double)tlog(x) - ref));   /* absolute error where the */
      if (a > maxa) maxa = a;                                  /* table terms cancel */
      continue;
    }
    double u = fabs((double)((long double)tlog(x) - ref)) / ldexp(1.0, ilogb((double)ref) - 52);
    if (u > maxu) maxu = u;
  }
  for (long n = 0; n < 2000000; ++n) {   /* ... and densely around 1 */
    double x = 1.0 + (drand48() * 2 - 1) * 0.0625 * (n %% 2 ? 1.0 : 1e-6);
    double a = fabs((double)((long double)tlog(x) - logl((long double)x)));
    if (a > maxa) maxa = a;
  }
  printf("%%.4f %%.4g\n", maxu, maxa);
  return 0;
}
'''


def test_table_log_stays_below_one_ulp():
    d = tempfile.mkdtemp()
    c, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
    open(c, "w").write(C_SRC % table_text())
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, c, "-lm"])
    worst, near_one = (float(v) for v in subprocess.check_output([exe]).decode().split())
    assert worst < 0.75, worst
    # within 1/16 of 1 (tlog_rate's Poisson terms only; tlog sends user expressions' logs through
    # mlog there): the absolute error
    assert near_one < 2.0 ** -56, near_one   # (measured 2^-57: one ulp of the table term)


GEXP_SRC = r'''
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static const double T[256][2] = {
#include "%s"
};
/* gexp() of csrc/mhx_device.hpp (the exp of user expressions) with C99 fma */
static double gexp(double x) {
  const double MAGIC = 0x1.8p44, L2E_HI = 0x1.71547652b82fep+0, L2E_LO = 0x1.777d0ffda0d24p-56;
  const double q3 = %s, q2 = %s, q1 = %s, q0 = %s;
  double kd = fma(x, L2E_HI, MAGIC), kf = kd - MAGIC;
  double r = fma(x, L2E_LO, fma(x, L2E_HI, -kf));
  uint64_t b; memcpy(&b, &kd, 8);
  int32_t lo = (int32_t)(uint32_t)b;
  double a = fma(r, q3, q2); a = fma(r, a, q1); a = fma(r, a, q0);
  double ee = fma(r, a, T[lo & 255][1]);
  double v = ldexp(fma(T[lo & 255][0], ee, T[lo & 255][0]), lo >> 8);
  if (!(fabs(x) <= 1000.0)) v = fabs(x) < INFINITY ? (x > 0.0 ? INFINITY : 0.0) : NAN;
  return v;
}
int main(void) {
  double maxu = 0; srand48(1);
  for (long n = 0; n < 4000000; ++n) {
    double x = (drand48() * 2 - 1) * (n %% 3 == 0 ? 708.0 : (n %% 3 == 1 ? 30.0 : 1.0));
    long double r = expl((long double)x);
    double u = fabs((double)(((long double)gexp(x) - r) / r)) / 0x1p-52;
    if (u > maxu) maxu = u;
  }
  int ok = gexp(710.0) == INFINITY && gexp(-746.0) == 0.0 && gexp(0.0) == 1.0 && isnan(gexp(NAN))
           && gexp(1e300) == INFINITY && gexp(-1e300) == 0.0 && gexp(999.9) == INFINITY
           && gexp(-999.9) == 0.0 && isnan(gexp(INFINITY)) && isnan(gexp(-INFINITY))
           && gexp(-745.0) > 0.0 /* a subnormal, not 0 */;
  printf("%%.4f %%d\n", maxu, ok);
  return 0;
}
'''


def test_expression_exp_stays_below_one_ulp():
    src = open(HDR).read()
    body = src[src.index("double gexp(double x)"):]
    body = body[:body.index("return v;")]
    q = [re.search(r"\b%s = (0x[0-9a-f.]+p[+-]\d+)" % n, body).group(1) for n in ("q3", "q2", "q1", "q0")]
    # ... the same constants as the Gaussians' table exp (Exp2K::pin)
    pin = src[src.index("struct Exp2K"):src.index("struct Exp2Head")]
    for c in q:
        assert c in pin, c
    d = tempfile.mkdtemp()
    c, exe = os.path.join(d, "g.c"), os.path.join(d, "g")
    open(c, "w").write(GEXP_SRC % ((EXP2_INC,) + tuple(q)))
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, c, "-lm"])
    worst, ok = subprocess.check_output([exe]).decode().split()
    assert float(worst) < 1.0 and ok == "1", (worst, ok)


# ---- mexp2_negsq: the table-driven 2^(-t^2) of the Gaussian peaks --------------------------------


def test_exp2_table_properties():
    """Th_j = RN(2^(j/256)) and Th_j (1 + rho_j) = 2^(j/256) to far below an ulp; the constants in
    the header are the ones tools/gen_exp2_table.py prints for N = 256, degree 4."""
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 300
    rows = re.findall(r"\{(-?0x[0-9a-f.]+p[+-]\d+), (-?0x[0-9a-f.]+p[+-]\d+)\}", open(EXP2_INC).read())
    assert len(rows) == 256
    for j, (a, b) in enumerate(rows):
        th, rho = float.fromhex(a), float.fromhex(b)
        true = mp.power(2, mp.mpf(j) / 256)
        assert th == float(true), j                                   # correctly rounded
        assert abs(mp.mpf(th) * (1 + mp.mpf(rho)) / true - 1) < mp.mpf(2) ** -100, j
        assert abs(rho) <= 2.0 ** -53
    src = open(HDR).read()
    out = subprocess.check_output(["python3", os.path.join(ROOT, "tools", "gen_exp2_table.py"), "256", "4"]).decode()
    q = re.findall(r"q\d = (0x[0-9a-f.]+p[+-]\d+)", out)
    assert len(q) == 4
    for c in q:
        assert c in src, c
    err = float(re.search(r"2\^-9: ([0-9.e-]+)", out).group(1))
    assert err < 1e-17  # the cubic's own error: 0.05 ulp


EXP2_SRC = r'''
#include <math.h>
#include <quadmath.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static const double T[256][2] = {
#include "%s"
};
/* mexp2_negsq() of csrc/mhx_device.hpp with C99 fma */
static double mexp2_negsq(double t) {
  const double MAGIC = 0x1.8p44;
  const double q3 = 0x1.3b2ab83eadfb0p-7, q2 = 0x1.c6b0902b5a0abp-5, q1 = 0x1.ebfbdff82c585p-3,
               q0 = 0x1.62e42fefa39d9p-1;
  double kd = fma(-t, t, MAGIC), kf = kd - MAGIC, r = fma(-t, t, -kf);
  uint64_t b; memcpy(&b, &kd, 8);
  int32_t lo = (int32_t)(uint32_t)b;
  double a = fma(r, q3, q2); a = fma(r, a, q1); a = fma(r, a, q0);
  double ee = fma(r, a, T[lo & 255][1]);
  return ldexp(fma(T[lo & 255][0], ee, T[lo & 255][0]), lo >> 8);
}
int main(void) {
  double maxu = 0, maxr = 0; srand48(7);
  long bad = 0;
  for (long n = 0; n < 3000000; ++n) {
    double t; int m = n %% 4;
    if (m == 0) t = (drand48() * 2 - 1) * 6.0;          /* the body of a peak */
    else if (m == 1) t = (drand48() * 2 - 1) * 26.5;    /* down to 2^-700 */
    else if (m == 2) t = (drand48() * 2 - 1) * 0.1;     /* the top */
    else t = ldexp(drand48(), -(int)(drand48() * 60));  /* tiny arguments */
    __float128 s = -(__float128)t * (__float128)t;      /* exact: 106 bits fit 113 */
    __float128 ref = exp2q(s);
    double got = mexp2_negsq(t);
    double ulp = ldexp(1.0, ilogb((double)ref) - 52);
    double u = (double)fabsq((__float128)got - ref) / ulp;
    if (u > maxu) maxu = u;
    double kd = fma(-t, t, 0x1.8p44), r = fma(-t, t, -(kd - 0x1.8p44));
    if (fabs(r) > maxr) maxr = fabs(r);
  }
  /* edges: exact 1 at 0, underflow to 0 (gradually), NaN propagates, bound of the fast path */
  int ok = mexp2_negsq(0.0) == 1.0 && mexp2_negsq(40.0) == 0.0 && mexp2_negsq(-2889.9) == 0.0
           && isnan(mexp2_negsq(NAN)) && mexp2_negsq(32.5) > 0.0 && mexp2_negsq(32.5) < 0x1p-1022
           && maxr <= 0x1p-9;
  printf("%%.4f %%d\n", maxu, ok);
  return 0;
}
'''


def test_table_exp2_stays_near_half_an_ulp():
    d = tempfile.mkdtemp()
    c, exe = os.path.join(d, "e.c"), os.path.join(d, "e")
    open(c, "w").write(EXP2_SRC % EXP2_INC)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, c, "-lquadmath", "-lm"])
    worst, ok = subprocess.check_output([exe]).decode().split()
    # 0.5 ulp from the final fma's rounding + the cubic's 0.04 + the roundings inside e
    assert float(worst) < 0.56 and ok == "1", (worst, ok)
