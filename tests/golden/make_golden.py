#!/usr/bin/env python3
"""Generates tests/golden/reference_kats.json.

Every vector here comes from ONE of:
  (a) a known-answer comment or usage example in the reference's own source (cited
      file:line; the numbers are data, not code), or
  (b) a closed form of a reference formula evaluated independently of the oracle with
      numpy / scipy / mpmath (50 digits), or
  (c) the published Random123 known-answer vectors for Philox4x32-10.
The reference itself cannot be executed (Common Lisp; no Lisp implementation in the build
container), so no vector is an output of the reference run here.

Run:  python tests/golden/make_golden.py
"""
import json
import math
import os

import mpmath as mp
import numpy as np
from scipy import stats

mp.mp.dps = 50
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")


def log_normal_mp(x, mu, sigma):
    # M:372-377
    x, mu, sigma = mp.mpf(x), mp.mpf(mu), mp.mpf(sigma)
    return float(-mp.log(2 * mp.pi) / 2 - mp.log(sigma) - ((x - mu) / sigma) ** 2 / 2)


def main():
    g = {}

    # ---- (a) M:729-733 example-lplist, M:745 covariance, M:749-751 Cholesky factor
    g["example_lplist"] = [[90.0, 60.0, 90.0], [90.0, 90.0, 30.0], [60.0, 60.0, 60.0],
                           [60.0, 60.0, 90.0], [30.0, 30.0, 30.0]]
    g["example_covariance"] = [[504.0, 360.0, 180.0], [360.0, 360.0, 0.0], [180.0, 0.0, 720.0]]
    g["example_l_matrix"] = [[22.44994432064365, 0.0, 0.0],
                             [16.035674514745462, 10.141851056742201, 0.0],
                             [8.017837257372731, -12.677313820927745, 22.248595461286993]]
    # independent re-derivation with numpy (population covariance, lower Cholesky)
    a = np.array(g["example_lplist"])
    g["example_covariance_numpy"] = np.cov(a.T, bias=True).tolist()
    g["example_l_matrix_numpy"] = np.linalg.cholesky(np.cov(a.T, bias=True)).tolist()

    # ---- (b) log-normal M:372-377
    g["log_normal"] = [
        {"x": x, "mu": mu, "sigma": s, "value": log_normal_mp(x, mu, s)}
        for (x, mu, s) in [(1.0, 0.0, 1.0), (0.0, 0.0, 0.2), (13.0, 19.0, 0.2), (-4.5, 2.25, 3.0),
                           (1e-7, -2e-7, 1e-7), (5.0, 5.0, 10.0)]
    ]
    g["log_normal_scipy"] = [
        {"x": x, "mu": mu, "sigma": s, "value": float(stats.norm.logpdf(x, mu, s))}
        for (x, mu, s) in [(1.0, 0.0, 1.0), (0.3, -0.2, 0.7)]
    ]

    # ---- (a)+(b) basic line fit M:1186: f = b + m x, data ((-4 -1 2 5 10) (0 2 5 9 13)),
    # params (:b -1 :m 2), :data-error 0.2, default likelihood normal, flat prior.
    # NB the Lisp reader makes 0.2 a SINGLE float; to-double-floats (M:833-835) coerces it,
    # so sigma = (double)0.2f0 = 0.20000000298023224.
    xs = [-4, -1, 2, 5, 10]
    ys = [0, 2, 5, 9, 13]
    for tag, sig in (("single", float(np.float32(0.2))), ("double", 0.2)):
        tot = mp.mpf(0)
        for x, y in zip(xs, ys):
            m = -1 + 2 * x
            tot += mp.mpf(log_normal_mp(y, m, sig))
        g["line_fit_initial_logpost_sigma_%s" % tag] = {"sigma": sig, "value": float(tot)}
    g["line_fit"] = {"x": xs, "y": ys, "theta": [-1.0, 2.0], "param_keys": ["b", "m"]}

    # ---- global fit example M:1198: poly3 (b m c d) and line (e, m+g); shared :m
    x1, y1 = [0, 1, 2, 3, 4], [4, 5, 6, 8, 4]
    x2, y2 = [10, 20, 30, 40, 50], [1, 2, 3, 4, 5]
    P = dict(b=-1.0, m=2.0, c=1.0, d=-1.0, e=0.5, g=-2.0)
    sig = float(np.float32(0.2))
    tot = mp.mpf(0)
    for x, y in zip(x1, y1):
        f = P["b"] + P["m"] * x + P["c"] * x * x + P["d"] * x * x * x
        tot += mp.mpf(log_normal_mp(y, f, sig))
    for x, y in zip(x2, y2):
        f = P["e"] + (P["m"] + P["g"]) * x
        tot += mp.mpf(log_normal_mp(y, f, sig))
    g["global_fit"] = {"x1": x1, "y1": y1, "x2": x2, "y2": y2, "sigma": sig,
                       "param_keys": ["b", "m", "c", "d", "e", "g"],
                       "theta": [P[k] for k in "bmcdeg"], "initial_logpost": float(tot)}

    # ---- bounds penalty M:358-360, literal double arithmetic exp(x)-1
    def pen(p, lo, hi):
        if lo < p < hi:
            return 0.0
        m = min(abs(p - hi), abs(p - lo))
        return -1e10 * (math.exp(m * 1e-5) - 1.0)
    g["bound_penalty"] = [{"p": p, "lo": lo, "hi": hi, "value": pen(p, lo, hi)}
                          for (p, lo, hi) in [(1.5, 0.0, 1.0), (-1e-3, 0.0, 1.0), (1.0, 0.0, 1.0),
                                              (0.0, 0.0, 1.0), (0.5, 0.0, 1.0), (250.0, 100.0, 150.0),
                                              (-26.0, -25.0, -24.0)]]
    # high-precision value of the same formula (shows the formula's own conditioning)
    g["bound_penalty_mp"] = [{"p": 1.5, "lo": 0.0, "hi": 1.0,
                              "value": float(-mp.mpf(10) ** 10 * (mp.exp(mp.mpf("0.5") * mp.mpf("1e-5")) - 1))}]

    # ---- Poisson M:379-383 vs scipy logpmf (double log-factorial variant)
    g["log_poisson"] = [{"lambda": lam, "k": k, "value": float(stats.poisson.logpmf(k, lam))}
                        for (lam, k) in [(5.0, 3), (200.0, 180), (0.5, 0), (12.25, 20), (1.0, 1)]]
    # single-float log-factorial as M:379-380 sums it: float32 accumulation of float32 logs
    lf = []
    for k in (0, 1, 2, 5, 20, 100, 170):
        acc = np.float32(0.0)
        for m_ in range(1, k + 1):
            acc = np.float32(acc + np.float32(math.log(m_)))
        lf.append({"k": k, "value": float(acc) if k > 0 else 0.0})
    g["log_factorial_single"] = lf

    # ---- temperature schedule M:873-878 (numpy, same left-to-right products)
    def temps(n, d, T):
        sts = 10 * max(50, d)
        ts = max(n, 10 * sts)
        x = np.arange(ts, dtype=np.float64)
        k = float(1 + 2 * (ts // 5000))
        arg = ((x * math.pi) * k) * (1.0 / float(2 * ts))
        return np.maximum(1.0, np.cos(arg) * T)
    t = temps(30000, 6, 10.0)
    g["temperature_schedule"] = {
        "n": 30000, "d": 6, "temperature": 10.0, "len": int(t.size),
        "samples": {str(i): float(t[i]) for i in (0, 1, 1000, 2160, 2161, 7070, 7071, 15000, 29999)},
        "count_above_1": int((t > 1.0).sum()),
        "first_clipped": int(np.argmax(t <= 1.0)),
    }
    t2 = temps(1000, 8, 1000.0)  # n < 10*sts -> temp-steps = 5000
    g["temperature_schedule_short"] = {"n": 1000, "d": 8, "temperature": 1000.0,
                                       "len": int(t2.size),
                                       "samples": {str(i): float(t2[i]) for i in (0, 1, 2, 2499, 2500, 4999)}}

    # ---- (c) Random123 kat_vectors, philox4x32 10 rounds: ctr[4] key[2] -> out[4]
    g["philox4x32_10"] = [
        {"ctr": [0, 0, 0, 0], "key": [0, 0],
         "out": [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]},
        {"ctr": [0xffffffff] * 4, "key": [0xffffffff] * 2,
         "out": [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]},
        {"ctr": [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], "key": [0xa4093822, 0x299f31d0],
         "out": [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]},
    ]

    with open(OUT, "w") as f:
        json.dump(g, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
