#!/usr/bin/env python3
"""Near-minimax polynomial for 2^f on [-1/2, 1/2] (Chebyshev-node interpolation in 60-digit
arithmetic, constant term pinned to 1), rounded to binary64; prints C hex-float constants and
the measured max relative error of the ROUNDED polynomial evaluated in exact arithmetic."""
import sys
import mpmath as mp

mp.mp.dps = 60


def fit(deg, half=mp.mpf(1) / 2):
    # interpolate g(f) = (2^f - 1)/f  (degree deg-1) at Chebyshev nodes, then p = 1 + f g(f)
    n = deg
    nodes = [half * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    def g(f):
        return (mp.power(2, f) - 1) / f if abs(f) > mp.mpf(10) ** -40 else mp.log(2)
    A = mp.matrix(n, n)
    b = mp.matrix(n, 1)
    for i, x in enumerate(nodes):
        for j in range(n):
            A[i, j] = x ** j
        b[i] = g(x)
    c = mp.lu_solve(A, b)
    return [mp.mpf(1)] + [c[j] for j in range(n)]


def max_err(coefs, half=mp.mpf(1) / 2, samples=4001):
    cd = [mp.mpf(float(c)) for c in coefs]
    worst = 0
    for i in range(samples):
        f = -half + 2 * half * i / (samples - 1)
        p = mp.mpf(0)
        for c in reversed(cd):
            p = p * f + c
        e = abs(p / mp.power(2, f) - 1)
        worst = max(worst, e)
    return worst


if __name__ == "__main__":
    for deg in (int(a) for a in sys.argv[1:] or ["10", "11", "12"]):
        c = fit(deg)
        print("degree", deg, "max rel err", mp.nstr(max_err(c), 5), "(2^-53 = 1.11e-16)")
        for j, v in enumerate(c):
            print("  c%-2d = %s  /* %.17g */" % (j, float(v).hex(), float(v)))
