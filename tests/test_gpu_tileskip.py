"""Tile-level skipping of Gaussian peaks (PeaksModel::tile_mask) must be an EXACT
transformation: a peak is left out of a 1024-point tile only when adding it could not change a
single bit of any model value in that tile.  The same engine with MHX_NO_TILE_SKIP=1 evaluates
every peak at every point; the two must agree bit for bit on every log-posterior, for benign and
for adversarial parameter vectors, and on whole walks."""
import os

import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def both_engines(mhx, spec, chains, **kw):
    """(skipping, not skipping); the switch is read when the problem is finalised"""
    out = []
    for flag in ("0", "1"):
        os.environ["MHX_NO_TILE_SKIP"] = flag
        try:
            e = spec.engine(mhx, chains, **kw)
            e.logpost(np.tile(spec.theta_star, (1, 1)))  # finalises the problem under this setting
        finally:
            os.environ.pop("MHX_NO_TILE_SKIP", None)
        out.append(e)
    return out


def adversarial_two_peak(theta_star, n, seed):
    rng = np.random.default_rng(seed)
    th = np.tile(theta_star, (n, 1))
    # columns: b0 b1 A1 mu1 w1 A2 mu2 w2
    th[:, 0] = rng.choice([0.5, 1e-3, 1e-12, 0.0, -0.2, -1e-3, -40.0, 40.0, 1e8], n)  # background level
    th[:, 1] = rng.choice([0.3, 0.0, -0.49, -0.6, 5.0, -1e-3], n)           # slope: bg may cross 0
    # amplitudes of either sign, from far below to far above the background (the rule's lower
    # bound on |f| loses what an evaluated peak of the wrong sign can take off it)
    th[:, 2] = rng.choice([1.0, 0.0, -1.0, 0.3, -0.3, 1e-30, 1e30, 1e300, 3e-310], n)
    th[:, 5] = rng.choice([0.7, 0.0, -0.7, 0.05, -0.05, 1e-300, 1e12], n)
    th[:, 3] = rng.uniform(-0.5, 1.5, n)                                    # centres in and out of range
    th[:, 6] = rng.uniform(-0.5, 1.5, n)
    th[:, 4] = 10.0 ** rng.uniform(-4, 0.5, n) * rng.choice([1, 1, 1, -1], n)  # widths, some negative
    th[:, 7] = 10.0 ** rng.uniform(-4, 0.5, n)
    th[:4] = theta_star * (1 + 0.01 * rng.standard_normal((4, 8)))         # and a few benign ones
    return th


@pytest.mark.parametrize("n,order", [(30000, "sorted"), (30000, "shuffled"), (1500, "sorted"), (1024, "sorted")])
def test_two_peak_logposts_identical_with_and_without_skipping(mhx, orc, n, order):
    s = pb.two_peak(n=n, seed=400 + n)
    if order == "shuffled":  # tiles then span the whole x range: nothing can be skipped, nothing may change
        x, y, sig, lik = s.data[0]
        p = np.random.default_rng(1).permutation(n)
        s.data[0] = (x[p], y[p], sig[p], lik)
    a, b = both_engines(mhx, s, 1)
    th = adversarial_two_peak(s.theta_star, 600, seed=n)
    with np.errstate(all="ignore"):
        ga, pa = a.logpost(th, parts=True)
        gb, pb_ = b.logpost(th, parts=True)
    assert np.array_equal(ga, gb, equal_nan=True)
    assert np.array_equal(pa, pb_, equal_nan=True)
    assert np.isfinite(ga[:4]).all()
    # ... and both equal the oracle's restatement of the kernel, which knows nothing of masks but
    # decides table / guarded exp and "exactly zero over this window" per window on its own
    op = s.oracle(orc)
    for i in range(0, 600, 5):
        with np.errstate(all="ignore"):
            ref = op.logpost_mirror(th[i])
        assert ga[i] == ref or (np.isnan(ga[i]) and np.isnan(ref)), (n, order, i, th[i], ga[i], ref)
    a.close()
    b.close()


def test_poisson_five_peaks_identical_with_and_without_skipping(mhx, orc):
    s = pb.poisson_peaks(n=40000, seed=9)        # BASELINE config 3's kernel: branches per peak
    a, b = both_engines(mhx, s, 1)
    rng = np.random.default_rng(3)
    th = s.theta_star * (1 + 0.05 * rng.standard_normal((200, s.d)))
    th[50:, 3::3] = 10.0 ** rng.uniform(-3.5, -0.5, (150, 5))      # widths from very narrow to broad
    th[100:, 0] = 10.0 ** rng.uniform(-9, 3, 100)                   # background over 12 decades
    th[150:, 1::3] = 10.0 ** rng.uniform(-20, 20, (50, 5))         # amplitudes over 40 decades
    th[190:, 1::3] *= -1.0                                          # ... some of them negative
    with np.errstate(all="ignore"):
        ga, gb = a.logpost(th), b.logpost(th)
    assert np.array_equal(ga, gb, equal_nan=True)
    assert np.isfinite(ga[:50]).all()
    # ... and the oracle's restatement of the kernel (which declines - NaN - where a rate comes
    # within 1/16 of 1: those go through the device's own log)
    op = s.oracle(orc)
    seen = 0
    for i in range(0, 200, 2):
        with np.errstate(all="ignore"):
            ref = op.logpost_mirror(th[i])
        if np.isnan(ref) and not np.isnan(ga[i]):
            continue
        seen += 1
        assert ga[i] == ref or (np.isnan(ga[i]) and np.isnan(ref)), (i, th[i], ga[i], ref)
    assert seen >= 60
    a.close()
    b.close()


def test_walks_identical_with_and_without_skipping(mhx):
    s = pb.two_peak(n=20000, seed=11)
    a, b = both_engines(mhx, s, 24, seed=5)
    th0 = pb.perturbed(s.theta_star, 24, 0.02, seed=6)
    for e in (a, b):
        e.init_chains(th0)
        e.adaptive_begin(1500, 10.0, 1)
        e.adaptive_advance(1 << 40)
    sa, sb = a.state(), b.state()
    for k in ("theta", "logpost", "age", "length"):
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(a.lmatrix(), b.lmatrix())
    a.close()
    b.close()
