"""The oracle's MIRROR mode (the GPU kernel's own arithmetic, restated on the CPU) against its
faithful mode (the reference's arithmetic): the stated tolerance, checked without a GPU.  The
GPU tests then require the device to EQUAL the mirror bit for bit."""
import numpy as np
import pytest

import problems as pb

REL = 1e-12
PRIOR_ULP = 2.0 ** -52 * 1e10


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 1024, 1025, 5000, 100000])
def test_mirror_within_tolerance_of_faithful(orc, n):
    s = pb.two_peak(n=n, seed=n + 1)
    op = s.oracle(orc)
    th = pb.perturbed(s.theta_star, 6, 0.03, seed=n)
    th[1] = s.theta_star * 1.7      # every bound violated
    th[2, 4] *= 0.3                 # a narrow peak
    for t in th:
        a, pa = op.logpost(t, parts=True)
        b, pm = op.logpost_mirror(t, parts=True)
        nv = int(((t <= s.bounds[0][1]) | (t >= s.bounds[0][2])).sum())
        assert abs(pa[0] - pm[0]) <= REL * op.abs_terms(t)
        assert abs(pa[1] - pm[1]) <= 4 * PRIOR_ULP * nv
        assert (pa[1] == 0.0) == (pm[1] == 0.0)


def test_mirror_declines_what_the_kernel_does_differently(orc):
    s = pb.two_peak(n=100, seed=3)
    op = s.oracle(orc)
    th = s.theta_star.copy()
    th[4] = 1e-6                     # |t| > kFastT somewhere: the chain's guarded path, mirrored
    a, m = op.logpost(th), op.logpost_mirror(th)
    assert np.isfinite(a) and abs(a - m) <= REL * op.abs_terms(th)
    th[4] = 1e-4                     # |t| ~ 3000-8000: beyond the table form's range (2^23 > t^2)
    a, m = op.logpost(th), op.logpost_mirror(th)
    assert np.isfinite(a) and abs(a - m) <= REL * op.abs_terms(th)
    # other models / likelihoods are not restated: the mirror says so with a NaN
    g = pb.global_fit(n_each=40, n_sets=2)
    assert np.isnan(g.oracle(orc).logpost_mirror(g.theta_star))
    # a Poisson rate within 1/16 of 1 is the table form's too (tlog_rate): mirrored, and within
    # the tolerance of the faithful sum
    sp = pb.poisson_peaks(n=50)
    t1 = sp.theta_star.copy()
    t1[0] = 1.01
    t1[1::3] = 0.0
    so = sp.oracle(orc)
    assert abs(so.logpost_mirror(t1) - so.logpost(t1)) <= REL * so.abs_terms(t1)


@pytest.mark.parametrize("n,logfact_double", [(1, False), (64, False), (1000, True), (1025, False),
                                              (40000, False), (100000, True)])
def test_poisson_mirror_within_tolerance_of_faithful(orc, n, logfact_double):
    """BASELINE config 3's kernel restated: five Gaussian peaks (recurrence where the grid is
    fine enough, direct table exp elsewhere), the table-driven log of the Poisson term, the
    single-float log-factorials of M:379-380"""
    s = pb.poisson_peaks(n=n, seed=n + 3)
    op = s.oracle(orc, logfact_double=logfact_double)
    th = pb.perturbed(s.theta_star, 6, 0.02, seed=n)
    th[2, 3] *= 0.05                # a narrow peak: direct form
    th[3, 6] *= 4.0                 # a wide one
    for t in th:
        a, pa = op.logpost(t, parts=True)
        b, pm = op.logpost_mirror(t, parts=True)
        assert np.isfinite(a) and np.isfinite(b)
        assert abs(pa[0] - pm[0]) <= REL * op.abs_terms(t), (n, pa[0], pm[0])
        assert pa[1] == pm[1] or abs(pa[1] - pm[1]) <= 4 * PRIOR_ULP * s.d


def test_mirror_walker_tracks_faithful_walker(orc):
    s = pb.two_peak(n=300, seed=5)
    op = s.oracle(orc)
    a = orc.Walker(op, s.theta_star)
    b = orc.Walker(op, s.theta_star, mirror=True)
    for w in (a, b):
        w.adaptive_begin(2500, 10.0, 1, seed=9, chain_id=4)
        assert w.adaptive_advance(1 << 40) == orc.DONE
    assert a.age == b.age
    assert np.array_equal(a.last()[0], b.last()[0])          # same accept decisions throughout
    assert abs(a.last()[1] - b.last()[1]) <= REL * op.abs_terms(a.last()[0])


def test_bounds_prior_far_outside_the_box(orc):
    """(exp x) of M:360 beyond the range of a double: the reference signals floating-point-overflow,
    the faithful mode and the kernel's restatement both answer -inf (which freezes the chain),
    whatever the size of the violation - a reduction left to itself answered +inf for a bound
    violated by 1e30 on the device and something finite in the restatement"""
    s = pb.two_peak(n=300, seed=5)
    op = s.oracle(orc)
    for k, v in ((2, 1e30), (2, 1e25), (0, -1e300), (4, 7.2e7), (7, 1e9)):
        t = s.theta_star.copy()
        t[k] = v
        with np.errstate(all="ignore"):
            a, pa = op.logpost(t, parts=True)
            b, pm = op.logpost_mirror(t, parts=True)
        assert pa[1] == -np.inf and pm[1] == -np.inf, (k, v, pa[1], pm[1])
    # ... and continuous up to there: just inside the range both are finite and agree
    t = s.theta_star.copy()
    t[2] = 1.5 + 6.5e7          # exp(650) = 1e282, times 1e10 still a double
    a, pa = op.logpost(t, parts=True)
    b, pm = op.logpost_mirror(t, parts=True)
    assert np.isfinite(pa[1]) and np.isfinite(pm[1]) and abs(pa[1] / pm[1] - 1.0) < 1e-12


@pytest.mark.parametrize("n", [1, 65, 1000, 1025, 5000, 100000])
def test_cutoff_mirror_within_tolerance_of_faithful(orc, n):
    """log-liklihood-normal-cutoff (M:419-427, each term clamped at -5000) in the kernel's
    arithmetic against the faithful restatement - with points whose terms ARE clamped"""
    s = pb.two_peak(n=n, seed=n + 5, lik=pb.CUTOFF)
    op = s.oracle(orc)
    th = pb.perturbed(s.theta_star, 6, 0.03, seed=n)
    th[1] = s.theta_star * 1.7      # every bound violated, most terms at the clamp
    th[2, 2] = 40.0                 # a peak 40 times too high: its points are clamped
    clamped = 0
    for t in th:
        a, pa = op.logpost(t, parts=True)
        b, pm = op.logpost_mirror(t, parts=True)
        assert np.isfinite(b)
        assert abs(pa[0] - pm[0]) <= REL * op.abs_terms(t)
        clamped += pa[0] <= -5000.0
    assert n < 65 or clamped >= 1


@pytest.mark.parametrize("n", [3000, 30000, 100000])
def test_per_window_grids_mirror_within_tolerance_of_faithful(orc, n):
    """Per-window grids (csrc/mhx_engine.cpp, PeaksModel::regrid): on an x that is three scans
    of different steps laid end to end plus one jittered window, every window on a grid takes the
    recurrence with ITS step, the junction and jittered windows the direct form.  The mirror's
    restatement against the faithful sum, and against the mirror without window grids (round 3's
    rule: such a dataset ran the direct form everywhere) - the same bound as recurrence vs direct."""
    s = pb.two_peak_piecewise(n=n, seed=n + 2)
    op = s.oracle(orc)
    th = pb.perturbed(s.theta_star, 6, 0.03, seed=n)
    th[2, 4] *= 0.3                 # a narrow peak
    th[3, 7] *= 3.0                 # a wide one
    differs = 0
    for t in th:
        a, pa = op.logpost(t, parts=True)
        b, pm = op.logpost_mirror(t, parts=True)
        orc.mirror_set_window_grids(False)
        try:
            c, pc = op.logpost_mirror(t, parts=True)
        finally:
            orc.mirror_set_window_grids(True)
        scale = op.abs_terms(t)
        assert abs(pa[0] - pm[0]) <= REL * scale
        assert abs(pm[0] - pc[0]) <= 1e-13 * scale
        differs += pm[0] != pc[0]
    assert n < 30000 or differs >= 3      # (about 615 grid points per peak width are needed)


@pytest.mark.slow
def test_mirror_within_tolerance_of_faithful_at_config_3_length(orc):
    """the mirror-vs-faithful bound at config 3's full length, N = 1e6 (the device side
    is held to the mirror bit for bit there: tests/test_gpu_fullsize.py)"""
    s = pb.poisson_peaks(n=1000000, seed=12)
    op = s.oracle(orc)
    th = pb.perturbed(s.theta_star, 2, 0.02, seed=6)
    th[1, 3] *= 0.05
    for t in th:
        a, pa = op.logpost(t, parts=True)
        b, pm = op.logpost_mirror(t, parts=True)
        assert np.isfinite(a) and np.isfinite(b)
        assert abs(pa[0] - pm[0]) <= REL * op.abs_terms(t)
