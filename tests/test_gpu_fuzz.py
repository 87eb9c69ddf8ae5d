"""Differential fuzz: log-posteriors of wildly perturbed parameter vectors - the proposals a walk
makes from `diag(theta)` in its first iterations: narrow and wide peaks, negative amplitudes and
widths, peaks outside the data - must equal the oracle's mirror bit for bit (configs 2 and 3's
kernels: every window variant, seeding class, guarded window and masked loop gets its share)."""
import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def fuzz(mhx, orc, s, chains, scales, seed, **okw):
    op = s.oracle(orc, **okw)
    e = s.engine(mhx, chains)
    rng = np.random.default_rng(seed)
    bad = []
    for sc in scales:
        th = s.theta_star[None, :] * (1.0 + sc * rng.standard_normal((chains, s.d)))
        got, parts = e.logpost(th, parts=True)
        for i, t in enumerate(th):
            ref, rp = op.logpost_mirror(t, parts=True)
            same = (got[i] == ref or (np.isnan(got[i]) and np.isnan(ref))) and \
                   (parts[i, 0] == rp[0] or (np.isnan(parts[i, 0]) and np.isnan(rp[0])))
            if not same:
                bad.append((sc, i, got[i], ref, parts[i, 0], rp[0]))
    e.close()
    return bad


@pytest.mark.parametrize("n", [5000, 70000])
def test_two_peak_normal_fuzz(mhx, orc, n):
    bad = fuzz(mhx, orc, pb.two_peak(n=n, seed=n), 512, (0.02, 0.1, 0.3, 1.0, 3.0), seed=n + 1)
    assert not bad, bad[:5]


def test_two_peak_cutoff_fuzz(mhx, orc):
    bad = fuzz(mhx, orc, pb.two_peak(n=20000, seed=4, lik=pb.CUTOFF), 384, (0.05, 0.3, 1.0), seed=9)
    assert not bad, bad[:5]


@pytest.mark.parametrize("n", [3000, 50000])
def test_five_peak_poisson_fuzz(mhx, orc, n):
    bad = fuzz(mhx, orc, pb.poisson_peaks(n=n, seed=n), 384, (0.02, 0.1, 0.3, 1.0), seed=n + 2)
    assert not bad, bad[:5]


@pytest.mark.parametrize("scale", [0.02, 0.1, 0.3])
def test_global_fit_pseudo_voigt_fuzz_within_tolerance(mhx, orc, scale):
    """config 4's kernel has no mirror (its reciprocal starts from v_rcp_f64): held to the
    faithful oracle within the stated 1e-12 sum |term| on perturbed parameter vectors - the fast
    path (one shared reciprocal, table exps) and, for the wide perturbations, the guarded one"""
    s = pb.global_fit(n_each=6000, n_sets=4, seed=11)
    op = s.oracle(orc)
    chains = 128
    e = s.engine(mhx, chains)
    rng = np.random.default_rng(int(scale * 1000))
    th = s.theta_star[None, :] * (1.0 + scale * rng.standard_normal((chains, s.d)))
    got, parts = e.logpost(th, parts=True)
    worst = 0.0
    for i, t in enumerate(th):
        ref, rp = op.logpost(t, parts=True)
        if not np.isfinite(ref):
            assert not np.isfinite(got[i]), (i, got[i], ref)
            continue
        tol = 1e-12 * op.abs_terms(t)
        assert abs(parts[i, 0] - rp[0]) <= tol, (scale, i, parts[i, 0], rp[0], tol)
        worst = max(worst, abs(parts[i, 0] - rp[0]) / tol)
    e.close()
    assert worst < 1.0


@pytest.mark.parametrize("npk,n", [(3, 30000), (4, 9000)])
def test_run_time_specialised_peaks_fuzz(mhx, orc, npk, n):
    """three and four Gaussian peaks on a linear background: no ahead-of-time kernel, compiled
    for gfx950 on first use (hiprtc, PeaksModel<2, npk>): the run-time-masked peak loops and the
    per-peak seeding constants in LDS, against the mirror, every bit"""
    rng = np.random.default_rng(npk)
    x = np.linspace(0.0, 1.0, n)
    sig = rng.uniform(0.05, 0.15, n)
    th = [0.5, 0.3]
    for k in range(npk):
        th += [rng.uniform(0.6, 1.2), (k + 0.6) / (npk + 0.2), rng.uniform(0.03, 0.07)]
    th = np.array(th)
    y = pb.model_eval_np(pb.GAUSS, (2, npk), th, x) + sig * rng.standard_normal(n)
    s = pb.Spec(len(th))
    lo, hi = np.minimum(th * 0.5, th * 1.5), np.maximum(th * 0.5, th * 1.5)
    s.add(pb.GAUSS, (2, npk), range(len(th)), x, y, sig, pb.NORMAL, (list(range(len(th))), lo, hi))
    s.theta_star = th
    e = s.engine(mhx, 1)
    assert "rtc[PeaksModel<2, %d, false>" % npk in e.kernel_name(), e.kernel_name()
    e.close()
    bad = fuzz(mhx, orc, s, 256, (0.02, 0.1, 0.3, 1.0), seed=npk + 40)
    assert not bad, bad[:5]
