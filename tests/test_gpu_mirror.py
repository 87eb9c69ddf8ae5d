"""Bit-for-bit parity of BASELINE config 2's kernel with the oracle's mirror mode: the mirror
restates the kernel's arithmetic on the CPU (and is itself held to the reference's arithmetic
within the stated tolerance by tests/test_oracle_mirror.py), so here EVERYTHING is compared for
equality: log-posteriors, their likelihood and prior parts, whole histories of
walker-adaptive-steps runs."""
import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


@pytest.mark.parametrize("n", [1, 64, 65, 1023, 1024, 1025, 3000, 100000])
def test_logpost_equals_mirror(mhx, orc, n):
    s = pb.two_peak(n=n, seed=100 + n)
    op = s.oracle(orc)
    e = s.engine(mhx, 1)
    th = pb.perturbed(s.theta_star, 13, 0.03, seed=n)
    th[1] = s.theta_star * 1.7
    th[2, 0] = -3.0
    got, parts = e.logpost(th, parts=True)
    for i, t in enumerate(th):
        ref, rp = op.logpost_mirror(t, parts=True)
        assert got[i] == ref and parts[i, 0] == rp[0] and parts[i, 1] == rp[1], (n, i)
    e.close()


def test_injected_steps_equal_mirror(mhx, orc):
    s = pb.two_peak(n=1500, seed=7)
    op = s.oracle(orc)
    C_, d = 9, s.d
    rng = np.random.default_rng(0)
    e = s.engine(mhx, C_)
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=1)
    e.init_chains(th0)
    ws = [orc.Walker(op, th0[c], mirror=True) for c in range(C_)]
    L = np.diag(0.02 * np.abs(s.theta_star))
    for it in range(80):
        z = rng.standard_normal((C_, d))
        u = 1.0 - rng.random(C_)
        T = rng.uniform(1.0, 10.0, C_)
        acc = e.step_injected(L, z, u, T)
        st = e.state()
        for c, w in enumerate(ws):
            assert w.take_step_injected(L, z[c], u[c], T[c]) == acc[c]
            th, pr = w.last()
            assert np.array_equal(st["theta"][c], th) and st["logpost"][c] == pr, (it, c)
    e.close()


def test_adaptive_run_history_equals_mirror(mhx, orc):
    """a complete walker-adaptive-steps run (annealing, bound excursions, L updates, auto
    shutdown): the newest 1000 (prob, theta) pairs, L, T and loop index all equal"""
    s = pb.two_peak(n=800, seed=8)
    op = s.oracle(orc)
    C_, n = 6, 5000
    e = s.engine(mhx, C_, seed=31)
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=2)
    e.init_chains(th0)
    e.adaptive_begin(n, 10.0, 1)
    e.adaptive_advance(1 << 40)
    st = e.state()
    status, loop_i = e.chain_status()
    Ls = e.lmatrix()
    for c in range(C_):
        w = orc.Walker(op, th0[c], mirror=True)
        w.adaptive_begin(n, 10.0, 1, seed=31, chain_id=c)
        assert w.adaptive_advance(1 << 40) == orc.DONE and status[c] == 1
        assert loop_i[c] == w.loop_index and st["age"][c] == w.age
        pg, tg = e.trace(c, 1000)
        po, to = w.trace(1000)
        assert np.array_equal(pg, po) and np.array_equal(tg, to), c
        assert np.array_equal(Ls[c], w.current_l())
        assert st["best_logpost"][c] == w.best()[1]
    e.close()


@pytest.mark.parametrize("n", [700, 5000])
def test_guarded_exp_path_is_the_chains_own_choice(mhx, orc, n):
    """Peaks so narrow that |t| leaves the table form's range (t^2 >= 2^23) inside a window where
    they still count send THAT chain through the guarded exp there; the choice is the wave's own,
    so the chains sharing its workgroup keep their bits (no workgroup vote), and both forms
    equal the mirror."""
    s = pb.two_peak(n=n, seed=21)
    op = s.oracle(orc)
    th = pb.perturbed(s.theta_star, 32, 0.02, seed=22)
    narrow = [3, 4, 17, 30]
    th[3, 4] = 1e-4      # w1: |t| up to ~1e4
    th[4, 7] = 1e-6      # w2: |t| up to ~1e6
    th[17, 4] = 2.0e-4
    th[30, 7] = -3e-5    # a negative width: |t| is what counts
    e = s.engine(mhx, 1)
    got, parts = e.logpost(th, parts=True)
    for i, t in enumerate(th):
        ref, rp = op.logpost_mirror(t, parts=True)
        assert got[i] == ref and parts[i, 0] == rp[0], (n, i)
        assert abs(got[i] - op.logpost(t)) <= 1e-12 * op.abs_terms(t) + 2.0 ** -52 * 1e10 * 8
    # the same vectors with the narrow ones replaced by benign ones: every other slot unchanged
    th2 = th.copy()
    th2[narrow] = s.theta_star
    got2 = e.logpost(th2)
    keep = [i for i in range(32) if i not in narrow]
    assert np.array_equal(got[keep], got2[keep])
    e.close()


def test_guarded_form_only_in_the_windows_that_need_it(mhx, orc):
    """The table exp / guarded form choice is made per 2048-point window: a peak whose |t| is at
    least 40 over a whole window is exactly zero there and left out, so a very narrow peak sends
    only the window that holds it (if any) through the guarded form.  Every bit equals the
    mirror, which restates the windows; MHX_NO_TILE_SKIP=1 changes nothing (the exact-zero rule
    is not an option); and such a step is no longer several times dearer than its neighbours."""
    import os
    n = 50000
    s = pb.two_peak(n=n, seed=23)
    op = s.oracle(orc)
    th = pb.perturbed(s.theta_star, 24, 0.02, seed=24)
    th[2, 4] = 1e-4                      # |t| up to ~6e3 at the far end, < 2890 inside its windows
    th[3, 7] = 2e-6                      # so narrow that even its own window runs |t| past 2890
    th[4, 4], th[4, 3] = 3e-5, 0.99999   # narrow, in the last (partly padded) window
    th[5, 7], th[5, 6] = 1e-4, 5.0       # narrow and far outside the data: zero everywhere
    th[6, 4], th[6, 0] = 1e-4, -0.3      # narrow with a negative background (no relative rule)
    th[7, 4], th[7, 2] = 1e-4, -2.0      # narrow with a negative amplitude
    th[8, 7], th[8, 5] = 1e-4, np.inf    # an infinite amplitude: nothing may be left out
    th[9, 4] = -1e-4
    e = s.engine(mhx, 1)
    os.environ["MHX_NO_TILE_SKIP"] = "1"
    try:
        e2 = s.engine(mhx, 1)
        with np.errstate(all="ignore"):
            got2 = e2.logpost(th)
    finally:
        os.environ.pop("MHX_NO_TILE_SKIP", None)
    with np.errstate(all="ignore"):
        got, parts = e.logpost(th, parts=True)
    assert np.array_equal(got, got2, equal_nan=True)
    for i, t in enumerate(th):
        with np.errstate(all="ignore"):
            ref, rp = op.logpost_mirror(t, parts=True)
            fa = op.logpost(t)
        assert (got[i] == ref and parts[i, 0] == rp[0]) or (np.isnan(got[i]) and np.isnan(ref)), i
        if np.isfinite(fa):
            assert abs(got[i] - fa) <= 1e-12 * op.abs_terms(t) + 2.0 ** -52 * 1e10 * 8, i
    assert not np.isfinite(got[8])
    e.close()
    e2.close()


def test_bounds_prior_far_outside_the_box(mhx, orc):
    """(exp x) of M:360 beyond the range of a double is -inf on the device, in the restatement and
    in the reference's own arithmetic (where SBCL would signal floating-point-overflow)"""
    s = pb.two_peak(n=700, seed=5)
    op = s.oracle(orc)
    th = np.tile(s.theta_star, (6, 1))
    th[0, 2], th[1, 2], th[2, 0], th[3, 4], th[4, 7] = 1e30, 1e25, -1e300, 7.2e7, 1e9
    th[5, 2] = 1.5 + 6.5e7     # exp(650): still inside
    e = s.engine(mhx, 1)
    with np.errstate(all="ignore"):
        got, parts = e.logpost(th, parts=True)
    for i, t in enumerate(th):
        with np.errstate(all="ignore"):
            ref, rp = op.logpost_mirror(t, parts=True)
            fa, fp = op.logpost(t, parts=True)
        assert parts[i, 1] == rp[1], (i, parts[i, 1], rp[1])
        assert (parts[i, 1] == -np.inf) == (i < 5) and (fp[1] == -np.inf) == (i < 5)
    e.close()


@pytest.mark.parametrize("n", [1, 64, 1025, 3000, 40000])
@pytest.mark.parametrize("logfact_double", [False, True])
def test_poisson_logpost_equals_mirror(mhx, orc, n, logfact_double):
    """BASELINE config 3's kernel (five Gaussian peaks, log-poisson, M:379-383 through
    M:402-416) against its restatement: table-driven log, masked pads, 4-point blocks, the
    recurrence where the grid allows it - every bit"""
    s = pb.poisson_peaks(n=n, seed=200 + n)
    op = s.oracle(orc, logfact_double=logfact_double)
    e = s.engine(mhx, 1, poisson_logfact_double=logfact_double)
    th = pb.perturbed(s.theta_star, 12, 0.02, seed=n)
    th[2, 3] *= 0.05
    th[3, 6] *= 4.0
    th[4] = s.theta_star * 1.6   # outside the bounds box
    got, parts = e.logpost(th, parts=True)
    for i, t in enumerate(th):
        ref, rp = op.logpost_mirror(t, parts=True)
        assert np.isfinite(ref)
        assert got[i] == ref and parts[i, 0] == rp[0] and parts[i, 1] == rp[1], (n, i)
    e.close()


@pytest.mark.parametrize("n", [64, 3000])
def test_poisson_rates_near_one_and_outside_the_log(mhx, orc, n):
    """tlog_rate (csrc/mhx_device.hpp) keeps the table form within 1/16 of 1 (where the user
    expressions' tlog detours through mlog) and answers everything that is not a positive normal
    number with a NaN: rates 0.94 .. 1.06 equal the mirror bit for bit and the faithful sum
    within the tolerance; a background that makes the rate negative somewhere is a NaN on the
    device, in the mirror and in the faithful oracle (M:383: log of a negative rate)"""
    s = pb.poisson_peaks(n=n, seed=5 + n)
    op = s.oracle(orc)
    e = s.engine(mhx, 1)
    th = np.tile(s.theta_star, (8, 1))
    for i in range(6):
        th[i, 0] = 0.94 + 0.024 * i          # the background alone ...
        th[i, 1::3] = 1e-3 * i               # ... under peaks of height 0 .. 5e-3
    th[6, 0] = -30.0                         # rate < 0 between the peaks
    th[7, 0] = 0.0
    th[7, 1::3] = 0.0                        # rate == 0 everywhere
    got, parts = e.logpost(th, parts=True)
    for i, t in enumerate(th):
        ref, rp = op.logpost_mirror(t, parts=True)
        if i < 6:
            assert np.isfinite(ref) and got[i] == ref and parts[i, 0] == rp[0], (n, i, got[i], ref)
            fa, fp = op.logpost(t, parts=True)   # (the likelihood part: the prior is far outside its box)
            assert abs(parts[i, 0] - fp[0]) <= 1e-12 * op.abs_terms(t), (n, i)
        else:
            assert np.isnan(got[i]) and np.isnan(ref), (n, i, got[i], ref)
    e.close()


def test_poisson_walk_equals_mirror(mhx, orc):
    s = pb.poisson_peaks(n=6000, seed=77)
    op = s.oracle(orc)
    C_, n = 4, 1500
    e = s.engine(mhx, C_, seed=23)
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=3)
    l0 = np.diag(0.002 * np.abs(s.theta_star))
    e.init_chains(th0)
    e.adaptive_begin(n, 10.0, 1, l_matrix=l0)
    e.adaptive_advance(1 << 40)
    st = e.state()
    for c in range(C_):
        w = orc.Walker(op, th0[c], mirror=True)
        w.adaptive_begin(n, 10.0, 1, l_matrix=l0, seed=23, chain_id=c)
        w.adaptive_advance(1 << 40)
        th, pr = w.last()
        assert st["age"][c] == w.age
        assert np.array_equal(st["theta"][c], th) and st["logpost"][c] == pr, c
    e.close()


@pytest.mark.parametrize("n", [1, 65, 1024, 2049, 5000, 100000])
def test_cutoff_logpost_and_walk_equal_mirror(mhx, orc, n):
    """log-liklihood-normal-cutoff (M:419-427; BASELINE names it beside the weighted normal
    likelihood): the four-array kernel (x, y/sigma, 1/sigma, the point's constant) bit for bit
    against the oracle's restatement, clamped terms included, on log-posteriors and on a walk"""
    s = pb.two_peak(n=n, seed=200 + n, lik=pb.CUTOFF)
    op = s.oracle(orc)
    e = s.engine(mhx, 3, seed=17)
    assert "cutoff" in e.kernel_name()
    th = pb.perturbed(s.theta_star, 13, 0.03, seed=n)
    th[1] = s.theta_star * 1.7
    th[2, 2] = 40.0
    th[3, 4] = 1e-3
    got, parts = e.logpost(th, parts=True)
    for i, t in enumerate(th):
        ref, rp = op.logpost_mirror(t, parts=True)
        assert got[i] == ref and parts[i, 0] == rp[0] and parts[i, 1] == rp[1], (n, i)
    th0 = pb.perturbed(s.theta_star, 3, 0.01, seed=4)
    n_it = 700 if n <= 5000 else 150  # (the CPU side walks 3 x n_it x n points)
    e.init_chains(th0)
    e.adaptive_begin(n_it, 10.0, 1)
    e.adaptive_advance(1 << 40)
    st = e.state()
    for c in range(3):
        w = orc.Walker(op, th0[c], mirror=True)
        w.adaptive_begin(n_it, 10.0, 1, seed=17, chain_id=c)
        w.adaptive_advance(1 << 40)
        th_w, pr_w = w.last()
        assert np.array_equal(st["theta"][c], th_w) and st["logpost"][c] == pr_w, (n, c)
        assert st["age"][c] == w.age
    e.close()
