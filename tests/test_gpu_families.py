"""The device code exists in two kernel families (8 and 16 chains per workgroup, 1024- and
2048-point LDS tiles; csrc/mhx_types.hpp).  Which one runs is a performance choice of the engine
(MHX_FAMILY_WPG pins it) and must not be visible in any result: the per-lane summation order does
not depend on the tile size, so everything is compared for equality."""
import os

import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def engines(make):
    out = []
    for wpg in ("8", "16"):
        os.environ["MHX_FAMILY_WPG"] = wpg
        try:
            out.append(make())
        finally:
            os.environ.pop("MHX_FAMILY_WPG", None)
    return out


SPECS = [
    ("two_peak", lambda: pb.two_peak(n=5000, seed=31)),
    ("two_peak_short", lambda: pb.two_peak(n=70, seed=32)),
    ("two_peak_cutoff", lambda: pb.two_peak(n=2049, seed=33, lik=pb.CUTOFF)),
    ("poisson", lambda: pb.poisson_peaks(n=4100, seed=34)),
    ("global_fit", lambda: pb.global_fit(n_each=700, n_sets=4, seed=35)),
    ("lorder", lambda: pb.lorder()),
]


@pytest.mark.parametrize("name,make_spec", SPECS, ids=[s[0] for s in SPECS])
def test_logposts_and_walks_do_not_depend_on_the_family(mhx, name, make_spec):
    s = make_spec()
    C_ = 19  # not a multiple of either workgroup size
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=3)

    def make():
        e = s.engine(mhx, C_, seed=8)
        e.init_chains(th0)  # finalises the problem under the pinned family
        return e
    a, b = engines(make)
    th = pb.perturbed(s.theta_star, 40, 0.03, seed=4)
    ga, pa = a.logpost(th, parts=True)
    gb, pb_ = b.logpost(th, parts=True)
    assert np.array_equal(ga, gb, equal_nan=True) and np.array_equal(pa, pb_, equal_nan=True)
    l0 = np.diag(0.003 * np.abs(s.theta_star) + 1e-6)
    for e in (a, b):
        e.adaptive_begin(1300, 10.0, 1, l_matrix=l0)
        e.adaptive_advance(1 << 40)
    sa, sb = a.state(), b.state()
    for k in ("theta", "logpost", "age", "length"):
        assert np.array_equal(sa[k], sb[k], equal_nan=True), (name, k)
    assert np.array_equal(a.lmatrix(), b.lmatrix(), equal_nan=True)
    assert np.array_equal(a.chain_status()[0], b.chain_status()[0])
    a.close()
    b.close()


def test_expression_kernels_in_both_families(mhx):
    rng = np.random.default_rng(12)
    n = 4500
    x = np.linspace(0, 4, n)
    sig = rng.uniform(0.05, 0.2, n)
    y = 2.0 * np.exp(-x / 1.5) + 0.3 + sig * rng.standard_normal(n)
    text = "(lambda (x &key a tau c &allow-other-keys) (+ c (* a (exp (/ (- x) tau)))))"

    def make():
        return mhx.walker_create(function=mhx.models.lisp(text), data=[x, y],
                                 params=[":a", 1.8, ":tau", 1.4, ":c", 0.35], data_error=sig,
                                 n_chains=5, seed=2)
    a, b = engines(make)
    assert a.last_step().prob == b.last_step().prob
    # exp(-x/tau) overflows once a proposal makes tau slightly negative (a floating-point trap in
    # the reference too): start from a small :l-matrix instead of diag(params)
    for w in (a, b):
        mhx.walker_adaptive_steps_full(w, n=1500, temperature=10, auto=":prob-settle",
                                       l_matrix=np.diag([0.02, 0.02, 0.01]))
    assert mhx.walker_get(a, get=":most-likely-params") == mhx.walker_get(b, get=":most-likely-params")
    assert np.array_equal(a.engine.state()["theta"], b.engine.state()["theta"])


def test_resident_single_tile_repeated_launches(mhx, orc):
    """A problem of one function and one tile walked by at most one workgroup keeps its tile in
    LDS (FnDesc::solo).  The wave that stages the tile sets a flag the others read: every launch -
    not only the first, whose cold caches hide the race - must wait for all readers before it
    does.  Repeated evaluations on such engines, for the 2-point and the 4-point inner loops,
    against the oracle."""
    for s in (pb.poisson_peaks(n=900, seed=3), pb.two_peak(n=1000, seed=4), pb.lorder()):
        op = s.oracle(orc)
        th = pb.perturbed(s.theta_star, 10, 0.01, seed=9)
        ref = np.array([op.logpost(t) for t in th])
        scale = np.array([op.abs_terms(t) for t in th])
        for chains in (1, 2, 8):
            e = s.engine(mhx, chains)
            for rep in range(6):
                got = e.logpost(th)
                assert np.all(np.abs(got - ref) <= 1e-12 * scale), (e.kernel_name(), chains, rep)
            e.close()


@pytest.mark.parametrize("n", [1500, 2048, 4097, 5000, 12288, 100000])
@pytest.mark.parametrize("wpg", ["8", "16"])
def test_yw_tiles_give_the_same_bits(mhx, orc, n, wpg):
    """Steps in which every running chain of a workgroup advances all its peaks and the
    background by the recurrence take two-array tiles of twice the points (sweep_yw: y/sigma and
    1/sigma only, seeds' x straight from L2; decided per sweep by a workgroup vote).  The layout
    must not show in any bit: against MHX_NO_YW=1 and against the oracle's mirror, for datasets of
    one window, whole windows, a ragged last window, an odd and an even number of windows, in
    both kernel families, on log-posteriors and on walks (where chains of one workgroup leave and
    re-enter the all-recurrence state from proposal to proposal)."""
    s = pb.two_peak(n=n, seed=40 + n % 7)
    op = s.oracle(orc)
    C_ = 37
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=3)
    out = []
    os.environ["MHX_FAMILY_WPG"] = wpg
    try:
        for flag in ("1", None):
            if flag:
                os.environ["MHX_NO_YW"] = flag
            try:
                e = s.engine(mhx, C_, seed=8)
                e.init_chains(th0)  # finalises the problem under these settings
            finally:
                os.environ.pop("MHX_NO_YW", None)
            out.append(e)
    finally:
        os.environ.pop("MHX_FAMILY_WPG", None)
    a, b = out
    th = pb.perturbed(s.theta_star, 64, 0.02, seed=4)     # all-recurrence vectors: yw tiles
    th[5, 4] = 1e-4                                        # ... but for a few: a narrow peak,
    th[21, 7] = -0.03                                      # a negative width,
    th[40, 3] = 7.5                                        # a centre far outside the data
    ga, gb = a.logpost(th), b.logpost(th)
    assert np.array_equal(ga, gb)
    for c in (0, 1, 5, 21, 40, 63):
        assert gb[c] == op.logpost_mirror(th[c]), (n, wpg, c)
    l0 = np.diag(0.003 * np.abs(s.theta_star))
    for e in (a, b):
        e.adaptive_begin(900, 10.0, 1, l_matrix=l0)
        e.adaptive_advance(1 << 40)
    sa, sb = a.state(), b.state()
    for k in ("theta", "logpost", "best_logpost", "age", "length"):
        assert np.array_equal(sa[k], sb[k]), (n, wpg, k)
    assert np.array_equal(a.lmatrix(), b.lmatrix())
    # ... and from the default diag(theta) start, where proposals are wild and the chains of a
    # workgroup disagree about the layout most of the time
    for e in (a, b):
        e.init_chains(th0)
        e.adaptive_begin(30000, 10.0, 1)
        e.adaptive_advance(60)
    sa, sb = a.state(), b.state()
    for k in ("theta", "logpost", "age"):
        assert np.array_equal(sa[k], sb[k]), (n, wpg, k)
    a.close()
    b.close()


DEAL_CASES = [
    ("two_peak", lambda: pb.two_peak(n=30000, seed=41), None),
    ("poisson5", lambda: pb.poisson_peaks(n=24000, seed=4), 0.002),
    ("global_fit", lambda: pb.global_fit(n_each=7000, n_sets=3, seed=5), None),
]


@pytest.mark.parametrize("name,make,lscale", DEAL_CASES, ids=[c[0] for c in DEAL_CASES])
@pytest.mark.parametrize("wpg", ["8", "16"])
def test_dealing_proposals_to_wave_slots_leaves_every_bit_alone(mhx, orc, name, make, lscale, wpg):
    """group_logpost ranks the proposals of a workgroup by estimated sweep cost and deals them to
    its wave slots in a snake over the SIMDs; wave slot s computes the likelihood sums of chain
    src(s) and hands them back.  A sum is one wave's lane-strided accumulation and butterfly
    whichever wave runs it: against MHX_NO_DEAL=1 bit for bit - log-posteriors of wildly different
    vectors (so that the deal really permutes), walks from the diag(theta) start where costs
    differ most, a workgroup that is not full, chains that have finished while others walk."""
    # (the product build leaves the dealing out - it measured slower where it matters,
    # csrc/mhx_kernels.hpp "DEALING" - so this test wants a library built with -DMHX_DEAL:
    # make -C lisp-mcmc_amd/csrc OUT=../libmhx_deal.so OUT_HOOKS=/tmp/h.so CXXFLAGS="... -DMHX_DEAL",
    # MHX_LIBRARY=.../libmhx_deal.so MHX_TEST_DEAL=1 pytest tests/test_gpu_families.py -k dealing)
    if not os.environ.get("MHX_TEST_DEAL"):
        pytest.skip("libmhx.so is built without -DMHX_DEAL")
    s = make()
    C_ = 37
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=3)
    out = []
    os.environ["MHX_FAMILY_WPG"] = wpg
    try:
        for flag in ("1", None):
            if flag:
                os.environ["MHX_NO_DEAL"] = flag
            try:
                e = s.engine(mhx, C_, seed=8)
                e.init_chains(th0)  # finalises the problem under these settings
            finally:
                os.environ.pop("MHX_NO_DEAL", None)
            out.append(e)
    finally:
        os.environ.pop("MHX_FAMILY_WPG", None)
    a, b = out
    rng = np.random.default_rng(5)
    th = s.theta_star[None, :] * (1.0 + 0.5 * rng.standard_normal((96, s.d)))
    if lscale is not None:                                   # Poisson rates must stay positive
        th = np.abs(th)
    ga, gb = a.logpost(th), b.logpost(th)
    assert np.array_equal(ga, gb, equal_nan=True)
    assert np.isfinite(gb).sum() > 48
    op = s.oracle(orc)
    for c in np.flatnonzero(np.isfinite(gb))[:6]:
        assert abs(gb[c] - op.logpost(th[c])) <= 1e-12 * op.abs_terms(th[c]) + 1e-5, (name, wpg, c)
    l0 = None if lscale is None else np.diag(lscale * np.abs(s.theta_star))
    for n_it in (60, 1 << 40):
        for e in (a, b):
            e.init_chains(th0)
            e.adaptive_begin(2500, 10.0, 1, l_matrix=l0)
            e.adaptive_advance(n_it)
        sa, sb = a.state(), b.state()
        for k in ("theta", "logpost", "best_logpost", "age", "length"):
            assert np.array_equal(sa[k], sb[k]), (name, wpg, n_it, k)
        assert np.array_equal(a.lmatrix(), b.lmatrix())
    a.close()
    b.close()
