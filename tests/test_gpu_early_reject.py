"""MHX_EARLY_REJECT=1: a sweep that stops where the growing sum of squares has already lost the
accept test (csrc/mhx_kernels.hpp, sweep(): "EXACT EARLY REJECTION").  Exact means: the walk is
the same walk, bit for bit - chains, histories, proposal factors - because a proposal that is left
early would have been rejected and nothing of its log-posterior is ever stored.  Asked for by the
environment, granted only to one function of a bounded enumerated model with the weighted normal
likelihood and no prior body, in a kernel compiled at run time."""
import os

import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def pair(mhx, spec, chains, **kw):
    out = []
    old = {k: os.environ.get(k) for k in ("MHX_EARLY_REJECT", "MHX_SPLIT")}
    os.environ["MHX_SPLIT"] = "0"  # (the batch kernels: where the early rejection lives)
    try:
        for flag in ("1", None):
            if flag:
                os.environ["MHX_EARLY_REJECT"] = flag
            else:
                os.environ.pop("MHX_EARLY_REJECT", None)
            e = spec.engine(mhx, chains, **kw)
            out.append((e, e.kernel_name()))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    return out


@pytest.mark.parametrize("n,chains,wpg", [(30000, 37, "8"), (100000, 64, "16"), (9001, 20, "8")])
def test_early_rejection_leaves_every_bit_alone(mhx, n, chains, wpg):
    s = pb.two_peak(n=n, seed=7)
    os.environ["MHX_FAMILY_WPG"] = wpg
    try:
        (a, na), (b, nb) = pair(mhx, s, chains, seed=11)
    finally:
        os.environ.pop("MHX_FAMILY_WPG", None)
    assert "+early-reject" in na and "rtc[" in na and "early-reject" not in nb, (na, nb)
    th0 = pb.perturbed(s.theta_star, chains, 0.01, seed=2)
    # the default start - diag(theta) proposals at T = 10: nothing is accepted for dozens of
    # iterations, every sweep is left early -, across the first adaptation ticks into the settled
    # walk, where nothing is left early; then a complete short run
    for e in (a, b):
        e.init_chains(th0)
        e.adaptive_begin(30000, 10.0, 1)
    for portion in (25, 400, 600):
        for e in (a, b):
            e.adaptive_advance(portion)
        sa, sb = a.state(), b.state()
        for k in ("theta", "logpost", "best_theta", "best_logpost", "age", "length"):
            assert np.array_equal(sa[k], sb[k]), (n, portion, k)
        assert np.array_equal(a.lmatrix(), b.lmatrix())
        assert np.array_equal(a.chain_status()[0], b.chain_status()[0])
    assert (a.chain_status()[0] != mhx.capi.CHAIN_FP_TRAP).all()
    for e in (a, b):
        e.init_chains(th0)
        e.adaptive_begin(1500, 10.0, 1)
        e.adaptive_advance(1 << 40)
    sa, sb = a.state(), b.state()
    for k in ("theta", "logpost", "best_theta", "best_logpost", "age", "length"):
        assert np.array_equal(sa[k], sb[k]), (n, "run", k)
    a.close()
    b.close()


def test_early_rejection_is_only_granted_where_it_is_exact(mhx):
    """not to a Poisson likelihood (its terms have both signs), not to a global fit (the threshold
    is on one sum), not to an expression (a later window may overflow where the reference traps)"""
    old = os.environ.get("MHX_EARLY_REJECT")
    os.environ["MHX_EARLY_REJECT"] = "1"
    try:
        for spec in (pb.poisson_peaks(n=9000, seed=4), pb.global_fit(n_each=3000, n_sets=2, seed=5)):
            e = spec.engine(mhx, 16)
            assert "early-reject" not in e.kernel_name(), e.kernel_name()
            e.close()
        x = np.linspace(0, 4, 6000)
        y = 2.0 * np.exp(-x / 1.5) + 0.3
        w = mhx.walker_create(function=mhx.models.lisp(
            "(lambda (x &key a tau c &allow-other-keys) (+ c (* a (exp (/ (- x) tau)))))"),
            data=[x, y], params=[":a", 1.8, ":tau", 1.4, ":c", 0.35], data_error=0.1, n_chains=4, seed=2)
        assert "early-reject" not in w.engine.kernel_name(), w.engine.kernel_name()
    finally:
        os.environ.pop("MHX_EARLY_REJECT", None)
        if old is not None:
            os.environ["MHX_EARLY_REJECT"] = old
