"""walker-get's post-processing selectors (mcmc-fitting.lisp:487-543) served from the device
trace - the ones no other test reads: :unique-steps (M:492-496), :param (M:509),
:stddev-params (M:525-539), :covariance-matrix (M:541) - against the oracle's walker on the same
walk (same Philox stream, so the histories are identical) and its KAT-pinned
lplist-covariance / l-matrix."""
import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def _orc_cov(orc, v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros((v.shape[1], v.shape[1]))
    assert orc.lib().orc_lplist_covariance(v.ctypes.data_as(orc.f64p), v.shape[0], v.shape[1],
                                           out.ctypes.data_as(orc.f64p)) == 0
    return out


def test_unique_steps_param_covariance_stddev_from_the_device_trace(mhx, orc):
    s = pb.two_peak(n=700, seed=12)
    op = s.oracle(orc)
    keys = ["b0", "b1", "a1", "mu1", "w1", "a2", "mu2", "w2"]
    params = []
    th0 = s.theta_star * (1.0 + 0.01 * np.random.default_rng(5).standard_normal(8))
    for k, v in zip(keys, th0):
        params += [":" + k, float(v)]
    x, y, sig, _ = s.data[0]
    idx, lo, hi = s.bounds[0]
    w = mhx.walker_create(function=mhx.models.gauss_peaks(keys[:2], [tuple(keys[2:5]), tuple(keys[5:8])]), data=[x, y], params=params,
                          data_error=sig,
                          log_prior=mhx.prior_bounds({keys[i]: (lo[i], hi[i]) for i in idx}),
                          seed=41, history_capacity=4096)
    ow = orc.Walker(op, th0, mirror=True)  # the kernels' own arithmetic: probs equal to the bit
    assert w.last_step().prob == ow.last()[1]
    # a young walker: :stddev-params is all zeros below 10 steps (M:528-529)
    assert mhx.walker_get(w, get=":stddev-params") == {k: 0.0 for k in keys}
    mhx.walker_adaptive_steps(w, 3000)
    ow.adaptive_begin(3000, 10.0, 1, seed=41, chain_id=0)
    ow.adaptive_advance(1 << 40)
    assert w.age() == ow.age and w.length() == ow.length
    for take in (None, 1, 2, 57, 900, 3000):
        t = ow.length if take is None else min(take, ow.length)
        oprob, oth = ow.trace(t)
        bits = oprob.view(np.uint64)
        keep = [i for i in range(t) if i + 1 >= t or bits[i] != bits[i + 1]]
        # :unique-steps (M:492-496): params of every step whose prob differs from the next older
        # one's (`equal` on doubles), the oldest of the window always kept
        uniq = mhx.walker_get(w, get=":unique-steps", take=take)
        assert len(uniq) == len(keep)
        got = np.array([[p[k] for k in keys] for p in uniq])
        assert np.array_equal(got, oth[keep]), take
        assert mhx.walker_get(w, get=":log-liklihoods", take=take) == oprob.tolist()
        # :param (M:509)
        for j in (0, 3, 7):
            col = mhx.walker_get(w, get=":param", take=take, param=":" + keys[j])
            assert np.array_equal(np.array(col), oth[:, j]), (take, j)
        # :covariance-matrix (M:541) = lplist-covariance of the unique steps
        if len(keep) >= 1:
            cov = mhx.walker_get(w, get=":covariance-matrix", take=take)
            assert np.array_equal(cov, _orc_cov(orc, oth[keep])), take
    # :stddev-params (M:525-539) = diagonal of (walker-get :l-matrix :take take)
    for take in (500, 1000, None):
        st, L, nf = ow.l_matrix(ow.length if take is None else take)
        assert st == 0 and nf > 10
        sd = mhx.walker_get(w, get=":stddev-params", take=take)
        assert [sd[k] for k in keys] == [L[j, j] for j in range(8)], take
    assert len(uniq) < ow.length  # (the walk did reject proposals: the selectors had work to do)


def test_a_window_beyond_the_history_ring_is_signalled(mhx):
    """the reference keeps every step (M:549); the device ring keeps the newest history_capacity
    (default 1024): asking for more returns what is there AND warns"""
    import warnings
    from lisp_mcmc_amd.walker import HistoryTruncated
    lf_x, lf_y = [-4, -1, 2, 5, 10], [0, 2, 5, 9, 13]
    w = mhx.walker_create(function=mhx.models.line("b", "m"), data=[lf_x, lf_y],
                          params=[":b", -1, ":m", 2], data_error=0.2, seed=3)
    mhx.walker_many_steps(w, 1500, l_matrix=np.diag([0.05, 0.02]))
    assert w.length() == 1501
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert len(mhx.walker_get(w, get=":steps", take=1024)) == 1024   # inside the ring: silent
    with pytest.warns(HistoryTruncated):
        steps = mhx.walker_get(w, get=":steps")                           # the whole walk: 1501
    assert len(steps) == 1024
    with pytest.warns(HistoryTruncated):
        mhx.walker_get(w, get=":median-params", take=1200)
