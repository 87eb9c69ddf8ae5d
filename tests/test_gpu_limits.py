"""The limits of include/mhx.h and the degenerate inputs, through the ABI against the oracle:
d = MHX_MAX_PARAMS = 63 parameters shared by K = MHX_MAX_FUNCTIONS = 16 functions of ragged
lengths, a function gathering MHX_MAX_FN_PARAMS = 32 parameters, MHX_MAX_BOUNDS = 64 bounds, an
empty dataset, a one-point dataset."""
import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu
REL = 1e-12


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def big_global_fit(seed=3):
    """16 functions over 63 parameters: 15 polynomials of 4 coefficients sharing nothing, plus
    one of 3; lengths 1 ... 2600 (ragged, some shorter than a wavefront, some several tiles)"""
    rng = np.random.default_rng(seed)
    d, K = 63, 16
    th = rng.uniform(-1.0, 1.0, d)
    s = pb.Spec(d)
    at = 0
    lengths = [1, 7, 63, 64, 65, 130, 500, 1023, 1024, 1025, 1500, 2047, 2048, 2049, 2600, 333]
    for k in range(K):
        npar = 4 if k < 15 else 3
        idx = list(range(at, at + npar))
        at += npar
        n = lengths[k]
        x = np.sort(rng.uniform(-1.0, 1.0, n))
        sig = rng.uniform(0.05, 0.2, n)
        y = pb.model_eval_np(pb.POLY, (), th[idx], x) + sig * rng.standard_normal(n)
        bounds = (idx, th[idx] - 2.0, th[idx] + 2.0)
        s.add(pb.POLY, (), idx, x, y, sig, pb.NORMAL, bounds)
    assert at == d
    s.theta_star = th
    return s


def test_63_parameters_16_ragged_functions(mhx, orc):
    s = big_global_fit()
    op = s.oracle(orc)
    C_ = 3
    e = s.engine(mhx, C_, seed=12)
    th = pb.perturbed(s.theta_star, 6, 0.05, seed=2)
    th[3, 10] += 5.0  # out of its bounds
    got, parts = e.logpost(th, parts=True)
    for i in range(len(th)):
        ref, rp = op.logpost(th[i], parts=True)
        assert abs(parts[i, 0] - rp[0]) <= REL * op.abs_terms(th[i]), i
        assert abs(parts[i, 1] - rp[1]) <= 1e-5
    # the controller at d = 63: 63 x 63 covariance, Cholesky, 630-step settle window
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=5)
    e.init_chains(th0)
    n = 2600
    L0 = np.diag(np.full(s.d, 0.01))
    e.adaptive_begin(n, 10.0, 1, l_matrix=L0)
    e.adaptive_advance(1 << 40)
    st = e.state()
    status, loop_i = e.chain_status()
    assert (status == mhx.capi.CHAIN_DONE).all()
    same = 0
    for c in range(C_):
        w = orc.Walker(op, th0[c])
        w.adaptive_begin(n, 10.0, 1, seed=12, chain_id=c, l_matrix=L0)
        w.adaptive_advance(1 << 40)
        assert st["age"][c] == w.age and st["length"][c] == w.length
        same += int(np.array_equal(st["theta"][c], w.last()[0]))
    assert same >= C_ - 1  # accept tests can differ only inside the 1e-12 band
    e.close()


def test_function_with_32_parameters_and_64_bounds(mhx, orc):
    rng = np.random.default_rng(8)
    d = 32
    th = rng.uniform(-0.5, 0.5, d) / (1 + np.arange(d))      # a tame degree-31 polynomial on [-1, 1]
    x = np.sort(rng.uniform(-1, 1, 900))
    sig = np.full(900, 0.1)
    y = pb.model_eval_np(pb.POLY, (), th, x) + 0.1 * rng.standard_normal(900)
    s = pb.Spec(d)
    # 64 bounds: every parameter twice (the second block tighter, so both kinds of excursion occur)
    idx = list(range(d)) + list(range(d))
    lo = np.concatenate([th - 1.0, th - 0.05])
    hi = np.concatenate([th + 1.0, th + 0.05])
    s.add(pb.POLY, (), range(d), x, y, sig, pb.NORMAL, (idx, lo, hi))
    s.theta_star = th
    e = s.engine(mhx, 2)
    op = s.oracle(orc)
    t = pb.perturbed(th, 8, 0.0, seed=1) + rng.normal(0, 0.04, (8, d))
    got, parts = e.logpost(t, parts=True)
    for i in range(len(t)):
        ref, rp = op.logpost(t[i], parts=True)
        assert abs(parts[i, 0] - rp[0]) <= REL * op.abs_terms(t[i]), i
        nv = int(((t[i][idx] <= lo) | (t[i][idx] >= hi)).sum())
        assert abs(parts[i, 1] - rp[1]) <= 1e-5 * max(1, nv)
    e.close()


def test_empty_and_single_point_datasets(mhx, orc):
    """(reduce #'+ (mapcar ...)) over no points is 0 (M:400): an empty dataset contributes only
    its prior; one point is one term"""
    s = pb.Spec(4)
    s.add(pb.POLY, (), [0, 1], [], [], None, pb.NORMAL, ([0, 1], [-1, -1], [1, 1]))
    s.add(pb.POLY, (), [2, 3], [0.5], [1.25], [0.3], pb.NORMAL)
    s.theta_star = np.array([0.2, 0.3, 1.0, 0.4])
    e = s.engine(mhx, 2, seed=1)
    op = s.oracle(orc)
    th = np.array([[0.2, 0.3, 1.0, 0.4], [2.0, 0.3, 0.9, 0.5]])
    got, parts = e.logpost(th, parts=True)
    for i in range(2):
        ref, rp = op.logpost(th[i], parts=True)
        assert parts[i, 0] == pytest.approx(rp[0], rel=1e-15, abs=0) and abs(parts[i, 1] - rp[1]) <= 1e-5
    r = (1.25 - (1.0 + 0.4 * 0.5)) / 0.3
    assert parts[0, 0] == pytest.approx(-0.5 * np.log(2 * np.pi) - np.log(0.3) - 0.5 * r * r, rel=1e-15)
    e.init_chains(th)
    e.adaptive_begin(700, 10.0, 1, l_matrix=np.diag(np.full(4, 0.05)))
    e.adaptive_advance(1 << 40)
    assert (e.chain_status()[0] == mhx.capi.CHAIN_DONE).all()
    e.close()


def test_limits_are_enforced(mhx):
    capi = mhx.capi
    for bad in (dict(n_params=64), dict(n_functions=17), dict(n_params=0)):
        kw = dict(n_chains=1, n_params=2, n_functions=1)
        kw.update(bad)
        with pytest.raises(mhx.MhxError) as ei:
            mhx.Engine(kw["n_chains"], kw["n_params"], kw["n_functions"])
        assert ei.value.code == capi.EINVAL
    e = mhx.Engine(1, 40)
    with pytest.raises(mhx.MhxError):
        e.set_function(0, capi.MODEL_POLY, (), list(range(33)))       # > MHX_MAX_FN_PARAMS
    with pytest.raises(mhx.MhxError):
        e.set_bounds(0, list(range(40)) + list(range(25)), np.zeros(65), np.ones(65))  # > MHX_MAX_BOUNDS
    with pytest.raises(mhx.MhxError):
        e.set_function(0, capi.MODEL_POLY, (), [0, 40])               # index outside the vector
    e.close()
