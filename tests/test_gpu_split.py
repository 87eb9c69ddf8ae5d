"""Split mode (csrc/mhx_kernels.hpp): with few chains and long datasets one chain's likelihood
sums are spread over many workgroups (two small launches per iteration).  It must walk like the
batch kernels: same proposals, same controller, log-posteriors equal to rounding - and be
deterministic."""
import os

import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu
REL = 1e-12


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def engine(mhx, spec, chains, split, **kw):
    old = os.environ.get("MHX_SPLIT")
    if split is None:
        os.environ.pop("MHX_SPLIT", None)
    else:
        os.environ["MHX_SPLIT"] = str(split)
    try:
        e = spec.engine(mhx, chains, **kw)
        name = e.kernel_name()  # finalises under this setting
    finally:
        if old is None:
            os.environ.pop("MHX_SPLIT", None)
        else:
            os.environ["MHX_SPLIT"] = old
    return e, name


def walk(e, th0, n, l0=None, plain=False):
    e.init_chains(th0)
    if plain:
        e.many_steps(n, l0)
    else:
        e.adaptive_begin(n, 10.0, 1, l_matrix=l0)
        e.adaptive_advance(1 << 40)
    return e.state(), e.chain_status()[0], e.lmatrix()


CASES = [
    ("two_peak", lambda: pb.two_peak(n=30000, seed=3), None),
    ("poisson", lambda: pb.poisson_peaks(n=24000, seed=4), 0.002),
    ("global_fit", lambda: pb.global_fit(n_each=20000, n_sets=3, seed=5), None),
]


@pytest.mark.parametrize("name,make,lscale", CASES, ids=[c[0] for c in CASES])
def test_split_walks_like_the_batch_kernels(mhx, orc, name, make, lscale):
    s = make()
    C_, n = 3, 1300
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=2)
    l0 = None if lscale is None else np.diag(lscale * np.abs(s.theta_star))
    batch, nb = engine(mhx, s, C_, 0, seed=9)
    split, ns = engine(mhx, s, C_, None, seed=9)
    assert "split" not in nb and "split x" in ns, (nb, ns)
    sb, stb, Lb = walk(batch, th0, n, l0)
    ss, sts, Ls = walk(split, th0, n, l0)
    assert np.array_equal(stb, sts) and (sts == mhx.capi.CHAIN_DONE).all()
    assert np.array_equal(sb["age"], ss["age"]) and np.array_equal(sb["length"], ss["length"])
    same = sum(int(np.array_equal(sb["theta"][c], ss["theta"][c])) for c in range(C_))
    assert same >= C_ - 1      # an accept test can differ only inside the rounding band
    op = s.oracle(orc)
    for c in range(C_):
        ref = op.logpost(ss["theta"][c])
        assert abs(ss["logpost"][c] - ref) <= REL * op.abs_terms(ss["theta"][c]) + 1e-5
    # deterministic: the partial sums are added in slot order
    split2, _ = engine(mhx, s, C_, None, seed=9)
    s2, _, L2 = walk(split2, th0, n, l0)
    for k in ("theta", "logpost", "age"):
        assert np.array_equal(ss[k], s2[k]), k
    assert np.array_equal(Ls, L2)
    for e in (batch, split, split2):
        e.close()


def test_split_single_walker_expression_and_many_steps(mhx):
    rng = np.random.default_rng(6)
    n = 60000
    x = np.linspace(0, 4, n)
    sig = rng.uniform(0.05, 0.2, n)
    y = 2.0 * np.exp(-x / 1.5) + 0.3 + sig * rng.standard_normal(n)
    text = "(lambda (x &key a tau c &allow-other-keys) (+ c (* a (exp (/ (- x) tau)))))"
    params = [":a", 1.8, ":tau", 1.4, ":c", 0.35]
    ws = []
    for split in ("0", None):
        if split is None:
            os.environ.pop("MHX_SPLIT", None)
        else:
            os.environ["MHX_SPLIT"] = split
        try:
            w = mhx.walker_create(function=mhx.models.lisp(text), data=[x, y], params=params,
                                  data_error=sig, seed=2)
            w.engine.kernel_name()
        finally:
            os.environ.pop("MHX_SPLIT", None)
        ws.append(w)
    assert "split x" in ws[1].engine.kernel_name() and "split" not in ws[0].engine.kernel_name()
    p0, p1 = ws[0].last_step().prob, ws[1].last_step().prob
    assert p0 == p1                                      # init uses the batch kernel in both
    L = np.diag([0.01, 0.01, 0.005])
    for w in ws:
        mhx.walker_many_steps(w, 400, L)
    a, b = ws[0].engine.state(), ws[1].engine.state()
    assert a["age"][0] == b["age"][0] == 401
    assert abs(a["logpost"][0] - b["logpost"][0]) <= 1e-9 * abs(a["logpost"][0])
    for w in ws:
        mhx.walker_adaptive_steps_full(w, n=1500, temperature=10, auto=":prob-settle", l_matrix=L)
    ml = [mhx.walker_get(w, get=":most-likely-params") for w in ws]
    for k in ("a", "tau", "c"):
        assert abs(ml[0][k] - ml[1][k]) <= 0.02 * abs(ml[0][k])
    assert abs(ml[1]["tau"] - 1.5) < 0.1


def test_split_mode_is_chosen_by_batch_and_dataset_size(mhx):
    s_long = pb.two_peak(n=100000, seed=1)
    s_short = pb.two_peak(n=3000, seed=1)
    for spec, chains, want in ((s_long, 1, "split x24"), (s_long, 256, "split x4"),
                               (s_long, 1024, "split x2"), (s_long, 2048, None),
                               (s_short, 1, None), (s_short, 64, None)):
        e, name = engine(mhx, spec, chains, None)
        assert (want in name) if want else ("split" not in name), (chains, name)
        e.close()
    # the pooled-covariance mode (multi-GPU bench) stays on the batch kernels
    e, name = engine(mhx, s_long, 8, None, adapt_mode=mhx.capi.ADAPT_POOLED)
    assert "split" not in name
    e.close()
