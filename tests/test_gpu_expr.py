"""The model-expression path on the GPU: reference-style closures and
prior-bounds-let bodies given as Lisp text, translated (sexpr.py), compiled with hiprtc into the
same fused kernels, and checked against the oracle / a test-side evaluation of the forms."""
import numpy as np
import pytest

import problems as pb
import sexpr_eval

pytestmark = pytest.mark.gpu
REL = 1e-12

TWO_PEAK = ("(lambda (x &key b0 b1 a1 mu1 w1 a2 mu2 w2 &allow-other-keys)"
            " (+ (+ b0 (* b1 x))"
            "    (* a1 (exp (- (expt (/ (- x mu1) w1) 2))))"
            "    (* a2 (exp (- (expt (/ (- x mu2) w2) 2))))))")
KEYS = ["b0", "b1", "a1", "mu1", "w1", "a2", "mu2", "w2"]


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def test_line_fit_from_lisp_text(mhx, golden):
    """mcmc-fitting.lisp:1186 with its own lambda"""
    lf = golden["line_fit"]
    ref = golden["line_fit_initial_logpost_sigma_single"]
    w = mhx.walker_create(function=mhx.models.lisp("(lambda (x &key m b &allow-other-keys) (+ b (* m x)))"),
                          data=[lf["x"], lf["y"]], params=[":b", -1, ":m", 2],
                          data_error=ref["sigma"], seed=1)
    assert w.last_step().prob == pytest.approx(ref["value"], rel=1e-14)
    mhx.walker_adaptive_steps(w, 3000)
    ml = mhx.walker_get(w, get=":most-likely-params")
    A = np.vstack([np.ones(5), lf["x"]]).T
    bm = np.linalg.lstsq(A, np.array(lf["y"], float), rcond=None)[0]
    assert abs(ml["b"] - bm[0]) < 0.2 and abs(ml["m"] - bm[1]) < 0.05


def test_single_item_parameter_styles(mhx, golden):
    """mcmc-fitting.lisp:1189-1195: one key holding a list / vector / d x 1 array, read with elt /
    aref inside the closure"""
    lf = golden["line_fit"]
    ref = golden["line_fit_initial_logpost_sigma_single"]
    for text, value in (
            ("(lambda (x &key params &allow-other-keys) (+ (elt params 0) (* (elt params 1) x)))", [-1, 2]),
            ("(lambda (x &key params &allow-other-keys) (+ (aref params 0 0) (* (aref params 1 0) x)))",
             np.array([[-1.0], [2.0]]))):
        w = mhx.walker_create(function=mhx.models.lisp(text), data=[lf["x"], lf["y"]],
                              params=[":params", value], data_error=ref["sigma"])
        assert w.param_keys == ["params_0", "params_1"]
        assert w.last_step().prob == pytest.approx(ref["value"], rel=1e-14)


def build_two_peak(mhx, s, C_, body=None, seed=0):
    keys, cexpr = __import__("lisp_mcmc_amd").sexpr.lambda_to_expr(TWO_PEAK)
    e = mhx.Engine(C_, 8, 1, seed=seed)
    e.set_function_expr(0, cexpr, keys, [KEYS.index(k) for k in keys])
    x, y, sig, lik = s.data[0]
    e.set_dataset(0, x, y, sig, lik)
    idx, lo, hi = s.bounds[0]
    e.set_bounds(0, idx, lo, hi)
    if body:
        e.set_prior_expr(0, __import__("lisp_mcmc_amd").sexpr.prior_body_to_expr(body), KEYS, range(8))
    return e


def test_two_peak_expression_vs_oracle(mhx, orc):
    s = pb.two_peak(n=3000, seed=71)
    op = s.oracle(orc)
    e = build_two_peak(mhx, s, 1)
    th = pb.perturbed(s.theta_star, 12, 0.02)
    th[2] = s.theta_star * 1.6
    got, parts = e.logpost(th, parts=True)
    for i in range(len(th)):
        ref, rp = op.logpost(th[i], parts=True)
        nv = int(((th[i] <= s.bounds[0][1]) | (th[i] >= s.bounds[0][2])).sum())
        assert abs(parts[i, 0] - rp[0]) <= REL * op.abs_terms(th[i]), i
        assert abs(parts[i, 1] - rp[1]) <= 1e-5 * max(1, nv)
    e.close()


def test_two_peak_expression_walk_vs_oracle(mhx, orc):
    s = pb.two_peak(n=500, seed=72)
    op = s.oracle(orc)
    C_, n = 6, 1500
    e = build_two_peak(mhx, s, C_, seed=13)
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=13)
    e.init_chains(th0)
    e.adaptive_begin(n, 10.0, 1)
    e.adaptive_advance(1 << 40)
    st = e.state()
    same = 0
    for c in range(C_):
        w = orc.Walker(op, th0[c])
        w.adaptive_begin(n, 10.0, 1, seed=13, chain_id=c)
        w.adaptive_advance(1 << 40)
        assert st["age"][c] == w.age
        same += int(np.array_equal(st["theta"][c], w.last()[0]))
    # the expression uses ocml exp/pow instead of the oracle's libm: logposts agree to ~1e-13
    # relative, so whole trajectories coincide unless an accept test falls inside that band
    assert same >= C_ - 1
    e.close()


def test_prior_body_with_cross_parameter_terms(mhx):
    """nv-specific.lisp:25-34 style: bounds-total plus constraints between parameters"""
    body = "(+ bounds-total (if (> mu1 mu2) -1e9 0e0) (if (< (- mu2 mu1) 0.3) -1e9 0e0) (if (not (< 1.1 (/ a1 a2) 1.8)) -1e9 0e0))"
    s = pb.two_peak(n=200, seed=73)
    e0 = build_two_peak(mhx, s, 1)
    e1 = build_two_peak(mhx, s, 1, body=body)
    from lisp_mcmc_amd import sexpr
    form = sexpr.parse(body)
    th = pb.perturbed(s.theta_star, 10, 0.05, seed=5)
    th[1, 3], th[1, 6] = 0.7, 0.3      # mu1 > mu2
    th[2, 6] = th[2, 3] + 0.1          # too close
    th[3, 2] = 2.0 * th[3, 5]          # amplitude ratio off
    base, pbase = e0.logpost(th, parts=True)
    got, pgot = e1.logpost(th, parts=True)
    assert np.array_equal(pbase[:, 0], pgot[:, 0])          # the likelihood is untouched
    for i in range(len(th)):
        env = dict(zip(KEYS, th[i]))
        env["bounds-total"] = pbase[i, 1]
        assert pgot[i, 1] == sexpr_eval.evaluate(form, env), i
    assert (pgot[1:4, 1] <= -1e9).all() and pgot[0, 1] == pbase[0, 1]
    e0.close()
    e1.close()


def test_prior_body_key_bound_variables(mhx, orc):
    """the docstring example of prior-bounds-let (mcmc-fitting.lisp:348):
    (+ bounds-total (* 4 x-bound)) - each key also binds <key>-bound"""
    s = pb.two_peak(n=100, seed=74)
    e = build_two_peak(mhx, s, 1, body="(+ bounds-total (* 4 mu1-bound) w2-bound)")
    idx, lo, hi = s.bounds[0]
    th = pb.perturbed(s.theta_star, 6, 0.01, seed=7)
    th[1, 3] = hi[3] + 0.2          # mu1 above its upper bound
    th[2, 7] = lo[7] - 0.01         # w2 below its lower bound
    th[3, 0] = hi[0] + 1.0          # b0 out: only bounds-total sees it
    got, parts = e.logpost(th, parts=True)
    pen = lambda v, k: orc.lib().orc_bound_penalty(float(v), float(lo[k]), float(hi[k]))  # noqa
    for i in range(len(th)):
        total = 0.0
        for k in range(8):
            b = pen(th[i, k], k)
            total = b if k == 0 else total + b
        want = (total + 4.0 * pen(th[i, 3], 3)) + pen(th[i, 7], 7)
        assert parts[i, 1] == pytest.approx(want, rel=1e-9, abs=1e-5), i
    assert parts[0, 1] == 0.0 and (parts[1:4, 1] < 0.0).all()
    e.close()


def test_expression_errors(mhx):
    e = mhx.Engine(1, 2)
    with pytest.raises(mhx.MhxError) as ei:
        e.set_function_expr(0, "b + m*x + q", ["b", "m"], [0, 1])
    assert ei.value.code == mhx.capi.EINVAL and "unknown identifier 'q'" in str(ei.value)
    with pytest.raises(mhx.MhxError):
        e.set_function_expr(0, "b + m*x; system(0)", ["b", "m"], [0, 1])
    with pytest.raises(mhx.MhxError):
        e.set_function_expr(0, "b + (m*x", ["b", "m"], [0, 1])
    e.set_function_expr(0, "b + m*x + 1/2", ["b", "m"], [0, 1])     # 1/2 must mean 0.5
    e.set_dataset(0, [0.0, 1.0], [0.5, 3.5], 1.0)
    v = e.logpost([[0.0, 3.0]])[0]
    assert v == pytest.approx(2 * (-0.5 * np.log(2 * np.pi)), abs=1e-15)
    e.close()
