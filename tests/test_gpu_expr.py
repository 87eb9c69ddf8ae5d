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
    for as_written in (False, True):   # recognised as the polynomial model / compiled as written
        w = mhx.walker_create(function=mhx.models.lisp("(lambda (x &key m b &allow-other-keys) (+ b (* m x)))",
                                                       as_written=as_written),
                              data=[lf["x"], lf["y"]], params=[":b", -1, ":m", 2],
                              data_error=ref["sigma"], seed=1)
        assert ("rtc[expr" in w.engine.kernel_name()) == as_written, w.engine.kernel_name()
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
    e.set_expr_recognition(False)     # these tests are about the text compiled as written
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


# ---- create-log-liklihood-function (M:402-416): the per-point likelihood as a closure --------
def _lik_walker(mhx, lik_text, model_text, x, y, sig, params, n_chains=1, seed=3):
    return mhx.walker_create(function=mhx.models.lisp(model_text), data=[x, y], params=params,
                             data_error=sig, n_chains=n_chains, seed=seed,
                             log_liklihood=mhx.create_log_liklihood_function(lik_text))


def test_custom_likelihood_equals_builtin_normal(mhx):
    """(log-normal y model error) through create-log-liklihood-function is log-liklihood-normal"""
    rng = np.random.default_rng(5)
    n = 2500  # three tiles, the last one ragged
    x = np.linspace(-2, 3, n)
    sig = rng.uniform(0.1, 0.4, n)
    y = 0.7 - 0.4 * x + 0.2 * x * x + sig * rng.standard_normal(n)
    model = "(lambda (x &key a b c &allow-other-keys) (+ a (* b x) (* c (expt x 2))))"
    params = [":a", 0.5, ":b", -0.3, ":c", 0.25]
    w0 = mhx.walker_create(function=mhx.models.lisp(model), data=[x, y], params=params, data_error=sig)
    w1 = _lik_walker(mhx, "(lambda (y model error) (log-normal y model error))", model, x, y, sig, params)
    m = 0.5 - 0.3 * x + 0.25 * x * x
    terms = -0.5 * np.log(2 * np.pi) - np.log(sig) - 0.5 * ((y - m) / sig) ** 2
    ref = float(np.sum(terms))
    tol = REL * float(np.sum(np.abs(terms)))
    assert abs(w1.last_step().prob - ref) <= tol
    assert abs(w1.last_step().prob - w0.last_step().prob) <= tol


def test_custom_likelihood_poisson_closure(mhx):
    """the README-style Poisson closure that ignores `error`; its sum differs from log-poisson's
    only by the parameter-independent sum of log k!"""
    rng = np.random.default_rng(6)
    n = 1500
    x = np.linspace(0, 1, n)
    lam = 30 + 80 * np.exp(-((x - 0.4) / 0.1) ** 2)
    y = rng.poisson(lam).astype(float)
    model = "(lambda (x &key bg a mu w &allow-other-keys) (+ bg (* a (exp (- (expt (/ (- x mu) w) 2))))))"
    params = [":bg", 28.0, ":a", 85.0, ":mu", 0.41, ":w", 0.105]
    lik = "(lambda (y model error) (declare (ignore error)) (- (* y (log model)) model))"
    w = _lik_walker(mhx, lik, model, x, y, None, params, n_chains=4)
    m = 28.0 + 85.0 * np.exp(-((x - 0.41) / 0.105) ** 2)
    terms = y * np.log(m) - m
    assert abs(w.last_step().prob - float(terms.sum())) <= REL * float(np.abs(terms).sum())
    wp = mhx.walker_create(function=mhx.models.lisp(model), data=[x, y], params=params,
                           log_liklihood="poisson", n_chains=4, seed=3)
    # same seed, same proposals; the two posteriors differ by a constant, so the walks coincide
    # a Poisson rate must stay positive ((log model) of a negative model is an error in the
    # reference as well): start from a small :l-matrix rather than diag(theta)
    l0 = np.diag(0.002 * np.array(params[1::2]))
    for ww in (w, wp):
        mhx.walker_adaptive_steps_full(ww, n=1200, temperature=10, auto=":prob-settle", l_matrix=l0)
    a = np.array(list(mhx.walker_get(w, get=":most-likely-params").values()))
    b = np.array(list(mhx.walker_get(wp, get=":most-likely-params").values()))
    assert np.allclose(a, b, rtol=1e-9)
    assert abs(a[2] - 0.4) < 0.02 and abs(a[1] - 80) < 15


def test_custom_likelihood_uses_error_and_branches(mhx):
    """a robust (Huber-like) term: error is the point's sigma, `if` picks the branch per point"""
    rng = np.random.default_rng(7)
    n = 700
    x = np.linspace(0, 5, n)
    sig = rng.uniform(0.2, 0.5, n)
    y = 1.0 + 2.0 * x + sig * rng.standard_normal(n)
    y[::50] += 25.0  # outliers
    model = "(lambda (x &key m b &allow-other-keys) (+ b (* m x)))"
    lik = ("(lambda (y model error) (if (< (abs (/ (- y model) error)) 2)"
           " (* -1/2 (expt (/ (- y model) error) 2))"
           " (- 2 (* 2 (abs (/ (- y model) error))))))")
    w = _lik_walker(mhx, lik, model, x, y, sig, [":b", 0.8, ":m", 2.1])
    r = np.abs((y - (0.8 + 2.1 * x)) / sig)
    terms = np.where(r < 2, -0.5 * r * r, 2 - 2 * r)
    assert abs(w.last_step().prob - float(terms.sum())) <= REL * float(np.abs(terms).sum())
    mhx.walker_adaptive_steps(w, 4000)
    ml = mhx.walker_get(w, get=":most-likely-params")
    assert abs(ml["b"] - 1.0) < 0.15 and abs(ml["m"] - 2.0) < 0.05  # the outliers do not pull it


def test_custom_likelihood_errors(mhx):
    x, y = [0.0, 1.0, 2.0], [1.0, 2.0, 3.0]
    lik = mhx.create_log_liklihood_function("(lambda (y model error) (- (abs (- y model))))")
    with pytest.raises(mhx.MhxError, match="expression model"):
        mhx.walker_create(function=mhx.models.line("b", "m"), data=[x, y], params=[":b", 0, ":m", 1],
                          log_liklihood=lik)
    with pytest.raises(mhx.MhxError, match="unknown identifier"):
        mhx.walker_create(function=mhx.models.lisp("(lambda (x &key m b &allow-other-keys) (+ b (* m x)))"),
                          data=[x, y], params=[":b", 0, ":m", 1],
                          log_liklihood=mhx.create_log_liklihood_function(
                              "(lambda (y model error) (* scale (- y model)))"))


def test_recognised_peak_closure_runs_on_the_peak_kernels(mhx, orc):
    """the recognition lives BELOW the C ABI (mhx_set_function_expr,
    csrc/mhx_expr.cpp).  models.lisp() only translates the closure's text - exactly what the Lisp
    shim's expr-model does - and libmhx serves background + Gaussian peaks with the enumerated
    model's kernel: same kernel name, same bits as models.gauss_peaks() over a 1200-step walk.
    as_written=True (mhx_set_expr_recognition) compiles the text and agrees within the stated
    tolerance; a body that is not of the shape is compiled as written whatever the switch says."""
    s = pb.two_peak(n=5000, seed=91)
    x, y, sig, _ = s.data[0]
    params = [":b0", 0.5, ":b1", 0.3, ":a1", 1.0, ":mu1", 0.3, ":w1", 0.05, ":a2", 0.7, ":mu2", 0.7, ":w2", 0.08]
    cross = TWO_PEAK.replace("(* a1 (exp", "(* a1 (+ 1 (* 0 b1)) (exp")           # a cross term
    cube = TWO_PEAK.replace("(expt (/ (- x mu2) w2) 2)", "(expt (/ (- x mu2) w2) 4)")   # not a Gaussian
    ws = [mhx.walker_create(function=f, data=[x, y], params=params, data_error=sig, n_chains=3, seed=5)
          for f in (mhx.models.lisp(TWO_PEAK, recognise=False),     # (the keyword of rounds 2-3: ignored)
                    mhx.models.gauss_peaks(["b0", "b1"], [("a1", "mu1", "w1"), ("a2", "mu2", "w2")]),
                    mhx.models.lisp(TWO_PEAK, as_written=True),
                    mhx.models.lisp(cross), mhx.models.lisp(cube))]
    names = [w.engine.kernel_name() for w in ws]
    assert names[0] == names[1] and "gauss22_normal" in names[0], names
    assert all("rtc[expr" in n for n in names[2:]), names
    p = [w.last_step().prob for w in ws]
    assert p[0] == p[1]
    op = s.oracle(orc)
    assert abs(p[2] - p[0]) <= 2 * REL * op.abs_terms(s.theta_star)
    assert abs(p[3] - p[0]) <= 2 * REL * op.abs_terms(s.theta_star)
    assert p[4] != p[0]
    for w in ws[:2]:
        mhx.walker_adaptive_steps(w, 1200)
    a, b = ws[0].engine.state(), ws[1].engine.state()
    for k in ("theta", "logpost", "best_theta", "age", "length"):
        assert np.array_equal(a[k], b[k]), k
    # an expression likelihood keeps its function an expression (it reads `model` from one)
    w = mhx.walker_create(function=mhx.models.lisp(TWO_PEAK), data=[x, y], params=params, data_error=sig,
                          log_liklihood=mhx.create_log_liklihood_function(
                              "(lambda (y model error) (* -1/2 (expt (/ (- y model) error) 2)))"),
                          n_chains=2, seed=5)
    assert "rtc[expr:expr" in w.engine.kernel_name(), w.engine.kernel_name()
    # a permuted key order in the plist: the gather map is permuted into the model's order
    perm = [":w2", 0.08, ":mu2", 0.7, ":a2", 0.7, ":b1", 0.3, ":w1", 0.05, ":mu1", 0.3, ":a1", 1.0, ":b0", 0.5]
    wp = mhx.walker_create(function=mhx.models.lisp(TWO_PEAK), data=[x, y], params=perm, data_error=sig,
                           n_chains=3, seed=5)
    assert "gauss22_normal" in wp.engine.kernel_name() and wp.last_step().prob == p[0]


def test_vector_valued_x_plane_fit(mhx):
    """"multiple or linked independent variables" (mcmc-fitting.lisp:1136-1137):
    each element of the x list is a vector and the closure reads (elt x 0), (elt x 1).  A plane
    z = a + b x0 + c x1 + d x0 x1 with per-point sigma through mhx_set_dataset_cols (the second
    column rides in the tiles' fourth array): log-posteriors against numpy within the stated
    tolerance, batch kernels and split mode, and a walk that recovers the plane; the weighted
    Poisson form with a custom likelihood; what is refused."""
    rng = np.random.default_rng(21)
    n = 5000
    X = np.column_stack([rng.uniform(-1, 2, n), rng.uniform(0, 3, n)])
    tstar = np.array([0.4, 1.3, -0.7, 0.25])
    sig = rng.uniform(0.05, 0.2, n)

    def plane(t, X):
        return t[0] + t[1] * X[:, 0] + t[2] * X[:, 1] + t[3] * X[:, 0] * X[:, 1]
    z = plane(tstar, X) + sig * rng.standard_normal(n)
    text = ("(lambda (x &key a b c d &allow-other-keys)"
            " (+ a (* b (elt x 0)) (* c (elt x 1)) (* d (elt x 0) (aref x 1))))")
    keys, cexpr = mhx.sexpr.lambda_to_expr(text)
    assert keys == ["a", "b", "c", "d"] and "xcol0" in cexpr and "xcol1" in cexpr
    params = [":a", 0.3, ":b", 1.0, ":c", -0.5, ":d", 0.2]
    w = mhx.walker_create(function=mhx.models.lisp(text), data=[X, z], params=params, data_error=sig,
                          n_chains=5, seed=4)
    assert "rtc[expr" in w.engine.kernel_name()
    th = np.array([[0.3, 1.0, -0.5, 0.2], tstar, tstar * 1.1, [0, 0, 0, 0], [1, -1, 2, 0.5]])
    got = w.engine.logpost(th)
    for i, t in enumerate(th):
        terms = -0.5 * np.log(2 * np.pi) - np.log(sig) - 0.5 * ((z - plane(t, X)) / sig) ** 2
        assert abs(got[i] - terms.sum()) <= REL * np.abs(terms).sum(), i
    mhx.walker_adaptive_steps(w, 4000)
    ml = mhx.walker_get(w, get=":most-likely-params")
    assert max(abs(ml[k] - v) for k, v in zip("abcd", tstar)) < 0.05, ml
    # split mode (a single walker on a long dataset): the same sums to rounding
    n2 = 60000
    X2 = np.column_stack([rng.uniform(-1, 2, n2), rng.uniform(0, 3, n2)])
    s2 = rng.uniform(0.05, 0.2, n2)
    z2 = plane(tstar, X2) + s2 * rng.standard_normal(n2)
    import os
    ws = []
    for split in ("0", None):
        if split is None:
            os.environ.pop("MHX_SPLIT", None)
        else:
            os.environ["MHX_SPLIT"] = split
        try:
            ws.append(mhx.walker_create(function=mhx.models.lisp(text), data=[X2, z2], params=params,
                                        data_error=s2, seed=4))
            ws[-1].engine.kernel_name()
        finally:
            os.environ["MHX_SPLIT"] = "0"
    assert "split x" in ws[1].engine.kernel_name() and "split" not in ws[0].engine.kernel_name()
    for w2 in ws:
        mhx.walker_adaptive_steps_full(w2, n=300, temperature=10, auto=":prob-settle",
                                       l_matrix=np.diag([0.01] * 4))
    a, b = ws[0].engine.state(), ws[1].engine.state()
    assert a["age"][0] == b["age"][0]
    t2 = b["theta"][0]
    terms = -0.5 * np.log(2 * np.pi) - np.log(s2) - 0.5 * ((z2 - plane(t2, X2)) / s2) ** 2
    assert abs(b["logpost"][0] - terms.sum()) <= REL * np.abs(terms).sum()
    # a Poisson rate over two variables with the closure likelihood of the README
    lam = np.exp(0.5 + 0.4 * X[:, 0] + 0.3 * X[:, 1])
    k = rng.poisson(lam).astype(float)
    wp = mhx.walker_create(
        function=mhx.models.lisp("(lambda (x &key u v w &allow-other-keys) (exp (+ u (* v (elt x 0)) (* w (elt x 1)))))"),
        data=[X, k], params=[":u", 0.45, ":v", 0.42, ":w", 0.28],
        log_liklihood=mhx.create_log_liklihood_function(
            "(lambda (y model error) (declare (ignore error)) (- (* y (log model)) model))"), n_chains=2, seed=1)
    m = np.exp(0.45 + 0.42 * X[:, 0] + 0.28 * X[:, 1])
    terms = k * np.log(m) - m
    assert abs(wp.last_step().prob - terms.sum()) <= REL * np.abs(terms).sum()
    # refused: a function that reads x1 on a dataset of one column; the cutoff likelihood with two
    e = mhx.Engine(1, 2, 1)
    e.set_function_expr(0, "a + b*xcol1", ["a", "b"], [0, 1])
    e.set_dataset(0, X[:, 0], z, sig)
    with pytest.raises(mhx.MhxError, match="one column"):
        e.init_chains(np.array([0.0, 1.0]))
    with pytest.raises(mhx.MhxError, match="cutoff"):
        e.set_dataset(0, X, z, sig, likelihood=mhx.capi.LIK_NORMAL_CUTOFF)
    e.close()
