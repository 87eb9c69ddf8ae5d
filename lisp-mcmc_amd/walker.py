"""The reference's own fitting API over the MI355X engine.

Host-side mirror of the exported Lisp surface of afranson/Lisp-MCMC for the
walker-adaptive-steps path (M: = mcmc-fitting.lisp):

    walker-create                M:1132-1163   walker_create
    mcmc-fit                     M:1165-1175   mcmc_fit
    walker-adaptive-steps        M:946-947     walker_adaptive_steps
    walker-adaptive-steps-full   M:862-942     walker_adaptive_steps_full
    walker-many-steps            M:849-853     walker_many_steps
    walker-take-step             M:1072-1095   walker_take_step
    walker-get                   M:487-543     walker_get
    walker-modify                M:547-580     walker_modify
    prior-bounds-let             M:346-369     prior_bounds
    mfit-walker-estop            M:860-861     request_stop

Same names (kebab-case -> snake_case), same argument meaning, same error behaviour where
the path defines one.  All stepping and every log-posterior is computed by libmhx.so on
the GPU; this file only lays data out the way clean-data / clean-data-error do
(M:774-825) and turns device read-backs into the reference's return shapes.

Extension: n_chains > 1 makes a walker SET that steps in one batch (the reference maps a
list of walkers sequentially, M:1029-1033); `chain=` selects a member on read-back.
"""
from fractions import Fraction

import warnings

import numpy as np

from . import _capi as capi
from .engine import Engine
from .models import Model

_LIKS = {
    None: capi.LIK_NORMAL,
    "normal": capi.LIK_NORMAL, "log-liklihood-normal": capi.LIK_NORMAL,
    "normal-weighted": capi.LIK_NORMAL, "log-liklihood-normal-weighted": capi.LIK_NORMAL,
    "normal-cutoff": capi.LIK_NORMAL_CUTOFF, "log-liklihood-normal-cutoff": capi.LIK_NORMAL_CUTOFF,
    "poisson": capi.LIK_POISSON, "log-poisson": capi.LIK_POISSON,
}


def _key(k):
    """:much-better-name -> much_better_name (the identifier used on both sides of the ABI)"""
    from .sexpr import mangle
    return mangle(str(k))


class WalkerStep:
    """(defstruct walker-step prob params) M:462-464; params is a {key: value} plist"""

    def __init__(self, prob, params):
        self.prob = float(prob)
        self.params = params

    def __repr__(self):
        return "#S(WALKER-STEP :PROB %r :PARAMS %r)" % (self.prob, self.params)


class PriorBounds:
    """The value of a prior whose body is `bounds-total` of (prior-bounds-let ((key lo hi) ...))"""

    def __init__(self, bounds, body=None):
        self.bounds = [(_key(k), float(lo), float(hi)) for k, (lo, hi) in dict(bounds).items()]
        self.body = body  # Lisp text of the prior-bounds-let body, None = bounds-total


def prior_bounds(bounds, body=None):
    """(prior-bounds-let ((:a lo hi) ...) BODY) M:346-369 as a log-prior designator.
    -1d10 (exp(1d-5 * distance) - 1) outside (lo, hi), strict at both ends; a key missing from
    the plist reads as 0d0 (M:353).  BODY is the Lisp text of the form's body and may use
    bounds-total and the parameter names, e.g. nv-specific.lisp:31-34's
    '(+ bounds-total (if (> mu1 mu2) -1e9 0e0))'; None means `bounds-total`."""
    return PriorBounds(bounds, body)


class LikelihoodExpr:
    """What (create-log-liklihood-function #'(lambda (y model error) ...)) returns, M:402-416"""

    def __init__(self, text):
        from . import sexpr
        self.text = text
        self.expr = sexpr.likelihood_lambda_to_expr(text)


def create_log_liklihood_function(text):
    """(create-log-liklihood-function log-liklihood-function) M:402-416 as a :log-liklihood
    designator.  TEXT is the Lisp text of the 3-argument closure, e.g.
    '(lambda (y model error) (declare (ignore error)) (- (* y (log model)) model))'; the
    log-likelihood is the sum of its values over the points.  `error` is the point's sigma (the
    docstring's meaning; the reference's code hands every call the whole stddev list).  The
    function the walker fits must then be an expression model (models.lisp / models.expr)."""
    return LikelihoodExpr(text)


log_prior_flat = None  # (log-prior-flat params data) => 0d0, M:340-343


def _force_list(item):  # M:755-759
    return list(item) if isinstance(item, (list, tuple)) else [item]


def _depth(t):  # get-depth M:761-772
    if isinstance(t, (int, float, np.floating, np.integer)):
        return 0
    if isinstance(t, np.ndarray):
        return t.ndim
    return 1 + _depth(t[0])


def _vector_x(ds):
    """a dataset [X, y] whose x elements are vectors - "multiple or linked independent variables",
    M:1136-1137: X [n][2] (one row per point), y [n]"""
    try:
        return (len(ds) == 2 and np.ndim(ds[0]) == 2 and np.ndim(ds[1]) == 1
                and np.shape(ds[0])[0] == np.shape(ds[1])[0] and np.shape(ds[0])[1] in (1, 2)
                and np.shape(ds[0])[0] != 2)
    except (TypeError, ValueError):
        return False


def _clean_data(data, n_fn):  # M:807-825
    if _vector_x(data):
        return _clean_data([data], n_fn)
    if len(data) == n_fn and all(_vector_x(ds) for ds in data):
        return [[np.asarray(ds[0], dtype=np.float64), np.asarray(ds[1], dtype=np.float64)] for ds in data]
    dep = _depth(data)
    if dep == 1:
        raise ValueError("clean-data: data is of insufficient depth or improperly structured.")
    if dep == 2:
        return _clean_data([data], n_fn)
    if len(data) == n_fn:
        return [[np.asarray(col, dtype=np.float64) for col in ds] for ds in data]
    raise ValueError("clean-data: insufficient number of datasets, %d, for the given number of "
                     "functions, %d." % (len(data), n_fn))


def _clean_data_error(stddev, ys):
    """clean-data-error M:774-805 for the layouts the path uses: a number broadcasts to every
    y; a structure equal to the y structure is taken as is; anything else broadcasts its
    first element."""
    def first(t):
        while not isinstance(t, (int, float, np.floating, np.integer)):
            t = t[0]
        return float(t)
    if isinstance(stddev, (int, float, np.floating, np.integer)):
        return [np.full(y.shape, float(stddev)) for y in ys]
    sd = _force_list(stddev)
    if len(sd) == len(ys) and all(
            not isinstance(s, (int, float, np.floating, np.integer)) and len(s) == len(y)
            for s, y in zip(sd, ys)):
        return [np.asarray(s, dtype=np.float64) for s in sd]
    if len(ys) == 1 and len(sd) == len(ys[0]) and all(
            isinstance(s, (int, float, np.floating, np.integer)) for s in sd):
        return [np.asarray(sd, dtype=np.float64)]
    return [np.full(y.shape, first(stddev)) for y in ys]


def _plist(params):
    """(:b -1 :m 2) / {'b': -1, 'm': 2} -> keys in plist order (plist-keys M:190-193), values"""
    if isinstance(params, dict):
        items = list(params.items())
    else:
        p = list(params)
        if len(p) % 2:
            raise ValueError("params plist must have an even number of elements")
        items = [(p[i], p[i + 1]) for i in range(0, len(p), 2)]
    keys, vals = [], []
    for k, v in items:
        k = _key(k)
        if k in keys:
            continue  # first value wins, like getf (M:195-198)
        if not isinstance(v, (int, float, np.floating, np.integer)):
            # :single-item styles (M:7-14, M:1153-1155): one key holding a list / vector /
            # d x 1 array; expanded to key_0, key_1, ... (what (elt key i) translates to)
            flat = np.asarray(v, dtype=np.float64).reshape(-1)
            for i, vi in enumerate(flat):
                keys.append("%s_%d" % (k, i))
                vals.append(float(vi))
            continue
        keys.append(k)
        vals.append(float(v))
    return keys, np.asarray(vals, dtype=np.float64)


class Walker:
    """(defstruct walker ...) M:467-479 backed by device state."""

    def __init__(self, engine, function, param_keys, data, data_error, log_liklihood, log_prior):
        self.engine = engine
        self.function = function
        self.param_keys = param_keys
        self.param_style = ":multiple-kwargs"
        self.data = data
        self.data_error = data_error
        self.log_liklihood = log_liklihood
        self.log_prior = log_prior
        self.n_chains = engine.n_chains

    # struct accessors (exported M:480)
    def _step(self, th, pr):
        return WalkerStep(pr, dict(zip(self.param_keys, (float(v) for v in th))))

    def last_step(self, chain=0):
        s = self.engine.state()
        return self._step(s["theta"][chain], s["logpost"][chain])

    def most_likely_step(self, chain=0):
        s = self.engine.state()
        return self._step(s["best_theta"][chain], s["best_logpost"][chain])

    def length(self, chain=0):
        return int(self.engine.state()["length"][chain])

    def age(self, chain=0):
        return int(self.engine.state()["age"][chain])

    def walk(self, chain=0, take=None):
        return walker_get(self, get=":steps", take=take, chain=chain)

    def status(self):
        return self.engine.chain_status()[0]

    def _raise_on_trap(self):
        st = self.status()
        if (st == capi.CHAIN_FP_TRAP).any():
            bad = np.flatnonzero(st == capi.CHAIN_FP_TRAP)
            raise FloatingPointError(
                "walker(s) %s: the reference would have signalled an unhandled floating-point "
                "trap here (non-finite log-posterior, or 0/0 in cholesky-decomp M:597); the "
                "walker is left where it stood" % bad[:8].tolist())


def walker_create(function=None, data=None, params=None, data_error=None, log_liklihood=None,
                  log_prior=None, param_bounds=None, n_chains=1, theta0=None, device=0, seed=0,
                  chain_offset=0, history_capacity=0, adapt_mode=capi.ADAPT_FAITHFUL,
                  poisson_logfact_double=False):
    """(walker-create &key function data params data-error log-liklihood log-prior param-bounds)
    M:1132-1163.  function: a models.Model or a list of them; everything else as the reference:
    each argument may be one item or a list with one item per function (global fit)."""
    del param_bounds  # (declare (ignorable param-bounds)) M:1141
    fns = _force_list(function)
    if not all(isinstance(f, Model) for f in fns):
        raise TypeError(":function must be a model designator (lisp_mcmc_amd.models), see "
                        "INTEGRATION.md: a Lisp/Python closure cannot run on the GPU")
    K = len(fns)
    dsets = _clean_data(data, K)
    ys = [ds[1] for ds in dsets]
    sig = _clean_data_error(1 if data_error is None else data_error, ys)
    keys, vals = _plist(params)
    liks = _force_list(log_liklihood) if isinstance(log_liklihood, (list, tuple)) else [log_liklihood] * K
    pris = _force_list(log_prior) if isinstance(log_prior, (list, tuple)) else [log_prior] * K
    if len(liks) != K or len(pris) != K:
        raise ValueError("one log-liklihood / log-prior per function")
    eng = Engine(n_chains, len(keys), K, device=device, seed=seed, chain_offset=chain_offset,
                 adapt_mode=adapt_mode, history_capacity=history_capacity,
                 poisson_logfact_double=poisson_logfact_double)
    if any(getattr(f, "as_written", False) for f in fns):
        eng.set_expr_recognition(False)
    for k, f in enumerate(fns):
        missing = [q for q in f.keys if q not in keys]
        if missing:
            raise KeyError("function %d reads keys %s that :params does not supply" % (k, missing))
        if f.model_id == capi.MODEL_EXPR:
            eng.set_function_expr(k, f.expr, f.keys, [keys.index(q) for q in f.keys])
        else:
            eng.set_function(k, f.model_id, f.shape, [keys.index(q) for q in f.keys])
        lk = liks[k]
        if isinstance(lk, LikelihoodExpr):
            eng.set_dataset(k, dsets[k][0], dsets[k][1], sig[k], capi.LIK_EXPR)
            eng.set_likelihood_expr(k, lk.expr)
        else:
            if isinstance(lk, str):
                lk = lk.lstrip("#':").lower()
            if lk not in _LIKS:
                raise capi.MhxError(capi.EUNSUPPORTED, "unknown :log-liklihood %r" % (liks[k],))
            eng.set_dataset(k, dsets[k][0], dsets[k][1], sig[k], _LIKS[lk])
        pr = pris[k]
        if pr is None:
            eng.set_bounds(k, [], [], [])
        elif isinstance(pr, PriorBounds):
            eng.set_bounds(k, [keys.index(q) if q in keys else -1 for q, _, _ in pr.bounds],
                           [lo for _, lo, _ in pr.bounds], [hi for _, _, hi in pr.bounds])
            if pr.body:
                from . import sexpr
                eng.set_prior_expr(k, sexpr.prior_body_to_expr(pr.body), keys, range(len(keys)))
        else:
            raise capi.MhxError(capi.EUNSUPPORTED,
                                ":log-prior must be None (log-prior-flat) or prior_bounds(...)")
    eng.init_chains(vals if theta0 is None else np.asarray(theta0, dtype=np.float64))
    w = Walker(eng, fns, keys, dsets, sig, liks, pris)
    w._raise_on_trap()
    return w


def walker_adaptive_steps_full(walker, n=100000, temperature=1e3, auto=":prob-settle",
                               sampling_optimization=":covariance", max_walker_length=None,
                               l_matrix=None):
    """(walker-adaptive-steps-full walker &key n temperature auto sampling-optimization
    max-walker-length l-matrix) M:862"""
    if sampling_optimization not in (":covariance", "covariance"):
        raise capi.MhxError(capi.EUNSUPPORTED, ":best-value is outside the accelerated path")
    if auto in (":slope-settle", "slope-settle"):
        raise capi.MhxError(capi.EUNSUPPORTED, ":slope-settle is outside the accelerated path")
    walker.engine.adaptive_steps_full(int(np.floor(n)), temperature, 1 if auto else 0,
                                      max_walker_length or 0, l_matrix)
    walker._raise_on_trap()
    return None


def walker_adaptive_steps(walker, n=30000):
    """(walker-adaptive-steps walker &optional (n 30000)) M:946-947"""
    return walker_adaptive_steps_full(walker, n=n, temperature=10, auto=":prob-settle")


def mcmc_fit(**kw):
    """(mcmc-fit &key ...) = walker-create + walker-adaptive-steps M:1165-1175"""
    w = walker_create(**kw)
    walker_adaptive_steps(w)
    return w


def walker_many_steps(walker, n, l_matrix=None):
    """(walker-many-steps the-walker n &optional l-matrix) M:849-853"""
    if l_matrix is None:  # M:851 diag(1e-2 (single float!) * median-params)
        med = walker_get(walker, get=":median-params")
        l_matrix = np.diag([float(np.float32(1e-2)) * v for v in med.values()])
    walker.engine.many_steps(n, l_matrix)
    walker._raise_on_trap()


def walker_take_step(walker, l_matrix=None, temperature=1, z=None, u=None, rng=None):
    """(walker-take-step walker &key l-matrix (temperature 1)) M:1072-1095.  z / u are the
    numbers alexandria:gaussian-random (M:687) and (random 1.0d0) (M:1092) would return
    (the parity hook, mhx_step_injected); when all are omitted the device draws its own
    (mhx_take_step), as the reference does."""
    e = walker.engine
    if l_matrix is None:  # M:1074
        ml = walker_get(walker, get=":most-likely-params", take=1000)
        l_matrix = np.diag([float(np.float32(1e-2)) * v for v in ml.values()])
    if z is None and u is None and rng is None:
        # the reference draws its own randomness (M:687, M:1092): so does the device (Philox,
        # the chain's next draw) - mhx_take_step
        e.take_step(l_matrix, temperature)
        walker._raise_on_trap()
        return None
    rng = rng or np.random.default_rng()
    if z is None:
        z = rng.standard_normal((e.n_chains, e.d))
    if u is None:
        u = 1.0 - rng.random(e.n_chains)
    acc = e.step_injected(l_matrix, z, u, temperature)
    walker._raise_on_trap()
    return acc


def request_stop(walker):
    """(setf mfit-walker-estop t) M:860-861"""
    walker.engine.request_stop()


def _percentile(n, seq):  # nth-percentile M:1493-1504
    copy = np.sort(np.asarray(seq, dtype=np.float64))
    pos = Fraction(n).limit_denominator(1000) * (len(copy) - 1) / 100
    lo = pos.numerator // pos.denominator
    if pos == lo:
        return float(copy[lo])
    return float((copy[lo] + copy[lo + 1]) / 2)


class HistoryTruncated(UserWarning):
    """walker-get was asked about more steps than the device history ring still holds"""


def lplist_covariance(v):
    """lplist-covariance M:614-643 of N parameter vectors [N, d], in the reference's own order of
    operations: averages (/ (reduce #'+ x) N) M:626, then per entry the serial sum over the points
    of (/ (* a_ik a_jk) N) M:636-643 - the division inside the sum.  np.cumsum accumulates strictly
    left to right (np.sum adds pairwise), so the result has the reference's bits."""
    v = np.ascontiguousarray(v, dtype=np.float64)
    n, d = v.shape
    avg = np.cumsum(v, axis=0)[-1] / n
    an = v - avg
    cov = np.empty((d, d))
    for i in range(d):
        for j in range(d):
            cov[i, j] = np.cumsum((an[:, i] * an[:, j]) / n)[-1]
    return cov


def walker_get(walker, get=":steps", take=None, param=None, chain=0):
    """(walker-get walker &key get take param) M:487-543, served from the device trace."""
    e = walker.engine
    g = str(get).lstrip(":").lower()
    keys = walker.param_keys
    cap = int(e.state()["length"][chain])
    t = cap if take is None else min(int(take), cap)
    if g == "most-likely-params":  # M:511-515 (struct slot, not windowed)
        return dict(walker.most_likely_step(chain).params)
    if g == "acceptance":  # M:506-508
        a = e.acceptance(max(t, 1))[chain]
        return Fraction(int(round(a * t)), t)
    if g == "l-matrix":  # M:543
        st, L, _ = e.proposal_factor(chain, max(t, 1))
        if st == capi.L_CAUGHT:
            raise ArithmeticError("(walker-get :l-matrix): type-error / division-by-zero / "
                                  "floating-point-overflow (the conditions M:891-894 handles)")
        if st == capi.L_INVALID:
            raise FloatingPointError("(walker-get :l-matrix): floating-point-invalid-operation")
        return L if st == capi.L_OK else np.zeros((0, 0))
    prob, th = e.trace(chain, t)
    if len(prob) < t:
        # the reference keeps every step of a walk (M:549); the engine keeps the newest
        # history_capacity in its device ring.  A window that reaches past the ring is answered
        # with what is there - and says so, instead of silently being about a shorter walk.
        warnings.warn(HistoryTruncated(
            "walker-get %s :take %d: the device history ring holds the newest %d of the walk's %d "
            "steps; create the walker with history_capacity >= the walk's length to keep them all"
            % (get, t, len(prob), cap)), stacklevel=2)
    steps = [walker._step(th[i], prob[i]) for i in range(len(prob))]
    if g == "steps":
        return steps
    if g == "log-liklihoods":  # M:540
        return [s.prob for s in steps]
    if g == "params":
        return [s.params for s in steps]
    if g == "param":
        return [s.params[_key(param)] for s in steps]
    if g == "unique-steps":  # M:492-496: `equal` on two double-floats is eql - the same BITS
        bits = np.asarray(prob, dtype=np.float64).view(np.uint64)  # (0.0 / -0.0 differ, NaN = NaN)
        return [steps[i].params for i in range(len(steps))
                if i + 1 >= len(steps) or bits[i] != bits[i + 1]]
    if g == "forward-steps":  # M:497-502
        return [steps[i].params for i in range(len(steps) - 1)
                if not steps[i].prob <= steps[i + 1].prob]
    if g == "most-likely-step":  # M:503-505 over the window; ties keep the later element
        best = steps[0]
        for s in steps[1:]:
            best = best if best.prob > s.prob else s
        return best
    if g == "median-params":  # M:516-523
        return {k: _percentile(50, th[:, j]) for j, k in enumerate(keys)}
    if g == "covariance-matrix":  # M:541: (lplist-covariance unique-steps)
        u = np.array([[p[k] for k in keys] for p in walker_get(walker, ":unique-steps", take, chain=chain)])
        return lplist_covariance(u)
    if g == "stddev-params":  # M:525-539 diagonal of the l-matrix
        if cap < 10:
            return {k: 0.0 for k in keys}
        L = walker_get(walker, ":l-matrix", take, chain=chain)
        return {k: float(L[j, j]) for j, k in enumerate(keys)}
    raise ValueError("unknown :get %r" % (get,))


def walker_modify(walker, modify=None, **kw):
    """(walker-modify ...) M:547-580: only :add-step is on the accelerated path and it is
    performed by the device inside walker-take-step; the list-surgery actions are host-side
    post-processing the engine does not take over yet."""
    m = str(modify).lstrip(":").lower()
    if m == "burn-walks":
        walker.engine.modify(m, kw["burn_number"])
    elif m == "keep-walks":
        walker.engine.modify(m, kw["keep_number"])
    elif m in ("reset", "reset-to-most-likely"):
        walker.engine.modify(m)
        return walker
    elif m == "delete":
        walker.engine.close()
    else:  # :add-step happens on the device inside walker-take-step; :add-walks is unused (M:556)
        raise capi.MhxError(capi.EUNSUPPORTED,
                            "walker-modify %s is not part of the accelerated path" % (modify,))
    return None
