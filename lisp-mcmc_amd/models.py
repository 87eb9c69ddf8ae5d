"""Model designators: what stands in for the reference's :function closures.

The reference takes (lambda (x &key m b &allow-other-keys) ...) (mcmc-fitting.lisp:1134-1137);
a Lisp closure cannot run on the GPU, so a function is named by an enumerated device model
(formulas: include/mhx.h) plus the KEYS it reads, in the model's local order.  The keys are
looked up in the walker's parameter plist, so functions of a global fit share parameters by
naming the same key (README "Global Parameter Fitting", test.lisp:54-72).
"""
from . import _capi as capi


class Model:
    def __init__(self, model_id, keys, shape=(), expr=None):
        self.model_id = int(model_id)
        from .sexpr import mangle
        self.keys = [mangle(str(k)) for k in keys]
        self.shape = tuple(int(s) for s in shape)
        self.expr = expr  # C-syntax body for MODEL_EXPR

    def __repr__(self):
        return "Model(%d, %r, %r)" % (self.model_id, self.keys, self.shape)


def lisp(lambda_text, recognise=True):
    """An arbitrary model from the TEXT of the reference-style closure, e.g.
    lisp('(lambda (x &key m b &allow-other-keys) (+ b (* m x)))').  The body is translated to a
    C expression (sexpr.py) and compiled for gfx950 at walker-create time (hiprtc).

    recognise: a body that IS a polynomial background plus Gaussian (or Lorentzian) peaks of the
    form a*exp(-((x-mu)/w)^2) (a/(1+((x-mu)/w)^2)) is handed to the engine as that enumerated
    model instead - same function, evaluated with the peak kernels' fused arithmetic (a few ulp
    from the closure's own rounding, well inside the parity tolerance) and their tile-level
    skipping.  recognise=False always compiles the expression as written."""
    from . import sexpr
    if recognise:
        hit = sexpr.recognise_peaks(lambda_text)
        if hit is not None:
            model_id, shape, keys = hit
            m = Model(model_id, keys, shape)
            # (an expression likelihood needs an expression model: walker_create falls back)
            m.source_expr = sexpr.lambda_to_expr(lambda_text)
            return m
    keys, expr = sexpr.lambda_to_expr(lambda_text)
    return Model(capi.MODEL_EXPR, keys, expr=expr)


def expr(c_expression, keys):
    """An arbitrary model from a C-syntax expression over x and the keys (include/mhx.h)."""
    from . import sexpr
    return Model(capi.MODEL_EXPR, [sexpr.mangle(k) for k in keys], expr=c_expression)


def poly(*keys):
    """f = c0 + c1 x + ... ; poly('b', 'm') is the line (+ b (* m x)) of mcmc-fitting.lisp:1186"""
    return Model(capi.MODEL_POLY, keys)


def line(b="b", m="m"):
    return poly(b, m)


def gauss_peaks(bg_keys, peak_keys):
    """bg_keys: background polynomial keys; peak_keys: [(A, mu, w), ...]"""
    keys = list(bg_keys) + [k for p in peak_keys for k in p]
    return Model(capi.MODEL_GAUSS_PEAKS, keys, (len(bg_keys), len(peak_keys)))


def lorentz_peaks(bg_keys, peak_keys):
    keys = list(bg_keys) + [k for p in peak_keys for k in p]
    return Model(capi.MODEL_LORENTZ_PEAKS, keys, (len(bg_keys), len(peak_keys)))


def lorder_mixed_bg(scale="scale", linewidth="linewidth", x0="x0", mix="mix", bg0="bg0", bg1="bg1"):
    """the six keys of test.lisp:16-17"""
    return Model(capi.MODEL_LORDER_MIXED, [scale, linewidth, x0, mix, bg0, bg1])


def exp_decay(a="a", tau="tau", c="c"):
    return Model(capi.MODEL_EXP_DECAY, [a, tau, c])


def sinusoid(a="a", omega="omega", phi="phi", c="c"):
    return Model(capi.MODEL_SINUSOID, [a, omega, phi, c])


def pvoigt2(a, b0, b1, mu1, w1, eta1, mu2, w2, eta2, rho, c2):
    return Model(capi.MODEL_PVOIGT2, [a, b0, b1, mu1, w1, eta1, mu2, w2, eta2, rho, c2])
