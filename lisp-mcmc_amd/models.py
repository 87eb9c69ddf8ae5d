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


def lisp(lambda_text, as_written=False, recognise=None):
    """An arbitrary model from the TEXT of the reference-style closure, e.g.
    lisp('(lambda (x &key m b &allow-other-keys) (+ b (* m x)))').  The body is translated to a
    C expression (sexpr.py) and handed to libmhx (mhx_set_function_expr), which compiles it for
    gfx950 at walker-create time (hiprtc) - or, when the body IS a polynomial background plus
    Gaussian / Lorentzian peaks a*exp(-((x-mu)/w)^2) (a/(1+((x-mu)/w)^2)), serves it with that
    enumerated model's kernels (recognised below the C ABI, csrc/mhx_expr.cpp: the Lisp shim's
    expr-model gets the same).

    as_written=True: the walker's engine compiles every expression exactly as written
    (mhx_set_expr_recognition).  `recognise` is accepted and ignored: until round 4 this module
    did the recognising itself."""
    del recognise
    from . import sexpr
    keys, expr = sexpr.lambda_to_expr(lambda_text)
    m = Model(capi.MODEL_EXPR, keys, expr=expr)
    m.as_written = bool(as_written)
    return m


def classify(lambda_text):
    """What libmhx makes of a closure's text (mhx_expr_classify; needs no GPU): (model id, shape,
    keys in that model's local order) - (MODEL_EXPR, (), the closure's keys) when it is compiled
    as written."""
    import ctypes as C
    from . import sexpr
    keys, cexpr = sexpr.lambda_to_expr(lambda_text)
    names = (C.c_char_p * max(len(keys), 1))(*[k.encode() for k in keys])
    model, n = C.c_int32(0), C.c_int32(0)
    shape = (C.c_int32 * 2)()
    order = (C.c_int32 * max(len(keys), 1))()
    capi.check(capi.lib().mhx_expr_classify(cexpr.encode(), names, len(keys), C.byref(model), shape,
                                            order, C.byref(n)))
    if model.value == capi.MODEL_EXPR:
        return capi.MODEL_EXPR, (), keys
    sh = () if model.value == capi.MODEL_POLY else (shape[0], shape[1])
    return model.value, sh, [keys[order[j]] for j in range(n.value)]


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
