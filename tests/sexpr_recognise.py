"""Test helper: an independent statement (on the Lisp form) of what libmhx's recogniser
(csrc/mhx_expr.cpp, on the C-syntax text) must make of a closure: polynomial background plus
Gaussian / Lorentzian peaks -> the enumerated model.  tests/test_sexpr.py holds the two against
each other over a corpus of closures.  (Until round 4 this lived in the Python mirror and was the
only recogniser that ever ran; the product's is below the C ABI now, for every host.)"""
import re

from lisp_mcmc_amd.sexpr import SexprError, mangle, number, parse


def recognise_peaks(text):
    """If the closure's body is  bg(x) + sum of peaks  with
         bg(x)  = c0 + c1 x + c2 x^2 ...   (terms: key, (* key x), (* key (expt x n)), (* key x x))
         peak   = (* a (exp (- (expt (/ (- x mu) w) 2))))            Gaussian, or
                  (/ a (+ 1 (expt (/ (- x mu) w) 2)))                Lorentzian
       (a, mu, w, c_i bare keys, each used once; (1+ ..), (* u u), (* -1 ..), either order of the
       factors are understood), return (model id, (nbg, npk), keys in the enumerated model's
       order); a body that is only the polynomial gives (0, (), keys by degree); otherwise None.
       MODEL ids as in include/mhx.h: 0 polynomial, 1 Gaussian, 2 Lorentzian peaks."""
    try:
        form = parse(text)
        if isinstance(form, list) and len(form) == 2 and form[0] in ("function", "quote"):
            form = form[1]
        if not (isinstance(form, list) and len(form) >= 3 and str(form[0]).lower() == "lambda"):
            return None
        ll = form[1]
        xname = ll[0].lower()
        keys, in_keys = set(), False
        for item in ll[1:]:
            if isinstance(item, list):
                return None  # default values: leave it to the expression path
            low = item.lower()
            if low.startswith("&"):
                in_keys = low == "&key"
            elif in_keys:
                keys.add(low)
        body = [b for b in form[2:] if not (isinstance(b, list) and b and b[0] == "declare")]
        if len(body) != 1:
            return None
    except (SexprError, AttributeError, IndexError):
        return None

    def is_key(f):
        return isinstance(f, str) and f.lower() in keys

    def is_x(f):
        return isinstance(f, str) and f.lower() == xname

    def is_num(f, v):
        if not isinstance(f, str):
            return False
        n = number(f)
        try:
            return n is not None and "/" not in n and float(n) == v
        except ValueError:
            return False

    def op(f):
        return f[0].lower() if isinstance(f, list) and f and isinstance(f[0], str) else None

    def terms(f):  # flatten nested sums
        if op(f) == "+":
            out = []
            for a in f[1:]:
                out += terms(a)
            return out
        return [f]

    def reduced(f):
        """(mu, w) if f is (/ (- x mu) w), else None"""
        if op(f) == "/" and len(f) == 3 and is_key(f[2]) and op(f[1]) == "-" and len(f[1]) == 3 \
                and is_x(f[1][1]) and is_key(f[1][2]):
            return f[1][2].lower(), f[2].lower()
        return None

    def squared(f):
        """(mu, w) if f is the square of a reduced coordinate"""
        if op(f) == "expt" and len(f) == 3 and (is_num(f[2], 2.0)):
            return reduced(f[1])
        if op(f) == "*" and len(f) == 3 and f[1] == f[2]:
            return reduced(f[1])
        return None

    def neg_squared(f):
        if op(f) == "-" and len(f) == 2:
            return squared(f[1])
        if op(f) == "*" and len(f) == 3:
            for a, b in ((f[1], f[2]), (f[2], f[1])):
                if is_num(a, -1.0):
                    return squared(b)
        return None

    def one_plus_squared(f):
        if op(f) == "1+" and len(f) == 2:
            return squared(f[1])
        if op(f) == "+" and len(f) == 3:
            for a, b in ((f[1], f[2]), (f[2], f[1])):
                if is_num(a, 1.0):
                    return squared(b)
        return None

    def x_power(f):
        """n if f is x^n written as x, (expt x n) or (* x x ...), else None"""
        if is_x(f):
            return 1
        if op(f) == "expt" and len(f) == 3 and is_x(f[1]) and isinstance(f[2], str) \
                and re.match(r"^[1-9]$", f[2]):
            return int(f[2])
        if op(f) == "*" and len(f) >= 3 and all(is_x(a) for a in f[1:]):
            return len(f) - 1
        return None

    bg, gauss, lorentz = {}, [], []
    for t in terms(body[0]):
        if is_key(t):
            deg, key = 0, t.lower()
        elif op(t) == "*" and len(t) >= 3:
            facs = t[1:]
            ks = [a for a in facs if is_key(a)]
            rest = [a for a in facs if not is_key(a)]
            if len(ks) != 1:
                return None
            key = ks[0].lower()
            g = rest[0] if len(rest) == 1 else None
            if g is not None and op(g) == "exp" and len(g) == 2 and neg_squared(g[1]):
                gauss.append((key,) + neg_squared(g[1]))
                continue
            if g is not None and op(g) == "/" and len(g) == 3 and is_num(g[1], 1.0) \
                    and one_plus_squared(g[2]):
                lorentz.append((key,) + one_plus_squared(g[2]))
                continue
            deg = x_power(rest[0]) if len(rest) == 1 else (len(rest) if all(is_x(a) for a in rest) and rest else None)
            if deg is None:
                return None
        elif op(t) == "/" and len(t) == 3 and is_key(t[1]) and one_plus_squared(t[2]):
            lorentz.append((t[1].lower(),) + one_plus_squared(t[2]))
            continue
        else:
            return None
        if deg in bg:
            return None
        bg[deg] = key
    peaks = gauss or lorentz
    nbg = len(bg)
    if sorted(bg) != list(range(nbg)):
        return None
    if not peaks:  # a plain polynomial c0 + c1 x + ...: MODEL_POLY (id 0), keys by degree
        if nbg < 1 or nbg > 16 or len(set(bg.values())) != nbg:
            return None
        return 0, (), [mangle(bg[i]) for i in range(nbg)]
    if (gauss and lorentz) or len(peaks) > 6 or nbg > 4:
        return None
    used = [bg[i] for i in range(nbg)] + [k for p in peaks for k in p]
    if len(set(used)) != len(used):
        return None  # a key used twice: not the enumerated model's independent parameters
    return (1 if gauss else 2), (nbg, len(peaks)), [mangle(k) for k in used]
