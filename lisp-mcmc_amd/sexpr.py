"""Lisp s-expression -> C expression, for model bodies and prior bodies.

The reference's :function is `(lambda (x &key m b &allow-other-keys) (+ b (* m x)))`
(mcmc-fitting.lisp:1134-1137) and its priors are prior-bounds-let bodies
(mcmc-fitting.lisp:346-369, nv-specific.lisp:25-34).  This module turns the TEXT of such a
form into the C-syntax expression `mhx_set_function_expr` / `mhx_set_prior_expr` take, which
libmhx compiles for gfx950 with hiprtc.  (The Common Lisp shim does the same walk on the form
itself; here the form arrives as a string because Python has no reader.)

Supported: numbers (1, 2.5, 1d-5, 1e3, 1/2), symbols, + - * / (n-ary, unary - and /), 1+ 1-,
expt, exp log sqrt sin cos tan atan tanh abs, max min, if, < > <= >= = /= (chains), and or
not, pi.  Anything else raises SexprError naming the operator.
"""
import re

_TOKEN = re.compile(r"""\s*(;[^\n]*|[()']|"(?:\\.|[^"])*"|[^\s()';]+)""")


class SexprError(ValueError):
    pass


def parse(text):
    """text -> nested lists of atoms (strings)"""
    toks = [t for t in _TOKEN.findall(text) if not t.startswith(";")]
    pos = 0

    def rd():
        nonlocal pos
        if pos >= len(toks):
            raise SexprError("unexpected end of form")
        t = toks[pos]
        pos += 1
        if t == "(":
            out = []
            while True:
                if pos >= len(toks):
                    raise SexprError("missing )")
                if toks[pos] == ")":
                    pos += 1
                    return out
                out.append(rd())
        if t == ")":
            raise SexprError("unexpected )")
        if t == "'":
            return ["quote", rd()]
        return t
    form = rd()
    if pos != len(toks):
        raise SexprError("trailing text after the form")
    return form


def mangle(sym):
    """Lisp symbol / keyword -> C identifier (also the name used for the parameter keys)"""
    s = sym.lstrip(":").lower()
    s = re.sub(r"[^a-z0-9_]", "_", s)
    if not s or s[0].isdigit():
        s = "k_" + s
    if s in ("x", "bounds_total"):
        return s
    return s


_NUM = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([edfslEDFSL][+-]?\d+)?$")
_RAT = re.compile(r"^([+-]?\d+)/(\d+)$")


def number(tok):
    """Lisp numeric token -> C literal text, or None"""
    m = _RAT.match(tok)
    if m:
        return "(%s.0/%s.0)" % (m.group(1), m.group(2))
    if _NUM.match(tok):
        t = re.sub(r"[dDfFsSlL]", "e", tok)
        if "e" in t.lower() and t.lower().endswith("e0"):
            t = t[:-2]
        if not any(c in t for c in ".eE"):
            t += ".0"
        elif t.endswith("."):
            t += "0"
        return t
    return None


_FUN1 = {"exp": "exp", "log": "log", "sqrt": "sqrt", "sin": "sin", "cos": "cos", "tan": "tan",
         "atan": "atan", "tanh": "tanh", "abs": "abs", "floor": "floor"}
_CMP = {"<": "<", ">": ">", "<=": "<=", ">=": ">=", "=": "==", "/=": "!="}


def to_c(form, rename=None):
    """nested form -> C expression string; rename maps Lisp symbols to C identifiers"""
    rename = rename or {}

    def sym(s):
        k = s.lower()
        if k in rename:
            return rename[k]
        if k == "pi":
            return "3.14159265358979323846"
        if k in ("t",):
            return "1.0"
        if k in ("nil",):
            return "0.0"
        return mangle(s)

    def go(f):
        if isinstance(f, str):
            n = number(f)
            return n if n is not None else sym(f)
        if not f:
            raise SexprError("empty form ()")
        op = f[0].lower() if isinstance(f[0], str) else None
        a = f[1:]
        if op in ("+", "*"):
            if not a:
                return "0.0" if op == "+" else "1.0"
            return "(" + (" %s " % op).join(go(v) for v in a) + ")"
        if op == "-":
            if len(a) == 1:
                return "(-" + go(a[0]) + ")"
            return "(" + " - ".join(go(v) for v in a) + ")"
        if op == "/":
            if len(a) == 1:
                return "(1.0 / " + go(a[0]) + ")"
            return "(" + " / ".join(go(v) for v in a) + ")"
        if op == "1+":
            return "(" + go(a[0]) + " + 1.0)"
        if op == "1-":
            return "(" + go(a[0]) + " - 1.0)"
        if op == "expt":
            # an INTEGER exponent is repeated multiplication in Common Lisp (exact intexp
            # order), a float exponent goes through pow like (expt q 2d0) of M:377
            if isinstance(a[1], str) and re.match(r"^[+-]?\d+$", a[1]) and abs(int(a[1])) <= 64:
                return "ipow(" + go(a[0]) + ", " + str(int(a[1])) + ")"
            # a float literal with an integral value ((expt q 2d0), M:377) is libm's pow there,
            # which is within an ulp of the product; the product is what the device forms
            # (ocml's pow costs ~220 instructions per point)
            lit = number(a[1]) if isinstance(a[1], str) else None
            if lit is not None and "/" not in lit:
                v = float(lit)
                if v == int(v) and abs(v) <= 64:
                    return "ipow(" + go(a[0]) + ", " + str(int(v)) + ")"
                if v == 0.5:
                    return "sqrt(" + go(a[0]) + ")"
            return "pow(" + go(a[0]) + ", " + go(a[1]) + ")"
        if op == "log-normal" and len(a) == 3:  # (log-normal x mu sigma) M:372-377, expanded
            return go(["+", ["*", "-1/2", ["log", ["*", "2", "pi"]]], ["*", "-1", ["log", a[2]]],
                       ["*", "-1/2", ["expt", ["/", ["-", a[0], a[1]], a[2]], "2d0"]]])
        if op in _FUN1 and len(a) == 1:
            return _FUN1[op] + "(" + go(a[0]) + ")"
        if op == "log" and len(a) == 2:
            return "(log(" + go(a[0]) + ") / log(" + go(a[1]) + "))"
        if op in ("max", "min"):
            out = go(a[0])
            for v in a[1:]:
                out = "%s(%s, %s)" % (op, out, go(v))
            return out
        if op == "if":
            els = go(a[2]) if len(a) > 2 else "0.0"
            return "((%s) ? %s : %s)" % (cond(a[0]), go(a[1]), els)
        if op in ("elt", "aref", "nth", "svref") and len(a) >= 2:
            # :single-item parameter styles (M:7-14, M:1189-1195): (elt params 1),
            # (aref params 1 0), (nth 1 params) name element 1 of the list/vector/column held
            # under the one key `params`; the host expands such a key into params_0, params_1...
            seq, ix = (a[1], a[0]) if op == "nth" else (a[0], a[1])
            if isinstance(seq, str) and isinstance(ix, str) and ix.isdigit():
                # (elt x 0), (elt x 1): the components of a vector-valued independent variable
                # (M:1136-1137) - columns xcol0, xcol1 of mhx_set_dataset_cols
                if sym(seq) == "x":
                    return "xcol" + ix
                return sym(seq) + "_" + ix
            raise SexprError("%s needs a literal index into a parameter sequence" % op)
        if op in ("the", "coerce", "float") and len(a) >= 2:
            return go(a[1] if op == "the" else a[0])
        raise SexprError("operator %r is not supported in a device expression" % (f[0],))

    def cond(f):
        if isinstance(f, list) and f and isinstance(f[0], str):
            op = f[0].lower()
            a = f[1:]
            if op in _CMP:
                vals = [go(v) for v in a]
                if len(vals) == 1:
                    return "1"
                return "(" + " && ".join("(%s %s %s)" % (vals[i], _CMP[op], vals[i + 1])
                                         for i in range(len(vals) - 1)) + ")"
            if op == "and":
                return "(" + " && ".join(cond(v) for v in a) + ")" if a else "1"
            if op == "or":
                return "(" + " || ".join(cond(v) for v in a) + ")" if a else "0"
            if op == "not":
                return "(!" + cond(a[0]) + ")"
        return "(" + go(f) + " != 0.0)"
    return go(form)


def lambda_to_expr(text):
    """'(lambda (x &key m b &allow-other-keys) body)' -> (keys ['m', 'b'], C expression)"""
    form = parse(text)
    if isinstance(form, list) and len(form) == 2 and form[0] in ("function", "quote"):
        form = form[1]
    if not (isinstance(form, list) and len(form) >= 3 and isinstance(form[0], str)
            and form[0].lower() == "lambda"):
        raise SexprError("expected (lambda (x &key ...) body)")
    ll = form[1]
    if not ll or not isinstance(ll[0], str):
        raise SexprError("the lambda list must start with the independent variable")
    xname = ll[0].lower()
    keys, in_keys = [], False
    for item in ll[1:]:
        name = item[0] if isinstance(item, list) else item     # (bg02 0d0) default forms
        low = name.lower()
        if low == "&key":
            in_keys = True
        elif low.startswith("&"):
            in_keys = low == "&key"
        elif in_keys:
            keys.append(mangle(name))
    body = [b for b in form[2:] if not (isinstance(b, list) and b and b[0] == "declare")]
    if len(body) != 1:
        raise SexprError("the lambda body must be one expression")
    cexpr = to_c(body[0], {xname: "x"})
    # a key used only through (elt key i) stands for the elements key_0 ... key_n
    out = []
    for k in keys:
        elems = sorted({int(m) for m in re.findall(r"\b%s_(\d+)\b" % re.escape(k), cexpr)})
        if elems and not re.search(r"\b%s\b" % re.escape(k), cexpr):
            out += ["%s_%d" % (k, i) for i in range(elems[-1] + 1)]
        else:
            out.append(k)
    return out, cexpr


def likelihood_lambda_to_expr(text):
    """The closure given to create-log-liklihood-function (M:402-416),
    '(lambda (y model error) (declare (ignore error)) body)' -> C expression over y, model, error
    (the three arguments may carry any names)."""
    form = parse(text)
    if isinstance(form, list) and len(form) == 2 and form[0] in ("function", "quote"):
        form = form[1]
    if not (isinstance(form, list) and len(form) >= 3 and isinstance(form[0], str)
            and form[0].lower() == "lambda"):
        raise SexprError("expected (lambda (y model error) body)")
    ll = form[1]
    if not (isinstance(ll, list) and len(ll) == 3 and all(isinstance(a, str) for a in ll)):
        raise SexprError("the function must accept 3 arguments: y, model, error (M:403)")
    body = [b for b in form[2:] if not (isinstance(b, list) and b and b[0] == "declare")]
    if len(body) != 1:
        raise SexprError("the lambda body must be one expression")
    return to_c(body[0], {ll[0].lower(): "y", ll[1].lower(): "model", ll[2].lower(): "error"})


def prior_body_to_expr(text):
    """prior-bounds-let body, e.g. '(+ bounds-total (if (> mu1 mu2) -1e9 0e0))'"""
    return to_c(parse(text), {"bounds-total": "bounds_total"})


def symbols_of(form, acc=None):
    acc = set() if acc is None else acc
    if isinstance(form, str):
        if number(form) is None:
            acc.add(form.lower())
    else:
        for i, v in enumerate(form):
            if i == 0 and isinstance(v, str):
                continue
            symbols_of(v, acc)
    return acc
