"""Test-side evaluator of the Lisp forms that lisp-mcmc_amd/sexpr.py translates: computes what
SBCL would (binary64, left-to-right n-ary folds, libm through Python's math).  Checker only."""
import math

from sexpr_shared import parse_number  # noqa: F401  (kept separate so product code is not imported)


def evaluate(form, env):
    """form: nested lists from a reader (any); env: {symbol(lower): float}"""
    if isinstance(form, str):
        v = parse_number(form)
        if v is not None:
            return float(v)
        k = form.lower()
        if k == "pi":
            return math.pi
        return env[k]
    op, a = form[0].lower(), form[1:]
    ev = lambda f: evaluate(f, env)  # noqa: E731
    if op == "+":
        acc = 0.0
        for i, v in enumerate(a):
            acc = ev(v) if i == 0 else acc + ev(v)
        return acc
    if op == "*":
        acc = 1.0
        for i, v in enumerate(a):
            acc = ev(v) if i == 0 else acc * ev(v)
        return acc
    if op == "-":
        if len(a) == 1:
            return -ev(a[0])
        acc = ev(a[0])
        for v in a[1:]:
            acc = acc - ev(v)
        return acc
    if op == "/":
        if len(a) == 1:
            return 1.0 / ev(a[0])
        acc = ev(a[0])
        for v in a[1:]:
            acc = acc / ev(v)
        return acc
    if op == "1+":
        return ev(a[0]) + 1.0
    if op == "1-":
        return ev(a[0]) - 1.0
    if op == "expt":
        if isinstance(a[1], str) and a[1].lstrip("+-").isdigit():   # SBCL intexp
            base, power = ev(a[0]), int(a[1])
            neg, power = power < 0, abs(power)
            nextn, total = power >> 1, (base if power & 1 else 1.0)
            while nextn:
                base = base * base
                if nextn & 1:
                    total = base * total
                nextn >>= 1
            return 1.0 / total if neg else total
        return math.pow(ev(a[0]), ev(a[1]))
    if op in ("exp", "sqrt", "sin", "cos", "tan", "atan", "tanh", "floor"):
        return float(getattr(math, op)(ev(a[0])))
    if op == "log":
        return math.log(ev(a[0])) if len(a) == 1 else math.log(ev(a[0])) / math.log(ev(a[1]))
    if op == "abs":
        return abs(ev(a[0]))
    if op == "max":
        return max(ev(v) for v in a)
    if op == "min":
        return min(ev(v) for v in a)
    if op == "if":
        return ev(a[1]) if truth(a[0], env) else (ev(a[2]) if len(a) > 2 else 0.0)
    raise ValueError("evaluate: unsupported operator %r" % op)


def truth(form, env):
    if isinstance(form, list) and form and isinstance(form[0], str):
        op, a = form[0].lower(), form[1:]
        cmp = {"<": lambda p, q: p < q, ">": lambda p, q: p > q, "<=": lambda p, q: p <= q,
               ">=": lambda p, q: p >= q, "=": lambda p, q: p == q, "/=": lambda p, q: p != q}
        if op in cmp:
            vals = [evaluate(v, env) for v in a]
            return all(cmp[op](vals[i], vals[i + 1]) for i in range(len(vals) - 1))
        if op == "and":
            return all(truth(v, env) for v in a)
        if op == "or":
            return any(truth(v, env) for v in a)
        if op == "not":
            return not truth(a[0], env)
    return evaluate(form, env) != 0.0
