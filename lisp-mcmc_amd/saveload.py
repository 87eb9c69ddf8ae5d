"""walker-save / walker-load in the plist text format the reference's
commented-out code sketches (mcmc-fitting.lisp:971-1001):

    (:FN names :DATA data :PARAM-KEYS keys :STDDEV stddev :LOG-LIKLIHOOD names
     :LOG-PRIOR names :WALKER ((:PROB p :PARAMS (:k v ...)) ...))

written with the Lisp printer's conventions (`with-standard-io-syntax`: upper-case symbols,
doubles with a d exponent) so that a Lisp `read` takes the file as is.  Functions are not
serialised, only named (test.lisp:42): loading needs the :function / :log-prior designators
again, exactly like the reference's walker-load.
"""
import re

import numpy as np

from . import sexpr
from .models import Model


def _d(v):
    """a double as the Lisp printer writes it under standard syntax: 1.5d0, 1.0d-7"""
    r = repr(float(v))
    if "inf" in r or "nan" in r:
        raise ValueError("non-finite value in a walker")
    if "e" in r:
        m, e = r.split("e")
        if "." not in m:
            m += ".0"
        return "%sd%d" % (m, int(e))
    return r + "d0"


def _list(vals):
    return "(" + " ".join(_d(v) for v in vals) + ")"


def _name(obj):
    if obj is None:
        return "NIL"
    if isinstance(obj, Model):
        return '"%s"' % (obj.expr if obj.expr else "MODEL-%d%s" % (obj.model_id, list(obj.shape)))
    if isinstance(obj, str):
        return obj.lstrip("#':").upper()
    return '"%s"' % type(obj).__name__


def walker_save(walker, filename, take=None, chain=0):
    """(walker-save walker filename &optional take) M:980-985"""
    from .walker import walker_get
    steps = walker_get(walker, get=":steps", take=take, chain=chain)
    keys = walker.param_keys
    out = ["(:FN (%s)" % " ".join(_name(f) for f in walker.function),
           " :DATA (%s)" % " ".join("(%s %s)" % (_list(ds[0]), _list(ds[1])) for ds in walker.data),
           " :PARAM-KEYS (%s)" % " ".join(":" + k.upper().replace("_", "-") for k in keys),
           " :STDDEV (%s)" % " ".join(_list(s) for s in walker.data_error),
           " :LOG-LIKLIHOOD (%s)" % " ".join(_name(x) for x in walker.log_liklihood),
           " :LOG-PRIOR (%s)" % " ".join(_name(x) for x in walker.log_prior),
           " :WALKER ("]
    for s in steps:
        plist = " ".join(":%s %s" % (k.upper().replace("_", "-"), _d(s.params[k])) for k in keys)
        out.append("  (:PROB %s :PARAMS (%s))" % (_d(s.prob), plist))
    out.append(" ))")
    with open(filename, "w") as f:
        f.write("\n".join(out) + "\n")
    return None


def _num(tok):
    return float(re.sub(r"[dD]", "e", tok))


def _getf(plist, key):
    for i in range(0, len(plist) - 1, 2):
        if isinstance(plist[i], str) and plist[i].upper() == key:
            return plist[i + 1]
    return None


def read_saved(filename):
    """the file as Python data: dict with data, param_keys, stddev, steps [(prob, [values])]"""
    text = open(filename).read().replace("#S(", "(")
    form = sexpr.parse(text)
    keys = [sexpr.mangle(k) for k in _getf(form, ":PARAM-KEYS")]
    steps = []
    for st in _getf(form, ":WALKER"):
        # (:PROB p :PARAMS (...)), what the Lisp shim's walker-save writes too; files of
        # round 1 have #S(WALKER-STEP :PROB p :PARAMS (...))
        body = st if str(st[0]).startswith(":") else st[1:]
        prob = _num(_getf(body, ":PROB"))
        pl = _getf(body, ":PARAMS")
        vals = {sexpr.mangle(pl[i]): _num(pl[i + 1]) for i in range(0, len(pl), 2)}
        steps.append((prob, [vals[k] for k in keys]))
    data = [[[_num(v) for v in col] for col in ds] for ds in _getf(form, ":DATA")]
    stddev = [[_num(v) for v in s] for s in _getf(form, ":STDDEV")]
    return {"data": data, "param_keys": keys, "stddev": stddev, "steps": steps,
            "fn": _getf(form, ":FN"), "log_liklihood": _getf(form, ":LOG-LIKLIHOOD"),
            "log_prior": _getf(form, ":LOG-PRIOR")}


def walker_load(filename, function=None, log_liklihood=None, log_prior=None, quiet=False, **kw):
    """(walker-load filename &key function log-liklihood log-prior quiet) M:987-1001: without
    the three designators it prints what the file recommends and returns None; with them it
    rebuilds the walker on the GPU and restores the saved walk."""
    from .walker import walker_create
    full = read_saved(filename)
    if not quiet:
        print("*Recommendations*\nfunction: %s\nlog-liklihood: %s\nlog-prior: %s" %
              (full["fn"], full["log_liklihood"], full["log_prior"]))
    if function is None:
        return None
    newest = full["steps"][0]
    params = [v for k, val in zip(full["param_keys"], newest[1]) for v in (":" + k, val)]
    data = full["data"] if len(full["data"]) > 1 else full["data"][0]
    stddev = full["stddev"] if len(full["stddev"]) > 1 else full["stddev"][0]
    w = walker_create(function=function, data=data, params=params, data_error=stddev,
                      log_liklihood=log_liklihood, log_prior=log_prior, **kw)
    prob = np.array([s[0] for s in full["steps"]])
    theta = np.array([s[1] for s in full["steps"]])
    w.engine.set_history(0, prob, theta)
    return w
