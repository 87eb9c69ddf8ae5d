"""What can be proven about the Common Lisp host shim WITHOUT a Lisp (none exists in the build
image or on the GPU box; DESIGN.md says so): its files are read with a small reader
(tests/lisp_reader.py) and checked against include/mhx.h -

  * every prototype of the header has a defcfun of the same arity whose argument and result
    types are of the same class (pointer / 32-bit int / 64-bit int / double / size_t / string),
    in the same order;
  * the two defcstructs lay their fields out exactly like the C structs (offsets and size from a
    compiled C probe against CFFI's natural-alignment rule);
  * every call of a %mhx-... function in the shim passes as many arguments as its defcfun
    declares;
  * every exported symbol is defined, every file reads to the end (balanced), the enum constants
    the shim restates equal the header's;
  * the reference's calling conventions are kept: walker-take-step and walker-many-steps have
    the reference's lambda lists, walker-adaptive-steps-full looks at mfit-walker-estop between
    launches, walker-get knows every selector of the reference.
"""
import os
import re
import subprocess
import tempfile

import pytest

from lisp_reader import Str, read_file, walk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LISP = os.path.join(ROOT, "lisp-mcmc_amd", "lisp")
HDR = os.path.join(ROOT, "include", "mhx.h")
FILES = ["package.lisp", "bindings.lisp", "models.lisp", "expr.lisp", "walker.lisp"]


def header_text():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    return src


def c_class(ctype):
    """class of a C parameter / result type as the ABI sees it"""
    t = " ".join(ctype.replace("const", " ").split())
    if "*" in t or "[" in t or t in ("mhx_allreduce_fn",):
        return "string-or-pointer" if t.replace(" ", "") == "char*" else "pointer"
    return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "i64", "size_t": "size",
            "double": "f64", "void": "void", "uint8_t": "i8"}[t]


def lisp_class(t):
    return {":POINTER": "pointer", ":STRING": "string", ":INT": "i32", ":INT32": "i32",
            ":INT64": "i64", ":UINT64": "i64", ":SIZE": "size", ":DOUBLE": "f64", ":VOID": "void",
            ":UINT8": "i8"}[t]


def header_prototypes():
    """name -> (result class, [argument classes])"""
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(mhx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;",
                         header_text(), flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef"):
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = "[" in a
                a = re.sub(r"\[[^\]]*\]", "", a)
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
                ctype = (mm.group(1) if mm and mm.group(1).strip() else a).strip()
                params.append(c_class(ctype + ("*" if arr else "")))
        out[name] = (c_class(ret), params)
    return out


def defcfuns():
    """c-name -> (lisp name, result class, [argument classes])"""
    out = {}
    for form in read_file(os.path.join(LISP, "bindings.lisp")):
        if isinstance(form, list) and form and form[0] == "CFFI:DEFCFUN":
            cname, lname = form[1][0], form[1][1]
            assert isinstance(cname, Str)
            args = [a for a in form[3:] if isinstance(a, list)]
            out[str(cname)] = (lname, lisp_class(form[2]), [lisp_class(a[1]) for a in args])
    return out


def compatible(c, l):
    if c == l:
        return True
    if c == "string-or-pointer":
        return l in ("string", "pointer")
    return False


def test_every_file_reads_to_the_end():
    for f in FILES + ["mcmc-fitting-amd.asd"]:
        forms = read_file(os.path.join(LISP, f))
        assert forms, f


def test_defcfuns_match_the_header_prototypes():
    protos, cfuns = header_prototypes(), defcfuns()
    assert len(protos) >= 55, len(protos)
    assert sorted(protos) == sorted(cfuns), sorted(set(protos) ^ set(cfuns))
    for name, (ret, params) in protos.items():
        lname, lret, largs = cfuns[name]
        assert lname == "%" + name.upper().replace("_", "-"), name
        assert compatible(ret, lret), (name, "result", ret, lret)
        assert len(params) == len(largs), (name, "arity", len(params), len(largs))
        for i, (c, l) in enumerate(zip(params, largs)):
            assert compatible(c, l), (name, "argument %d" % i, c, l)


STRUCTS = {"MHX-CONFIG": "mhx_config", "MHX-RUN-OPTS": "mhx_run_opts"}
SIZES = {":INT64": 8, ":UINT64": 8, ":INT32": 4, ":DOUBLE": 8, ":POINTER": 8, ":INT": 4}


def test_defcstructs_match_the_c_layout():
    # the C side: offsetof / sizeof from a compiled probe
    src = open(HDR).read()
    fields = {}
    for lname, cname in STRUCTS.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields[cname] = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*;", body)
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % HDR, "int main(void) {"]
    for cname, fs in fields.items():
        for f in fs:
            prog.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
        prog.append('  printf("%s . %%zu\\n", sizeof(%s));' % (cname, cname))
    prog.append("  return 0;\n}")
    d = tempfile.mkdtemp()
    c, exe = os.path.join(d, "p.c"), os.path.join(d, "p")
    open(c, "w").write("\n".join(prog))
    subprocess.check_call(["gcc", "-o", exe, c])
    c_layout = {}
    for line in subprocess.check_output([exe]).decode().split("\n"):
        if line:
            s, f, off = line.split()
            c_layout[(s, f)] = int(off)
    # the Lisp side: CFFI lays a defcstruct out with natural alignment, in declaration order
    seen = 0
    for form in read_file(os.path.join(LISP, "bindings.lisp")):
        if isinstance(form, list) and form and form[0] == "CFFI:DEFCSTRUCT" and form[1] in STRUCTS:
            cname = STRUCTS[form[1]]
            off, align = 0, 1
            slots = [s for s in form[2:] if isinstance(s, list)]
            assert [s[0].lower().replace("-", "_") for s in slots] == fields[cname], cname
            for s in slots:
                size = SIZES[s[1]]
                off = (off + size - 1) // size * size
                assert c_layout[(cname, s[0].lower().replace("-", "_"))] == off, (cname, s[0])
                off += size
                align = max(align, size)
            assert c_layout[(cname, ".")] == (off + align - 1) // align * align, cname
            seen += 1
    assert seen == 2


def all_forms():
    return {f: read_file(os.path.join(LISP, f)) for f in FILES}


def test_foreign_calls_pass_the_declared_number_of_arguments():
    arity = {v[0]: len(v[2]) for v in defcfuns().values()}
    calls = 0
    for f, forms in all_forms().items():
        for top in forms:
            for lst in walk(top):
                if lst and isinstance(lst[0], str) and not isinstance(lst[0], Str) and \
                        lst[0].startswith("%MHX-") and lst[0] in arity:
                    if top[0] == "CFFI:DEFCFUN":
                        continue
                    assert len(lst) - 1 == arity[lst[0]], (f, lst[0], len(lst) - 1, arity[lst[0]])
                    calls += 1
                # (apply #'%mhx-x fixed... rest): at least the fixed arguments must fit
                if lst and lst[0] == "APPLY" and isinstance(lst[1], list) and lst[1][0] == "FUNCTION" \
                        and isinstance(lst[1][1], str) and lst[1][1].startswith("%MHX-"):
                    assert lst[1][1] in arity, lst[1][1]
                    assert len(lst) - 3 <= arity[lst[1][1]], (f, lst[1][1])
                    calls += 1
    assert calls >= 40, calls
    # ... and nothing calls a %mhx- function that is not bound
    for f, forms in all_forms().items():
        for top in forms:
            for lst in walk(top):
                for a in lst:
                    if isinstance(a, str) and not isinstance(a, Str) and a.startswith("%MHX-"):
                        assert a in arity, (f, a)


def definitions():
    names = set()
    for forms in all_forms().values():
        for top in forms:
            if not isinstance(top, list) or not top or not isinstance(top[0], str):
                continue
            h = top[0]
            if h in ("DEFUN", "DEFMACRO", "DEFVAR", "DEFPARAMETER", "DEFCONSTANT", "DEFGENERIC",
                     "DEFINE-CONDITION"):
                names.add(top[1])
                if h == "DEFINE-CONDITION":
                    for slot in top[3]:
                        for i, tok in enumerate(slot):
                            if tok == ":READER":
                                names.add(slot[i + 1])
            elif h == "DEFSTRUCT":
                n = top[1][0] if isinstance(top[1], list) else top[1]
                names |= {n, "MAKE-" + n, n + "-P", "COPY-" + n}
                for slot in top[2:]:
                    if isinstance(slot, Str):
                        continue
                    names.add(n + "-" + (slot[0] if isinstance(slot, list) else slot))
    return names


def test_every_exported_symbol_is_defined():
    pkg = read_file(os.path.join(LISP, "package.lisp"))[0]
    exports = [s for clause in pkg if isinstance(clause, list) and clause[0] == ":EXPORT"
               for s in clause[1:]]
    assert len(exports) > 50
    have = definitions()
    # symbols that are only designators (named in alists / case clauses), not definitions
    designators = {"LOG-LIKLIHOOD-NORMAL", "LOG-LIKLIHOOD-NORMAL-WEIGHTED",
                   "LOG-LIKLIHOOD-NORMAL-CUTOFF", "LOG-LIKLIHOOD-POISSON", "BOUNDS-TOTAL",
                   "LOG-NORMAL"}
    missing = [s for s in exports if s not in have and s not in designators]
    assert not missing, missing
    # what the reference exports on this path (mcmc-fitting.lisp:465, 480, 544, 581, 861, 1176,
    # 1566) is there under the same name
    for s in ("WALKER-CREATE", "WALKER-ADAPTIVE-STEPS", "WALKER-ADAPTIVE-STEPS-FULL",
              "WALKER-MANY-STEPS", "WALKER-TAKE-STEP", "WALKER-GET", "WALKER-MODIFY", "WALKER-LOAD",
              "WALKER-SAVE", "MFIT-WALKER-ESTOP", "MCMC-FIT", "PRIOR-BOUNDS"):
        assert s in exports, s


def test_enum_constants_equal_the_header():
    src = header_text()
    vals = {}
    for body in re.findall(r"enum\s*\{(.*?)\}", src, re.S):
        nxt = 0
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = [t.strip() for t in item.split("=")]
                nxt = int(v, 0)
            else:
                k = item
            vals[k] = nxt
            nxt += 1
    want = {"+MHX-OK+": "MHX_OK", "+LIK-NORMAL+": "MHX_LIK_NORMAL",
            "+LIK-NORMAL-CUTOFF+": "MHX_LIK_NORMAL_CUTOFF", "+LIK-POISSON+": "MHX_LIK_POISSON",
            "+LIK-EXPR+": "MHX_LIK_EXPR", "+CHAIN-RUNNING+": "MHX_CHAIN_RUNNING",
            "+CHAIN-DONE+": "MHX_CHAIN_DONE", "+CHAIN-FP-TRAP+": "MHX_CHAIN_FP_TRAP",
            "+CHAIN-STOPPED+": "MHX_CHAIN_STOPPED"}
    seen = 0
    for form in read_file(os.path.join(LISP, "bindings.lisp")):
        if isinstance(form, list) and form and form[0] == "DEFCONSTANT" and form[1] in want:
            assert int(form[2]) == vals[want[form[1]]], form[1]
            seen += 1
    assert seen == len(want)
    # model ids of models.lisp's constructors
    ids = {"POLY-MODEL": "MHX_MODEL_POLY", "GAUSS-PEAKS-MODEL": "MHX_MODEL_GAUSS_PEAKS",
           "LORENTZ-PEAKS-MODEL": "MHX_MODEL_LORENTZ_PEAKS",
           "LORDER-MIXED-BG-MODEL": "MHX_MODEL_LORDER_MIXED", "EXP-DECAY-MODEL": "MHX_MODEL_EXP_DECAY",
           "SINUSOID-MODEL": "MHX_MODEL_SINUSOID", "PVOIGT2-MODEL": "MHX_MODEL_PVOIGT2"}
    for form in read_file(os.path.join(LISP, "models.lisp")):
        if isinstance(form, list) and form and form[0] == "DEFUN" and form[1] in ids:
            mk = [l for l in walk(form) if l and l[0] == "MAKE-MODEL"][0]
            assert int(mk[mk.index(":ID") + 1]) == vals[ids[form[1]]], form[1]


def defun(name):
    for form in read_file(os.path.join(LISP, "walker.lisp")):
        if isinstance(form, list) and form[:2] == ["DEFUN", name]:
            return form
    raise AssertionError("no (defun %s" % name)


def test_reference_calling_conventions():
    # (defun walker-take-step (walker &key l-matrix (temperature 1)) M:1072
    assert defun("WALKER-TAKE-STEP")[2] == ["WALKER", "&KEY", "L-MATRIX", ["TEMPERATURE", "1"]]
    # (defun walker-many-steps (the-walker n &optional l-matrix) M:849
    assert defun("WALKER-MANY-STEPS")[2] == ["WALKER", "N", "&OPTIONAL", "L-MATRIX"]
    # (defun walker-adaptive-steps-full (walker &key (n 100000) (temperature 1d3) (auto ...)
    #   (sampling-optimization ...) max-walker-length l-matrix) M:862
    ll = defun("WALKER-ADAPTIVE-STEPS-FULL")[2]
    assert ll[:2] == ["WALKER", "&KEY"]
    keys = [k[0] if isinstance(k, list) else k for k in ll[2:]]
    assert keys == ["N", "TEMPERATURE", "AUTO", "SAMPLING-OPTIMIZATION", "MAX-WALKER-LENGTH", "L-MATRIX"]
    assert ["N", "100000"] in ll and ["TEMPERATURE", "1D3"] in ll
    # walker-adaptive-steps: n defaults to 30000, temperature 10, :auto :prob-settle M:946-947
    f = defun("WALKER-ADAPTIVE-STEPS")
    assert f[2] == ["WALKER", "&OPTIONAL", ["N", "30000"]]
    flat = [l for l in walk(f) if l and l[0] == "WALKER-ADAPTIVE-STEPS-FULL"][0]
    assert flat[flat.index(":TEMPERATURE") + 1] == "10" and flat[flat.index(":AUTO") + 1] == ":PROB-SETTLE"
    # the e-stop variable is cleared on entry and looked at inside the stepping loop, which
    # advances in bounded launches (begin + advance), not in one blocking call
    body = defun("WALKER-ADAPTIVE-STEPS-FULL")
    sub = list(walk(body))
    assert ["SETF", "MFIT-WALKER-ESTOP", "NIL"] in sub
    loops = [l for l in sub if l and l[0] == "LOOP"]
    assert any("MFIT-WALKER-ESTOP" in str(l) and "%MHX-ADAPTIVE-ADVANCE" in str(l) for l in loops)
    assert not any(l and l[0] == "%MHX-ADAPTIVE-STEPS-FULL" for l in sub)
    # walker-get: every selector of the reference's lambda list (M:487)
    g = defun("WALKER-GET")
    case = [l for l in walk(g) if l and l[0] == "CASE" and l[1] == "GET"][0]
    have = {c[0] for c in case[2:] if isinstance(c, list)}
    for sel in (":STEPS", ":UNIQUE-STEPS", ":FORWARD-STEPS", ":MOST-LIKELY-STEP", ":ACCEPTANCE",
                ":PARAM", ":PARAMS", ":MOST-LIKELY-PARAMS", ":MEDIAN-PARAMS", ":STDDEV-PARAMS",
                ":LOG-LIKLIHOODS", ":COVARIANCE-MATRIX", ":L-MATRIX"):
        assert sel in have, sel
    # one chain's accessors read one chain (mhx_get_chain), not the state of all of them
    st = defun("%STATE")
    assert any(l and l[0] == "%MHX-GET-CHAIN" for l in walk(st))
    assert not any(l and l[0] == "%MHX-GET-STATE" for l in walk(st))


def test_no_reference_text_in_the_shim():
    """the reference is GPLv3 and the shim is not a copy of it: its mapcon selectors and
    clean-data messages are written afresh"""
    txt = open(os.path.join(LISP, "walker.lisp")).read()
    assert "mapcon" not in txt
    assert "insufficient depth or improperly structured" not in txt
    assert "insufficient number of datasets" not in txt


# ---- what a compiler would say at load time ------------------------------------------------------
def _shim_sources():
    return [os.path.join(LISP, f) for f in ("package.lisp", "bindings.lisp", "expr.lisp",
                                            "models.lisp", "walker.lisp")]


def test_no_undefined_function_wrong_arity_or_unbound_variable():
    """tests/lisp_lint.py walks the shim the way an evaluator would (special forms, the standard
    binding macros, loop, CFFI's with-foreign-*, the shim's own macros) and reports calls of
    functions defined nowhere, calls of the shim's own functions (defuns, defcfuns, struct
    constructors and accessors) with an argument count or keyword their lambda list does not
    take, and variables bound nowhere - what SBCL would print as warnings when loading a file
    that has never been loaded."""
    import lisp_lint
    problems = lisp_lint.lint(_shim_sources(), own_packages=("mcmc-fitting-amd", "mfit-amd"))
    assert problems == [], "\n".join(problems)


def test_the_walker_does_find_such_mistakes(tmp_path):
    """... and it is not vacuous: three planted mistakes of those kinds are all reported"""
    import lisp_lint
    src = open(os.path.join(LISP, "walker.lisp")).read()
    bad = (src.replace("(walker-n-params walker)", "(walker-nparams walker)", 1)
           .replace("(check (%mhx-get-trace e c take pr th n-out))",
                    "(check (%mhx-get-trace e c take pr th))", 1)
           .replace("(cffi:mem-ref n-out :int)", "(cffi:mem-ref nout :int)", 1))
    assert bad != src
    p = tmp_path / "walker.lisp"
    p.write_text(bad)
    problems = lisp_lint.lint(_shim_sources()[:-1] + [str(p)],
                              own_packages=("mcmc-fitting-amd", "mfit-amd"))
    text = "\n".join(problems)
    assert "WALKER-NPARAMS, which is defined nowhere" in text
    assert "%MHX-GET-TRACE called with 5 argument(s), needs 6" in text
    assert "variable NOUT is bound nowhere" in text
    assert len(problems) == 3, text
