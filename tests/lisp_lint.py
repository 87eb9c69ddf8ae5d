"""A small Common Lisp CODE WALKER for checking the CFFI shim without a Lisp (no Lisp
implementation exists in the build image).  On top of tests/lisp_reader.py it follows the
evaluation rules of the special forms and macros the shim uses and reports

  * calls to functions / macros that are defined nowhere (the shim, the COMMON-LISP package, a
    package-qualified symbol of CFFI / ALEXANDRIA / SB-INT),
  * calls of the shim's own functions with an argument count or a keyword their lambda list
    does not take,
  * references to variables that are bound nowhere (lexically, or by defvar / defparameter /
    defconstant).

These are the mistakes a compiler would report at load time with a full warning or a
style-warning, and the ones a never-run file is most likely to carry.  The walker is deliberately
conservative: anything it does not understand (reader conditionals, unknown macros of other
packages) is walked as a plain call or skipped, never guessed at."""
import lisp_reader as lr

CL_FUNCTIONS = set("""
+ - * / = /= < > <= >= 1+ 1- abs acos adjoin alpha-char-p alphanumericp and append apply aref
array-dimension array-dimensions array-rank arrayp ash asin assoc atan atom boundp butlast car cdr
caar cadr cdar cddr caddr cdddr case ccase ceiling char char= char-code char-upcase char-downcase
characterp code-char coerce concatenate cond cons consp copy-list copy-seq cos cosh count
count-if decf declaim declare defconstant defgeneric define-condition defmacro defmethod
defpackage defparameter defstruct defun defvar delete delete-if denominator destructuring-bind
digit-char-p do do* dolist dotimes ecase eighth elt endp eq eql equal equalp error etypecase eval
evenp every exp expt fboundp fifth fill find find-if find-package first flet float floatp floor
format fourth fround funcall function gensym get getf gethash handler-bind handler-case identity
if ignore-errors in-package incf integerp intern isqrt keywordp labels lambda last length let let*
list list* listp log loop macrolet make-array make-hash-table make-list make-string map mapc
mapcan mapcar maphash max member min minusp mod multiple-value-bind multiple-value-list
multiple-value-setq nconc ninth not notany nreverse nth nth-value nthcdr null numberp numerator
oddp or parse-integer plusp pop position position-if prin1 prin1-to-string princ print progn prog1
psetf push pushnew quote rationalp read read-from-string realp reduce rem remove remove-duplicates
remove-if remove-if-not replace rest return return-from reverse round second setf setq seventh
signal sin sinh sixth some sort sqrt stable-sort string string-downcase string-upcase string=
string< stringp subseq symbol-name symbol-value symbolp tan tanh tenth terpri the third truncate
typecase typep unless unwind-protect values values-list vector vectorp warn when
with-open-file with-output-to-string with-standard-io-syntax write-line write-string zerop
make-symbol string-equal char-equal search mismatch complement constantly princ-to-string
write write-char write-to-string read-line finish-output fresh-line
""".upper().split())

# (min, max or None) positional arguments of the standard functions the shim calls most; keyword
# arguments, where a function takes them, count from `max` on and are not checked
CL_ARITY = {k.upper(): v for k, v in {
    "length": (1, 1), "first": (1, 1), "second": (1, 1), "third": (1, 1), "fourth": (1, 1),
    "car": (1, 1), "cdr": (1, 1), "cadr": (1, 1), "cddr": (1, 1), "rest": (1, 1), "last": (1, 2),
    "nth": (2, 2), "nthcdr": (2, 2), "elt": (2, 2), "aref": (1, None), "getf": (2, 3),
    "cons": (2, 2), "list": (0, None), "append": (0, None), "mapcar": (2, None), "mapc": (2, None),
    "map": (3, None), "every": (2, None), "some": (2, None), "reduce": (2, None),
    "apply": (2, None), "funcall": (1, None), "format": (2, None), "error": (1, None),
    "warn": (1, None), "coerce": (2, 2), "floor": (1, 2), "round": (1, 2), "fround": (1, 2),
    "truncate": (1, 2), "max": (1, None), "min": (1, None), "abs": (1, 1), "sqrt": (1, 1),
    "exp": (1, 1), "log": (1, 2), "expt": (2, 2), "sin": (1, 1), "cos": (1, 1), "tan": (1, 1),
    "atan": (1, 2), "tanh": (1, 1), "1+": (1, 1), "1-": (1, 1), "zerop": (1, 1), "not": (1, 1),
    "null": (1, 1), "consp": (1, 1), "atom": (1, 1), "symbolp": (1, 1), "numberp": (1, 1),
    "integerp": (1, 1), "floatp": (1, 1), "realp": (1, 1), "rationalp": (1, 1),
    "stringp": (1, 1), "arrayp": (1, 1), "array-rank": (1, 1), "eq": (2, 2), "eql": (2, 2),
    "equal": (2, 2), "symbol-name": (1, 1), "intern": (1, 2), "find-package": (1, 1),
    "gensym": (0, 1), "numerator": (1, 1), "denominator": (1, 1), "char": (2, 2),
    "string-downcase": (1, None), "string-upcase": (1, None), "prin1-to-string": (1, 1),
    "princ-to-string": (1, 1), "concatenate": (1, None), "subseq": (2, 3), "nreverse": (1, 1),
    "reverse": (1, 1), "make-list": (1, None), "make-array": (1, None), "assoc": (2, None),
    "member": (2, None), "position": (2, None), "remove-if": (2, None), "count": (2, None),
    "values": (0, None), "identity": (1, 1), "write-string": (1, None), "digit-char-p": (1, 2),
    "alphanumericp": (1, 1), "char=": (1, None), "string=": (2, None),
}.items()}

CL_VARIABLES = set("""T NIL PI *PACKAGE* *READ-DEFAULT-FLOAT-FORMAT* *STANDARD-OUTPUT*
*ERROR-OUTPUT* MOST-POSITIVE-FIXNUM MOST-POSITIVE-DOUBLE-FLOAT""".split())

LAMBDA_KEYWORDS = {"&OPTIONAL", "&KEY", "&REST", "&BODY", "&AUX", "&ALLOW-OTHER-KEYS", "&WHOLE",
                   "&ENVIRONMENT"}
LOOP_KEYWORDS = set("""FOR AS WITH IN ON ACROSS FROM BELOW TO UPTO DOWNTO DOWNFROM ABOVE BY THEN =
BEING THE OF USING DO DOING COLLECT COLLECTING APPEND APPENDING NCONC NCONCING SUM SUMMING COUNT
COUNTING MAXIMIZE MAXIMIZING MINIMIZE MINIMIZING INTO WHILE UNTIL WHEN IF UNLESS ELSE END AND
RETURN FINALLY INITIALLY REPEAT ALWAYS NEVER THEREIS NAMED IT""".split())


def is_symbol(x):
    return isinstance(x, str) and not isinstance(x, lr.Str)


def is_number(tok):
    t = tok.replace("D", "E").replace("F", "E")
    try:
        float(t)
        return True
    except ValueError:
        pass
    if "/" in tok:
        a, _, b = tok.partition("/")
        return a.lstrip("+-").isdigit() and b.isdigit()
    return False


def self_evaluating(tok):
    return (tok.startswith(":") or tok.startswith("#\\") or is_number(tok) or tok in CL_VARIABLES)


class LambdaList:
    """required / optional counts and keywords of an ordinary lambda list"""

    def __init__(self, ll):
        self.required, self.optional, self.rest, self.keys, self.allow = 0, 0, False, set(), False
        self.vars = []
        mode = "req"
        for item in ll:
            if is_symbol(item) and item in LAMBDA_KEYWORDS:
                mode = {"&OPTIONAL": "opt", "&KEY": "key", "&REST": "rest", "&BODY": "rest",
                        "&AUX": "aux", "&ALLOW-OTHER-KEYS": "allow"}.get(item, mode)
                if item == "&ALLOW-OTHER-KEYS":
                    self.allow = True
                continue
            name = item
            supplied = None
            if isinstance(item, list) and mode == "req":  # a destructuring pattern (macros)
                self.required += 1
                self.vars += [v for v in flatten(item) if v not in LAMBDA_KEYWORDS]
                continue
            if isinstance(item, list):  # (var default supplied-p) / ((:key var) default)
                name = item[0]
                if isinstance(name, list):
                    name = name[1]
                if len(item) > 2:
                    supplied = item[2]
            if mode == "req":
                self.required += 1
            elif mode == "opt":
                self.optional += 1
            elif mode == "rest":
                self.rest = True
            elif mode == "key":
                kw = item[0][0] if isinstance(item, list) and isinstance(item[0], list) else ":" + name
                self.keys.add(kw)
            if isinstance(name, list):  # destructuring (macro lambda lists)
                self.vars += [v for v in flatten(name) if v not in LAMBDA_KEYWORDS]
            else:
                self.vars.append(name)
            if supplied:
                self.vars.append(supplied)

    def defaults(self, ll):
        for item in ll:
            if isinstance(item, list) and len(item) > 1 and not (is_symbol(item[0]) and item[0] in LAMBDA_KEYWORDS):
                yield item[1]


def flatten(x):
    if isinstance(x, list):
        for y in x:
            yield from flatten(y)
    elif is_symbol(x):
        yield x


class Linter:
    def __init__(self, files, own_packages=()):
        self.forms = []
        for f in files:
            for form in lr.read_file(f):
                self.forms.append((f, form))
        self.own = tuple(p.upper() + ":" for p in own_packages)
        self.functions = {}   # name -> LambdaList or None (unknown shape)
        self.macros = {}      # name -> raw lambda list
        self.variables = set()
        self.problems = []
        self.where = ""
        self.collect()

    # ---- pass 1: what the files define -------------------------------------------------------
    def strip(self, sym):
        for p in self.own:
            if sym.startswith(p):
                return sym[len(p):].lstrip(":")
        return sym

    def collect(self):
        for _, form in self.forms:
            self.collect_form(form)

    def collect_form(self, form):
        if not isinstance(form, list) or not form or not is_symbol(form[0]):
            return
        h = form[0]
        if h in ("FEATURE+", "FEATURE-"):
            self.collect_form(form[2])
        elif h in ("PROGN", "EVAL-WHEN"):
            for f in form[1:]:
                self.collect_form(f)
        elif h == "DEFUN" and is_symbol(form[1]):
            self.functions[form[1]] = LambdaList(form[2])
        elif h == "DEFUN":  # (defun (setf name) ...)
            pass
        elif h == "DEFGENERIC":
            self.functions[form[1]] = None
        elif h == "DEFMACRO":
            self.macros[form[1]] = form[2]
        elif h in ("DEFVAR", "DEFPARAMETER", "DEFCONSTANT"):
            self.variables.add(form[1])
        elif h == "CFFI:DEFCFUN":
            name = form[1][1] if isinstance(form[1], list) else form[1]
            args = [a for a in form[3:] if isinstance(a, list)]
            self.functions[name] = LambdaList([a[0] for a in args])
        elif h == "DEFSTRUCT":
            spec = form[1]
            name = spec[0] if isinstance(spec, list) else spec
            conc = name + "-"
            ctor = "MAKE-" + name
            if isinstance(spec, list):
                for opt in spec[1:]:
                    if isinstance(opt, list) and opt[0] == ":CONC-NAME":
                        conc = opt[1] if len(opt) > 1 and opt[1] != "NIL" else ""
                    if isinstance(opt, list) and opt[0] == ":CONSTRUCTOR" and len(opt) > 1:
                        ctor = opt[1]
            slots = [s[0] if isinstance(s, list) else s for s in form[2:] if not isinstance(s, lr.Str)]
            self.functions[ctor] = LambdaList(["&KEY"] + slots)
            self.functions[name + "-P"] = LambdaList(["X"])
            self.functions["COPY-" + name] = LambdaList(["X"])
            for s in slots:
                self.functions[conc + s] = LambdaList(["X"])
        elif h == "DEFINE-CONDITION":
            for slot in form[3]:
                if isinstance(slot, list):
                    for i in range(1, len(slot) - 1, 2):
                        if slot[i] in (":READER", ":ACCESSOR"):
                            self.functions[slot[i + 1]] = LambdaList(["X"])

    # ---- pass 2: walk ------------------------------------------------------------------------
    def problem(self, msg):
        self.problems.append("%s: %s" % (self.where, msg))

    def run(self):
        for f, form in self.forms:
            self.file = f
            self.where = f
            self.toplevel(form)
        return self.problems

    def toplevel(self, form):
        if not isinstance(form, list) or not form or not is_symbol(form[0]):
            return
        h = form[0]
        if h in ("FEATURE+", "FEATURE-"):
            return self.toplevel(form[2])
        if h in ("DEFPACKAGE", "IN-PACKAGE", "DECLAIM", "DEFSTRUCT", "DEFINE-CONDITION",
                 "CFFI:DEFCFUN", "CFFI:DEFCSTRUCT", "CFFI:DEFINE-FOREIGN-LIBRARY",
                 "CFFI:USE-FOREIGN-LIBRARY", "CFFI:LOAD-FOREIGN-LIBRARY", "CFFI:DEFCTYPE",
                 "CFFI:DEFCENUM", "EXPORT"):
            return
        if h in ("DEFUN", "DEFMACRO"):
            self.where = "%s (%s %s)" % (self.file.split("/")[-1], h.lower(), form[1])
            ll = LambdaList(form[2])
            env = set(ll.vars)
            for d in ll.defaults(form[2]):
                self.expr(d, env, set())
            self.body(form[3:], env, set())
            return
        if h in ("DEFVAR", "DEFPARAMETER", "DEFCONSTANT"):
            self.where = "%s (%s %s)" % (self.file.split("/")[-1], h.lower(), form[1])
            if len(form) > 2:
                self.expr(form[2], set(), set())
            return
        self.where = self.file.split("/")[-1] + " (top level)"
        self.expr(form, set(), set())

    def body(self, forms, env, fenv):
        for f in forms:
            if isinstance(f, list) and f and f[0] == "DECLARE":
                continue
            if isinstance(f, lr.Str):
                continue
            self.expr(f, env, fenv)

    def bind_list(self, bindings, env, fenv, sequential):
        """let-style bindings -> new env"""
        new = set(env)
        for b in bindings:
            if isinstance(b, list):
                if len(b) > 1:
                    self.expr(b[1], new if sequential else env, fenv)
                new.add(b[0])
            else:
                new.add(b)
        return new

    def expr(self, x, env, fenv):
        if isinstance(x, lr.Str):
            return
        if is_symbol(x):
            return self.variable(x, env)
        if not isinstance(x, list) or not x:
            return
        h = x[0]
        if isinstance(h, list):
            if h and h[0] == "LAMBDA":  # ((lambda ...) args)
                self.expr(h, env, fenv)
                for a in x[1:]:
                    self.expr(a, env, fenv)
            return
        if not is_symbol(h):
            return
        h = self.strip(h)
        m = getattr(self, "f_" + h.replace("-", "_").replace("*", "_STAR").replace(":", "__"), None)
        if m is not None:
            return m(x, env, fenv)
        if h in self.macros and h not in fenv:
            return self.own_macro(h, x, env, fenv)
        self.call(h, x[1:], env, fenv)

    def variable(self, s, env):
        s = self.strip(s)
        if self_evaluating(s) or s in env or s in self.variables:
            return
        if ":" in s:  # another package's variable / constant
            return
        self.problem("variable %s is bound nowhere" % s)

    def call(self, h, args, env, fenv):
        for a in args:
            self.expr(a, env, fenv)
        if h in fenv:
            return
        if h in self.functions:
            ll = self.functions[h]
            if ll is not None:
                self.check_args(h, ll, args)
            return
        if h in CL_FUNCTIONS:
            lo_hi = CL_ARITY.get(h)
            if lo_hi and (len(args) < lo_hi[0] or (lo_hi[1] is not None and len(args) > lo_hi[1])):
                self.problem("%s called with %d argument(s)" % (h, len(args)))
            return
        if ":" in h and not h.startswith(":"):
            return  # package-qualified: CFFI, ALEXANDRIA, SB-INT ... (not ours: stripped above)
        self.problem("call of %s, which is defined nowhere" % h)

    def check_args(self, h, ll, args):
        n = len(args)
        if n < ll.required:
            return self.problem("%s called with %d argument(s), needs %d" % (h, n, ll.required))
        if ll.rest:
            return
        extra = args[ll.required + ll.optional:]
        if not ll.keys and not ll.allow:
            if extra:
                self.problem("%s called with %d arguments, takes at most %d" % (h, n, ll.required + ll.optional))
            return
        if len(extra) % 2:
            return self.problem("%s: odd number of keyword arguments" % h)
        for k in extra[0::2]:
            if is_symbol(k) and k.startswith(":") and k not in ll.keys and not ll.allow:
                self.problem("%s does not take the keyword %s" % (h, k))

    def own_macro(self, h, x, env, fenv):
        """a macro of the shim: arguments in a destructuring position that binds variables are
        taken as binding lists ((var ...) ...), everything else as code"""
        ll = self.macros[h]
        args = x[1:]
        new = set(env)
        i = 0
        for p in ll:
            if is_symbol(p) and p in LAMBDA_KEYWORDS:
                break
            if i >= len(args):
                break
            if isinstance(p, list) and isinstance(args[i], list):
                # e.g. ((var ...) &body body): a binding spec - first symbols are variables
                for b in args[i] if args[i] and isinstance(args[i][0], list) else [args[i]]:
                    if isinstance(b, list) and b and is_symbol(b[0]):
                        new.add(b[0])
                        for e in b[1:]:
                            self.expr(e, env, fenv)
                    elif is_symbol(b):
                        new.add(b)
            else:
                self.expr(args[i], env, fenv)
            i += 1
        self.body(args[i:], new, fenv)

    # ---- special forms and standard macros ---------------------------------------------------
    def f_QUOTE(self, x, env, fenv):
        return

    def f_DECLARE(self, x, env, fenv):
        return

    def f_FUNCTION(self, x, env, fenv):
        f = x[1]
        if isinstance(f, list):
            return self.expr(f, env, fenv)
        f = self.strip(f)
        if f in fenv or f in self.functions or f in CL_FUNCTIONS or (":" in f and not f.startswith(":")):
            return
        self.problem("#'%s names a function that is defined nowhere" % f)

    def f_QUASIQUOTE(self, x, env, fenv):
        def unq(t):
            if isinstance(t, list) and t:
                if t[0] in ("UNQUOTE", "UNQUOTE-SPLICING"):
                    self.expr(t[1], env, fenv)
                else:
                    for u in t:
                        unq(u)
        unq(x[1])

    def f_LAMBDA(self, x, env, fenv):
        ll = LambdaList(x[1])
        for d in ll.defaults(x[1]):
            self.expr(d, env, fenv)
        self.body(x[2:], env | set(ll.vars), fenv)

    def f_LET(self, x, env, fenv):
        self.body(x[2:], self.bind_list(x[1], env, fenv, False), fenv)

    def f_LET_STAR(self, x, env, fenv):
        self.body(x[2:], self.bind_list(x[1], env, fenv, True), fenv)

    def f_FLET(self, x, env, fenv, rec=False):
        names = {d[0] for d in x[1]}
        inner = fenv | names
        for d in x[1]:
            ll = LambdaList(d[1])
            self.body(d[2:], env | set(ll.vars), inner if rec else fenv)
        self.body(x[2:], env, inner)

    def f_LABELS(self, x, env, fenv):
        self.f_FLET(x, env, fenv, rec=True)

    def f_MACROLET(self, x, env, fenv):
        self.body(x[2:], env, fenv | {d[0] for d in x[1]})

    def f_MULTIPLE_VALUE_BIND(self, x, env, fenv):
        self.expr(x[2], env, fenv)
        self.body(x[3:], env | set(x[1]), fenv)

    def f_DESTRUCTURING_BIND(self, x, env, fenv):
        self.expr(x[2], env, fenv)
        self.body(x[3:], env | {v for v in flatten(x[1]) if v not in LAMBDA_KEYWORDS}, fenv)

    def f_DOLIST(self, x, env, fenv):
        for e in x[1][1:]:
            self.expr(e, env, fenv)
        self.body(x[2:], env | {x[1][0]}, fenv)

    f_DOTIMES = f_DOLIST

    def f_DO(self, x, env, fenv):
        new = set(env) | {b[0] if isinstance(b, list) else b for b in x[1]}
        for b in x[1]:
            if isinstance(b, list):
                for e in b[1:]:
                    self.expr(e, new, fenv)
        self.body(x[2], new, fenv)
        self.body(x[3:], new, fenv)

    f_DO_STAR = f_DO

    def f_COND(self, x, env, fenv):
        for clause in x[1:]:
            self.body(clause, env, fenv)

    def f_CASE(self, x, env, fenv):
        self.expr(x[1], env, fenv)
        for clause in x[2:]:
            self.body(clause[1:], env, fenv)

    f_ECASE = f_CCASE = f_TYPECASE = f_ETYPECASE = f_CASE

    def f_HANDLER_CASE(self, x, env, fenv):
        self.expr(x[1], env, fenv)
        for clause in x[2:]:
            vars_ = set(clause[1]) if len(clause) > 1 and isinstance(clause[1], list) else set()
            self.body(clause[2:], env | vars_, fenv)

    def f_HANDLER_BIND(self, x, env, fenv):
        for b in x[1]:
            self.expr(b[1], env, fenv)
        self.body(x[2:], env, fenv)

    def f_THE(self, x, env, fenv):
        self.expr(x[2], env, fenv)

    def f_CHECK_TYPE(self, x, env, fenv):
        self.expr(x[1], env, fenv)

    def f_RETURN_FROM(self, x, env, fenv):
        self.body(x[2:], env, fenv)

    def f_SETF(self, x, env, fenv):
        for i in range(1, len(x) - 1, 2):
            place, val = x[i], x[i + 1]
            self.expr(val, env, fenv)
            if isinstance(place, list) and place and is_symbol(place[0]):
                p = self.strip(place[0])
                if p in ("VALUES",):
                    for a in place[1:]:
                        self.expr(a, env, fenv)
                else:  # (accessor args): the accessor must exist as a function (setf-able)
                    for a in place[1:]:
                        self.expr(a, env, fenv)
                    if not (p in self.functions or p in CL_FUNCTIONS or p in fenv or
                            (":" in p and not p.startswith(":"))):
                        self.problem("(setf (%s ...)): %s is defined nowhere" % (p, p))
            else:
                self.expr(place, env, fenv)

    f_SETQ = f_SETF

    def f_WITH_OUTPUT_TO_STRING(self, x, env, fenv):
        self.body(x[2:], env | {x[1][0]}, fenv)

    def f_WITH_OPEN_FILE(self, x, env, fenv):
        for e in x[1][1:]:
            self.expr(e, env, fenv)
        self.body(x[2:], env | {x[1][0]}, fenv)

    def f_WITH_STANDARD_IO_SYNTAX(self, x, env, fenv):
        self.body(x[1:], env, fenv)

    def f_CFFI__WITH_FOREIGN_OBJECTS(self, x, env, fenv):
        new = set(env)
        for b in x[1]:
            for e in b[1:]:
                self.expr(e, new, fenv)
            new.add(b[0])
        self.body(x[2:], new, fenv)

    def f_CFFI__WITH_FOREIGN_OBJECT(self, x, env, fenv):
        for e in x[1][1:]:
            self.expr(e, env, fenv)
        self.body(x[2:], env | {x[1][0]}, fenv)

    def f_CFFI__WITH_FOREIGN_STRING(self, x, env, fenv):
        self.f_CFFI__WITH_FOREIGN_OBJECT(x, env, fenv)

    def f_CFFI__FOREIGN_SLOT_VALUE(self, x, env, fenv):
        self.expr(x[1], env, fenv)  # (ptr 'type 'slot): the rest is quoted data

    def f_CFFI__FOREIGN_TYPE_SIZE(self, x, env, fenv):
        return

    def f_CFFI__USE_FOREIGN_LIBRARY(self, x, env, fenv):
        return  # (the library's NAME, not evaluated)

    f_CFFI__LOAD_FOREIGN_LIBRARY = f_CFFI__USE_FOREIGN_LIBRARY

    def f_CFFI__MEM_REF(self, x, env, fenv):
        for a in x[1:]:
            if not (isinstance(a, list) and a and a[0] == ":STRUCT"):
                self.expr(a, env, fenv)

    f_CFFI__MEM_AREF = f_CFFI__MEM_REF

    def f_SB_INT__WITH_FLOAT_TRAPS_MASKED(self, x, env, fenv):
        self.body(x[2:], env, fenv)

    def f_FEATURE_PLUS(self, x, env, fenv):
        self.expr(x[2], env, fenv)

    def f_LOOP(self, x, env, fenv):
        """extended loop: variables after FOR / AS / WITH (and INTO) are bound for the whole form;
        every other sub-form is code"""
        items = x[1:]
        new = set(env)
        i = 0
        while i < len(items):
            it = items[i]
            if is_symbol(it) and it in ("FOR", "AS", "WITH") and i + 1 < len(items):
                new |= {v for v in flatten(items[i + 1]) if v != "NIL"}
                i += 2
                continue
            if is_symbol(it) and it == "INTO" and i + 1 < len(items):
                new.add(items[i + 1])
                i += 2
                continue
            i += 1
        prev = None
        for it in items:
            if is_symbol(it):
                if it in LOOP_KEYWORDS or (prev in ("FOR", "AS", "WITH", "INTO", "NAMED")):
                    prev = it
                    continue
                self.variable(it, new)
            elif isinstance(it, list) and prev in ("FOR", "AS", "WITH"):
                pass  # a destructuring variable spec
            else:
                self.expr(it, new, fenv)
            prev = it


# the names with characters Python identifiers cannot carry
Linter.f_FEATURE_ = Linter.f_FEATURE_PLUS
setattr(Linter, "f_FEATURE+", Linter.f_FEATURE_PLUS)


def lint(files, own_packages=()):
    return Linter(files, own_packages).run()
