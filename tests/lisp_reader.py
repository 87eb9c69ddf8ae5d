"""A small Common Lisp READER (syntax only) for checking the CFFI shim without a Lisp: no Lisp
implementation exists in the build image, so tests/test_lisp_shim.py reads the shim's files with
this and checks them against include/mhx.h.  Handles what the shim uses: lists, strings,
; and #| |# comments, #\\c characters, ' ` , ,@ #' #: #+ #- and dotted pairs.  Atoms come back
as strings (symbols upper-cased like the reader does, strings as Str instances)."""


class Str(str):
    """a Lisp string literal (as opposed to a symbol / number token)"""


class ReadError(ValueError):
    pass


def read_all(text):
    pos = 0
    n = len(text)

    def skip():
        nonlocal pos
        while pos < n:
            c = text[pos]
            if c.isspace():
                pos += 1
            elif c == ";":
                while pos < n and text[pos] != "\n":
                    pos += 1
            elif text.startswith("#|", pos):
                depth, pos = 1, pos + 2
                while depth:
                    if pos >= n:
                        raise ReadError("unterminated #| comment")
                    if text.startswith("#|", pos):
                        depth, pos = depth + 1, pos + 2
                    elif text.startswith("|#", pos):
                        depth, pos = depth - 1, pos + 2
                    else:
                        pos += 1
            else:
                return

    def token():
        nonlocal pos
        start = pos
        while pos < n and not text[pos].isspace() and text[pos] not in "()'`,\";":
            if text[pos] == "\\":
                pos += 1
            elif text[pos] == "|":  # |multiple escape|
                pos += 1
                while pos < n and text[pos] != "|":
                    pos += 1
            pos += 1
        if pos == start:
            raise ReadError("empty token at %d: %r" % (pos, text[pos:pos + 20]))
        return text[start:pos].upper()

    def read():
        nonlocal pos
        skip()
        if pos >= n:
            raise ReadError("unexpected end of file")
        c = text[pos]
        if c == "(":
            pos += 1
            out = []
            while True:
                skip()
                if pos >= n:
                    raise ReadError("missing ) for a list opened before line %d"
                                    % (text[:pos].count("\n") + 1))
                if text[pos] == ")":
                    pos += 1
                    return out
                out.append(read())
        if c == ")":
            raise ReadError("unexpected ) on line %d" % (text[:pos].count("\n") + 1))
        if c == '"':
            pos += 1
            buf = []
            while True:
                if pos >= n:
                    raise ReadError("unterminated string")
                if text[pos] == "\\":
                    buf.append(text[pos + 1])
                    pos += 2
                elif text[pos] == '"':
                    pos += 1
                    return Str("".join(buf))
                else:
                    buf.append(text[pos])
                    pos += 1
        if c == "'":
            pos += 1
            return ["QUOTE", read()]
        if c == "`":
            pos += 1
            return ["QUASIQUOTE", read()]
        if c == ",":
            pos += 1
            if pos < n and text[pos] == "@":
                pos += 1
                return ["UNQUOTE-SPLICING", read()]
            return ["UNQUOTE", read()]
        if c == "#":
            d = text[pos + 1] if pos + 1 < n else ""
            if d == "'":
                pos += 2
                return ["FUNCTION", read()]
            if d == "\\":
                pos += 2
                start = pos
                pos += 1
                while pos < n and (text[pos].isalnum() or text[pos] == "-"):
                    pos += 1
                return "#\\" + text[start:pos]
            if d == ":":
                pos += 2
                return token()
            if d in "+-":
                pos += 2
                feature = read()
                form = read()
                return ["FEATURE" + d, feature, form]
            if d == "(":
                pos += 1
                return ["VECTOR"] + read()
        return token()

    forms = []
    while True:
        skip()
        if pos >= n:
            return forms
        forms.append(read())


def read_file(path):
    return read_all(open(path).read())


def walk(form):
    """every sub-list of FORM, FORM first"""
    if isinstance(form, list):
        yield form
        for f in form:
            yield from walk(f)
