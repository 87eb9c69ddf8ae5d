#!/usr/bin/env python3
"""Prints the id mhx_build_id() of a libmhx.so built from the sources as they are now would
report (csrc/Makefile: SRC_ID): "csrc:" + the first 16 hex digits of the SHA-256 over the
library's sources in the Makefile's order."""
import hashlib
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lisp-mcmc_amd", "csrc")


def source_id():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    var = {}
    for name in ("DEV", "SRC_FILES"):
        var[name] = re.search(r"^%s\s*=\s*(.*)$" % name, mk, re.M).group(1)
    files = var["SRC_FILES"].replace("$(DEV)", var["DEV"]).split()
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return "csrc:" + h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_id())
