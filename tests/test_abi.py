"""CPU-only checks of the drop-in boundary: libmhx.so loads (without a GPU) and exports every
entry point include/mhx.h declares; the ctypes table covers the header; compute calls fail
loudly without a device (there is no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mhx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mhx_[a-z0-9_]+)\s*\(", src)) - {"mhx_allreduce_fn"})


def test_library_exports_every_header_symbol():
    import lisp_mcmc_amd
    capi = lisp_mcmc_amd.capi
    lib = capi.lib()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
        assert n in capi.SIGNATURES, "ctypes table misses %s" % n
    assert sorted(capi.SIGNATURES) == names
    assert lib.mhx_version() == 200


def test_no_cpu_fallback_without_device():
    import lisp_mcmc_amd
    capi = lisp_mcmc_amd.capi
    n = C.c_int(-1)
    rc = capi.lib().mhx_device_count(C.byref(n))
    if rc == capi.OK and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(lisp_mcmc_amd.MhxError) as ei:
        lisp_mcmc_amd.Engine(4, 2)
    assert ei.value.code == capi.EDEVICE
    assert "no CPU path" in str(ei.value)


def test_product_never_touches_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/"""
    pkg = os.path.join(ROOT, "lisp-mcmc_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".inc", ".h", ".lisp", ".asd")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oraclelib" not in txt and "mhx_oracle" not in txt and "orc_" not in txt, f


def test_lisp_shim_binds_every_header_symbol():
    """the CFFI shim (never executed here: no Lisp in the image) at least declares a defcfun for
    every entry point and its files are paren-balanced"""
    d = os.path.join(ROOT, "lisp-mcmc_amd", "lisp")
    bound = set(re.findall(r'defcfun\s+\(?"(mhx_[a-z0-9_]+)"', open(os.path.join(d, "bindings.lisp")).read()))
    missing = [n for n in header_functions() if n not in bound]
    assert not missing, missing
    for f in sorted(os.listdir(d)):
        if not f.endswith(".lisp"):
            continue
        s, depth, i = open(os.path.join(d, f)).read(), 0, 0
        while i < len(s):
            c = s[i]
            if c == ";":
                while i < len(s) and s[i] != "\n":
                    i += 1
                continue
            if c == '"':
                i += 1
                while s[i] != '"':
                    i += 2 if s[i] == "\\" else 1
            elif c == "#" and s[i + 1] == "\\":
                i += 2
            elif c == "#" and s[i + 1] == "|":
                i = s.index("|#", i) + 1
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
                assert depth >= 0, (f, s[:i].count("\n") + 1)
            i += 1
        assert depth == 0, f


def test_the_library_in_the_tree_was_built_from_the_sources_in_the_tree():
    """mhx_build_id() = the hash of csrc/ as it is now: a stale libmhx.so (the .so travels to the
    GPU box as a built file) would otherwise pass or fail the GPU suite for the wrong sources,
    and bench.py would pair its timings with another build's instruction counts"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import source_id
    from importlib import import_module
    capi = import_module("lisp-mcmc_amd._capi")
    assert capi.lib().mhx_build_id().decode() == source_id.source_id()
