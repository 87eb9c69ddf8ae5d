"""Host-side formats either side of the path, CPU only: the data-file
reader (read-file->data, mcmc-fitting.lisp:1425-1477) and the saved-walker plist text
(mcmc-fitting.lisp:971-1001)."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def write_example_like(path, n=334, pages=1, delim="\t", crlf=True):
    """a file in the layout of the reference's example-data.xls: one header line, 9 columns,
    numbers printed like 2.000E+3, CRLF line ends (synthetic values)"""
    rng = np.random.default_rng(4)
    eol = "\r\n" if crlf else "\n"
    cols = ["Set Magnetic Field (Oe)", "Probe field", "FMR", "Phase", "X", "Y", "FMR frequency",
            "FMR Power", "Integrated X"]
    rows = []
    with open(path, "w", newline="") as f:
        f.write(delim.join(cols) + eol)
        for p in range(pages):
            for i in range(n):
                r = [2000.0 + 3 * i, 2000.0 + 2.985 * i] + list(rng.normal(0, 4e-7, 7))
                rows.append(r)
                f.write(delim.join("%.3E" % v for v in r) + eol)
            if pages > 1:
                f.write(eol)
    return np.array([[float("%.3E" % v) for v in r] for r in rows])


def test_read_file_to_data_example_layout(mhx, tmp_path):
    p = str(tmp_path / "example-like.xls")
    want = write_example_like(p)
    specs = mhx.ingest.file_to_file_specs(p)
    assert specs == {"file-lines": 335, "header-lines": 1, "data-length": 9, "data-rows": 334,
                     "num-pages": 1}
    data = mhx.read_file_to_data(p)
    assert len(data) == 9 and all(len(c) == 334 for c in data)
    assert np.array_equal(np.array(data).T, want)
    x, y = mhx.create_walker_data(data, 1, 4)          # test.lisp:15
    assert np.array_equal(x, want[:, 1]) and np.array_equal(y, want[:, 4])
    rows = mhx.read_file_to_data(p, transpose=False)
    assert len(rows) == 334 and rows[0] == list(want[0])


def test_read_file_pages_and_semicolons(mhx, tmp_path):
    p = str(tmp_path / "nv.csv")
    want = write_example_like(p, n=10, pages=3, delim=";", crlf=False)
    specs = mhx.ingest.file_to_file_specs(p, delim=";")
    assert specs["data-rows"] == 10 and specs["num-pages"] == 3 and specs["data-length"] == 9
    pages = mhx.read_file_to_data(p, delim=";", pages=True)
    assert len(pages) == 3 and len(pages[0]) == 9 and len(pages[0][0]) == 10
    assert np.array_equal(np.array(pages[2]).T, want[20:30])


def test_saved_walker_text_roundtrip_without_gpu(mhx, tmp_path):
    """the plist text of M:971-978 as a Lisp printer writes it"""
    txt = """(:FN ("MODEL-0[]") :DATA (((-4.0d0 -1.0d0 2.0d0) (0.0d0 2.0d0 5.0d0)))
 :PARAM-KEYS (:B :MUCH-BETTER-M) :STDDEV ((0.2d0 0.2d0 0.2d0))
 :LOG-LIKLIHOOD (LOG-LIKLIHOOD-NORMAL) :LOG-PRIOR (NIL)
 :WALKER (
  #S(WALKER-STEP :PROB -1.5d0 :PARAMS (:B 1.25d0 :MUCH-BETTER-M 2.0d-7))
  #S(WALKER-STEP :PROB -3.0d2 :PARAMS (:B -1.0d0 :MUCH-BETTER-M 2.0d0))
 ))"""
    p = tmp_path / "w.wlk"
    p.write_text(txt)
    full = mhx.saveload.read_saved(str(p))
    assert full["param_keys"] == ["b", "much_better_m"]
    assert full["steps"] == [(-1.5, [1.25, 2e-7]), (-300.0, [-1.0, 2.0])]
    assert full["data"] == [[[-4.0, -1.0, 2.0], [0.0, 2.0, 5.0]]] and full["stddev"] == [[0.2, 0.2, 0.2]]
    assert mhx.saveload._d(1e-7) == "1.0d-7" and mhx.saveload._d(2.5) == "2.5d0"
    assert mhx.walker_load(str(p), quiet=True) is None     # no designators: recommendations only


def _orc_cov(orc, v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros((v.shape[1], v.shape[1]))
    assert orc.lib().orc_lplist_covariance(v.ctypes.data_as(orc.f64p), v.shape[0], v.shape[1],
                                           out.ctypes.data_as(orc.f64p)) == 0
    return out


def test_lplist_covariance_python_equals_the_pinned_oracle(orc):
    """host mirror of lplist-covariance = the oracle's (which reproduces the reference's own
    known answer M:745), bit for bit, also on ill-conditioned input"""
    from importlib import import_module
    lplist_covariance = import_module("lisp-mcmc_amd.walker").lplist_covariance
    rng = np.random.default_rng(3)
    for n, d in ((5, 3), (2, 2), (333, 8), (1000, 6)):
        v = rng.standard_normal((n, d)) * 10.0 ** rng.integers(-6, 6, d) + rng.standard_normal(d) * 1e3
        assert np.array_equal(lplist_covariance(v), _orc_cov(orc, v)), (n, d)


