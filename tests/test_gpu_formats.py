"""The formats either side of the path, end to end on the GPU:
BASELINE config 1's flow (test.lisp:12-24) from a data file in the reference's layout, and a
walker-save / walker-load round trip."""
import numpy as np
import pytest

import problems as pb
from test_host_formats import write_example_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def test_config1_flow_from_file(mhx, tmp_path):
    """read-file->data -> create-walker-data 1 4 -> walker-create -> walker-adaptive-steps,
    with test.lisp's six keys and uniform data-error 1d-7"""
    s = pb.lorder()
    x, y, sig, _ = s.data[0]
    p = str(tmp_path / "example-data.xls")
    with open(p, "w", newline="") as f:
        f.write("\t".join("c%d" % i for i in range(9)) + "\r\n")
        for xi, yi in zip(x, y):
            f.write("\t".join(["%.3E" % xi, "%.6E" % xi, "0", "0", "%.9E" % yi, "0", "0", "0", "0"]) + "\r\n")
    data = mhx.read_file_to_data(p)
    assert len(data) == 9 and len(data[0]) == 334
    w = mhx.walker_create(function=mhx.models.lorder_mixed_bg(),
                          data=mhx.create_walker_data(data, 1, 4),
                          params=[":scale", 1e-5, ":linewidth", 60, ":x0", 2790, ":mix", 0.9,
                                  ":bg0", 1e-7, ":bg1", 1e-10],
                          data_error=1e-7, log_liklihood="log-liklihood-normal", seed=2)
    p0 = w.last_step().prob
    mhx.walker_adaptive_steps(w, 6000)
    best = w.most_likely_step()
    assert best.prob > p0 and w.age() > 2000
    assert abs(best.params["x0"] - 2790.0) < 5.0 and abs(best.params["linewidth"] - 60.0) < 10.0


def test_walker_save_load_roundtrip(mhx, tmp_path, golden):
    lf = golden["line_fit"]
    kw = dict(function=mhx.models.line("b", "m"), data=[lf["x"], lf["y"]], data_error=0.2)
    w = mhx.walker_create(params=[":b", -1, ":m", 2], seed=5, **kw)
    mhx.walker_many_steps(w, 300, np.diag([0.05, 0.02]))
    path = str(tmp_path / "walker001.wlk")
    mhx.walker_save(w, path, 200)
    text = open(path).read()
    assert text.startswith("(:FN (") and "(:PROB " in text and ":PARAM-KEYS (:B :M)" in text
    w2 = mhx.walker_load(path, function=mhx.models.line("b", "m"), quiet=True, seed=5)
    a = mhx.walker_get(w, get=":steps", take=200)
    b = mhx.walker_get(w2, get=":steps", take=1000)
    assert len(b) == 200 and w2.length() == 200 and w2.age() == 200
    for s1, s2 in zip(a, b):
        assert s1.prob == s2.prob and s1.params == s2.params      # repr round-trips doubles
    assert w2.last_step().prob == w.last_step().prob
    best = max(a, key=lambda s: s.prob)
    assert w2.most_likely_step().prob == best.prob
    # the loaded walker keeps walking
    mhx.walker_many_steps(w2, 50, np.diag([0.05, 0.02]))
    assert w2.length() == 250
