"""Single-walker latency probe: microseconds per step of one chain against dataset size and model
(run on the GPU box: python tools/latency_probe.py)."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lisp_mcmc_amd as mhx
import problems as pb
def run(spec, label, its=20000):
    e = spec.engine(mhx, 1, seed=1)
    e.init_chains(spec.theta_star[None, :])
    e.adaptive_begin(10**9, 10.0, 0, l_matrix=np.diag(0.01 * np.abs(spec.theta_star) + 1e-9))
    e.adaptive_advance(2000, count=False)
    e.kernel_timing(reset=True)
    t0 = time.perf_counter(); e.adaptive_advance(its, count=False); t1 = time.perf_counter()
    kt = e.kernel_timing()
    print(label, e.kernel_name(), "us/step kernel %.2f wall %.2f" % (kt["total_ms"] * 1e3 / its, (t1 - t0) * 1e6 / its))
    e.close()
for n in (5, 64, 334, 1024, 2048, 4096):
    rng = np.random.default_rng(0)
    x = np.linspace(0, 1, n); y = 1 + 2 * x + 0.1 * rng.standard_normal(n)
    s = pb.Spec(2); s.add(pb.POLY, (), [0, 1], x, y, np.full(n, 0.1), pb.NORMAL); s.theta_star = np.array([1.0, 2.0])
    run(s, "line n=%d" % n)
run(pb.lorder(), "lorder 334")
s = pb.two_peak(n=334, seed=1); run(s, "two_peak 334 (8 params, bounds)")
s = pb.two_peak(n=334, seed=1, bounds=False); run(s, "two_peak 334 (no bounds)")
