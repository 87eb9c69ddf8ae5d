"""Where a config-2 iteration's time goes: microseconds per loop iteration against the number of
tiles (fixed cost per iteration = controller + proposal + ring; slope = cost per tile).
Run on the GPU box: python tools/debug/phase_probe.py [chains]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lisp_mcmc_amd as mhx
import problems as pb

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
warm, its = 200, 200
rows = []
for n in (2048, 4096, 8192, 16384, 32768, 65536, 102400):
    rng = np.random.Generator(np.random.Philox(key=7))
    x = np.linspace(0.0, 1.0, n)
    sig = rng.uniform(0.05, 0.15, n)
    th = np.array([0.5, 0.3, 1.0, 0.3, 0.05, 0.7, 0.7, 0.08])
    y = pb.model_eval_np(pb.GAUSS, (2, 2), th, x) + sig * rng.standard_normal(n)
    s = pb.Spec(8)
    lo, hi = np.minimum(th * 0.5, th * 1.5), np.maximum(th * 0.5, th * 1.5)
    s.add(pb.GAUSS, (2, 2), range(8), x, y, sig, pb.NORMAL, (list(range(8)), lo, hi))
    s.theta_star = th
    os.environ["MHX_FAMILY_WPG"] = "16"
    e = s.engine(mhx, chains, seed=1)
    th0 = th[None, :] * (1.0 + 0.01 * rng.standard_normal((chains, 8)))
    e.init_chains(th0)
    e.adaptive_begin(30000, 10.0, 1)
    e.adaptive_advance(warm, count=False)
    e.kernel_timing(reset=True)
    e.adaptive_advance(its, count=False)
    kt = e.kernel_timing()
    us = kt["total_ms"] * 1e3 / its
    rows.append((n // 2048, us))
    print("n = %6d (%2d tiles) %s: %.2f us per iteration" % (n, n // 2048, e.kernel_name(), us), flush=True)
    e.close()
t = np.array([r[0] for r in rows], float); u = np.array([r[1] for r in rows])
b, a = np.polyfit(t, u, 1)
print("fit: %.2f us fixed + %.3f us per tile (%.0f cycles at 2.3 GHz; VALU work of a tile at 11.2 instr/point: 5734 cycles)" % (a, b, b * 2300))
