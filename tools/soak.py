"""Soak run on the GPU box: complete default walker-adaptive-steps runs (n = 30000) of the bench
workloads on a few hundred chains each; every chain must finish untrapped and agree with the
generating parameters.  python tools/soak.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import lisp_mcmc_amd as mhx  # noqa: E402

ok = True
# SOAK_BIG=1: the batch kernels of the 16-wave family instead (4096 chains each)
# SOAK_SMALL=1: the persistent split modes (1 ... 256 walkers: one launch per portion of iterations)
BIG = os.environ.get("SOAK_BIG") == "1"
SMALL = os.environ.get("SOAK_SMALL") == "1"
for name, chains, small_l in ((("c2", 4096, False), ("c3", 4096, True), ("c4", 4096, False),
                               ("g23", 4096, False), ("poly7", 4096, False)) if BIG else
                              (("c2", 1, False), ("c2", 5, False), ("c2", 64, False), ("c2", 256, False),
                               ("c3", 16, True), ("c4", 64, False), ("g23", 24, False),
                               ("poly7", 40, False)) if SMALL else
                              (("c2", 1024, False), ("c3", 256, True), ("c4", 512, False),
                               ("g23", 512, False), ("poly7", 1024, False))):
    spec, _, _, desc = bench.synth_workload(name)
    rng = np.random.Generator(np.random.Philox(key=123))
    th0 = spec.theta_star[None, :] * (1.0 + 0.01 * rng.standard_normal((chains, spec.d)))
    e = spec.engine(mhx, chains, seed=99)
    e.init_chains(th0)
    t0 = time.perf_counter()
    l0 = np.diag(0.002 * np.abs(spec.theta_star)) if small_l else None
    e.adaptive_steps_full(30000, 10.0, 1, 0, l0)
    dt = time.perf_counter() - t0
    st, _ = e.chain_status()
    s = e.state()
    steps = int(e.counters()[0])
    rel = np.abs(np.median(s["best_theta"], axis=0) / spec.theta_star - 1.0)
    acc = float(np.median(e.acceptance(1000)))
    good = bool((st == mhx.capi.CHAIN_DONE).all() and np.isfinite(s["logpost"]).all()
                and rel.max() < (0.2 if name == "poly7" else 0.05)  # poly7's x^7 term is barely constrained
                and 0.1 < acc < 0.6)
    ok = ok and good
    print("%-6s %-34s chains %5d  steps %9d  %6.1f s  %.3g chain-steps/s  done %d trapped %d  "
          "max|rel dev| %.2e  acceptance %.2f  %s"
          % (name, e.kernel_name(), chains, steps, dt, steps / dt, int((st == 1).sum()),
             int((st == 2).sum()), rel.max(), acc, "ok" if good else "FAILED"), flush=True)
    e.close()
sys.exit(0 if ok else 1)
