#!/usr/bin/env python3
"""Wall time of COMPLETE (walker-adaptive-steps w) runs (n = 30000, :prob-settle) of many chains:
the walks end at very different loop indices, so what is measured is the engine's handling of
finished chains (mhx_engine.cpp: compact_slots) as much as the kernel.
    python3 tools/complete_run.py [--workload c2] [--chains 16384] [--adapt faithful|pooled]
prints one line per setting (slot repacking on / off)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import lisp_mcmc_amd as mhx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c2")
ap.add_argument("--chains", type=int, default=16384)
ap.add_argument("--adapt", default="pooled", choices=["faithful", "pooled"])
ap.add_argument("--settings", default="compact,nocompact")
args = ap.parse_args()
spec, _, _, desc = bench.synth_workload(args.workload)
rng = np.random.Generator(np.random.Philox(key=123))
th0 = spec.theta_star[None, :] * (1.0 + 0.01 * rng.standard_normal((args.chains, spec.d)))
mode = mhx.capi.ADAPT_POOLED if args.adapt == "pooled" else mhx.capi.ADAPT_FAITHFUL
ref = None
for setting in args.settings.split(","):
    if setting == "nocompact":
        os.environ["MHX_NO_COMPACT"] = "1"
    else:
        os.environ.pop("MHX_NO_COMPACT", None)
    e = spec.engine(mhx, args.chains, seed=99, adapt_mode=mode)
    e.init_chains(th0)
    t0 = time.perf_counter()
    e.adaptive_steps_full(30000, 10.0, 1, 0, None)
    dt = time.perf_counter() - t0
    st, _ = e.chain_status()
    s = e.state()
    steps = int(e.counters()[0])
    same = "" if ref is None else ("  same results as the first setting: %s"
                                   % bool(np.array_equal(ref["theta"], s["theta"]) and
                                          np.array_equal(ref["age"], s["age"])))
    ref = ref or s
    print("%-4s %-9s %-9s chains %6d  steps %10d  %7.2f s  %.3g chain-steps/s  done %d trapped %d  "
          "ages %d..%d%s" % (args.workload, args.adapt, setting, args.chains, steps, dt, steps / dt,
                            int((st == 1).sum()), int((st == 2).sum()), int(s["age"].min()),
                            int(s["age"].max()), same), flush=True)
    e.close()
