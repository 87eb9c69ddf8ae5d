"""Per-chain persistent split mode (MHX_SPLIT=s) against the tile-sliced persistent mode
(MHX_PERSIST_TS=1): us per iteration."""
import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np
import lisp_mcmc_amd as mhx
import problems as pb
def run(spec, chains):
    e = spec.engine(mhx, chains, seed=9)
    e.init_chains(pb.perturbed(spec.theta_star, chains, 0.01, seed=2))
    e.adaptive_begin(30000, 10.0, 1)
    e.adaptive_advance(256)
    t0 = time.perf_counter()
    e.adaptive_advance(1024)
    r = "%s %.2f us" % (e.kernel_name().replace("w8/gauss22_normal ", ""), (time.perf_counter() - t0) / 1024 * 1e6)
    e.close()
    return r
for n in (20000, 50000, 100000, 1000000):
    spec = pb.two_peak(n=n, seed=3)
    for chains in (8, 16, 32, 64):
        out = []
        os.environ.pop("MHX_SPLIT", None); os.environ["MHX_PERSIST_TS"] = "1"
        out.append(run(spec, chains))
        os.environ.pop("MHX_PERSIST_TS", None)
        for s in (24, 12, 6, 3):
            if chains * (1 + s) <= 460:
                os.environ["MHX_SPLIT"] = str(s)
                out.append(run(spec, chains))
        os.environ.pop("MHX_SPLIT", None)
        print("n %7d chains %3d: %s" % (n, chains, " | ".join(out)), flush=True)
