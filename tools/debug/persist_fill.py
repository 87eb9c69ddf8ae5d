"""1024 (and 768, 1536) walkers: the persistent tile-sliced form when it may fill the GPU to the last
workgroup slot (MHX_PERSIST_FILL=100) against the default 90 % and two launches."""
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
for n in (100000, 1000000):
    spec = pb.two_peak(n=n, seed=3)
    for chains in (512, 768, 1024, 1536):
        out = []
        for fill, pts in (("90", None), ("100", None), ("100", "1"), ("90", "0")):
            os.environ["MHX_PERSIST_FILL"] = fill
            if pts is None: os.environ.pop("MHX_PERSIST_TS", None)
            else: os.environ["MHX_PERSIST_TS"] = pts
            out.append("fill %s ts %s: %s" % (fill, pts, run(spec, chains)))
        print("n %7d chains %4d: %s" % (n, chains, " | ".join(out)), flush=True)
