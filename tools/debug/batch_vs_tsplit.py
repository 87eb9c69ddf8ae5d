"""1100 ... 2047 walkers: the engine's choice against the batch kernels (MHX_SPLIT=0)."""
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
    e.adaptive_advance(512)
    r = "%-28s %7.2f us" % (e.kernel_name().replace("w8/gauss22_normal", "").strip() or "batch kernel", (time.perf_counter() - t0) / 512 * 1e6)
    e.close()
    return r
for n in (50000, 100000, 1000000):
    spec = pb.two_peak(n=n, seed=3)
    for chains in (1024, 1100, 1280, 1536, 1800, 2047):
        os.environ.pop("MHX_SPLIT", None)
        a = run(spec, chains)
        os.environ["MHX_SPLIT"] = "0"
        b = run(spec, chains)
        os.environ.pop("MHX_SPLIT", None)
        print("n %7d chains %4d: %s | %s" % (n, chains, a, b), flush=True)
