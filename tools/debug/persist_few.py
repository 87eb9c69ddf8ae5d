"""Fewer than 8 walkers: the per-chain persistent form (default) against the tile-sliced one forced
(MHX_TSPLIT=<windows>)."""
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
    r = "%s %.2f us" % (e.kernel_name().replace("w8/", ""), (time.perf_counter() - t0) / 1024 * 1e6)
    e.close()
    return r
for spec, n in ((pb.two_peak(n=20000, seed=3), 20000), (pb.two_peak(n=100000, seed=3), 100000),
                (pb.two_peak(n=1000000, seed=3), 1000000), (pb.poisson_peaks(n=200000, seed=4), 200000)):
    nwin = min((n + 2047) // 2048, 459)
    for chains in (1, 4):
        os.environ.pop("MHX_TSPLIT", None)
        a = run(spec, chains)
        os.environ["MHX_TSPLIT"] = str(nwin)
        b = run(spec, chains)
        os.environ.pop("MHX_TSPLIT", None)
        print("n %7d chains %d: %s | %s" % (n, chains, a, b), flush=True)
