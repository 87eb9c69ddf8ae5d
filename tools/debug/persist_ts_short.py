"""Short datasets: the default mode against the tile-sliced persistent mode forced
(MHX_TSPLIT=<windows> MHX_PERSIST_TS=1): us per iteration."""
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
    r = (e.kernel_name(), (time.perf_counter() - t0) / 1024 * 1e6)
    e.close()
    return r
for n in (4096, 8192, 20000, 50000):
    spec = pb.two_peak(n=n, seed=3)
    nwin = (n + 2047) // 2048
    for chains in (8, 32, 128, 256, 512, 1024, 2048):
        os.environ.pop("MHX_TSPLIT", None); os.environ.pop("MHX_PERSIST_TS", None)
        a = run(spec, chains)
        os.environ["MHX_TSPLIT"] = str(nwin); os.environ["MHX_PERSIST_TS"] = "1"
        b = run(spec, chains)
        print("n %6d chains %5d: %-40s %8.2f us | %-40s %8.2f us" % (n, chains, a[0], a[1], b[0], b[1]), flush=True)
