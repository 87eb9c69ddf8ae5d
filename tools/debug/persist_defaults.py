"""What the engine chooses by default against MHX_NO_PERSIST=1 (rounds 1-3's rules): us per iteration."""
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
    r = "%-28s %7.2f us" % (e.kernel_name().replace("w8/gauss22_normal", "").strip() or "batch kernel", (time.perf_counter() - t0) / 1024 * 1e6)
    e.close()
    return r
sizes = [int(v) for v in sys.argv[1:]] or [4096, 8192, 20000, 50000, 100000, 1000000]
for n in sizes:
    spec = pb.two_peak(n=n, seed=3)
    for chains in (1, 8, 64, 256, 512, 1024, 1536):
        os.environ.pop("MHX_NO_PERSIST", None)
        a = run(spec, chains)
        os.environ["MHX_NO_PERSIST"] = "1"
        b = run(spec, chains)
        os.environ.pop("MHX_NO_PERSIST", None)
        print("n %7d chains %4d: %s | round-3 rules: %s" % (n, chains, a, b), flush=True)
