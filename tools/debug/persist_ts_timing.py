import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np
import lisp_mcmc_amd as mhx
import problems as pb
os.environ.pop("MHX_SPLIT", None)
os.environ["MHX_PERSIST_TS"] = "1"
# (needs a library built with -DMHX_PERSIST_TIMING [-DMHX_X_TIMING=1]: MHX_LIBRARY=...)
big = pb.two_peak(n=100000, seed=3)
for chains in (1, 64):
    e = big.engine(mhx, chains, seed=9)
    print(e.kernel_name())
    e.init_chains(pb.perturbed(big.theta_star, chains, 0.01, seed=2))
    e.adaptive_begin(30000, 10.0, 1)
    e.adaptive_advance(512)
    t0 = time.perf_counter()
    e.adaptive_advance(2048)
    print("us per iteration", (time.perf_counter() - t0) / 2048 * 1e6)
    e.close()
