"""Tile-sliced split mode, one launch per portion (persistent) against two launches per
iteration: us per iteration over walker counts and dataset sizes."""
import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np
import lisp_mcmc_amd as mhx
import problems as pb
for n in (100000, 1000000, 20000):
    spec = pb.two_peak(n=n, seed=3)
    for chains in (8, 16, 64, 128, 256, 512, 1024):
        row = []
        for flag in ("1", "0"):
            os.environ["MHX_PERSIST_TS"] = flag
            e = spec.engine(mhx, chains, seed=9)
            e.init_chains(pb.perturbed(spec.theta_star, chains, 0.01, seed=2))
            e.adaptive_begin(30000, 10.0, 1)
            e.adaptive_advance(256)
            t0 = time.perf_counter()
            e.adaptive_advance(1024)
            row.append((e.kernel_name(), (time.perf_counter() - t0) / 1024 * 1e6))
            e.close()
        print("n %8d chains %5d: %-42s %8.2f us | %-36s %8.2f us" % (n, chains, row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)
