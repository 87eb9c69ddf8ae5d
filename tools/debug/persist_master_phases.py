"""Cycles per step of the persistent kernels' master waves, from a -DMHX_X_TIMING=2 build
(MHX_TIMC in csrc/mhx_kernels.hpp; the sums of the last launch come back in the chains'
most-likely-parameter rows):  MHX_LIBRARY=.../libmhx_timc.so python tools/debug/persist_master_phases.py"""
import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np
import lisp_mcmc_amd as mhx
import problems as pb
os.environ["MHX_PERSIST_TS"] = "1"
names = ["vote + shut-down test (loop top)", "random numbers", "proposal (L z + theta)",
         "waiting for the sums + prior", "accept test", ":add-step", "annealing + adaptation test",
         "publishing + draws made ahead"]
big = pb.two_peak(n=100000, seed=3)
for chains in (1, 64):
    e = big.engine(mhx, chains, seed=9)
    e.init_chains(pb.perturbed(big.theta_star, chains, 0.01, seed=2))
    e.adaptive_begin(30000, 10.0, 1)
    e.adaptive_advance(512)
    a0 = int(e.state()["age"].mean())
    e.adaptive_advance(512)
    its = int(e.state()["age"].mean()) - a0
    t = e.state()["best_theta"][:, :8]
    print(e.kernel_name(), "iterations", its)
    for k in range(8):
        print("  %-36s %8.0f cycles per iteration" % (names[k], t[:, k].mean() / its))
    print("  %-36s %8.0f" % ("sum", t.sum(axis=1).mean() / its))
    e.close()
