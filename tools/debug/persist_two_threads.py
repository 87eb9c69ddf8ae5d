"""Two engines of 64 walkers each on ONE GPU, advanced at the same time from two host threads:
do their persistent launches get in each other's way?"""
import os, sys, time, threading
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np
import lisp_mcmc_amd as mhx
import problems as pb
spec = pb.two_peak(n=100000, seed=3)
def make(seed):
    e = spec.engine(mhx, 64, seed=seed)
    e.init_chains(pb.perturbed(spec.theta_star, 64, 0.01, seed=2))
    e.adaptive_begin(30000, 10.0, 1)
    e.adaptive_advance(64)
    return e
es = [make(9), make(10)]
print([e.kernel_name() for e in es], flush=True)
res = [None, None]
def work(i):
    t0 = time.perf_counter()
    try:
        for _ in range(8):
            es[i].adaptive_advance(512)
        res[i] = "ok %.1f ms" % ((time.perf_counter() - t0) * 1e3)
    except Exception as ex:
        res[i] = "FAILED after %.1f ms: %s" % ((time.perf_counter() - t0) * 1e3, ex)
ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
print("together: %.1f ms" % ((time.perf_counter() - t0) * 1e3), res, [e.kernel_name() for e in es], flush=True)
t0 = time.perf_counter()
for i in range(2):
    try:
        for _ in range(8):
            es[i].adaptive_advance(512)
    except Exception as ex:
        print("sequential", i, "FAILED", ex)
print("one after the other: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
