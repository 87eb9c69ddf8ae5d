"""How many proposals of config 2's walk are accepted in its first iterations, and how far over the
accept threshold the rejected ones are (on the CPU side: logpost of proposals is not stored, so
this looks at the chains' own log-posteriors and acceptance counts)."""
import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np
import lisp_mcmc_amd as mhx
import bench
spec, chains, b_pt, desc = bench.synth_workload("c2")
rng = np.random.Generator(np.random.Philox(key=0x5EED0002))
th0 = spec.theta_star[None, :] * (1.0 + 0.01 * rng.standard_normal((chains, spec.d)))
e = spec.engine(mhx, chains, seed=0x5EED0003)
e.init_chains(th0)
lp0 = e.state()["logpost"].copy()
e.adaptive_begin(30000, 10.0, 1)
prev = e.state()["theta"].copy()
moved_total = 0
for it in range(1, 26):
    e.adaptive_advance(1)
    th = e.state()["theta"]
    moved = (th != prev).any(axis=1)
    moved_total += moved.sum()
    prev = th.copy()
    if it in (1, 2, 5, 10, 25):
        lp = e.state()["logpost"]
        print("iteration %2d: accepted this step %.3f of chains; logpost median %.1f (start %.1f)" % (it, moved.mean(), np.median(lp), np.median(lp0)), flush=True)
print("accepted over 25 iterations: %.3f" % (moved_total / 25.0 / chains))
print(e.kernel_name(), "L[0] diag:", np.diag(e.lmatrix()[0]) if e.lmatrix().ndim == 3 else np.diag(e.lmatrix()))
print("theta*:", spec.theta_star)
