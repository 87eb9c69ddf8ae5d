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
e.adaptive_begin(30000, 10.0, 1)
e.adaptive_advance(2)
print(e.kernel_name())
