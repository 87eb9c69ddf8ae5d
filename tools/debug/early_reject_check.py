"""State of config 2's walk (4096 chains from the bench's start) after 30 and 450 iterations, as a
digest: run with and without MHX_EARLY_REJECT=1 and compare the lines."""
import hashlib, os, sys
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np
import lisp_mcmc_amd as mhx
import bench
spec, chains, b_pt, desc = bench.synth_workload("c2")
chains = 4096
rng = np.random.Generator(np.random.Philox(key=0x5EED0002))
th0 = spec.theta_star[None, :] * (1.0 + 0.01 * rng.standard_normal((chains, spec.d)))
e = spec.engine(mhx, chains, seed=0x5EED0003)
e.init_chains(th0)
e.adaptive_begin(30000, 10.0, 1)
for n in (30, 420):
    e.adaptive_advance(n)
    s = e.state()
    h = hashlib.sha256()
    for k in ("theta", "logpost", "best_theta", "best_logpost", "age", "length"):
        h.update(np.ascontiguousarray(s[k]).tobytes())
    h.update(np.ascontiguousarray(e.lmatrix()).tobytes())
    st = e.chain_status()[0]
    print("after %d more iterations: %s  accepted-ish mean age %.1f  trapped %d  kernel %s" % (
        n, h.hexdigest()[:16], float(s["age"].mean()), int((st == mhx.capi.CHAIN_FP_TRAP).sum()), e.kernel_name()), flush=True)
