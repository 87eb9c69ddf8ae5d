"""Cycles per phase of a config-2 iteration, from a -DMHX_X_TIMING build of libmhx (MHX_TIM in
csrc/mhx_kernels.hpp): MHX_LIBRARY=.../libmhx_tim.so python tools/debug/phase_cycles.py [warmup steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lisp_mcmc_amd as mhx
import bench

warm = int(sys.argv[1]) if len(sys.argv) > 1 else 200
its = int(sys.argv[2]) if len(sys.argv) > 2 else 200
uniform = len(sys.argv) > 3 and sys.argv[3] == "uniform"   # every chain at theta*, tiny proposals
workload = sys.argv[4] if len(sys.argv) > 4 else "c2"
spec, chains, b_pt, desc = bench.synth_workload(workload)
e = spec.engine(mhx, chains, seed=0x5EED)
rng = np.random.Generator(np.random.Philox(key=0x5EED0002))
if uniform:
    e.init_chains(np.tile(spec.theta_star, (chains, 1)))
    e.adaptive_begin(30000, 10.0, 1, l_matrix=np.diag(1e-6 * np.abs(spec.theta_star)))
else:
    e.init_chains(spec.theta_star[None, :] * (1.0 + 0.01 * rng.standard_normal((chains, spec.d))))
    e.adaptive_begin(30000, 10.0, 1)
e.adaptive_advance(warm, count=False)
e.kernel_timing(reset=True)
age0 = int(e.state()["age"].mean())
e.adaptive_advance(its, count=False)
ms = e.kernel_timing()["total_ms"]
t = e.state()["best_theta"][:, :8]   # [chains][8] cycle sums of the LAST launch (needs d >= 8... or fewer phases)
if os.environ.get("MHX_TIMC"):
    names = ["vote + shut-down test", "random numbers", "proposal (L z + theta)", "log-posterior (all of it)",
             "accept test", ":add-step", "annealing + adaptation test", "loop overhead"]
else:
  names = ["controller+proposal", "park+barrier before sweep", "prepare (loglik prep)", "tile 0 DMA + barrier",
           "tile compute", "tile-end wait+barrier", "accept+add_step+adapt", "butterfly+prior+unpark"]
tot = t.sum(axis=1)
its = int(e.state()["age"].mean()) - age0
print("%s: %d iterations in %.3f ms; cycle counter: %.1f MHz" % (e.kernel_name(), its, ms, tot.mean() / ms / 1e3))
for k in range(t.shape[1]):
    col = t[:, k]
    print("  %-28s mean %6.2f %%  (per iteration %8.0f cycles; min %5.2f %% max %5.2f %% over chains)" %
          (names[k], 100 * col.mean() / tot.mean(), col.mean() / its, 100 * col.min() / tot.mean(), 100 * col.max() / tot.mean()))
if chains % 16:
    e.close()
    sys.exit(0)
wg = t.reshape(-1, 16, 8)
print("  tile compute, spread inside workgroups: mean over WGs of (max - min)/mean = %.3f" %
      ((wg[:, :, 4].max(1) - wg[:, :, 4].min(1)) / wg[:, :, 4].mean(1)).mean())
print("  by wave slot (mean over workgroups): tile compute %, wait %")
for w in range(16):
    print("   wave %2d (SIMD %d): %5.1f  %5.1f" % (w, w % 4, 100 * wg[:, w, 4].mean() / tot.mean(), 100 * wg[:, w, 5].mean() / tot.mean()))
th = e.state()["theta"]
print("  widths of the chains' peaks now: w1 %.4f..%.4f  w2 %.4f..%.4f" % (np.abs(th[:, 4]).min(), np.abs(th[:, 4]).max(), np.abs(th[:, 7]).min(), np.abs(th[:, 7]).max()))
e.close()
