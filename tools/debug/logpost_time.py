import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, lisp_mcmc_amd as mhx, problems as pb
import bench
spec, chains, b_pt, desc = bench.synth_workload("c2")
e = spec.engine(mhx, chains, seed=1)
th = spec.theta_star[None,:]*(1+0.002*np.random.default_rng(0).standard_normal((chains, spec.d)))
e.init_chains(th)
for _ in range(5): e.logpost(th)
torch.cuda.synchronize()
t=time.time(); N=50
for _ in range(N): e.logpost(th)
torch.cuda.synchronize()
print("logpost per call %.1f us" % ((time.time()-t)/N*1e6), e.kernel_name())
