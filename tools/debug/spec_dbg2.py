"""debug: is the 3-peak (P = 4, w8) direct path wrong by order, by solo mode, or at random?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lisp_mcmc_amd as mhx
import oraclelib as orc
import problems as pb
import test_gpu_specialise as tg

def run(tag, s, chains, reps=3, env=None):
    env = env or {}
    for k, v in env.items():
        os.environ[k] = v
    try:
        op = s.oracle(orc)
        th = pb.perturbed(s.theta_star, 10, 0.01, seed=9)
        ref = np.array([op.logpost(t) for t in th])
        e = s.engine(mhx, chains)
        errs = []
        for r in range(reps):
            got = e.logpost(th)
            errs.append(float(np.max(np.abs(got - ref))))
        print(tag, env, e.kernel_name(), "chains", chains, "errs", ["%.3g" % v for v in errs], flush=True)
        e.close()
    finally:
        for k in env:
            os.environ.pop(k, None)

s3 = tg.three_peaks(n=900, seed=8)
for i in range(4):
    run("A%d" % i, s3, 2)
run("B", s3, 16)
run("B64", s3, 64)
run("Cnorec", s3, 2, env={"MHX_NO_RECURRENCE": "1"})
run("Cnorec16", s3, 16, env={"MHX_NO_RECURRENCE": "1"})
run("Cnoskip", s3, 2, env={"MHX_NO_TILE_SKIP": "1"})
run("Cw16", s3, 2, env={"MHX_FAMILY_WPG": "16", "MHX_NO_RECURRENCE": "1"})
run("Cgen", s3, 2, env={"MHX_NO_RTC_SPECIALISE": "1"})
s5 = pb.poisson_peaks(n=900, seed=3)
run("P5", s5, 2)
run("P5norec", s5, 2, env={"MHX_NO_RECURRENCE": "1"})
s3b = tg.three_peaks(n=3000, seed=8)
run("D3000", s3b, 2)
run("D3000norec", s3b, 2, env={"MHX_NO_RECURRENCE": "1"})
