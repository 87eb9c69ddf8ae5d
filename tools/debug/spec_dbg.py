"""debug: a run-time specialised 3-peak / 6-peak problem under several switches"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lisp_mcmc_amd as mhx
import oraclelib as orc
import problems as pb
import test_gpu_specialise as tg

rng = np.random.default_rng(2024)
s = tg.random_peaks_problem(rng, pb.GAUSS, 3, 6, pb.NORMAL, 2500)
op = s.oracle(orc)
th = pb.perturbed(s.theta_star, 10, 0.02, seed=1)
ref = np.array([op.logpost(t) for t in th])
for env in ({}, {"MHX_NO_RTC_SPECIALISE": "1"}, {"MHX_NO_TILE_SKIP": "1"}, {"MHX_NO_RECURRENCE": "1"},
            {"MHX_FAMILY_WPG": "16"}):
    for k, v in env.items():
        os.environ[k] = v
    try:
        e = s.engine(mhx, 2)
        got = e.logpost(th)
        print(env, e.kernel_name(), "max |got-ref| =", np.max(np.abs(got - ref)), got[:3], ref[:3])
        e.close()
    finally:
        for k in env:
            os.environ.pop(k, None)
s3 = tg.three_peaks(n=900, seed=8)
op3 = s3.oracle(orc)
th3 = pb.perturbed(s3.theta_star, 10, 0.01, seed=9)
ref3 = np.array([op3.logpost(t) for t in th3])
for env in ({}, {"MHX_NO_RTC_SPECIALISE": "1"}, {"MHX_NO_TILE_SKIP": "1"}, {"MHX_NO_RECURRENCE": "1"}):
    for k, v in env.items():
        os.environ[k] = v
    try:
        e = s3.engine(mhx, 2)
        got = e.logpost(th3)
        print("3pk", env, e.kernel_name(), "max |got-ref| =", np.max(np.abs(got - ref3)))
        e.close()
    finally:
        for k in env:
            os.environ.pop(k, None)
