"""MHX_EARLY_REJECT=1 where the engine chooses a split mode (the define is in the program, the
threshold never set): the walk must equal the one without, in every mode."""
import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np
import lisp_mcmc_amd as mhx
import problems as pb
ok = True
for n, chains in ((30000, 20), (100000, 3), (9001, 5), (30000, 300), (50000, 1100)):
    s = pb.two_peak(n=n, seed=7)
    th0 = pb.perturbed(s.theta_star, chains, 0.01, seed=2)
    res = []
    for flag in ("1", None):
        if flag: os.environ["MHX_EARLY_REJECT"] = flag
        else: os.environ.pop("MHX_EARLY_REJECT", None)
        e = s.engine(mhx, chains, seed=11)
        name = e.kernel_name()
        e.init_chains(th0)
        e.adaptive_begin(900, 10.0, 1)
        e.adaptive_advance(1 << 40)
        res.append((name, e.state(), e.lmatrix(), e.chain_status()[0]))
        e.close()
    same = all(np.array_equal(res[0][1][k], res[1][1][k]) for k in ("theta", "logpost", "best_theta", "age", "length")) \
        and np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
    ok = ok and same
    print(n, chains, res[0][0], "|", res[1][0], "->", "same" if same else "DIFFERENT", flush=True)
sys.exit(0 if ok else 1)
