"""Problems without an ahead-of-time kernel are compiled at run time (hiprtc) with one
compile-time specialisation per function - expression or enumerated model - instead of running on
the run-time-dispatched generic kernels.  Both must agree with the oracle; the specialised ones
must really be in use; the generic ones stay reachable."""
import os

import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu
REL = 1e-12


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def three_peaks(n=6000, seed=5, model=pb.GAUSS, lik=pb.NORMAL):
    rng = np.random.default_rng(seed)
    th = np.array([0.5, 0.3, 1.0, 0.25, 0.04, 0.7, 0.5, 0.06, 0.9, 0.8, 0.05])
    x = np.linspace(0.0, 1.0, n)
    f = pb.model_eval_np(model, (2, 3), th, x)
    s = pb.Spec(11)
    lo, hi = np.minimum(th * 0.5, th * 1.5), np.maximum(th * 0.5, th * 1.5)
    if lik == pb.POISSON:
        s.add(model, (2, 3), range(11), x, rng.poisson(40 * f).astype(float), None, lik,
              (list(range(11)), lo, hi))
        th = th.copy()
        th[[0, 1, 2, 5, 8]] *= 40
        s.bounds[0] = (list(range(11)), np.minimum(th * 0.5, th * 1.5), np.maximum(th * 0.5, th * 1.5))
    else:
        sig = rng.uniform(0.05, 0.15, n)
        s.add(model, (2, 3), range(11), x, f + sig * rng.standard_normal(n), sig, lik,
              (list(range(11)), lo, hi))
    s.theta_star = th
    return s


CASES = [("gauss_normal", pb.GAUSS, pb.NORMAL), ("gauss_cutoff", pb.GAUSS, pb.CUTOFF),
         ("gauss_poisson", pb.GAUSS, pb.POISSON), ("lorentz_normal", pb.LORENTZ, pb.NORMAL)]


@pytest.mark.parametrize("name,model,lik", CASES, ids=[c[0] for c in CASES])
def test_specialised_and_generic_kernels_agree_with_the_oracle(mhx, orc, name, model, lik):
    s = three_peaks(model=model, lik=lik)
    op = s.oracle(orc)
    spec = s.engine(mhx, 4)
    assert "rtc[PeaksModel<2, 3, %s>" % ("true" if model == pb.LORENTZ else "false") in spec.kernel_name()
    os.environ["MHX_NO_RTC_SPECIALISE"] = "1"
    try:
        gen = s.engine(mhx, 4)
        assert gen.kernel_name().endswith("/generic")
    finally:
        os.environ.pop("MHX_NO_RTC_SPECIALISE")
    th = pb.perturbed(s.theta_star, 24, 0.02, seed=3)
    th[5] = s.theta_star * 1.6  # outside the bounds box
    a, pa = spec.logpost(th, parts=True)
    b, pb_ = gen.logpost(th, parts=True)
    for i in range(len(th)):
        ref, rp = op.logpost(th[i], parts=True)
        tol = REL * op.abs_terms(th[i])
        assert abs(pa[i, 0] - rp[0]) <= tol and abs(pb_[i, 0] - rp[0]) <= tol, (name, i)
        nv = int(((th[i] <= s.bounds[0][1]) | (th[i] >= s.bounds[0][2])).sum())
        assert abs(pa[i, 1] - rp[1]) <= 1e-5 * max(1, nv) and pa[i, 1] == pb_[i, 1]
    spec.close()
    gen.close()


def test_specialised_walk_matches_the_oracle_controller(mhx, orc):
    """whole walker-adaptive-steps runs on a specialised kernel: loop index, age and the
    positions agree with the oracle (same proposals; accept tests could differ only inside the
    1e-12 band between the two log-posteriors)"""
    s = three_peaks(n=900, seed=8)
    op = s.oracle(orc)
    C_, n = 5, 1400
    e = s.engine(mhx, C_, seed=4)
    assert "rtc[" in e.kernel_name()
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=9)
    e.init_chains(th0)
    e.adaptive_begin(n, 10.0, 1)
    e.adaptive_advance(1 << 40)
    st = e.state()
    same = 0
    for c in range(C_):
        w = orc.Walker(op, th0[c])
        w.adaptive_begin(n, 10.0, 1, seed=4, chain_id=c)
        w.adaptive_advance(1 << 40)
        assert st["age"][c] == w.age
        same += int(np.array_equal(st["theta"][c], w.last()[0]))
    assert same >= C_ - 1
    e.close()


def test_mixed_problem_every_function_specialised(mhx, orc):
    """a global fit mixing enumerated models with different shapes: one kernel, one
    specialisation per function"""
    rng = np.random.default_rng(2)
    x1, x2 = np.linspace(0, 1, 1500), np.linspace(0, 2, 700)
    th = np.array([0.4, 1.2, 0.5, 0.07, 0.3, 1.5, 0.8])   # bg A mu w | c0 c1 c2
    s = pb.Spec(7)
    sig1, sig2 = np.full(1500, 0.1), np.full(700, 0.2)
    y1 = pb.model_eval_np(pb.GAUSS, (1, 1), th[:4], x1) + 0.1 * rng.standard_normal(1500)
    y2 = pb.model_eval_np(pb.POLY, (), th[4:], x2) + 0.2 * rng.standard_normal(700)
    s.add(pb.GAUSS, (1, 1), [0, 1, 2, 3], x1, y1, sig1, pb.NORMAL)
    s.add(pb.POLY, (), [4, 5, 6], x2, y2, sig2, pb.NORMAL)
    s.theta_star = th
    e = s.engine(mhx, 3)
    assert e.kernel_name().endswith("rtc[PeaksModel<1, 1, false>:normal, PolyModel<3>:normal]")
    op = s.oracle(orc)
    t = pb.perturbed(th, 9, 0.03, seed=1)
    got = e.logpost(t)
    for i in range(len(t)):
        assert abs(got[i] - op.logpost(t[i])) <= REL * op.abs_terms(t[i])
    e.close()


def random_peaks_problem(rng, model, nbg, npk, lik, n):
    th = []
    for j in range(nbg):
        th.append([0.6, 0.25, -0.15, 0.08][j] * rng.uniform(0.8, 1.2))
    for k in range(npk):
        th += [rng.uniform(0.4, 1.5), (k + rng.uniform(0.3, 0.7)) / npk, rng.uniform(0.02, 0.2)]
    th = np.array(th if th else [0.0])
    x = np.sort(rng.uniform(0.0, 1.0, n))
    f = pb.model_eval_np(model, (nbg, npk), th, x)
    d = len(th)
    s = pb.Spec(d)
    lo, hi = np.minimum(th * 0.5, th * 1.5) - 1e-9, np.maximum(th * 0.5, th * 1.5) + 1e-9
    if lik == pb.POISSON:
        scale = 60.0 / max(f.max(), 1e-9)
        th = th.copy()
        idx = list(range(nbg)) + [nbg + 3 * k for k in range(npk)]
        th[idx] *= scale
        lam = pb.model_eval_np(model, (nbg, npk), th, x)
        lam = np.where(lam > 0.5, lam, 0.5)
        s.add(model, (nbg, npk), range(d), x, rng.poisson(lam).astype(float), None, lik,
              (list(range(d)), np.minimum(th * 0.5, th * 1.5) - 1e-9, np.maximum(th * 0.5, th * 1.5) + 1e-9))
    else:
        sig = rng.uniform(0.05, 0.2, n)
        s.add(model, (nbg, npk), range(d), x, f + sig * rng.standard_normal(n), sig, lik,
              (list(range(d)), lo, hi))
    s.theta_star = th
    return s


def test_random_shapes_against_the_oracle(mhx, orc):
    """every (background terms, peaks, line shape, likelihood) the run-time specialiser accepts,
    sampled at random: each is its own template instantiation"""
    rng = np.random.default_rng(2024)
    seen = set()
    for trial in range(14):
        model = pb.GAUSS if rng.random() < 0.6 else pb.LORENTZ
        nbg, npk = int(rng.integers(0, 5)), int(rng.integers(1, 7))
        lik = [pb.NORMAL, pb.CUTOFF, pb.POISSON][int(rng.integers(0, 3))]
        if lik == pb.POISSON and nbg == 0:
            nbg = 1  # a Poisson rate needs a positive floor between the peaks
        if (model, nbg, npk, lik) in seen or nbg + 3 * npk > 32:
            continue
        seen.add((model, nbg, npk, lik))
        n = int(rng.choice([700, 1024, 2500, 5000]))
        s = random_peaks_problem(rng, model, nbg, npk, lik, n)
        e = s.engine(mhx, 2)
        name = e.kernel_name()
        want = "rtc[PeaksModel<%d, %d, %s>" % (nbg, npk, "true" if model == pb.LORENTZ else "false")
        assert want in name or ("rtc" not in name and "generic" not in name), name  # or ahead-of-time
        op = s.oracle(orc)
        th = pb.perturbed(s.theta_star, 10, 0.02, seed=trial)
        got, parts = e.logpost(th, parts=True)
        for i in range(len(th)):
            ref, rp = op.logpost(th[i], parts=True)
            if not np.isfinite(ref):
                assert not np.isfinite(got[i]) or abs(got[i]) > 1e9, (name, i)
                continue
            assert abs(parts[i, 0] - rp[0]) <= REL * op.abs_terms(th[i]), (name, i, parts[i, 0], rp[0])
        e.close()
    assert len(seen) >= 8


CACHE_SCRIPT = r"""
import sys, time
sys.path.insert(0, %r)
import numpy as np
import lisp_mcmc_amd as mhx
rng = np.random.default_rng(77)
x = np.linspace(0, 1, 400)
y = 0.3 + np.exp(-((x - 0.47) / 0.09) ** 2) + 0.05 * rng.standard_normal(400)
text = "(lambda (x &key c a m s &allow-other-keys) (+ c (* 1.0000077 a (exp (- (expt (/ (- x m) s) 2))))))"
t0 = time.perf_counter()
w = mhx.walker_create(function=mhx.models.lisp(text), data=[x, y],
                      params=[":c", 0.3, ":a", 1.0, ":m", 0.47, ":s", 0.09], data_error=0.05)
print("RESULT %%.6f %%r %%s" %% (time.perf_counter() - t0, w.last_step().prob, w.engine.kernel_name()))
"""


def test_compiled_kernels_are_cached_on_disk(tmp_path):
    """a second PROCESS asking for the same problem loads the code object instead of compiling
    (MHX_RTC_CACHE_DIR); an entry that does not load is rebuilt, not trusted"""
    import glob
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MHX_RTC_CACHE_DIR=str(tmp_path))

    def run():
        out = subprocess.check_output([sys.executable, "-c", CACHE_SCRIPT % root], env=env).decode()
        line = [ln for ln in out.splitlines() if ln.startswith("RESULT")][0].split(" ", 3)
        return float(line[1]), line[2], line[3]
    t1, p1, k1 = run()
    files = glob.glob(str(tmp_path / "*.co"))
    assert len(files) == 1 and os.path.getsize(files[0]) > 10000 and "rtc[expr" in k1
    assert open(files[0], "rb").read(6) == b"MHXC1\n"
    t2, p2, k2 = run()
    assert p2 == p1 and k2 == k1 and len(glob.glob(str(tmp_path / "*.co"))) == 1
    assert t2 < 0.6 * t1, (t1, t2)        # loading beats compiling
    blob = open(files[0], "rb").read()
    open(files[0], "wb").write(blob[: len(blob) // 2])   # truncated: must be rebuilt
    t3, p3, _ = run()
    assert p3 == p1 and os.path.getsize(files[0]) == len(blob)
