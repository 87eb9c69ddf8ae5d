"""Seeded synthetic problems, built identically on the engine (C ABI) and on the oracle."""
import numpy as np

POLY, GAUSS, LORENTZ, LORDER, EXPDECAY, SINUS, PVOIGT2 = 0, 1, 2, 3, 4, 5, 6
NORMAL, CUTOFF, POISSON = 0, 1, 2


class Spec:
    """d, functions [(model, shape, idx)], datasets [(x, y, sigma, lik)], bounds [(idx, lo, hi)]"""

    def __init__(self, d):
        self.d = d
        self.fns, self.data, self.bounds = [], [], []
        self.theta_star = None

    @property
    def K(self):
        return len(self.fns)

    def add(self, model, shape, idx, x, y, sigma, lik, bounds=None):
        self.fns.append((model, tuple(shape), list(idx)))
        self.data.append((np.asarray(x, float), np.asarray(y, float),
                          None if sigma is None else np.asarray(sigma, float), lik))
        self.bounds.append(bounds)
        return self

    def apply(self, target):
        """target: lisp_mcmc_amd.Engine or oraclelib.Problem (same method names)"""
        for k, ((model, shape, idx), (x, y, s, lik), b) in enumerate(
                zip(self.fns, self.data, self.bounds)):
            if hasattr(target, "set_function") and hasattr(target, "logpost_many"):
                target.set_function(k, model, shape, idx)            # oracle Problem
                target.set_dataset(k, x, y, s, lik)
            else:
                target.set_function(k, model, shape, idx)            # Engine
                target.set_dataset(k, x, y, s, lik)
            if b is not None:
                target.set_bounds(k, b[0], b[1], b[2])
        return target

    def oracle(self, orc, logfact_double=False):
        p = orc.Problem(self.d, self.K)
        self.apply(p)
        p.set_logfact_double(logfact_double)
        return p

    def engine(self, mod, n_chains, **kw):
        e = mod.Engine(n_chains, self.d, self.K, **kw)
        self.apply(e)
        return e


def model_eval_np(model, shape, p, x):
    x = np.asarray(x, float)
    if model == POLY:
        f = np.full_like(x, p[-1])
        for c in p[-2::-1]:
            f = f * x + c
        return f
    if model in (GAUSS, LORENTZ):
        nbg, npk = shape
        f = np.zeros_like(x)
        if nbg:
            f = np.full_like(x, p[nbg - 1])
            for c in p[nbg - 2::-1] if nbg > 1 else []:
                f = f * x + c
        for k in range(npk):
            A, mu, w = p[nbg + 3 * k: nbg + 3 * k + 3]
            t = (x - mu) / w
            f = f + (A * np.exp(-t * t) if model == GAUSS else A / (1 + t * t))
        return f
    raise NotImplementedError


def two_peak(n=4000, seed=1, lik=NORMAL, bounds=True, sigma_lo=0.05, sigma_hi=0.15):
    """BASELINE config 2's problem: 2 Gaussian peaks + linear background, 8 params, weighted
    normal likelihood, x = linspace(0,1), per-point sigma in U(0.05,0.15)."""
    rng = np.random.default_rng(seed)
    th = np.array([0.5, 0.3, 1.0, 0.3, 0.05, 0.7, 0.7, 0.08])  # b0 b1 A1 mu1 w1 A2 mu2 w2
    x = np.linspace(0.0, 1.0, n)
    sig = rng.uniform(sigma_lo, sigma_hi, n)
    y = model_eval_np(GAUSS, (2, 2), th, x) + sig * rng.standard_normal(n)
    s = Spec(8)
    b = None
    if bounds:
        lo, hi = np.minimum(th * 0.5, th * 1.5), np.maximum(th * 0.5, th * 1.5)
        b = (list(range(8)), lo, hi)
    s.add(GAUSS, (2, 2), range(8), x, y, sig, lik, b)
    s.theta_star = th
    return s


def piecewise_x(n, seed=0, jitter_window=True):
    """x of `n` points as a real instrument file often is: three scans of different steps laid
    end to end (each a grid on its own; the windows that hold a junction are not), one 2048-point
    window in the middle re-sampled with jitter, and a last stretch on the first scan's step
    again.  Ascending, inside [0, 1]."""
    rng = np.random.default_rng(seed)
    cuts = [0, int(0.37 * n), int(0.61 * n), int(0.83 * n), n]
    rel = [1.0, 0.6, 1.7, 1.0]                 # relative steps of the four stretches
    units = sum((cuts[i + 1] - cuts[i]) * rel[i] for i in range(4))
    h0 = 1.0 / units
    x = np.empty(n)
    at = 0.0
    for i in range(4):
        m = cuts[i + 1] - cuts[i]
        x[cuts[i]:cuts[i + 1]] = at + h0 * rel[i] * np.arange(m)
        at = x[cuts[i + 1] - 1] + h0 * rel[i] * 0.5        # (a junction: half a step)
    if jitter_window and n >= 8 * 2048:
        w = (cuts[1] // 2048) // 2                          # a window inside the first scan
        lo = w * 2048
        x[lo + 1:lo + 2047] += h0 * 0.2 * rng.uniform(-1, 1, 2046)
    return x


def two_peak_piecewise(n=30000, seed=1, lik=NORMAL):
    """config 2's problem on a piecewise-uniform x (piecewise_x): per-window grids"""
    s = two_peak(n=n, seed=seed, lik=lik)
    rng = np.random.default_rng(seed + 1000)
    x = piecewise_x(n, seed)
    _, _, sig, lk = s.data[0]
    y = model_eval_np(GAUSS, (2, 2), s.theta_star, x) + sig * rng.standard_normal(n)
    s.data[0] = (x, y, sig, lk)
    return s


def poisson_peaks(n=3000, seed=2, npk=5):
    """BASELINE config 3's problem: lambda = bg + 5 Gaussian peaks, counts ~ Poisson"""
    rng = np.random.default_rng(seed)
    th = [20.0]
    for k in range(npk):
        th += [rng.uniform(40, 150), (k + 0.5) / npk, rng.uniform(0.02, 0.05)]
    th = np.array(th)
    x = np.linspace(0.0, 1.0, n)
    lam = model_eval_np(GAUSS, (1, npk), th, x)
    y = rng.poisson(lam).astype(float)
    s = Spec(1 + 3 * npk)
    lo, hi = th * 0.5, th * 1.5
    s.add(GAUSS, (1, npk), range(1 + 3 * npk), x, y, None, POISSON,
          (list(range(1 + 3 * npk)), lo, hi))
    s.theta_star = th
    return s


def pvoigt_eval(p, x):
    A, b0, b1, mu1, w1, e1, mu2, w2, e2, rho, c2 = p
    u1, u2 = (x - mu1) / w1, (x - mu2) / w2
    pv1 = e1 / (1 + u1 * u1) + (1 - e1) * np.exp(-u1 * u1)
    pv2 = e2 / (1 + u2 * u2) + (1 - e2) * np.exp(-u2 * u2)
    return b0 + b1 * x + c2 * x * x + A * (pv1 + rho * pv2)


def global_fit(n_each=500, n_sets=8, seed=3):
    """BASELINE config 4's shape: n_sets datasets x functions sharing one vector of
    8 + 3*n_sets parameters (8 shared shape terms, 3 local per dataset); the same bounds
    prior is listed once per function (counted n_sets times, M:1069)."""
    rng = np.random.default_rng(seed)
    shared = np.array([0.35, 0.06, 0.4, 0.65, 0.09, 0.6, 0.8, 0.1])  # mu1 w1 eta1 mu2 w2 eta2 rho c2
    d = 8 + 3 * n_sets
    th = np.zeros(d)
    th[:8] = shared
    s = Spec(d)
    lo = np.full(d, -10.0)
    hi = np.full(d, 10.0)
    specs = []
    for k in range(n_sets):
        A, b0, b1 = rng.uniform(0.5, 2.0), rng.uniform(-0.2, 0.2), rng.uniform(-0.3, 0.3)
        th[8 + 3 * k: 11 + 3 * k] = (A, b0, b1)
        # local order of PVOIGT2: A b0 b1 mu1 w1 eta1 mu2 w2 eta2 rho c2
        idx = [8 + 3 * k, 9 + 3 * k, 10 + 3 * k, 0, 1, 2, 3, 4, 5, 6, 7]
        specs.append(idx)
    for k in range(n_sets):
        idx = specs[k]
        x = np.sort(rng.uniform(0, 1, n_each))
        sig = rng.uniform(0.03, 0.08, n_each)
        y = pvoigt_eval(th[idx], x) + sig * rng.standard_normal(n_each)
        s.add(PVOIGT2, (), idx, x, y, sig, NORMAL, (list(range(d)), lo, hi))
    s.theta_star = th
    return s


def line_fit(golden, sigma=None):
    lf = golden["line_fit"]
    s = Spec(2)
    sg = golden["line_fit_initial_logpost_sigma_single"]["sigma"] if sigma is None else sigma
    s.add(POLY, (), [0, 1], lf["x"], lf["y"], np.full(len(lf["x"]), sg), NORMAL)
    s.theta_star = np.array(lf["theta"])
    return s


def lorder(n=334, seed=5):
    """BASELINE config 1's shape (test.lisp:12-21): 334 points, 6 params, uniform sigma 1e-7"""
    rng = np.random.default_rng(seed)
    th = np.array([1e-5, 60.0, 2790.0, 0.9, 1e-7, 1e-10])
    x = np.linspace(2000.0, 2997.0, n)
    u = (x - th[2]) / th[1]
    q = 1 + u * u
    y = th[0] * (np.cos(th[3]) * (-2 * u) + np.sin(th[3]) * (1 - u * u)) / (q * q) + th[4] + th[5] * x
    y = y + 1e-7 * rng.standard_normal(n)
    s = Spec(6)
    s.add(LORDER, (), range(6), x, y, np.full(n, 1e-7), NORMAL,
          ([0, 1, 2, 3], [-1e-3, 1.0, 2000.0, -10.0], [1e-3, 500.0, 3000.0, 10.0]))
    s.theta_star = th
    return s


def perturbed(theta_star, n, scale=0.01, seed=11):
    rng = np.random.default_rng(seed)
    return theta_star[None, :] * (1.0 + scale * rng.standard_normal((n, theta_star.size)))
