"""Split mode (csrc/mhx_kernels.hpp): with few chains and long datasets one chain's likelihood
sums are spread over many workgroups (two small launches per iteration).  It must walk like the
batch kernels: same proposals, same controller, log-posteriors equal to rounding - and be
deterministic."""
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


def engine(mhx, spec, chains, split, **kw):
    old = os.environ.get("MHX_SPLIT")
    if split is None:
        os.environ.pop("MHX_SPLIT", None)
    else:
        os.environ["MHX_SPLIT"] = str(split)
    try:
        e = spec.engine(mhx, chains, **kw)
        name = e.kernel_name()  # finalises under this setting
    finally:
        if old is None:
            os.environ.pop("MHX_SPLIT", None)
        else:
            os.environ["MHX_SPLIT"] = old
    return e, name


def walk(e, th0, n, l0=None, plain=False):
    e.init_chains(th0)
    if plain:
        e.many_steps(n, l0)
    else:
        e.adaptive_begin(n, 10.0, 1, l_matrix=l0)
        e.adaptive_advance(1 << 40)
    return e.state(), e.chain_status()[0], e.lmatrix()


CASES = [
    ("two_peak", lambda: pb.two_peak(n=30000, seed=3), None),
    ("poisson", lambda: pb.poisson_peaks(n=24000, seed=4), 0.002),
    ("global_fit", lambda: pb.global_fit(n_each=20000, n_sets=3, seed=5), None),
]


@pytest.mark.parametrize("name,make,lscale", CASES, ids=[c[0] for c in CASES])
def test_split_walks_like_the_batch_kernels(mhx, orc, name, make, lscale):
    s = make()
    C_, n = 3, 1300
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=2)
    l0 = None if lscale is None else np.diag(lscale * np.abs(s.theta_star))
    batch, nb = engine(mhx, s, C_, 0, seed=9)
    split, ns = engine(mhx, s, C_, None, seed=9)
    assert "split" not in nb and "split x" in ns, (nb, ns)
    sb, stb, Lb = walk(batch, th0, n, l0)
    ss, sts, Ls = walk(split, th0, n, l0)
    assert np.array_equal(stb, sts) and (sts == mhx.capi.CHAIN_DONE).all()
    assert np.array_equal(sb["age"], ss["age"]) and np.array_equal(sb["length"], ss["length"])
    same = sum(int(np.array_equal(sb["theta"][c], ss["theta"][c])) for c in range(C_))
    assert same >= C_ - 1      # an accept test can differ only inside the rounding band
    op = s.oracle(orc)
    for c in range(C_):
        ref = op.logpost(ss["theta"][c])
        assert abs(ss["logpost"][c] - ref) <= REL * op.abs_terms(ss["theta"][c]) + 1e-5
    # deterministic: the partial sums are added in slot order
    split2, _ = engine(mhx, s, C_, None, seed=9)
    s2, _, L2 = walk(split2, th0, n, l0)
    for k in ("theta", "logpost", "age"):
        assert np.array_equal(ss[k], s2[k]), k
    assert np.array_equal(Ls, L2)
    for e in (batch, split, split2):
        e.close()


def ts_engine(mhx, spec, chains, ts, **kw):
    """an engine finalised under MHX_TSPLIT=<ts> (None: the engine's own choice)"""
    old = {k: os.environ.get(k) for k in ("MHX_TSPLIT", "MHX_SPLIT")}
    os.environ.pop("MHX_SPLIT", None)
    if ts is None:
        os.environ.pop("MHX_TSPLIT", None)
    else:
        os.environ["MHX_TSPLIT"] = str(ts)
        if ts == 0:
            os.environ["MHX_SPLIT"] = "0"
    try:
        e = spec.engine(mhx, chains, **kw)
        name = e.kernel_name()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return e, name


TS_CASES = [   # (..., chains, slices asked for: None = the engine's own choice)
    ("two_peak", lambda: pb.two_peak(n=30000, seed=3), None, 20, None),
    ("two_peak_ragged", lambda: pb.two_peak(n=9001, seed=13), None, 9, 4),  # 5 windows, a short last one
    ("poisson", lambda: pb.poisson_peaks(n=24000, seed=4), 0.002, 11, None),
    ("global_fit", lambda: pb.global_fit(n_each=7000, n_sets=3, seed=5), None, 16, None),
]


@pytest.mark.parametrize("name,make,lscale,chains,ask", TS_CASES, ids=[c[0] for c in TS_CASES])
def test_tile_sliced_split_walks_like_the_batch_kernels(mhx, orc, name, make, lscale, chains, ask):
    """k_split_tsweep: groups of chains on slices of whole windows (the engine's choice from 8
    chains on, below the batch kernels' range): same proposals, same controller, log-posteriors
    equal to rounding, deterministic; slices that do not exist (fewer windows than slices in one
    function of a global fit), a ragged last slice, a last group that is not full"""
    s = make()
    n = 900
    th0 = pb.perturbed(s.theta_star, chains, 0.01, seed=2)
    l0 = None if lscale is None else np.diag(lscale * np.abs(s.theta_star))
    batch, nb = ts_engine(mhx, s, chains, 0, seed=9)
    ts, nt = ts_engine(mhx, s, chains, ask, seed=9)
    assert "split" not in nb and "tsplit x" in nt, (nb, nt)
    sb, stb, Lb = walk(batch, th0, n, l0)
    ss, sts, Ls = walk(ts, th0, n, l0)
    assert np.array_equal(stb, sts) and (sts == mhx.capi.CHAIN_DONE).all()
    assert np.array_equal(sb["age"], ss["age"]) and np.array_equal(sb["length"], ss["length"])
    same = sum(int(np.array_equal(sb["theta"][c], ss["theta"][c])) for c in range(chains))
    assert same >= chains - 2      # an accept test can differ only inside the rounding band
    op = s.oracle(orc)
    for c in range(chains):
        ref = op.logpost(ss["theta"][c])
        assert abs(ss["logpost"][c] - ref) <= REL * op.abs_terms(ss["theta"][c]) + 1e-5
    ts2, _ = ts_engine(mhx, s, chains, ask, seed=9)
    s2, _, L2 = walk(ts2, th0, n, l0)
    for k in ("theta", "logpost", "age"):
        assert np.array_equal(ss[k], s2[k]), k
    assert np.array_equal(Ls, L2)
    # plain steps (mhx_many_steps: no controller) go through the same launches
    lp = np.diag(0.003 * np.abs(s.theta_star))
    batch.many_steps(60, lp)
    ts.many_steps(60, lp)
    pb_, ps_ = batch.state(), ts.state()
    assert np.array_equal(pb_["age"], ps_["age"]) and (ps_["age"] == ss["age"] + 60).all()
    assert sum(int(np.array_equal(pb_["theta"][c], ps_["theta"][c])) for c in range(chains)) >= chains - 3
    # ... and with another number of slices the same walk again, to rounding
    ts3, n3 = ts_engine(mhx, s, chains, 3, seed=9)
    assert "tsplit x3" in n3 or "tsplit x2" in n3, n3
    s3, st3, _ = walk(ts3, th0, n, l0)
    assert np.array_equal(st3, sts)
    assert sum(int(np.array_equal(s3["theta"][c], ss["theta"][c])) for c in range(chains)) >= chains - 2
    for e in (batch, ts, ts2, ts3):
        e.close()


def test_tile_sliced_split_repacks_the_chains_still_walking(mhx, orc, monkeypatch):
    """complete walker-adaptive-steps runs end at different loop indices (:prob-settle): between
    portions of launches the chains still walking are packed into fewer groups and the functions
    cut into more slices (compact_tsplit).  The sums are then grouped differently - results to
    rounding - so what is asserted is what a run must deliver either way: every chain done, the
    same ages as without repacking for the chains whose accept tests never sat in the rounding
    band, final log-posteriors equal to the oracle's at the final parameters, parameters
    recovered; and a second run on the same engine starts from the original slicing again."""
    s = pb.two_peak(n=30000, seed=31)
    chains, n = 72, 7000
    th0 = pb.perturbed(s.theta_star, chains, 0.01, seed=8)
    res = {}
    for label, nc in (("packed", None), ("fixed", "1")):
        if nc is None:
            monkeypatch.delenv("MHX_NO_COMPACT", raising=False)
        else:
            monkeypatch.setenv("MHX_NO_COMPACT", nc)
        e, name = ts_engine(mhx, s, chains, None, seed=17)
        assert "tsplit x" in name, name
        e.init_chains(th0)
        e.adaptive_begin(n, 10.0, 1)
        left = 1
        while left:
            left = e.adaptive_advance(600)     # portions: the engine looks at the states between them
        st, stat = e.state(), e.chain_status()[0]
        if label == "packed":                  # ... and once more on the same engine
            e.init_chains(th0)
            e.adaptive_begin(n, 10.0, 1)
            e.adaptive_advance(1 << 40)
            st2, stat2 = e.state(), e.chain_status()[0]
            assert (stat2 == mhx.capi.CHAIN_DONE).all()
            assert np.array_equal(st2["age"] > 0, st["age"] > 0)
        res[label] = (st, stat)
        e.close()
    op = s.oracle(orc)
    for label, (st, stat) in res.items():
        assert (stat == mhx.capi.CHAIN_DONE).all(), label
        assert len(set(st["age"].tolist())) > 3, label          # they did end at different times
        for c in range(0, chains, 5):
            ref = op.logpost(st["theta"][c])
            assert abs(st["logpost"][c] - ref) <= REL * op.abs_terms(st["theta"][c]) + 1e-5, (label, c)
        rel = np.abs(np.median(st["best_theta"], axis=0) / s.theta_star - 1.0)
        assert rel.max() < 0.05, (label, rel)
    same = int((res["packed"][0]["age"] == res["fixed"][0]["age"]).sum())
    assert same >= chains // 2, same


def test_tile_sliced_split_in_the_16_wave_family(mhx, orc, monkeypatch):
    """groups of 16 chains on 2048-point tiles (MHX_FAMILY_WPG=16; the engine's own choice below
    4096 chains is the 8-wave family): the same walk to rounding"""
    s = pb.two_peak(n=30000, seed=21)
    chains, n = 40, 500
    th0 = pb.perturbed(s.theta_star, chains, 0.01, seed=6)
    res = {}
    for fam in ("8", "16"):
        monkeypatch.setenv("MHX_FAMILY_WPG", fam)
        e, name = ts_engine(mhx, s, chains, None, seed=3)
        assert name.startswith("w%s/" % fam) and "tsplit x" in name, name
        res[fam] = walk(e, th0, n)
        e.close()
    (s8, st8, _), (s16, st16, _) = res["8"], res["16"]
    assert np.array_equal(st8, st16) and (st16 == mhx.capi.CHAIN_DONE).all()
    assert np.array_equal(s8["age"], s16["age"])
    assert sum(int(np.array_equal(s8["theta"][c], s16["theta"][c])) for c in range(chains)) >= chains - 3
    op = s.oracle(orc)
    for c in range(chains):
        ref = op.logpost(s16["theta"][c])
        assert abs(s16["logpost"][c] - ref) <= REL * op.abs_terms(s16["theta"][c]) + 1e-5


def test_tile_sliced_split_with_a_model_compiled_at_run_time(mhx, orc):
    """the same through hiprtc (mhx_user_split_tsweep): config 2's model given as its closure
    text, 24 chains"""
    s = pb.two_peak(n=30000, seed=8)
    keys, cexpr = mhx.sexpr.lambda_to_expr(
        "(lambda (x &key b0 b1 a1 mu1 w1 a2 mu2 w2 &allow-other-keys)"
        " (+ (+ b0 (* b1 x)) (* a1 (exp (- (expt (/ (- x mu1) w1) 2))))"
        "    (* a2 (exp (- (expt (/ (- x mu2) w2) 2))))))")
    chains, n = 24, 700
    th0 = pb.perturbed(s.theta_star, chains, 0.01, seed=4)
    runs = {}
    for label, env in (("batch", {"MHX_SPLIT": "0"}), ("tsplit", {})):
        old = {k: os.environ.get(k) for k in ("MHX_SPLIT", "MHX_TSPLIT")}
        for k in old:
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            e = s.engine(mhx, chains, seed=5)
            e.set_expr_recognition(False)   # (as written: the run-time compiled sweep is the subject)
            e.set_function_expr(0, cexpr, keys, list(range(8)))
            name = e.kernel_name()
        finally:
            for k, v in old.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v
        assert ("tsplit x" in name) == (label == "tsplit") and "rtc[" in name, name
        runs[label] = walk(e, th0, n)
        e.close()
    (sb, stb, _), (ss, sts, _) = runs["batch"], runs["tsplit"]
    assert np.array_equal(stb, sts) and (sts == mhx.capi.CHAIN_DONE).all()
    assert np.array_equal(sb["age"], ss["age"])
    assert sum(int(np.array_equal(sb["theta"][c], ss["theta"][c])) for c in range(chains)) >= chains - 2
    op = s.oracle(orc)
    for c in range(chains):
        ref = op.logpost(ss["theta"][c])
        assert abs(ss["logpost"][c] - ref) <= REL * op.abs_terms(ss["theta"][c]) + 1e-5


def test_split_single_walker_expression_and_many_steps(mhx):
    rng = np.random.default_rng(6)
    n = 60000
    x = np.linspace(0, 4, n)
    sig = rng.uniform(0.05, 0.2, n)
    y = 2.0 * np.exp(-x / 1.5) + 0.3 + sig * rng.standard_normal(n)
    text = "(lambda (x &key a tau c &allow-other-keys) (+ c (* a (exp (/ (- x) tau)))))"
    params = [":a", 1.8, ":tau", 1.4, ":c", 0.35]
    ws = []
    for split in ("0", None):
        if split is None:
            os.environ.pop("MHX_SPLIT", None)
        else:
            os.environ["MHX_SPLIT"] = split
        try:
            w = mhx.walker_create(function=mhx.models.lisp(text), data=[x, y], params=params,
                                  data_error=sig, seed=2)
            w.engine.kernel_name()
        finally:
            os.environ.pop("MHX_SPLIT", None)
        ws.append(w)
    assert "split x" in ws[1].engine.kernel_name() and "split" not in ws[0].engine.kernel_name()
    p0, p1 = ws[0].last_step().prob, ws[1].last_step().prob
    assert p0 == p1                                      # init uses the batch kernel in both
    L = np.diag([0.01, 0.01, 0.005])
    for w in ws:
        mhx.walker_many_steps(w, 400, L)
    a, b = ws[0].engine.state(), ws[1].engine.state()
    assert a["age"][0] == b["age"][0] == 401
    assert abs(a["logpost"][0] - b["logpost"][0]) <= 1e-9 * abs(a["logpost"][0])
    for w in ws:
        mhx.walker_adaptive_steps_full(w, n=1500, temperature=10, auto=":prob-settle", l_matrix=L)
    ml = [mhx.walker_get(w, get=":most-likely-params") for w in ws]
    for k in ("a", "tau", "c"):
        assert abs(ml[0][k] - ml[1][k]) <= 0.02 * abs(ml[0][k])
    assert abs(ml[1]["tau"] - 1.5) < 0.1


def test_split_mode_is_chosen_by_batch_and_dataset_size(mhx):
    s_long = pb.two_peak(n=100000, seed=1)
    s_mid = pb.two_peak(n=20000, seed=1)
    s_1e6 = pb.two_peak(n=1000000, seed=1)
    s_8k = pb.two_peak(n=8192, seed=1)
    s_short = pb.two_peak(n=3000, seed=1)
    # one chain's points over many workgroups below 8 chains on datasets of fewer than 12 windows
    # (and everywhere with MHX_NO_PERSIST=1); otherwise groups of (up to) 8 chains on
    # slices of whole windows ("tsplit": about 512 workgroups in the sweep launch) - as ONE
    # persistent launch per portion of iterations where the GPU holds all its workgroups at once
    # with at least three quarters of the default slicing (fewer slices than the default only
    # up to 48 windows per slice), else as two launches per iteration;
    # the batch kernels from 256 workgroups on, for short datasets, and where only two or three
    # slices of a dataset that is not long would be left to two launches.  (Slices that would stay
    # empty are not asked for: 49 windows in 16 slices are 13 slices of 4 windows.)  Shorter datasets only as
    # a persistent launch: per chain where that fits, else tile-sliced from 4 windows on.
    for spec, chains, want in ((s_long, 1, "persistent tsplit x49"), (s_long, 4, "persistent tsplit x49"),
                               (s_mid, 1, "persistent split x4"),
                               (s_long, 16, "persistent tsplit x49"), (s_long, 256, "persistent tsplit x13"),
                               (s_long, 512, "persistent tsplit x7"), (s_long, 1024, "persistent tsplit x3"),
                               (s_1e6, 1024, " tsplit x4"), (s_1e6, 256, "persistent tsplit x15"), (s_long, 1536, None), (s_1e6, 1536, " tsplit x2"),
                               (s_long, 2048, None), (s_short, 1, None), (s_short, 64, None),
                               (s_mid, 32, "persistent split x4"), (s_mid, 256, "persistent tsplit x10"),
                               (s_mid, 1024, "persistent tsplit x3"), (s_mid, 1536, None),
                               (s_8k, 4, "persistent split x2"), (s_8k, 256, "persistent tsplit x4"),
                               (s_8k, 1024, "persistent tsplit x2"), (s_8k, 1536, None)):
        e, name = engine(mhx, spec, chains, None)
        assert (want in name) if want else ("split" not in name), (chains, name)
        e.close()
    # MHX_PERSIST_TS=0 / MHX_NO_PERSIST=1: the rules of rounds 1-3
    for var in ("MHX_PERSIST_TS", "MHX_NO_PERSIST"):
        os.environ[var] = "0" if var == "MHX_PERSIST_TS" else "1"
        try:
            for spec, chains, want in ((s_long, 256, " tsplit x13"), (s_long, 512, " tsplit x7"),
                                       (s_8k, 256, None), (s_mid, 256, " split x4"), (s_long, 4, " split x24")):
                e, name = engine(mhx, spec, chains, None)
                assert ((want in name) if want else ("split" not in name)) and "persistent t" not in name, (var, chains, name)
                e.close()
        finally:
            os.environ.pop(var, None)
    # MHX_TSPLIT=0: the per-chain split mode where it applies
    os.environ["MHX_TSPLIT"] = "0"
    try:
        e, name = engine(mhx, s_long, 256, None)
        assert " split x4" in name, name
        e.close()
    finally:
        os.environ.pop("MHX_TSPLIT", None)
    # the pooled-covariance mode (multi-GPU bench) stays on the batch kernels
    e, name = engine(mhx, s_long, 8, None, adapt_mode=mhx.capi.ADAPT_POOLED)
    assert "split" not in name
    e.close()


def _ragged_global_fit(seed=3):
    """4 functions sharing 10 parameters, lengths 1 / 700 / 40000 / 17000: in split mode the short
    ones occupy a single slot, the long ones all of them"""
    rng = np.random.default_rng(seed)
    th = np.array([0.4, 1.1, 0.5, 0.08, 0.3, 0.2, -0.1, 0.9, 0.25, 0.06])
    s = pb.Spec(10)
    for k, (n, model, shape, idx) in enumerate([
            (1, pb.POLY, (), [4, 5]), (700, pb.POLY, (), [4, 5, 6]),
            (40000, pb.GAUSS, (1, 1), [0, 1, 2, 3]), (17000, pb.LORENTZ, (1, 1), [0, 7, 8, 9])]):
        x = np.sort(rng.uniform(0, 1, n))
        sig = rng.uniform(0.05, 0.2, n)
        y = pb.model_eval_np(model, shape, th[idx], x) + sig * rng.standard_normal(n)
        s.add(model, shape, idx, x, y, sig, pb.CUTOFF if k == 1 else pb.NORMAL,
              (idx, th[idx] - 1.0, th[idx] + 1.0))
    s.theta_star = th
    return s


def test_split_run_time_specialised_ragged_global_fit(mhx, orc):
    s = _ragged_global_fit()
    C_, n = 2, 900
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=4)
    batch, nb = engine(mhx, s, C_, 0, seed=3)
    split, ns = engine(mhx, s, C_, None, seed=3)
    assert "rtc[" in ns and "split x" in ns and "split" not in nb
    l0 = np.diag(np.full(10, 0.004))
    sb, stb, _ = walk(batch, th0, n, l0)
    ss, sts, _ = walk(split, th0, n, l0)
    assert np.array_equal(stb, sts) and np.array_equal(sb["age"], ss["age"])
    op = s.oracle(orc)
    for c in range(C_):
        ref = op.logpost(ss["theta"][c])
        assert abs(ss["logpost"][c] - ref) <= REL * op.abs_terms(ss["theta"][c]) + 1e-5
    assert sum(int(np.array_equal(sb["theta"][c], ss["theta"][c])) for c in range(C_)) >= C_ - 1
    batch.close()
    split.close()


def test_split_with_expression_likelihood(mhx):
    rng = np.random.default_rng(6)
    n = 50000
    x = np.linspace(0, 1, n)
    lam = 30 + 80 * np.exp(-((x - 0.4) / 0.1) ** 2)
    y = rng.poisson(lam).astype(float)
    model = "(lambda (x &key bg a mu w &allow-other-keys) (+ bg (* a (exp (- (expt (/ (- x mu) w) 2))))))"
    lik = mhx.create_log_liklihood_function(
        "(lambda (y model error) (declare (ignore error)) (- (* y (log model)) model))")
    params = [":bg", 28.0, ":a", 85.0, ":mu", 0.41, ":w", 0.105]
    ws = []
    for split in ("0", None):
        if split is None:
            os.environ.pop("MHX_SPLIT", None)
        else:
            os.environ["MHX_SPLIT"] = split
        try:
            w = mhx.walker_create(function=mhx.models.lisp(model), data=[x, y], params=params,
                                  log_liklihood=lik, seed=3)
            w.engine.kernel_name()
        finally:
            os.environ.pop("MHX_SPLIT", None)
        ws.append(w)
    assert "expr:expr" in ws[1].engine.kernel_name() and "split x" in ws[1].engine.kernel_name()
    l0 = np.diag(0.002 * np.array(params[1::2]))
    for w in ws:
        mhx.walker_adaptive_steps_full(w, n=1000, temperature=10, auto=":prob-settle", l_matrix=l0)
    a, b = ws[0].engine.state(), ws[1].engine.state()
    assert a["age"][0] == b["age"][0]
    # one flipped accept test sends two walks apart for good, so the walks are not compared with
    # each other: each engine's stored log-posterior must be what the batch kernel (mhx_logpost)
    # and numpy say at the stored position
    for st, w in ((a, ws[0]), (b, ws[1])):
        th = st["theta"][0]
        m = th[0] + th[1] * np.exp(-((x - th[2]) / th[3]) ** 2)
        ref = float(np.sum(y * np.log(m) - m))
        assert abs(st["logpost"][0] - ref) <= 1e-11 * float(np.sum(np.abs(y * np.log(m)) + m))
        assert abs(w.engine.logpost(th[None, :])[0] - st["logpost"][0]) <= 1e-11 * abs(ref)
    assert abs(b["theta"][0][2] - 0.4) < 0.02
    # ... and the split chain must really have WALKED: with a wrong split-mode log-posterior
    # (e.g. log() of the expression reading a table the sweep kernel never staged) every
    # proposal is rejected, the stored log-posterior stays k_init's and all of the above holds
    start = np.array(params[1::2])
    for st, w in ((a, ws[0]), (b, ws[1])):
        assert not np.array_equal(st["theta"][0], start)
        acc = w.engine.acceptance(1000)[0]
        assert 0.05 < acc < 0.7, acc
        assert st["best_logpost"][0] > w.engine.logpost(start[None, :])[0]
    # (the stored log-posterior of a chain that moved is the split sweep's value at an accepted
    # proposal, so the comparison with mhx_logpost above is a comparison of the two sweeps)


@pytest.mark.parametrize("persist_ts", [False, True], ids=["two_launches", "persistent"])
def test_tile_sliced_repacking_does_not_depend_on_how_the_host_chunks_its_calls(mhx, persist_ts):
    """the tile-sliced mode regroups the partial sums when it repacks, so WHEN it
    repacks must be a function of the walk, not of the host's call sequence.  Repacking is
    considered only where the iterations since begin are a multiple of the portion length
    (launch_steps_enqueue cuts its launches there whatever the caller asked for): complete runs
    driven in chunks of 600, of 137 with a count after each, and in one call end in the same
    bits."""
    s = pb.two_peak(n=30000, seed=31)
    chains, n = 72, 7000
    th0 = pb.perturbed(s.theta_star, chains, 0.01, seed=8)
    res = []
    # (the persistent form - the default where it fits; MHX_PERSIST_TS=0: the two launches - in its
    # many short launches is also where a torn
    # read of the handshake showed in round 4: one chain in 72 x 7000 iterations went another way)
    for chunk, count in ((600, True), (137, True), (1 << 40, False), (512, False)):
        os.environ["MHX_PERSIST_TS"] = "1" if persist_ts else "0"
        try:
            e, name = ts_engine(mhx, s, chains, None, seed=17)
        finally:
            os.environ.pop("MHX_PERSIST_TS", None)
        assert "tsplit x" in name and ("persistent" in name) == persist_ts, name
        e.init_chains(th0)
        e.adaptive_begin(n, 10.0, 1)
        for _ in range(200):
            left = e.adaptive_advance(chunk, count=count)
            if (count and left == 0) or (not count and (e.chain_status()[0] != mhx.capi.CHAIN_RUNNING).all()):
                break
        st, stat = e.state(), e.chain_status()[0]
        assert (stat == mhx.capi.CHAIN_DONE).all()
        res.append(st)
        e.close()
    assert len(set(res[0]["age"].tolist())) > 3          # they did end at different times: repacked
    for other in res[1:]:
        for k in ("theta", "logpost", "best_theta", "best_logpost", "age", "length"):
            assert np.array_equal(res[0][k], other[k]), k


def test_slot_map_grows_when_the_problem_changes_the_mode(mhx):
    """a tile-sliced engine runs to completion (compact_tsplit allocates the
    slot map for its own needs), the caller swaps in a SHORT dataset - finalize picks the batch
    kernels, whose initial deal writes a map of cus * W entries - and runs again.  The map is
    sized per deal now (put_slot_map) and the identity map restored when a problem is finalised;
    before, the second run wrote 2048 ints into a 1032-int allocation."""
    long_, short = pb.two_peak(n=30000, seed=3), pb.two_peak(n=900, seed=3)
    for chains in (64, 1024):
        th0 = pb.perturbed(long_.theta_star, chains, 0.01, seed=2)
        e, name = ts_engine(mhx, long_, chains, None, seed=21)
        assert "tsplit x" in name, name
        e.init_chains(th0)
        e.adaptive_begin(3000, 10.0, 1)
        while e.adaptive_advance(512):
            pass
        x, y, sg, lik = short.data[0]
        e.set_dataset(0, x, y, sg, lik)
        # a run that was begun on the old problem is over
        with pytest.raises(mhx.MhxError):
            e.adaptive_advance(10)
        name2 = e.kernel_name()
        assert "split" not in name2, name2
        e.init_chains(th0)
        e.adaptive_begin(2500, 10.0, 1)
        while e.adaptive_advance(700):
            pass
        st = e.state()
        ref = short.engine(mhx, chains, seed=21)
        ref.init_chains(th0)
        ref.adaptive_begin(2500, 10.0, 1)
        ref.adaptive_advance(1 << 40)
        rs = ref.state()
        for k in ("theta", "logpost", "age"):
            assert np.array_equal(st[k], rs[k]), (chains, k)
        e.close()
        ref.close()


def persist_pair(mhx, make_engine):
    """(the per-chain persistent split mode - the default for a handful of chains on datasets of
    fewer than 12 windows; MHX_TSPLIT=0 keeps the longer ones of these tests in it -, the two
    launches per iteration of rounds 1-3: MHX_NO_PERSIST=1), finalised under their settings"""
    out = []
    for flag in (None, "1"):
        old = {k: os.environ.get(k) for k in ("MHX_SPLIT", "MHX_TSPLIT", "MHX_NO_PERSIST")}
        for k in old:
            os.environ.pop(k, None)
        os.environ["MHX_TSPLIT"] = "0"  # (the per-chain form is what these tests are about)
        if flag:
            os.environ["MHX_NO_PERSIST"] = flag
        try:
            e = make_engine()
            name = e.kernel_name()
        finally:
            for k, v in old.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v
        out.append((e, name))
    assert "persistent split x" in out[0][1] and "persistent" not in out[1][1] and "split x" in out[1][1], \
        [n for _, n in out]
    return out[0][0], out[1][0]


@pytest.mark.parametrize("name,make,lscale", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("chains", [1, 3, 7])
def test_persistent_split_mode_equals_the_two_launch_mode(mhx, name, make, lscale, chains):
    """ONE launch for many iterations of a handful of chains (k_persist: the
    chain's master wave and its sweep workgroups hand each other the proposal and the partial sums
    through memory, release / acquire at agent scope) - same slots, same points, same order of
    the sums as k_split_sweep + k_split_step: the same bits, over complete walker-adaptive-steps
    runs driven in uneven portions, a second run on the same engine, and plain steps."""
    s = make()
    th0 = pb.perturbed(s.theta_star, chains, 0.01, seed=2)
    l0 = None if lscale is None else np.diag(lscale * np.abs(s.theta_star))
    a, b = persist_pair(mhx, lambda: s.engine(mhx, chains, seed=9))
    res = []
    for e in (a, b):
        e.init_chains(th0)
        e.adaptive_begin(1300, 10.0, 1, l_matrix=l0)
        for portion in (1, 7, 200, 64, 1 << 40):
            if e.adaptive_advance(portion) == 0:
                break
        first = (e.state(), e.chain_status()[0], e.lmatrix())
        e.init_chains(th0)                       # ... and once more on the same engine
        e.adaptive_begin(600, 10.0, 1, l_matrix=l0)
        e.adaptive_advance(1 << 40)
        lp = np.diag(0.003 * np.abs(s.theta_star))
        e.many_steps(45, lp)
        res.append((first, (e.state(), e.chain_status()[0], e.lmatrix())))
    for (sa, sta, La), (sb, stb, Lb) in zip(res[0], res[1]):
        assert np.array_equal(sta, stb) and (sta == mhx.capi.CHAIN_DONE).all()
        for k in ("theta", "logpost", "best_theta", "best_logpost", "age", "length"):
            assert np.array_equal(sa[k], sb[k]), (name, chains, k)
        assert np.array_equal(La, Lb)
    assert (res[0][1][0]["age"] == res[0][0][0]["age"] * 0 + res[0][1][0]["age"][0]).all()
    for e in (a, b):
        e.close()


def test_persistent_split_mode_with_a_model_compiled_at_run_time_and_the_stop_flag(mhx):
    """the same through hiprtc (mhx_user_persist), a single walker - the reference's own way of
    working (test.lisp:24) - and mfit-walker-estop: a stop raised between launches ends the walk"""
    rng = np.random.default_rng(12)
    n = 60000
    x = np.linspace(0, 4, n)
    sig = rng.uniform(0.05, 0.2, n)
    y = 2.0 * np.exp(-x / 1.5) + 0.3 + sig * rng.standard_normal(n)
    text = "(lambda (x &key a tau c &allow-other-keys) (+ c (* a (exp (/ (- x) tau)))))"
    params = [":a", 1.8, ":tau", 1.4, ":c", 0.35]
    ws = []
    for flag in (None, "1"):
        for k in ("MHX_SPLIT", "MHX_TSPLIT", "MHX_NO_PERSIST"):
            os.environ.pop(k, None)
        os.environ["MHX_TSPLIT"] = "0"  # (the per-chain form)
        if flag:
            os.environ["MHX_NO_PERSIST"] = flag
        try:
            w = mhx.walker_create(function=mhx.models.lisp(text), data=[x, y], params=params,
                                  data_error=sig, seed=2)
            w.engine.kernel_name()
        finally:
            os.environ.pop("MHX_NO_PERSIST", None)
            os.environ.pop("MHX_TSPLIT", None)
        ws.append(w)
    assert "persistent split x" in ws[0].engine.kernel_name() and "rtc[" in ws[0].engine.kernel_name()
    assert "persistent" not in ws[1].engine.kernel_name()
    L = np.diag([0.01, 0.01, 0.005])
    for w in ws:
        mhx.walker_adaptive_steps_full(w, n=900, temperature=10, auto=":prob-settle", l_matrix=L)
    sa, sb = ws[0].engine.state(), ws[1].engine.state()
    for k in ("theta", "logpost", "age"):
        assert np.array_equal(sa[k], sb[k]), k
    e = ws[0].engine
    e.init_chains(np.array(params[1::2], float))
    e.adaptive_begin(100000, 10.0, 1, l_matrix=L)
    assert e.adaptive_advance(50) == 1
    e.request_stop()
    assert e.adaptive_advance(1 << 40) == 0
    assert e.chain_status()[0][0] == mhx.capi.CHAIN_STOPPED and 50 <= e.state()["age"][0] - 1 <= 50 + 16


def test_single_walker_step_time(mhx):
    """what a step of ONE walker costs on config 2's 1e5 points, both modes (printed; the
    persistent kernel must not be slower than the two launches it replaces)"""
    import time
    s = pb.two_peak(n=100000, seed=3)
    a, b = persist_pair(mhx, lambda: s.engine(mhx, 1, seed=9))
    out = []
    for e in (a, b):
        e.init_chains(s.theta_star)
        e.adaptive_begin(30000, 10.0, 1)
        e.adaptive_advance(300)
        t0 = time.perf_counter()
        e.adaptive_advance(3000)
        out.append((time.perf_counter() - t0) / 3000 * 1e6)
        e.close()
    print("single walker, 1e5 points: %.2f us per step persistent, %.2f us two launches" % tuple(out))
    assert out[0] <= out[1] * 1.05


def ts_persist_pair(mhx, spec, chains, ts, **kw):
    """(k_persist_ts - MHX_PERSIST_TS=1: wherever two slices of it fit the GPU -, the two launches of
    MHX_PERSIST_TS=0)"""
    out = []
    for flag in ("1", "0"):
        os.environ["MHX_PERSIST_TS"] = flag
        try:
            e, name = ts_engine(mhx, spec, chains, ts, **kw)
        finally:
            os.environ.pop("MHX_PERSIST_TS", None)
        out.append((e, name))
    assert "persistent tsplit x" in out[0][1] and "persistent" not in out[1][1] and "tsplit x" in out[1][1], \
        [n for _, n in out]
    return out[0][0], out[1][0]


TS_PERSIST_CASES = TS_CASES + [   # fewer walkers than a workgroup has waves: the default from 12 windows on
    ("two_peak_single_walker", lambda: pb.two_peak(n=30000, seed=3), None, 1, 15),
    ("two_peak_three_walkers", lambda: pb.two_peak(n=40000, seed=23), None, 3, 20),
]


@pytest.mark.parametrize("name,make,lscale,chains,ask", TS_PERSIST_CASES, ids=[c[0] for c in TS_PERSIST_CASES])
def test_persistent_tile_sliced_mode_equals_the_two_launch_form(mhx, name, make, lscale, chains, ask):
    """VERDICT r3 item 6, the batches of 8 ... a few hundred walkers: ONE launch per portion of
    iterations (k_persist_ts) - the 8 waves of a group's master workgroup run the chains'
    controllers, the group's sweep workgroups walk their slices through sweep() round after
    round, proposals and {sum, round} pairs handed over through memory.  Same slices, same sweep,
    same order of the sums: bit for bit the two launches per iteration (MHX_NO_PERSIST=1) under
    the same slicing - complete runs in uneven portions, chains ending at different times inside
    a group, a last group that is not full, a second run, plain steps."""
    s = make()
    th0 = pb.perturbed(s.theta_star, chains, 0.01, seed=2)
    l0 = None if lscale is None else np.diag(lscale * np.abs(s.theta_star))
    a, b = ts_persist_pair(mhx, s, chains, ask if ask is not None else 5, seed=9)
    res = []
    for e in (a, b):
        e.init_chains(th0)
        e.adaptive_begin(2600, 10.0, 1, l_matrix=l0)
        for portion in (3, 200, 1 << 40):
            if e.adaptive_advance(portion) == 0:
                break
        first = (e.state(), e.chain_status()[0], e.lmatrix())
        e.init_chains(th0)
        e.adaptive_begin(500, 10.0, 1, l_matrix=l0)
        e.adaptive_advance(1 << 40)
        e.many_steps(45, np.diag(0.003 * np.abs(s.theta_star)))
        res.append((first, (e.state(), e.chain_status()[0], e.lmatrix())))
    for (sa, sta, La), (sb, stb, Lb) in zip(res[0], res[1]):
        assert np.array_equal(sta, stb) and (sta == mhx.capi.CHAIN_DONE).all()
        for k in ("theta", "logpost", "best_theta", "best_logpost", "age", "length"):
            assert np.array_equal(sa[k], sb[k]), (name, k)
        assert np.array_equal(La, Lb)
    for e in (a, b):
        e.close()


def test_small_batch_iteration_time(mhx):
    """64, 256 and 1024 walkers on config 2's 1e5 points: microseconds per iteration, the persistent
    tile-sliced form (the default down to three quarters of the default slicing; forced here)
    against the two launches of MHX_PERSIST_TS=0 (printed; asserted: quicker at 64 and 256)"""
    import time
    big = pb.two_peak(n=100000, seed=3)
    for chains in (64, 256, 1024):
        out = []
        for flag in ("1", "0"):
            os.environ["MHX_PERSIST_TS"] = flag
            try:
                e, name = ts_engine(mhx, big, chains, None, seed=9)
            finally:
                os.environ.pop("MHX_PERSIST_TS", None)
            e.init_chains(pb.perturbed(big.theta_star, chains, 0.01, seed=2))
            e.adaptive_begin(30000, 10.0, 1)
            e.adaptive_advance(512)
            t0 = time.perf_counter()
            e.adaptive_advance(2048)
            out.append(((time.perf_counter() - t0) / 2048 * 1e6, name))
            e.close()
        print("%d walkers, 1e5 points: %.2f us per iteration %s (%.3g chain-steps/s), %.2f us %s"
              % (chains, out[0][0], out[0][1].split("/")[1], chains / out[0][0] * 1e6, out[1][0],
                 out[1][1].split("/")[1]))
        assert "persistent" in out[0][1] and "persistent" not in out[1][1], out
        if chains <= 256:  # (measured 11.9 against 22.6 and 20.4 against 34.7 us)
            assert out[0][0] < 0.8 * out[1][0], out
