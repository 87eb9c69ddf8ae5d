"""Parity of the HIP path (through the C ABI) with the oracle.  `-m gpu`: needs an MI355X.

Stated tolerance: |gpu - oracle| <= 1e-12 * sum_i |term_i| on the likelihood sum
(fp64, lane-partial + butterfly order instead of the reference's serial order, 1/sigma and
1/w multiplications instead of divisions, device exp/log within 1 ulp), plus
2^-52 * 1e10 per violated bound for the prior, whose formula -1e10 (exp(x) - 1) amplifies a
1-ulp difference of exp to 2.2e-6 (mcmc-fitting.lisp:360).  Proposals theta' = L z + theta are
bit-exact; accept decisions are identical except inside that tolerance band.
"""
import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu

REL = 1e-12
PRIOR_ULP = 2.0 ** -52 * 1e10


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def tol_for(op, theta, n_viol=0):
    return REL * op.abs_terms(theta) + 4 * PRIOR_ULP * n_viol + 1e-300


def n_violations(spec, theta):
    n = 0
    for b in spec.bounds:
        if b is None:
            continue
        idx, lo, hi = b
        v = np.array([theta[i] if i >= 0 else 0.0 for i in idx])
        n += int((~((np.asarray(lo) < v) & (v < np.asarray(hi)))).sum())
    return n


def check_logpost(mhx, orc, spec, thetas, **ekw):
    e = spec.engine(mhx, 1, **ekw)
    op = spec.oracle(orc, logfact_double=ekw.get("poisson_logfact_double", False))
    got, parts = e.logpost(thetas, parts=True)
    worst = 0.0
    for i, th in enumerate(thetas):
        ref, rparts = op.logpost(th, parts=True)
        tol = tol_for(op, th, n_violations(spec, th))
        assert abs(got[i] - ref) <= tol, (i, got[i], ref, tol)
        assert abs(parts[i, 0] - rparts[0]) <= tol
        assert abs(parts[i, 1] - rparts[1]) <= 4 * PRIOR_ULP * max(1, n_violations(spec, th))
        worst = max(worst, abs(got[i] - ref) / tol)
    e.close()
    return worst


def test_golden_line_fit_through_abi(mhx, golden):
    for tag in ("single", "double"):
        ref = golden["line_fit_initial_logpost_sigma_%s" % tag]
        s = pb.line_fit(golden, ref["sigma"])
        e = s.engine(mhx, 1)
        got = e.logpost([golden["line_fit"]["theta"]])[0]
        assert got == pytest.approx(ref["value"], rel=1e-14)
        e.init_chains(golden["line_fit"]["theta"])
        st = e.state()
        assert st["logpost"][0] == got and st["length"][0] == 1 and st["age"][0] == 1
        e.close()


def test_golden_global_fit_through_abi(mhx, golden):
    gf = golden["global_fit"]
    s = pb.Spec(6)
    sig = np.full(5, gf["sigma"])
    s.add(pb.POLY, (), [0, 1, 2, 3], gf["x1"], gf["y1"], sig, pb.NORMAL)
    s.add(pb.POLY, (), [4, 5], gf["x2"], gf["y2"], sig, pb.NORMAL)
    th = list(gf["theta"])
    th[5] = th[1] + th[5]
    e = s.engine(mhx, 1)
    assert e.logpost([th])[0] == pytest.approx(gf["initial_logpost"], rel=1e-13)
    e.close()


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 1023, 1024, 1025, 2048, 5000])
def test_logpost_two_peak_sizes(mhx, orc, n):
    s = pb.two_peak(n=n, seed=n)
    th = pb.perturbed(s.theta_star, 19, 0.02)
    th[3] = s.theta_star * 1.7   # every bound violated
    th[4, 2] = -0.5              # one bound violated
    check_logpost(mhx, orc, s, th)


def test_logpost_generic_equals_fixed(mhx, orc, monkeypatch):
    s = pb.two_peak(n=3000, seed=4)
    th = pb.perturbed(s.theta_star, 9, 0.02)
    e1 = s.engine(mhx, 1)
    a = e1.logpost(th)
    e1.close()
    monkeypatch.setenv("MHX_FORCE_GENERIC", "1")
    e2 = s.engine(mhx, 1)
    b = e2.logpost(th)
    e2.close()
    op = s.oracle(orc)
    for i in range(len(th)):
        assert abs(a[i] - b[i]) <= tol_for(op, th[i])


@pytest.mark.parametrize("lik", [pb.NORMAL, pb.CUTOFF])
def test_logpost_cutoff(mhx, orc, lik):
    s = pb.two_peak(n=2500, seed=7, lik=lik, sigma_lo=0.001, sigma_hi=0.002)
    th = pb.perturbed(s.theta_star, 6, 0.2)   # far off: many terms hit the -5000 floor
    check_logpost(mhx, orc, s, th)


@pytest.mark.parametrize("dbl", [False, True])
def test_logpost_poisson(mhx, orc, dbl):
    s = pb.poisson_peaks(n=3000)
    th = pb.perturbed(s.theta_star, 8, 0.02)
    check_logpost(mhx, orc, s, th, poisson_logfact_double=dbl)


def test_logpost_global_fit(mhx, orc):
    s = pb.global_fit(n_each=700, n_sets=8)
    assert s.d == 32 and s.K == 8
    th = pb.perturbed(s.theta_star, 5, 0.01)
    th[2, 0] = 11.0   # shared parameter out of bounds: the prior counts once PER FUNCTION
    check_logpost(mhx, orc, s, th)


def test_logpost_other_models(mhx, orc):
    rng = np.random.default_rng(0)
    x = np.linspace(-1, 2, 777)
    sig = rng.uniform(0.1, 0.2, x.size)
    cases = [
        (pb.POLY, (), np.array([0.3, -1.0, 0.5, 0.2, -0.1])),
        (pb.POLY, (), np.array([0.3, -1.0, 0.5, 0.2, -0.1, 0.05, 0.01, -0.02])),
        (pb.LORENTZ, (1, 2), np.array([0.1, 1.0, 0.2, 0.1, 0.5, 1.2, 0.3])),
        (pb.GAUSS, (3, 1), np.array([0.1, 0.2, -0.1, 1.0, 0.4, 0.2])),
        (pb.GAUSS, (0, 3), np.array([1.0, 0.0, 0.2, 0.5, 1.0, 0.1, 0.8, 1.5, 0.3])),
        (pb.EXPDECAY, (), np.array([2.0, 0.7, 0.1])),
        (pb.SINUS, (), np.array([1.5, 3.0, 0.4, -0.2])),
        (pb.PVOIGT2, (), np.array([1.2, 0.1, -0.2, 0.3, 0.1, 0.4, 1.1, 0.2, 0.7, 0.8, 0.05])),
    ]
    for model, shape, p in cases:
        s = pb.Spec(len(p))
        y = np.array([orc.lib().orc_model_eval(model, pb.np.asarray(shape or (0,), dtype=np.int32).ctypes.data_as(orc.i32p),
                                               p.ctypes.data_as(orc.f64p), len(p), float(xi)) for xi in x])
        y = y + sig * rng.standard_normal(x.size)
        s.add(model, shape, range(len(p)), x, y, sig, pb.NORMAL)
        s.theta_star = p
        check_logpost(mhx, orc, s, pb.perturbed(p, 4, 0.01))
    s = pb.lorder()
    check_logpost(mhx, orc, s, pb.perturbed(s.theta_star, 5, 0.001))


def test_chain_slot_independence(mhx):
    """the same theta gives the same bits in every wave / workgroup slot"""
    s = pb.two_peak(n=5000, seed=9)
    th = np.repeat(pb.perturbed(s.theta_star, 1, 0.01), 37, axis=0)
    e = s.engine(mhx, 1)
    got = e.logpost(th)
    assert np.all(got == got[0])
    e.close()


def run_injected(mhx, orc, spec, n_chains, n_steps, seed=0, T=None, scale=0.02):
    rng = np.random.default_rng(seed)
    d = spec.d
    e = spec.engine(mhx, n_chains)
    op = spec.oracle(orc)
    th0 = pb.perturbed(spec.theta_star, n_chains, 0.01, seed=seed + 1)
    e.init_chains(th0)
    ws = [orc.Walker(op, th0[c]) for c in range(n_chains)]
    L = np.array([np.tril(rng.normal(size=(d, d))) * scale * np.abs(spec.theta_star)[:, None]
                  for _ in range(n_chains)])
    flips = 0
    for it in range(n_steps):
        z = rng.standard_normal((n_chains, d))
        u = 1.0 - rng.random(n_chains)
        Ts = np.ones(n_chains) if T is None else T[it]
        acc = e.step_injected(L, z, u, Ts)
        st = e.state()
        for c in range(n_chains):
            a = ws[c].take_step_injected(L[c], z[c], u[c], Ts[c])
            th_o, pr_o = ws[c].last()
            if a != acc[c]:
                # legitimate only inside the tolerance band |(p1-p0)/T - log u| <= tol/T
                flips += 1
                pytest.fail("accept decision differs (chain %d step %d)" % (c, it))
            assert np.array_equal(st["theta"][c], th_o), (c, it)
            assert abs(st["logpost"][c] - pr_o) <= tol_for(op, th_o, n_violations(spec, th_o))
    for c in range(n_chains):
        assert st["length"][c] == ws[c].length and st["age"][c] == ws[c].age
        bt, bp = ws[c].best()
        assert np.array_equal(st["best_theta"][c], bt)
        n1, d1 = ws[c].acceptance(50)
        assert e.acceptance(50)[c] == n1 / d1
        pr, th = e.trace(c, 30)
        opr, oth = ws[c].trace(30)
        assert np.array_equal(th, oth) and np.allclose(pr, opr, rtol=0, atol=1e-6)
    e.close()
    return flips


def test_step_injected_trace(mhx, orc):
    s = pb.two_peak(n=1500, seed=21)
    run_injected(mhx, orc, s, n_chains=11, n_steps=60)


def test_step_injected_tempered_global(mhx, orc):
    s = pb.global_fit(n_each=200, n_sets=4)
    rng = np.random.default_rng(5)
    T = rng.uniform(1.0, 10.0, size=(25, 5))
    run_injected(mhx, orc, s, n_chains=5, n_steps=25, T=T, scale=0.002)


def test_philox_stream_and_many_steps(mhx, orc):
    """device Philox + Box-Muller draws are bit-identical to the oracle's: walker-many-steps
    trajectories coincide"""
    s = pb.two_peak(n=700, seed=31)
    C, n = 10, 120
    e = s.engine(mhx, C, seed=1234, chain_offset=5)
    op = s.oracle(orc)
    th0 = pb.perturbed(s.theta_star, C, 0.01)
    e.init_chains(th0)
    L = np.diag(0.01 * np.abs(s.theta_star))
    e.many_steps(n, L)
    st = e.state()
    for c in range(C):
        w = orc.Walker(op, th0[c])
        assert w.many_steps(n, L, seed=1234, chain_id=5 + c) == orc.DONE
        th, pr = w.last()
        assert np.array_equal(st["theta"][c], th), c
        assert st["age"][c] == w.age == n + 1
        n1, d1 = w.acceptance(100)
        assert e.acceptance(100)[c] == n1 / d1
    assert e.counters()[0] == C * n
    e.close()


def test_walker_modify_actions(mhx, orc):
    """:burn-walks / :keep-walks / :reset / :reset-to-most-likely (M:566-578) on the device ring,
    interleaved with stepping, against the oracle's list surgery"""
    s = pb.two_peak(n=300, seed=33)
    C_ = 5
    e = s.engine(mhx, C_, seed=77)
    op = s.oracle(orc)
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=3)
    e.init_chains(th0)
    ws = [orc.Walker(op, th0[c]) for c in range(C_)]
    L = np.diag(0.01 * np.abs(s.theta_star))

    def step(n):
        e.many_steps(n, L)
        for c, w in enumerate(ws):
            w.many_steps(n, L, seed=77, chain_id=c)

    def same():
        st = e.state()
        for c, w in enumerate(ws):
            th, pr = w.last()
            assert np.array_equal(st["theta"][c], th) and st["length"][c] == w.length
            assert st["age"][c] == w.age
            n1, d1 = w.acceptance(1000)
            assert e.acceptance(1000)[c] == n1 / d1
            assert e.proposal_factor(c, 500)[0] == w.l_matrix(500)[0]
            pg, tg = e.trace(c, 40)
            po, to = w.trace(40)
            assert np.array_equal(tg, to) and len(pg) == len(po)

    step(300)
    same()
    for action, n in (("burn-walks", 100), ("keep-walks", 50)):
        e.modify(action, n)
        assert all(w.modify(action, n) == 0 for w in ws)
        same()
        step(30)
        same()
    with pytest.raises(mhx.MhxError):
        e.modify("keep-walks", 10 ** 6)      # (subseq walk 0 keep-number) beyond the walk
    assert ws[0].modify("keep-walks", 10 ** 6) == -1
    e.modify("reset")
    for w in ws:
        w.modify("reset")
    same()
    assert (e.state()["length"] == 1).all()
    step(60)
    same()
    e.modify("reset-to-most-likely")
    for w in ws:
        w.modify("reset-to-most-likely")
    st = e.state()
    assert np.array_equal(st["theta"], st["best_theta"]) and np.array_equal(st["logpost"], st["best_logpost"])
    step(40)
    same()
    e.close()


def compare_adaptive(mhx, orc, spec, C, n, seed, auto=1, temperature=10.0, checkpoints=(),
                     l_matrix=None, max_walker_length=0, history_capacity=0):
    e = spec.engine(mhx, C, seed=seed, history_capacity=history_capacity)
    op = spec.oracle(orc)
    th0 = pb.perturbed(spec.theta_star, C, 0.01, seed=seed)
    e.init_chains(th0)
    ws = [orc.Walker(op, th0[c]) for c in range(C)]
    e.adaptive_begin(n, temperature, auto, max_walker_length, l_matrix)
    for c, w in enumerate(ws):
        w.adaptive_begin(n, temperature, auto, max_walker_length, l_matrix, seed=seed, chain_id=c)
    assert np.array_equal(e.lmatrix(), np.array([w.current_l() for w in ws]))
    done = 0
    marks = list(checkpoints) + [1 << 40]
    for m in marks:
        step = m - done
        running = e.adaptive_advance(step)
        for w in ws:
            w.adaptive_advance(step)
        done = m
        st = e.state()
        status, loop_i = e.chain_status()
        Ls = e.lmatrix()
        Ts = e.temperature()
        for c, w in enumerate(ws):
            th, pr = w.last()
            assert loop_i[c] == w.loop_index, (c, m)
            assert status[c] == w.status, (c, m)
            assert np.array_equal(st["theta"][c], th), (c, m)
            assert st["age"][c] == w.age and st["length"][c] == w.length
            assert np.array_equal(Ls[c], w.current_l()), (c, m)
            assert Ts[c] == w.temperature
        if running == 0:
            break
    assert all(w.status == orc.DONE for w in ws)
    return e, ws


def test_adaptive_matches_oracle_small(mhx, orc):
    s = pb.two_peak(n=600, seed=41)
    e, ws = compare_adaptive(mhx, orc, s, C=6, n=3200, seed=77,
                             checkpoints=(1, 199, 200, 201, 1000, 1001, 2000))
    # read-backs after the run
    for c, w in enumerate(ws):
        st, L, nf = e.proposal_factor(c, 500)
        ost, oL, onf = w.l_matrix(500)
        assert (st, nf) == (ost, onf)
        if st == 0:
            assert np.array_equal(L, oL)
    e.close()


def test_adaptive_short_n_quirk(mhx, orc):
    """n < 2000: the shutdown branch fires at i = 1 and rewinds i to n - 2000 < 0 (M:906, 917)"""
    s = pb.line_fit({"line_fit": {"x": [-4, -1, 2, 5, 10], "y": [0, 2, 5, 9, 13], "theta": [-1.0, 2.0]},
                     "line_fit_initial_logpost_sigma_single": {"sigma": 0.2}})
    e, ws = compare_adaptive(mhx, orc, s, C=3, n=100, seed=5, checkpoints=(1, 2, 500))
    assert e.state()["age"][0] == 2000 + 1
    e.close()


def test_adaptive_given_l_and_max_walker_length(mhx, orc):
    s = pb.two_peak(n=300, seed=43)
    L = np.diag(0.003 * np.abs(s.theta_star))
    e, ws = compare_adaptive(mhx, orc, s, C=4, n=12500, seed=9, auto=1, temperature=1e3,
                             checkpoints=(9999, 10000, 10001, 11026), l_matrix=L,
                             max_walker_length=2050, history_capacity=2048)
    e.close()


def test_adaptive_second_call_uses_window_covariance(mhx, orc):
    """a walker that already has >= steps-to-settle steps and acceptance(100) >= 0.1 starts the
    next walker-adaptive-steps from get-optimal-mcmc-l-matrix, not diag(theta) (M:896-901)"""
    s = pb.two_peak(n=300, seed=45)
    C_ = 5
    e = s.engine(mhx, C_, seed=21)
    op = s.oracle(orc)
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=21)
    e.init_chains(th0)
    ws = [orc.Walker(op, th0[c]) for c in range(C_)]
    L0 = np.diag(0.004 * np.abs(s.theta_star))
    for rnd in range(2):
        e.adaptive_begin(2600, 10.0, 1, 0, L0 if rnd == 0 else None)
        for c, w in enumerate(ws):
            w.adaptive_begin(2600, 10.0, 1, 0, L0 if rnd == 0 else None, seed=21, chain_id=c)
        Ls = e.lmatrix()
        for c, w in enumerate(ws):
            assert np.array_equal(Ls[c], w.current_l()), (rnd, c)
        if rnd == 1:  # the second start is a scaled Cholesky factor: lower triangular, not diagonal
            assert any(np.count_nonzero(np.tril(Ls[c], -1)) > 0 for c in range(C_))
        e.adaptive_advance(1 << 40)
        st = e.state()
        for c, w in enumerate(ws):
            w.adaptive_advance(1 << 40)
            assert np.array_equal(st["theta"][c], w.last()[0]), (rnd, c)
            assert st["age"][c] == w.age
    e.close()


def test_cholesky_invalid_operation_freezes_the_walker(mhx, orc):
    """a parameter that never moves gives a zero pivot and 0/0 in cholesky-decomp (M:597):
    floating-point-invalid-operation is NOT among the conditions M:891-894 handles, so the
    reference run would end in the debugger; the chain is frozen with MHX_CHAIN_FP_TRAP"""
    s = pb.two_peak(n=200, seed=46, bounds=False)
    C_ = 6
    L0 = np.diag(0.004 * np.abs(s.theta_star))
    L0[0, 0] = 0.0                                   # theta_0 is never proposed away
    e = s.engine(mhx, C_, seed=31)
    op = s.oracle(orc)
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=31)
    e.init_chains(th0)
    ws = [orc.Walker(op, th0[c]) for c in range(C_)]
    e.adaptive_begin(20000, 1.0, 0, 0, L0)
    for c, w in enumerate(ws):
        w.adaptive_begin(20000, 1.0, 0, 0, L0, seed=31, chain_id=c)
    e.adaptive_advance(3000)
    status, loop_i = e.chain_status()
    st = e.state()
    for c, w in enumerate(ws):
        w.adaptive_advance(3000)
        assert status[c] == w.status and loop_i[c] == w.loop_index, c
        assert np.array_equal(st["theta"][c], w.last()[0])
    assert (status == mhx.capi.CHAIN_FP_TRAP).any(), "the scenario should trap at least once"
    e.close()


def test_estop(mhx):
    s = pb.two_peak(n=200, seed=3)
    e = s.engine(mhx, 4, seed=1)
    e.init_chains(s.theta_star)
    e.adaptive_begin(100000, 10.0, 1)
    e.adaptive_advance(10)
    e.request_stop()
    assert e.adaptive_advance(10) == 0
    st, li = e.chain_status()
    assert (st == 3).all() and (li == 11).all()
    e.close()


def test_errors_through_abi(mhx):
    capi = mhx.capi
    with pytest.raises(mhx.MhxError) as ei:
        mhx.Engine(0, 2)
    assert ei.value.code == capi.EINVAL
    e = mhx.Engine(2, 2)
    with pytest.raises(mhx.MhxError) as ei:
        e.init_chains([0.0, 1.0])
    assert ei.value.code == capi.ESTATE and b"never set" in capi.lib().mhx_last_error()
    e.set_function(0, capi.MODEL_POLY, (), [0, 1])
    with pytest.raises(mhx.MhxError):
        e.set_dataset(0, [0.0, 1.0], [1.0, 2.0], [0.1, 0.0])   # sigma must be > 0 (M:376)
    with pytest.raises(mhx.MhxError):
        e.set_function(0, capi.MODEL_GAUSS_PEAKS, (1, 1), [0, 1])
    e.set_dataset(0, [0.0, 1.0], [1.0, 2.0], 0.1)
    with pytest.raises(mhx.MhxError) as ei:
        e.adaptive_advance(1)
    assert ei.value.code == capi.ESTATE
    # a non-finite log-posterior is where the reference traps: the walker freezes
    e.init_chains([[0.0, 1.0], [1e308, 1e308]])
    st, _ = e.chain_status()
    assert st[0] == capi.CHAIN_DONE and st[1] == capi.CHAIN_FP_TRAP
    e.close()


def test_walker_api_line_fit(mhx, golden):
    """The reference's own example (mcmc-fitting.lisp:1186) through the mirrored API"""
    m = mhx
    lf = golden["line_fit"]
    w = m.walker_create(function=m.models.line("b", "m"), data=[lf["x"], lf["y"]],
                        params=[":b", -1, ":m", 2],
                        data_error=golden["line_fit_initial_logpost_sigma_single"]["sigma"], seed=3)
    assert w.param_keys == ["b", "m"]
    assert w.last_step().prob == pytest.approx(
        golden["line_fit_initial_logpost_sigma_single"]["value"], rel=1e-14)
    assert w.length() == 1 and w.age() == 1
    m.walker_adaptive_steps(w, 4000)
    ml = m.walker_get(w, get=":most-likely-params")
    # least squares of the example data: b = 3.4778, m = 0.9676
    A = np.vstack([np.ones(5), lf["x"]]).T
    bm = np.linalg.lstsq(A, np.array(lf["y"], float), rcond=None)[0]
    assert abs(ml["b"] - bm[0]) < 0.2 and abs(ml["m"] - bm[1]) < 0.05
    acc = m.walker_get(w, get=":acceptance", take=1000)
    assert 0 < acc <= 1
    steps = m.walker_get(w, get=":steps", take=10)
    assert len(steps) == 10 and steps[0].prob == w.last_step().prob
    med = m.walker_get(w, get=":median-params", take=1000)
    assert abs(med["m"] - bm[1]) < 0.1
    L = m.walker_get(w, get=":l-matrix", take=500)
    assert L.shape == (2, 2) and L[0, 1] == 0.0


def test_pooled_rccl_hook_on_device_buffer(mhx):
    """the N > 1 exchange step as bench.py wires it: torch.distributed 'nccl' (= RCCL)
    all-reduce directly on the engine's device buffer (1-rank group on this 1-GPU box)"""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from lisp_mcmc_amd import distributed as mdist
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        s = pb.two_peak(n=300, seed=62)
        C_, d = 16, s.d
        e = s.engine(mhx, C_, seed=6, adapt_mode=mhx.capi.ADAPT_POOLED)
        e.set_allreduce(mdist.torch_allreduce_hook(dist), device_buffer=True)
        e.init_chains(pb.perturbed(s.theta_star, C_, 0.01))
        e.adaptive_begin(5000, 10.0, 0, l_matrix=np.diag(0.01 * np.abs(s.theta_star)))
        e.adaptive_advance(200)
        p = e.pooled()
        assert p["refreshes"] == 1 and p["valid"]
        assert np.allclose(p["stats"], host_pooled_stats(e, C_, d), rtol=1e-12, atol=1e-300)
        e.close()
    finally:
        dist.destroy_process_group()


def host_pooled_stats(e, C, d, take=500):
    out = np.zeros(1 + d + d * d)
    for c in range(C):
        prob, th = e.trace(c, take)
        fwd = [i for i in range(len(prob) - 1) if prob[i] > prob[i + 1]]
        if len(fwd) < 2:
            continue
        v = np.array([th[fwd[k + 1]] - th[fwd[k]] for k in range(len(fwd) - 1)])
        out[0] += len(v)
        out[1:1 + d] += v.sum(0)
        out[1 + d:] += (v.T @ v).ravel()
    return out


@pytest.mark.parametrize("ranks", [1, 2])
def test_pooled_adaptation_tick(mhx, ranks):
    """MHX_ADAPT_POOLED (extension): every 200 iterations the forward-step displacement
    statistics of all chains are summed (k_pool_stats/k_pool_reduce), all-reduced through the
    hook, and turned into one shared factor (2.38^2/d) chol(cov)"""
    s = pb.two_peak(n=400, seed=61)
    C_, d = 24, s.d
    e = s.engine(mhx, C_, seed=5, adapt_mode=mhx.capi.ADAPT_POOLED)
    calls = []

    def hook(buf, n, dev):   # stands for `ranks` identical ranks: sum = ranks * local
        import ctypes
        a = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_double)), shape=(n,))
        calls.append((n, dev, a.copy()))
        a *= ranks
        return 0
    e.set_allreduce(hook, device_buffer=False)
    e.init_chains(pb.perturbed(s.theta_star, C_, 0.01))
    e.adaptive_begin(5000, 10.0, 0, l_matrix=np.diag(0.01 * np.abs(s.theta_star)))
    assert e.adaptive_advance(150) == C_ and e.pooled()["refreshes"] == 0
    e.adaptive_advance(50)                      # iteration 200: first tick
    p = e.pooled()
    assert p["refreshes"] == 1 and len(calls) == 1 and calls[0][0] == 1 + d + d * d
    host = host_pooled_stats(e, C_, d)
    assert np.allclose(calls[0][2], host, rtol=1e-12, atol=1e-300)
    assert np.allclose(p["stats"], ranks * host, rtol=1e-12, atol=1e-300)
    n = p["stats"][0]
    mean = p["stats"][1:1 + d] / n
    cov = p["stats"][1 + d:].reshape(d, d) / n - np.outer(mean, mean)
    assert p["valid"]
    assert np.allclose(p["L"], (2.38 ** 2 / d) * np.linalg.cholesky(cov), rtol=1e-7, atol=1e-12)
    # a chain re-estimates its factor only at i % (2*steps-to-settle) = 0 with acceptance in
    # (0.2, 0.4) (M:932-938); in pooled mode it then adopts the factor of the latest tick
    e.adaptive_advance(600)
    p800 = e.pooled()
    assert p800["refreshes"] == 4 and p800["valid"]
    e.adaptive_advance(200)
    acc = e.acceptance(200)
    Ls = e.lmatrix()
    adopted = [c for c in range(C_) if np.array_equal(Ls[c], p800["L"])]
    lo, hi = float(np.float32(0.2)), float(np.float32(0.4))   # single-float literals of M:934
    band = [c for c in range(C_) if lo < acc[c] < hi]
    assert set(band) == set(adopted)
    st, _ = e.chain_status()
    assert (st == mhx.capi.CHAIN_RUNNING).all()
    e.close()
