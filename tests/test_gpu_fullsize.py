"""Parity at BASELINE.json's full sizes (config 2: 4096 chains, 8 params, 1e5 points, fp64).

The oracle is serial CPU code, so at full size it checks a SAMPLE of chains directly and the
rest through size-independent properties: slot independence (the same theta gives the same
bits in any of the 4096 waves), additivity over a split of the dataset, invariance under a
permutation of the points (within the stated tolerance), and chain-count independence
(chain c of a 4096-chain batch walks exactly like chain c of an 8-chain batch: Philox counters
use global ids)."""
import os
import sys

import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REL = 1e-12


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


@pytest.fixture(scope="module")
def c2():
    import bench
    spec, chains, b_pt, desc = bench.synth_workload("c2")
    assert chains == 4096 and len(spec.data[0][0]) == 100000 and spec.d == 8
    return spec


def test_c2_logpost_sample_vs_oracle_and_slot_independence(mhx, orc, c2):
    op = c2.oracle(orc)
    e = c2.engine(mhx, 4096)
    th = pb.perturbed(c2.theta_star, 4096, 0.01, seed=1)
    th[1000:] = th[np.arange(3096) % 8]            # 8 distinct vectors repeated over the slots
    got = e.logpost(th)
    for c in (0, 1, 2, 3, 17, 511, 999):
        ref = op.logpost(th[c])
        assert abs(got[c] - ref) <= REL * op.abs_terms(th[c]), c
        assert got[c] == op.logpost_mirror(th[c]), c  # the restated kernel: every bit
    for c in range(1000, 4096):
        assert got[c] == got[(c - 1000) % 8]
    e.close()


def test_c2_additivity_and_permutation(mhx, orc, c2):
    x, y, s, lik = c2.data[0]
    th = pb.perturbed(c2.theta_star, 16, 0.01, seed=2)
    e = c2.engine(mhx, 1)
    whole, parts = e.logpost(th, parts=True)
    e.close()
    # the same points as two functions of one global fit (57 344 + 42 656 points)
    cut = 57344
    sp = pb.Spec(8)
    sp.add(pb.GAUSS, (2, 2), range(8), x[:cut], y[:cut], s[:cut], lik, c2.bounds[0])
    sp.add(pb.GAUSS, (2, 2), range(8), x[cut:], y[cut:], s[cut:], lik, None)
    e2 = sp.engine(mhx, 1)
    split = e2.logpost(th)
    e2.close()
    # and in a random order
    perm = np.random.default_rng(3).permutation(x.size)
    sq = pb.Spec(8)
    sq.add(pb.GAUSS, (2, 2), range(8), x[perm], y[perm], s[perm], lik, c2.bounds[0])
    e3 = sq.engine(mhx, 1)
    shuf = e3.logpost(th)
    e3.close()
    op = c2.oracle(orc)
    for i in range(len(th)):
        tol = REL * op.abs_terms(th[i])
        assert abs(split[i] - whole[i]) <= tol
        assert abs(shuf[i] - whole[i]) <= tol


def test_c2_full_batch_walks_like_small_batch_and_oracle(mhx, orc, c2):
    """60 iterations of walker-adaptive-steps on all 4096 chains; chains 0..7 must coincide
    bit for bit with an 8-chain engine, and chains 0..2 with the oracle"""
    n_it = 60
    th0 = pb.perturbed(c2.theta_star, 4096, 0.01, seed=4)
    big = c2.engine(mhx, 4096, seed=11)
    big.init_chains(th0)
    big.adaptive_begin(30000, 10.0, 1)
    assert big.adaptive_advance(n_it) == 4096
    sb = big.state()
    small = c2.engine(mhx, 8, seed=11)
    small.init_chains(th0[:8])
    small.adaptive_begin(30000, 10.0, 1)
    small.adaptive_advance(n_it)
    ss = small.state()
    assert np.array_equal(sb["theta"][:8], ss["theta"])
    assert np.array_equal(sb["logpost"][:8], ss["logpost"])
    assert (sb["age"] == n_it + 1).all() and big.counters()[0] == 4096 * n_it
    op = c2.oracle(orc)
    for c in range(3):
        w = orc.Walker(op, th0[c])
        w.adaptive_begin(30000, 10.0, 1, seed=11, chain_id=c)
        w.adaptive_advance(n_it)
        th, pr = w.last()
        assert np.array_equal(sb["theta"][c], th), c
        assert abs(sb["logpost"][c] - pr) <= REL * op.abs_terms(th)
    # a different partition of the same global ids gives the same chains
    shard = c2.engine(mhx, 8, seed=11, chain_offset=2048)
    shard.init_chains(th0[2048:2056])
    shard.adaptive_begin(30000, 10.0, 1)
    shard.adaptive_advance(n_it)
    assert np.array_equal(shard.state()["theta"], sb["theta"][2048:2056])
    for e in (big, small, shard):
        e.close()


def test_c5_per_gpu_share_65536_chains(mhx, orc, c2):
    """BASELINE config 5 puts 524288 chains on 8 GPUs = 65536 per GPU.  One rank's share, taken
    from the MIDDLE of the global id range (rank 3: ids 196608...), with the pooled-covariance
    mode the multi-GPU bench uses, THROUGH the first pooled tick (iteration 200: statistics of
    all 65536 chains, reduction, factorisation on the engine's stream): chains must coincide
    with an 8-chain engine owning the same global ids (a pooled factor is first adopted at
    iteration 400, M:929's cadence), every chain must have moved the same number of steps, and
    the pooled factor must be the Cholesky factor of the covariance its statistics describe.
    (The walk starts from a small :l-matrix: from the default diag(theta) of M:899 hardly a chain
    has accepted anything by iteration 200, and a covariance of fewer displacements than
    parameters has no Cholesky factor - the tick then rightly reports `valid = 0`.)"""
    C_, n_it, off = 65536, 212, 3 * 65536
    l0 = np.diag(0.01 * np.abs(c2.theta_star))
    rng = np.random.Generator(np.random.Philox(key=99))
    th0 = c2.theta_star[None, :] * (1.0 + 0.01 * rng.standard_normal((C_, c2.d)))
    big = c2.engine(mhx, C_, seed=21, chain_offset=off, adapt_mode=mhx.capi.ADAPT_POOLED)
    big.init_chains(th0)
    big.adaptive_begin(30000, 10.0, 1, l_matrix=l0)
    assert big.adaptive_advance(12) == C_
    assert big.pooled()["refreshes"] == 0
    assert big.adaptive_advance(n_it - 12) == C_     # ... across the tick at iteration 200
    sb = big.state()
    assert (sb["age"] == n_it + 1).all() and big.counters()[0] == C_ * n_it
    assert np.isfinite(sb["logpost"]).all()
    st, _ = big.chain_status()
    assert (st == mhx.capi.CHAIN_RUNNING).all()
    # the tick: (n, sum delta, sum delta delta^T) over all chains -> covariance -> (2.38^2/d) chol
    pool = big.pooled()
    d = c2.d
    assert pool["refreshes"] == 1 and pool["valid"]
    n = pool["stats"][0]
    assert n == np.floor(n) and n > C_                # displacement count: an exact integer
    mean = pool["stats"][1:1 + d] / n
    cov = pool["stats"][1 + d:].reshape(d, d) / n - np.outer(mean, mean)
    want = (2.38 ** 2 / d) * np.linalg.cholesky(cov)
    assert np.allclose(pool["L"], want, rtol=1e-9, atol=1e-18)
    # ... the two halves of the share, as two engines, see the same statistics between them
    halves = np.zeros(1 + d + d * d)
    for h in range(2):
        e = c2.engine(mhx, C_ // 2, seed=21, chain_offset=off + h * (C_ // 2),
                      adapt_mode=mhx.capi.ADAPT_POOLED)
        e.init_chains(th0[h * (C_ // 2):(h + 1) * (C_ // 2)])
        e.adaptive_begin(30000, 10.0, 1, l_matrix=l0)
        e.adaptive_advance(200)
        halves += e.pooled()["stats"]
        e.close()
    assert halves[0] == n
    assert np.allclose(halves, pool["stats"], rtol=1e-11, atol=1e-18)
    for lo in (0, 40000, C_ - 8):
        small = c2.engine(mhx, 8, seed=21, chain_offset=off + lo, adapt_mode=mhx.capi.ADAPT_POOLED)
        small.init_chains(th0[lo:lo + 8])
        small.adaptive_begin(30000, 10.0, 1, l_matrix=l0)
        small.adaptive_advance(n_it)
        ss = small.state()
        assert np.array_equal(sb["theta"][lo:lo + 8], ss["theta"]), lo
        assert np.array_equal(sb["logpost"][lo:lo + 8], ss["logpost"]), lo
        small.close()
    op = c2.oracle(orc)
    w = orc.Walker(op, th0[40001])
    w.adaptive_begin(30000, 10.0, 1, l_matrix=l0, seed=21, chain_id=off + 40001)
    w.adaptive_advance(n_it)
    assert np.array_equal(sb["theta"][40001], w.last()[0])
    big.close()


def test_c2_complete_default_run_4096_chains(mhx, c2):
    """(walker-adaptive-steps w) with its default n = 30000 on every one of the 4096 chains of
    BASELINE config 2: annealing, L-matrix adaptation and the automatic shut-down all the way to
    the end of the do loop.  Every chain must finish (none trapped), sit at the posterior mode
    within its width, and sample with a healthy acceptance rate."""
    C_ = 4096
    rng = np.random.Generator(np.random.Philox(key=7))
    th0 = c2.theta_star[None, :] * (1.0 + 0.01 * rng.standard_normal((C_, c2.d)))
    e = c2.engine(mhx, C_, seed=77)
    e.init_chains(th0)
    e.adaptive_steps(30000)
    st, _ = e.chain_status()
    assert (st == mhx.capi.CHAIN_DONE).all(), np.bincount(st)
    s = e.state()
    assert np.isfinite(s["logpost"]).all()
    # posterior widths at 1e5 points are ~1e-3 relative: the chains agree with each other and
    # with the generating parameters far inside the 50 % bounds box
    best = s["best_theta"] if "best_theta" in s else s["theta"]
    rel = np.abs(np.median(best, axis=0) / c2.theta_star - 1.0)
    assert (rel < 0.02).all(), rel
    spread = np.std(s["theta"], axis=0) / np.abs(c2.theta_star)
    assert (spread < 0.02).all(), spread
    acc = e.acceptance(1000)
    assert 0.1 < float(np.median(acc)) < 0.6, float(np.median(acc))
    # every chain used the whole loop or shut itself down through :prob-settle (M:905-917), which
    # jumps to the last (tail) iterations: ages of 7000-16000 are typical here
    assert (s["age"] > 2000).all() and (s["age"] <= 30001).all()
    assert (s["age"] < 30001).any()
    e.close()


# ---------------------------------------------------------------------------------------------
# BASELINE config 3 at full size: 65536 chains, 16 params, log-poisson, 1e6 points (M:379-383
# through M:402-416) and config 4: 4096 chains, 8 datasets x 12500 points sharing 32 params
# (M:1067-1070).  Same plan as config 2: a sample against the faithful oracle, the rest through
# size-independent properties.
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c3():
    import bench
    spec, chains, b_pt, desc = bench.synth_workload("c3")
    assert chains == 65536 and len(spec.data[0][0]) == 1000000 and spec.d == 16
    return spec


@pytest.fixture(scope="module")
def c4():
    import bench
    spec, chains, b_pt, desc = bench.synth_workload("c4")
    assert chains == 4096 and spec.K == 8 and spec.d == 32
    assert all(len(d[0]) == 12500 for d in spec.data)
    return spec


def _skip_and_noskip(mhx, spec, chains, **kw):
    """(engine with tile-level peak skipping, engine without); read when the problem is finalised"""
    out = []
    for flag in ("0", "1"):
        os.environ["MHX_NO_TILE_SKIP"] = flag
        try:
            e = spec.engine(mhx, chains, **kw)
            e.kernel_name()  # finalises the problem under this setting
        finally:
            os.environ.pop("MHX_NO_TILE_SKIP", None)
        out.append(e)
    return out


def test_c3_logpost_sample_vs_oracle_and_slot_independence(mhx, orc, c3):
    op = c3.oracle(orc)
    C_ = 65536
    e = c3.engine(mhx, C_)
    assert "gauss15_poisson" in e.kernel_name() and e.kernel_name().startswith("w16/")
    th8 = pb.perturbed(c3.theta_star, 8, 0.01, seed=31)
    got8, parts8 = e.logpost(th8, parts=True)
    for c in range(8):
        ref, pr = op.logpost(th8[c], parts=True)
        tol = REL * op.abs_terms(th8[c])
        assert abs(got8[c] - ref) <= tol, (c, got8[c], ref)
        assert abs(parts8[c][0] - pr[0]) <= tol and parts8[c][1] == pr[1] == 0.0
        assert got8[c] == op.logpost_mirror(th8[c]), c  # ... and the restated kernel: every bit
    # a vector outside its bounds box: the prior part carries the penalty (M:360)
    out = th8[0].copy()
    out[2] = c3.theta_star[2] * 1.7
    g, gp = e.logpost(out[None, :], parts=True)
    ref, pr = op.logpost(out, parts=True)
    assert abs(gp[0][1] - pr[1]) <= 2.0 ** -52 * 1e10 and pr[1] < -1.0
    assert abs(g[0] - ref) <= REL * op.abs_terms(out) + 2.0 ** -52 * 1e10
    # the same 8 vectors in every one of the 65536 wave slots of ONE launch: same bits
    got = e.logpost(np.tile(th8, (C_ // 8, 1)))
    assert np.array_equal(got, np.tile(got8, C_ // 8))
    e.close()


def test_c3_additivity_and_permutation(mhx, orc, c3):
    x, y, s, lik = c3.data[0]
    th = pb.perturbed(c3.theta_star, 8, 0.01, seed=32)
    e = c3.engine(mhx, 1)
    whole = e.logpost(th)
    e.close()
    cut = 571392  # 279 tiles of 2048 points, then a ragged rest
    sp = pb.Spec(16)
    sp.add(pb.GAUSS, (1, 5), range(16), x[:cut], y[:cut], None, lik, c3.bounds[0])
    sp.add(pb.GAUSS, (1, 5), range(16), x[cut:], y[cut:], None, lik, None)
    e2 = sp.engine(mhx, 1)
    split = e2.logpost(th)
    e2.close()
    perm = np.random.default_rng(33).permutation(x.size)
    sq = pb.Spec(16)
    sq.add(pb.GAUSS, (1, 5), range(16), x[perm], y[perm], None, lik, c3.bounds[0])
    e3 = sq.engine(mhx, 1)
    shuf = e3.logpost(th)
    e3.close()
    op = c3.oracle(orc)
    for i in range(len(th)):
        tol = REL * op.abs_terms(th[i])
        assert abs(split[i] - whole[i]) <= tol
        assert abs(shuf[i] - whole[i]) <= tol


def test_c3_tile_skipping_is_exact_on_489_tiles(mhx, c3):
    """1e6 points = 489 tiles of 2048: the 64-tile mask refresh of sweep() (lane i takes tile
    t + i) runs 8 times per sweep for the Poisson kernel; with and without skipping the bits
    must agree, for benign vectors, for narrow / wide / out-of-range peaks and on a walk"""
    a, b = _skip_and_noskip(mhx, c3, 64, seed=5)
    rng = np.random.default_rng(34)
    th = pb.perturbed(c3.theta_star, 64, 0.01, seed=35)
    for r in range(16, 64):
        k = rng.integers(0, 5)
        th[r, 2 + 3 * k] = rng.uniform(-0.2, 1.2)                 # centre, in and out of range
        th[r, 3 + 3 * k] = 10.0 ** rng.uniform(-4.5, 0.3)         # width over 5 decades
        th[r, 1 + 3 * k] = rng.choice([0.0, 1e-12, 3.0, 150.0, 1e6])  # amplitude
        th[r, 0] = rng.choice([20.0, 1e-3, 5e4])
    ga, gb = a.logpost(th), b.logpost(th)
    assert np.array_equal(ga, gb) and np.isfinite(ga).all()
    th0 = pb.perturbed(c3.theta_star, 64, 0.01, seed=36)
    l0 = np.diag(0.002 * np.abs(c3.theta_star))
    for e in (a, b):
        e.init_chains(th0)
        e.adaptive_begin(30000, 10.0, 1, l_matrix=l0)
        e.adaptive_advance(6)
    sa, sb = a.state(), b.state()
    assert np.array_equal(sa["theta"], sb["theta"]) and np.array_equal(sa["logpost"], sb["logpost"])
    assert (sa["age"] == 7).all()
    a.close()
    b.close()


def test_c3_full_batch_walks_like_small_batch_and_oracle(mhx, orc, c3):
    """20 iterations of walker-adaptive-steps on all 65536 chains x 1e6 points; chains of three
    8-chain engines with the same global ids must coincide bit for bit, and chains 0..2 with the
    oracle (positions equal, log-posteriors within the stated tolerance)"""
    C_, n_it = 65536, 20
    rng = np.random.Generator(np.random.Philox(key=0xC3))
    th0 = c3.theta_star[None, :] * (1.0 + 0.01 * rng.standard_normal((C_, c3.d)))
    l0 = np.diag(0.002 * np.abs(c3.theta_star))
    big = c3.engine(mhx, C_, seed=13)
    big.init_chains(th0)
    big.adaptive_begin(30000, 10.0, 1, l_matrix=l0)
    left = n_it
    while left > 0:  # launches of <= 5 iterations (0.14 s each at this size)
        assert big.adaptive_advance(min(left, 5)) == C_
        left -= 5
    sb = big.state()
    assert (sb["age"] == n_it + 1).all() and big.counters()[0] == C_ * n_it
    assert np.isfinite(sb["logpost"]).all()
    st, _ = big.chain_status()
    assert (st == mhx.capi.CHAIN_RUNNING).all()
    assert (sb["theta"] != th0).any(axis=1).mean() > 0.5  # most chains have accepted something
    for lo in (0, 30000, C_ - 8):
        small = c3.engine(mhx, 8, seed=13, chain_offset=lo)
        small.init_chains(th0[lo:lo + 8])
        small.adaptive_begin(30000, 10.0, 1, l_matrix=l0)
        small.adaptive_advance(n_it)
        ss = small.state()
        assert np.array_equal(sb["theta"][lo:lo + 8], ss["theta"]), lo
        assert np.array_equal(sb["logpost"][lo:lo + 8], ss["logpost"]), lo
        small.close()
    op = c3.oracle(orc)
    for c in range(3):
        w = orc.Walker(op, th0[c])
        w.adaptive_begin(30000, 10.0, 1, l_matrix=l0, seed=13, chain_id=c)
        w.adaptive_advance(n_it)
        th, pr = w.last()
        assert np.array_equal(sb["theta"][c], th), c
        assert abs(sb["logpost"][c] - pr) <= REL * op.abs_terms(th)
        assert w.age == sb["age"][c]
    big.close()


def test_c4_logpost_sample_vs_oracle_slots_additivity_permutation(mhx, orc, c4):
    op = c4.oracle(orc)
    e = c4.engine(mhx, 4096)
    assert "pvoigt2_normal" in e.kernel_name()
    th8 = pb.perturbed(c4.theta_star, 8, 0.01, seed=41)
    th8[7, 9] = 11.0  # outside (-10, 10): the bounds block is listed once per function, so the
    #                   penalty is counted 8 times (M:1069)
    got8, parts8 = e.logpost(th8, parts=True)
    for c in range(8):
        ref, pr = op.logpost(th8[c], parts=True)
        slack = 8 * 2.0 ** -52 * 1e10 if c == 7 else 0.0
        assert abs(got8[c] - ref) <= REL * op.abs_terms(th8[c]) + slack, c
        assert abs(parts8[c][1] - pr[1]) <= slack
    assert parts8[7][1] < -8.0 * 1e4
    got = e.logpost(np.tile(th8, (512, 1)))
    assert np.array_equal(got, np.tile(got8, 512))
    e.close()
    # every dataset cut in two (the functions of a 16-function global fit) and shuffled
    sp, sq = pb.Spec(32), pb.Spec(32)
    rng = np.random.default_rng(42)
    halves = []
    for k in range(8):
        x, y, s, lik = c4.data[k]
        model, shape, idx = c4.fns[k]
        cut = 6144 + 512 * k
        halves.append((model, shape, idx, x[cut:], y[cut:], s[cut:], lik))
        sp.add(model, shape, idx, x[:cut], y[:cut], s[:cut], lik, c4.bounds[k])
        p = rng.permutation(x.size)
        sq.add(model, shape, idx, x[p], y[p], s[p], lik, c4.bounds[k])
    for (model, shape, idx, x, y, s, lik) in halves:
        sp.add(model, shape, idx, x, y, s, lik, None)
    e2 = sp.engine(mhx, 1)
    e3 = sq.engine(mhx, 1)
    split, shuf = e2.logpost(th8[:7]), e3.logpost(th8[:7])
    for i in range(7):
        tol = REL * op.abs_terms(th8[i])
        assert abs(split[i] - got8[i]) <= tol and abs(shuf[i] - got8[i]) <= tol
    e2.close()
    e3.close()


def test_c4_full_batch_walks_like_small_batch_and_oracle(mhx, orc, c4):
    C_, n_it = 4096, 40
    rng = np.random.Generator(np.random.Philox(key=0xC4))
    th0 = c4.theta_star[None, :] * (1.0 + 0.01 * rng.standard_normal((C_, c4.d)))
    big = c4.engine(mhx, C_, seed=14)
    big.init_chains(th0)
    big.adaptive_begin(30000, 10.0, 1)
    assert big.adaptive_advance(n_it) == C_
    sb = big.state()
    assert (sb["age"] == n_it + 1).all() and big.counters()[0] == C_ * n_it
    for lo in (0, 2048, C_ - 8):
        small = c4.engine(mhx, 8, seed=14, chain_offset=lo)
        small.init_chains(th0[lo:lo + 8])
        small.adaptive_begin(30000, 10.0, 1)
        small.adaptive_advance(n_it)
        ss = small.state()
        assert np.array_equal(sb["theta"][lo:lo + 8], ss["theta"]), lo
        assert np.array_equal(sb["logpost"][lo:lo + 8], ss["logpost"]), lo
        small.close()
    op = c4.oracle(orc)
    for c in range(3):
        w = orc.Walker(op, th0[c])
        w.adaptive_begin(30000, 10.0, 1, seed=14, chain_id=c)
        w.adaptive_advance(n_it)
        th, pr = w.last()
        assert np.array_equal(sb["theta"][c], th), c
        assert abs(sb["logpost"][c] - pr) <= REL * op.abs_terms(th)
    big.close()
