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
    mode the multi-GPU bench uses: chains must coincide with an 8-chain engine owning the same
    global ids, and every chain must have moved the same number of steps."""
    C_, n_it, off = 65536, 12, 3 * 65536
    rng = np.random.Generator(np.random.Philox(key=99))
    th0 = c2.theta_star[None, :] * (1.0 + 0.01 * rng.standard_normal((C_, c2.d)))
    big = c2.engine(mhx, C_, seed=21, chain_offset=off, adapt_mode=mhx.capi.ADAPT_POOLED)
    big.init_chains(th0)
    big.adaptive_begin(30000, 10.0, 1)
    assert big.adaptive_advance(n_it) == C_
    sb = big.state()
    assert (sb["age"] == n_it + 1).all() and big.counters()[0] == C_ * n_it
    assert np.isfinite(sb["logpost"]).all()
    st, _ = big.chain_status()
    assert (st == mhx.capi.CHAIN_RUNNING).all()
    for lo in (0, 40000, C_ - 8):
        small = c2.engine(mhx, 8, seed=21, chain_offset=off + lo, adapt_mode=mhx.capi.ADAPT_POOLED)
        small.init_chains(th0[lo:lo + 8])
        small.adaptive_begin(30000, 10.0, 1)
        small.adaptive_advance(n_it)
        ss = small.state()
        assert np.array_equal(sb["theta"][lo:lo + 8], ss["theta"]), lo
        assert np.array_equal(sb["logpost"][lo:lo + 8], ss["logpost"]), lo
        small.close()
    op = c2.oracle(orc)
    w = orc.Walker(op, th0[40001])
    w.adaptive_begin(30000, 10.0, 1, seed=21, chain_id=off + 40001)
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
