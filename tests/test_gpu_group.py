"""One host process, several GPUs (mhx_group_*), native RCCL (mhx_comm_*), and the single-walker
entry points added for the Lisp shim (mhx_take_step, mhx_get_chain) - on the ONE GPU a test box
has: a group of one device must be the engine; a group naming device 0 twice rehearses the
multi-engine code path (chain ranges, concurrent launches, the pooled tick) with the host-staged
sum that stands in for RCCL between engines of one device; an engine that joined a 1-rank RCCL
communicator runs its pooled tick through ncclAllReduce on its own stream."""
import os

import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def _walk(obj, th0, iters, l_matrix=None):
    obj.init_chains(th0)
    obj.adaptive_begin(30000, 10.0, 1, l_matrix=l_matrix)
    left = iters
    while left > 0:
        obj.adaptive_advance(min(left, 150))
        left -= 150
    return obj.state()


def test_partition_matches_the_python_sharding(mhx):
    from lisp_mcmc_amd import distributed as mdist
    for total, parts in ((13, 2), (524288, 8), (7, 7), (100, 3)):
        got = [mhx.partition(total, parts, i) for i in range(parts)]
        assert got == [mdist.shard(total, parts, i) for i in range(parts)]
        assert got[0][0] == 0 and sum(c for _, c in got) == total
        assert all(got[i][0] + got[i][1] == got[i + 1][0] for i in range(parts - 1))


@pytest.mark.parametrize("pooled", [False, True])
def test_group_of_one_device_is_the_engine(mhx, pooled):
    s = pb.two_peak(n=2000, seed=3)
    mode = mhx.capi.ADAPT_POOLED if pooled else mhx.capi.ADAPT_FAITHFUL
    C_ = 48
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=5)
    e = s.engine(mhx, C_, seed=9, adapt_mode=mode)
    g = mhx.Group(C_, s.d, s.K, devices=[0], seed=9, adapt_mode=mode)
    s.apply(g)
    a, b = _walk(e, th0, 450), _walk(g, th0, 450)
    for k in ("theta", "logpost", "best_theta", "age", "length"):
        assert np.array_equal(a[k], b[k]), k
    assert e.counters()[0] == g.counters()[0] == C_ * 450
    if pooled:
        assert e.pooled()["refreshes"] == g.engines[0].pooled()["refreshes"] == 2
    e.close()
    g.close()


def test_two_engines_of_a_group_walk_like_one_engine(mhx):
    """faithful mode: no exchange, so the chains of a 2-engine group (ranges 0..24, 25..48 of 49)
    must coincide bit for bit with one engine holding all 49 (global-id Philox counters)"""
    s = pb.two_peak(n=2500, seed=4)
    C_ = 49
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=6)
    e = s.engine(mhx, C_, seed=10)
    g = mhx.Group(C_, s.d, s.K, devices=[0, 0], seed=10)
    assert g.ranges == [(0, 25), (25, 24)]
    s.apply(g)
    a, b = _walk(e, th0, 400), _walk(g, th0, 400)
    for k in ("theta", "logpost", "age"):
        assert np.array_equal(a[k], b[k]), k
    # one walker of the second engine through the per-engine entry points
    c = g.engines[1].chain(3)
    assert np.array_equal(c["theta"], a["theta"][28]) and c["age"] == a["age"][28]
    e.close()
    g.close()


def test_pooled_tick_over_two_engines(mhx):
    """pooled mode over two engines of one device: the statistics every engine ends up with are
    the sum over ALL chains (host-staged here; ncclAllReduce inside ncclGroupStart/End between
    distinct devices), equal to a single engine's up to the order of the additions"""
    s = pb.two_peak(n=2000, seed=5)
    C_ = 64
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=7)
    mode = mhx.capi.ADAPT_POOLED
    e = s.engine(mhx, C_, seed=11, adapt_mode=mode)
    g = mhx.Group(C_, s.d, s.K, devices=[0, 0], seed=11, adapt_mode=mode)
    s.apply(g)
    l0 = np.diag(0.01 * np.abs(s.theta_star))  # (from diag(theta), M:899, nothing would move yet)
    a, b = _walk(e, th0, 200, l0), _walk(g, th0, 200, l0)
    assert np.array_equal(a["theta"], b["theta"])  # nothing pooled has been adopted yet
    pe, p0, p1 = e.pooled(), g.engines[0].pooled(), g.engines[1].pooled()
    assert pe["refreshes"] == p0["refreshes"] == p1["refreshes"] == 1
    assert np.array_equal(p0["stats"], p1["stats"]) and np.array_equal(p0["L"], p1["L"])
    assert pe["stats"][0] == p0["stats"][0] > C_  # displacement count: an exact integer
    assert np.allclose(pe["stats"], p0["stats"], rtol=1e-11, atol=1e-18)
    assert pe["valid"] and p0["valid"] and np.allclose(pe["L"], p0["L"], rtol=1e-8, atol=1e-16)
    # ... and the walk goes on with the pooled factor on both
    for obj in (e, g):
        obj.adaptive_advance(1000)
    sa, sb = e.state(), g.state()
    assert (sa["age"] == sb["age"]).all() and np.isfinite(sb["logpost"]).all()
    e.close()
    g.close()


def test_native_rccl_all_reduce_on_one_rank(mhx):
    """mhx_comm_get_unique_id / mhx_comm_init_rank with one rank: the pooled tick goes through
    ncclAllReduce on the engine's stream (no host hook, no host synchronisation); a sum over one
    rank is the identity, so the walk must equal the engine without a communicator"""
    s = pb.two_peak(n=2000, seed=6)
    C_ = 32
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=8)
    mode = mhx.capi.ADAPT_POOLED
    plain = s.engine(mhx, C_, seed=12, adapt_mode=mode)
    rccl = s.engine(mhx, C_, seed=12, adapt_mode=mode)
    uid = mhx.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    rccl.comm_init_rank(uid, 0, 1)
    a, b = _walk(plain, th0, 1300), _walk(rccl, th0, 1300)
    for k in ("theta", "logpost", "age"):
        assert np.array_equal(a[k], b[k]), k
    pa, pb_ = plain.pooled(), rccl.pooled()
    assert pa["refreshes"] == pb_["refreshes"] == 6 and np.array_equal(pa["stats"], pb_["stats"])
    plain.close()
    rccl.close()


def test_a_hook_set_after_the_communicator_replaces_it(mhx):
    """bench.py's fall-back: when some rank cannot join the RCCL communicator, EVERY rank switches
    to the host hook - also the ones that did join (mhx_set_allreduce destroys the communicator)"""
    s = pb.two_peak(n=2000, seed=6)
    C_ = 32
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=8)
    e = s.engine(mhx, C_, seed=12, adapt_mode=mhx.capi.ADAPT_POOLED)
    e.comm_init_rank(mhx.comm_unique_id(), 0, 1)
    calls = []

    def hook(buf, n, dev):
        calls.append((n, dev))
        return 0
    e.set_allreduce(hook, device_buffer=False)
    plain = s.engine(mhx, C_, seed=12, adapt_mode=mhx.capi.ADAPT_POOLED)
    a, b = _walk(plain, th0, 700), _walk(e, th0, 700)
    assert len(calls) == 3 and calls[0] == (1 + 8 + 64, 0)
    for k in ("theta", "logpost", "age"):
        assert np.array_equal(a[k], b[k]), k
    e.close()
    plain.close()


def test_take_step_and_get_chain(mhx, orc):
    """(walker-take-step w :l-matrix L) with the device's randomness = one iteration of
    walker-many-steps (M:852-853); mhx_get_chain = one row of mhx_get_state"""
    s = pb.two_peak(n=700, seed=7)
    op = s.oracle(orc)
    C_ = 5
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=9)
    e = s.engine(mhx, C_, seed=13)
    e.init_chains(th0)
    L = np.diag(0.01 * np.abs(s.theta_star))
    ws = [orc.Walker(op, th0[c]) for c in range(C_)]
    for c, w in enumerate(ws):
        w.many_steps(7, L, seed=13, chain_id=c)
    for _ in range(7):
        e.take_step(L)
    st = e.state()
    for c, w in enumerate(ws):
        th, pr = w.last()
        assert np.array_equal(st["theta"][c], th) and st["age"][c] == w.age == 8
        one = e.chain(c)
        assert np.array_equal(one["theta"], st["theta"][c]) and one["logpost"] == st["logpost"][c]
        assert np.array_equal(one["best_theta"], st["best_theta"][c])
        assert one["best_logpost"] == st["best_logpost"][c]
        assert one["length"] == st["length"][c] and one["age"] == st["age"][c]
    # a hot step (T = 50) accepts what a cold one (T = 1) may refuse: same proposal either way
    hot, cold = s.engine(mhx, C_, seed=14), s.engine(mhx, C_, seed=14)
    for x in (hot, cold):
        x.init_chains(th0)
    big = np.diag(0.2 * np.abs(s.theta_star))
    for _ in range(20):
        hot.take_step(big, temperature=1e6)
        cold.take_step(big, temperature=1.0)
    assert (hot.acceptance(20) >= cold.acceptance(20)).all()
    assert hot.acceptance(20).mean() > cold.acceptance(20).mean()
    for x in (e, hot, cold):
        x.close()


def test_trace_across_the_ring_boundary(mhx, orc):
    """mhx_get_trace copies only the slots asked for: one run or two when they wrap"""
    s = pb.two_peak(n=300, seed=8)
    op = s.oracle(orc)
    th0 = pb.perturbed(s.theta_star, 2, 0.01, seed=3)
    e = s.engine(mhx, 2, seed=15)  # ring of 1024 steps
    e.init_chains(th0)
    L = np.diag(0.01 * np.abs(s.theta_star))
    w = orc.Walker(op, th0[1])
    for n_steps in (600, 700, 431):  # 1 + 600, then past 1024, then once more round
        e.many_steps(n_steps, L)
        w.many_steps(n_steps, L, seed=15, chain_id=1)
        for take in (1, 5, 300, 1000):
            gp, gt = e.trace(1, take)
            op_, ot = w.trace(take)
            assert np.array_equal(gt, ot) and len(gp) == min(take, w.length), (n_steps, take)
    e.close()


def test_sliding_temperature_window(mhx, orc):
    """the annealing schedule (M:878) lives on the device as a window that follows the loop index:
    a complete walker-adaptive-steps run with a 777-entry window equals the oracle's"""
    os.environ["MHX_TEMPS_WINDOW"] = "777"
    try:
        import subprocess, sys
        code = (
            "import sys, numpy as np; sys.path[:0] = [%r, %r]\n"
            "import lisp_mcmc_amd as mhx, oraclelib as orc, problems as pb\n"
            "s = pb.two_peak(n=500, seed=9); op = s.oracle(orc)\n"
            "th0 = pb.perturbed(s.theta_star, 3, 0.01, seed=4)\n"
            "e = s.engine(mhx, 3, seed=16); e.init_chains(th0)\n"
            "e.adaptive_begin(6000, 10.0, 1); e.adaptive_advance(1 << 40)\n"
            "st = e.state(); T = e.temperature()\n"
            "for c in range(3):\n"
            "    w = orc.Walker(op, th0[c]); w.adaptive_begin(6000, 10.0, 1, seed=16, chain_id=c)\n"
            "    w.adaptive_advance(1 << 40)\n"
            "    assert np.array_equal(st['theta'][c], w.last()[0]) and st['age'][c] == w.age, c\n"
            "    assert T[c] == w.temperature\n"
            "print('ok')\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                               os.path.dirname(os.path.abspath(__file__))))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                             env=dict(os.environ, MHX_SPLIT="0"))
        assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
    finally:
        os.environ.pop("MHX_TEMPS_WINDOW", None)
