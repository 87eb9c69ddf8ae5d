"""Engine lifecycle on the device: handles are independent (two engines stepped in an interleaved
order give what each gives alone), and creating / destroying engines does not leak device
memory."""
import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mhx():
    import lisp_mcmc_amd
    return lisp_mcmc_amd


def test_interleaved_engines_are_independent(mhx):
    s1 = pb.two_peak(n=3000, seed=1)
    s2 = pb.poisson_peaks(n=2500, seed=2)
    th1 = pb.perturbed(s1.theta_star, 9, 0.01, seed=3)
    th2 = pb.perturbed(s2.theta_star, 5, 0.01, seed=4)
    l2 = np.diag(0.002 * np.abs(s2.theta_star))

    def fresh():
        a = s1.engine(mhx, 9, seed=7)
        b = s2.engine(mhx, 5, seed=8)
        a.init_chains(th1)
        b.init_chains(th2)
        a.adaptive_begin(1500, 10.0, 1)
        b.adaptive_begin(1500, 10.0, 1, l_matrix=l2)
        return a, b
    a, b = fresh()
    a.adaptive_advance(1 << 40)
    b.adaptive_advance(1 << 40)
    ref = (a.state(), b.state())
    a.close()
    b.close()
    a, b = fresh()
    for _ in range(40):            # ping-pong in uneven chunks
        a.adaptive_advance(37)
        b.adaptive_advance(101)
    a.adaptive_advance(1 << 40)
    b.adaptive_advance(1 << 40)
    got = (a.state(), b.state())
    for r, g in zip(ref, got):
        for k in ("theta", "logpost", "age", "length"):
            assert np.array_equal(r[k], g[k]), k
    a.close()
    b.close()


def test_create_destroy_does_not_leak_device_memory(mhx):
    torch = pytest.importorskip("torch")
    s = pb.two_peak(n=20000, seed=5)
    th0 = pb.perturbed(s.theta_star, 512, 0.01, seed=6)

    def cycle():
        e = s.engine(mhx, 512, seed=1, history_capacity=4096)
        e.init_chains(th0)
        e.adaptive_begin(300, 10.0, 1)
        e.adaptive_advance(50)
        e.close()
    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(40):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 32 << 20, (free0, free1)   # each engine holds ~170 MB while alive


@pytest.mark.parametrize("pooled", [False, True])
def test_finished_chains_give_up_their_slots_without_changing_anything(mhx, pooled):
    """Complete walker-adaptive-steps runs of 200 chains end at different loop indices; launches
    with more workgroups than the GPU holds at once are repacked between launches with the chains
    still walking (compact_slots; MHX_COMPACT_ALWAYS=1 makes it happen for this small one too).
    A chain's walk must not depend on it: the same run with MHX_NO_COMPACT=1.  In both adaptation
    modes: the pooled statistics index chains, not wave slots, so the pooled factors - and with
    them every chain - are the same wherever the chains walk."""
    import os
    os.environ["MHX_COMPACT_ALWAYS"] = "1"
    s = pb.two_peak(n=1200, seed=3)
    th0 = pb.perturbed(s.theta_star, 200, 0.01, seed=4)
    mode = mhx.capi.ADAPT_POOLED if pooled else mhx.capi.ADAPT_FAITHFUL
    out = []
    for flag in ("1", None):
        if flag:
            os.environ["MHX_NO_COMPACT"] = flag
        try:
            e = s.engine(mhx, 200, seed=21, adapt_mode=mode)
            e.init_chains(th0)
            e.adaptive_begin(6000, 10.0, 1)
            left = 1
            while left:
                left = e.adaptive_advance(250)   # many launches: many chances to repack
            st = e.state()
            out.append((st, e.lmatrix(), e.chain_status()[0], e.acceptance(1000)))
            if pooled:
                assert e.pooled()["refreshes"] >= 10 and e.pooled()["valid"]
            e.close()
        finally:
            os.environ.pop("MHX_NO_COMPACT", None)
    (a, la, sa, aa), (b, lb, sb, ab) = out
    os.environ.pop("MHX_COMPACT_ALWAYS", None)
    assert len(set(a["age"].tolist())) >= 2           # the walks really end at different times
    for k in ("theta", "logpost", "best_theta", "best_logpost", "age", "length"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(la, lb) and np.array_equal(sa, sb) and np.array_equal(aa, ab)
    # a second run on the same engine starts from the identity map again
    os.environ["MHX_COMPACT_ALWAYS"] = "1"
    e = s.engine(mhx, 40, seed=22)
    e.init_chains(th0[:40])
    for _ in range(2):
        e.adaptive_begin(3000, 10.0, 1)
        while e.adaptive_advance(500):
            pass
        assert (e.chain_status()[0] == mhx.capi.CHAIN_DONE).all()
    e.close()
    os.environ.pop("MHX_COMPACT_ALWAYS", None)
