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
