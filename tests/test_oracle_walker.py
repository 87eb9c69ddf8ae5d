"""CPU tests of the oracle's walker and controller against hand-derivable behaviour of the
reference (mcmc-fitting.lisp, M:).  No GPU."""
import math

import numpy as np
import pytest

import problems as pb


def flat_walker(orc, d=4, y=1e6):
    """posterior depends on the LAST parameter only (constant model vs one datum)"""
    p = orc.Problem(d, 1)
    p.set_function(0, pb.POLY, (), [d - 1])
    p.set_dataset(0, [0.0], [y], 1.0, pb.NORMAL)
    return p


def test_take_step_accept_rule(orc):
    p = flat_walker(orc, d=2, y=0.0)
    w = orc.Walker(p, [5.0, 3.0])          # prob0 = -1/2 log 2pi - 9/2
    L = np.eye(2)
    prob0 = w.last()[1]
    assert prob0 == pytest.approx(-0.5 * math.log(2 * math.pi) - 4.5, rel=1e-15)
    # uphill: accepted whatever u is (M:1091 first clause)
    assert w.take_step_injected(L, [7.0, -1.0], 1e-300) == 1
    th, pr = w.last()
    assert np.array_equal(th, [12.0, 2.0]) and pr > prob0
    assert w.length == 2 and w.age == 2
    # downhill by delta: accepted iff delta/T > log u (M:1092)
    delta = (-0.5 * 9.0) - (-0.5 * 4.0)
    u_edge = math.exp(delta)
    w2 = orc.Walker(p, [0.0, 2.0])
    assert w2.take_step_injected(L, [0.0, 1.0], u_edge * 1.01) == 0   # log u > delta
    th, pr = w2.last()
    assert np.array_equal(th, [0.0, 2.0])                              # previous step pushed again
    assert w2.length == 2 and w2.age == 2
    assert w2.take_step_injected(L, [0.0, 1.0], u_edge * 0.99) == 1
    # temperature divides the log ratio
    w3 = orc.Walker(p, [0.0, 2.0])
    assert w3.take_step_injected(L, [0.0, 1.0], math.exp(delta / 10) * 0.99, 10.0) == 1
    w4 = orc.Walker(p, [0.0, 2.0])
    assert w4.take_step_injected(L, [0.0, 1.0], math.exp(delta / 10) * 1.01, 10.0) == 0
    # most-likely step needs a strictly greater prob (M:553)
    bt, bp = w2.best()
    assert np.array_equal(bt, [0.0, 2.0])


def test_acceptance_is_runs_over_take(orc):
    p = flat_walker(orc, d=1, y=0.0)
    w = orc.Walker(p, [10.0])
    L = np.eye(1)
    pattern = [1, 1, 0, 0, 0, 1, 0, 1, 1, 0]     # 1 = move downhill-but-accepted (u tiny) / uphill
    for a in pattern:
        if a:
            assert w.take_step_injected(L, [-0.5], 1e-300) == 1   # towards 0: uphill
        else:
            assert w.take_step_injected(L, [+50.0], 1.0) == 0     # far downhill, log u = 0
    # walk newest-first has 11 entries; runs = 1 + number of changes
    num, den = w.acceptance(100)
    assert den == 11 and num == 1 + sum(pattern)
    num, den = w.acceptance(4)      # newest 4 probs: steps 10,9,8,7 -> pattern[9]=0 keeps, ...
    assert den == 4
    probs, _ = w.trace(4)
    runs = 1 + sum(probs[i] != probs[i + 1] for i in range(3))
    assert num == runs
    assert w.forward_count(100) == sum(pattern)   # every accepted move here was uphill


def test_l_matrix_reproduces_reference_kat(orc, golden):
    """forward-step displacements equal to example-lplist (M:729-733) must give the factor the
    reference prints at M:749-751 in the leading 3x3 block"""
    p = flat_walker(orc, d=4)
    th = np.zeros(4)
    w = orc.Walker(p, th)
    L = np.eye(4)
    rows = golden["example_lplist"]
    # 6 uphill steps -> 6 forward steps -> 5 displacements, newest first = rows[0..4] in the
    # order the reference's example lists them (the first step's own displacement is unused)
    for r in [rows[0]] + rows[::-1]:
        z = np.array([-r[0], -r[1], -r[2], 1.0])   # displacement (older - newer) = +row
        assert w.take_step_injected(L, z, 0.5) == 1
    st, Lm, nf = w.l_matrix(500)
    assert st == orc.L_OK and nf == 6
    assert np.array_equal(Lm[:3, :3], np.array(golden["example_l_matrix"]))
    assert np.all(Lm[3] == 0.0) and np.all(Lm[:, 3] == 0.0)
    # window semantics: the oldest step of the window is never a forward step (M:498)
    assert w.forward_count(3) == 2
    st, _, nf = w.l_matrix(2)
    assert (st, nf) == (orc.L_EMPTY, 1)
    st, _, nf = w.l_matrix(1)
    assert (st, nf) == (orc.L_CAUGHT, 0)     # (elt nil 0) -> type-error, handled at M:894


def test_l_matrix_matches_numpy(orc):
    s = pb.two_peak(n=200, seed=2)
    op = s.oracle(orc)
    w = orc.Walker(op, s.theta_star)
    w.many_steps(600, np.diag(0.01 * np.abs(s.theta_star)), seed=5, chain_id=0)
    st, Lm, nf = w.l_matrix(500)
    assert st == orc.L_OK
    prob, th = w.trace(500)
    fwd = [i for i in range(len(prob) - 1) if prob[i] > prob[i + 1]]
    assert nf == len(fwd)
    diffs = np.array([th[fwd[k + 1]] - th[fwd[k]] for k in range(len(fwd) - 1)])
    cov = np.cov(diffs.T, bias=True)
    assert np.allclose(Lm @ Lm.T, cov, rtol=1e-9, atol=1e-18)


def test_controller_rewind_quirk_and_schedule(orc):
    """n < 2000: at i = 1 the tail test fires, i <- n - 2000 < 0, 2000 steps follow at T = 1"""
    p = flat_walker(orc, d=1, y=0.0)
    w = orc.Walker(p, [1.0])
    w.adaptive_begin(100, 10.0, 1, seed=1)
    assert np.array_equal(w.current_l(), [[1.0]])      # diag(most-likely params), M:899
    assert w.temperature == 10.0 and w.loop_index == 1
    w.adaptive_advance(1)
    assert w.loop_index == 100 - 2000 + 1 and w.temperature == 1.0
    assert w.adaptive_advance(1 << 40) == orc.DONE
    assert w.age == 2001


def test_controller_annealing_follows_schedule(orc):
    s = pb.two_peak(n=50, seed=3)
    op = s.oracle(orc)
    w = orc.Walker(op, s.theta_star)
    n = 30000
    w.adaptive_begin(n, 10.0, 0, seed=2)
    t = orc.temperature_schedule(n, s.d, 10.0)
    for i in (1, 2, 50, 199):
        w.adaptive_advance(i - (w.loop_index - 1) - 0) if False else None
    w.adaptive_advance(150)
    # after the body of iteration i the temperature is temps[i] (M:920-921); loop_index = i + 1
    assert w.loop_index == 151 and w.temperature == t[150]


def test_controller_scales_l_on_acceptance(orc):
    """all proposals rejected -> acceptance(200) = 1/200 < 0.2 -> L <- 0.1 L at i = 200 (M:939-940);
    all accepted -> 200/200 > 0.4 -> L <- 1.9 L (M:941-942)"""
    p = flat_walker(orc, d=1, y=0.0)
    w = orc.Walker(p, [0.0])
    big = np.array([[1e9]])
    w.adaptive_begin(30000, 1.0, 0, l_matrix=big, seed=3)
    w.adaptive_advance(199)
    assert np.array_equal(w.current_l(), big)
    w.adaptive_advance(1)
    assert w.acceptance(200) == (1, 200)
    assert w.current_l()[0, 0] == 0.1 * 1e9
    # a nearly flat posterior: almost every proposal is taken
    p2 = orc.Problem(1, 1)
    p2.set_function(0, pb.POLY, (), [0])
    p2.set_dataset(0, [0.0], [0.0], 100.0, pb.NORMAL)
    w2 = orc.Walker(p2, [1.0])
    w2.adaptive_begin(30000, 1.0, 0, l_matrix=np.array([[1.0]]), seed=4)
    w2.adaptive_advance(200)
    num, den = w2.acceptance(200)
    assert num / den > 0.4
    assert w2.current_l()[0, 0] == 1.9


def test_acceptance_thresholds_are_single_floats(orc):
    """(< acc 0.2) compares the exact rational with the SINGLE float 0.2 = 0.2000000029...:
    40/200 counts as below 0.2 (M:930, 939)"""
    assert float(np.float32(0.2)) > 0.2
    assert 40 < float(np.float32(0.2)) * 200
    assert not (80 > float(np.float32(0.4)) * 200)


def test_estop_and_status(orc):
    p = flat_walker(orc, d=1, y=0.0)
    w = orc.Walker(p, [1.0])
    w.adaptive_begin(10000, 10.0, 1, seed=1)
    w.adaptive_advance(5)
    w.request_stop()
    assert w.adaptive_advance(5) == orc.STOPPED and w.loop_index == 6
    # begin clears the flag (M:865)
    w.adaptive_begin(10000, 10.0, 1, seed=1)
    assert w.adaptive_advance(5) == orc.RUNNING


def test_non_finite_logpost_is_a_trap(orc):
    p = orc.Problem(2, 1)
    p.set_function(0, pb.POLY, (), [0, 1])
    p.set_dataset(0, [0.0, 1.0], [0.0, 1.0], 1.0, pb.NORMAL)
    w = orc.Walker(p, [0.0, 1.0])
    assert w.take_step_injected(np.eye(2) * 1e308, [1e10, 1e10], 0.5) == -1
    assert w.length == 1
