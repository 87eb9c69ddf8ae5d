"""The communicator branch of the multi-GPU group (mhx_group_create -> ncclCommInitAll, one
ncclAllReduce per engine inside ncclGroupStart/End on the engines' own streams) on the ONE GPU a
test box has.  The real librccl wants a device per rank, so these tests run libmhx against the
stand-in of tests/stub_rccl (same eight entry points and calling conventions, host-staged sum in
rank order; MHX_RCCL_LIBRARY selects it, MHX_GROUP_FORCE_RCCL=1 lets engines that share a device
take the branch).  That switch exists only in tests/hooks/libmhx_hooks.so - the same objects as
libmhx.so with -DMHX_DEBUG_HOOKS, chosen through MHX_LIBRARY; the release library has no test
switches.  libmhx opens its RCCL once per process, so every case is a child process.

And the REAL librccl through the same branch: a forced group of ONE device (which the real
library accepts) runs ncclCommInitAll / ncclGroupStart / ncclAllReduce / ncclGroupEnd - the three
prototypes libmhx restates by hand - on the one GPU of the box.

Also here: the process-exit regression of round 2 (a full-suite process that had used RCCL and
hiprtc aborted with `double free or corruption` after its work was done, while librccl was opened
RTLD_GLOBAL) - one child process with the REAL library, torch imported, asserting exit status 0."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STUB = os.path.join(HERE, "stub_rccl", "librccl_stub.so")
HOOKS = os.path.join(HERE, "hooks", "libmhx_hooks.so")

PRELUDE = """
import os, sys
sys.path[:0] = [%r, %r]
import numpy as np
import lisp_mcmc_amd as mhx
import problems as pb

def walk(obj, th0, iters, l_matrix=None, chunk=150):
    obj.init_chains(th0)
    obj.adaptive_begin(30000, 10.0, 1, l_matrix=l_matrix)
    left = iters
    while left > 0:
        obj.adaptive_advance(min(left, chunk))
        left -= chunk
    return obj.state()
""" % (ROOT, HERE)


def run_child(body, tmp_path, stub=True, force=True, fail=None, extra_env=None):
    log = tmp_path / "rccl_calls.log"
    env = dict(os.environ, MHX_SPLIT="0")
    env.pop("MHX_STUB_RCCL_FAIL", None)
    if stub:
        assert os.path.exists(STUB), "tests/stub_rccl/librccl_stub.so is not built (__graft_entry__.build())"
        env["MHX_RCCL_LIBRARY"] = STUB
        env["MHX_STUB_RCCL_LOG"] = str(log)
    if force:
        assert os.path.exists(HOOKS), "tests/hooks/libmhx_hooks.so is not built (__graft_entry__.build())"
        env["MHX_GROUP_FORCE_RCCL"] = "1"
        env["MHX_LIBRARY"] = HOOKS
    if fail:
        env["MHX_STUB_RCCL_FAIL"] = fail
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(body)],
                         capture_output=True, text=True, env=env, timeout=600)
    calls = log.read_text().splitlines() if log.exists() else []
    return out, calls


def test_two_engines_through_the_communicator_branch(tmp_path):
    """group of two engines, pooled mode, ncclCommInitAll + grouped ncclAllReduce (stub):
    * bit for bit the group whose pool vectors are summed by libmhx itself through the host
      (group_pool_sum_local adds in rank order too) over six ticks, pooled factors adopted;
    * one engine holding all chains: same walk until the first pooled factor is adopted, pooled
      statistics equal up to the order of the additions;
    * the call pattern a multi-GPU node will see."""
    out, calls = run_child("""
        s = pb.two_peak(n=2000, seed=5)
        C_ = 64
        th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=7)
        mode = mhx.capi.ADAPT_POOLED
        l0 = np.diag(0.01 * np.abs(s.theta_star))
        g = mhx.Group(C_, s.d, s.K, devices=[0, 0], seed=11, adapt_mode=mode)
        s.apply(g)
        e = s.engine(mhx, C_, seed=11, adapt_mode=mode)
        a, b = walk(e, th0, 200, l0), walk(g, th0, 200, l0)
        assert np.array_equal(a["theta"], b["theta"]) and np.array_equal(a["age"], b["age"])
        pe, p0, p1 = e.pooled(), g.engines[0].pooled(), g.engines[1].pooled()
        assert pe["refreshes"] == p0["refreshes"] == p1["refreshes"] == 1
        assert np.array_equal(p0["stats"], p1["stats"]) and np.array_equal(p0["L"], p1["L"])
        assert pe["stats"][0] == p0["stats"][0] > C_
        assert np.allclose(pe["stats"], p0["stats"], rtol=1e-11, atol=1e-18)
        assert p0["valid"] and np.allclose(pe["L"], p0["L"], rtol=1e-8, atol=1e-16)
        for _ in range(8):
            g.adaptive_advance(150)
        sg = g.state()
        np.save(os.environ["OUT_NPY"], np.concatenate([sg["theta"].ravel(), sg["logpost"],
                                                       sg["age"].astype(float),
                                                       g.engines[1].pooled()["stats"]]))
        assert g.engines[0].pooled()["refreshes"] == 7
        e.close(); g.close()
        print("ok")
    """, tmp_path, extra_env={"OUT_NPY": str(tmp_path / "stub.npy")})
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
    # the same walk with libmhx's own host-staged sum between the two engines (no communicator)
    out2, calls2 = run_child("""
        s = pb.two_peak(n=2000, seed=5)
        C_ = 64
        th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=7)
        l0 = np.diag(0.01 * np.abs(s.theta_star))
        g = mhx.Group(C_, s.d, s.K, devices=[0, 0], seed=11, adapt_mode=mhx.capi.ADAPT_POOLED)
        s.apply(g)
        walk(g, th0, 200, l0)
        for _ in range(8):
            g.adaptive_advance(150)
        sg = g.state()
        np.save(os.environ["OUT_NPY"], np.concatenate([sg["theta"].ravel(), sg["logpost"],
                                                       sg["age"].astype(float),
                                                       g.engines[1].pooled()["stats"]]))
        g.close()
        print("ok")
    """, tmp_path, stub=False, force=False, extra_env={"OUT_NPY": str(tmp_path / "local.npy")})
    assert out2.returncode == 0 and "ok" in out2.stdout, out2.stderr[-3000:]
    import numpy as np
    assert np.array_equal(np.load(tmp_path / "stub.npy"), np.load(tmp_path / "local.npy"))
    # call pattern: one communicator per engine from ONE ncclCommInitAll; per tick a group of
    # exactly two all-reduces of 1 + d + d^2 doubles; both communicators destroyed at close
    assert calls[0] == "CommInitAll ndev=2 devices=0,0"
    ticks = [i for i, c in enumerate(calls) if c.startswith("GroupStart")]
    assert len(ticks) == 7
    for i in ticks:
        assert calls[i] == "GroupStart depth=1"
        assert calls[i + 1].startswith("AllReduce rank=0 comm_device=0 current_device=0 count=73 in_group=1")
        assert calls[i + 2].startswith("AllReduce rank=1 comm_device=0 current_device=0 count=73 in_group=1")
        assert calls[i + 3] == "GroupEnd depth=0 pending=2"
        assert calls[i + 4] == "flush clique_of=2 count=73 rc=0"
    assert [c for c in calls if c.startswith("CommDestroy")] == [
        "CommDestroy rank=0 device=0", "CommDestroy rank=1 device=0"]


@pytest.mark.parametrize("where", ["allreduce", "groupend"])
def test_a_failing_collective_surfaces_as_ecomm_and_ends_the_run(tmp_path, where):
    out, calls = run_child("""
        s = pb.two_peak(n=1500, seed=5)
        C_ = 32
        th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=7)
        g = mhx.Group(C_, s.d, s.K, devices=[0, 0], seed=3, adapt_mode=mhx.capi.ADAPT_POOLED)
        s.apply(g)
        g.init_chains(th0)
        g.adaptive_begin(30000, 10.0, 1)
        g.adaptive_advance(199)                      # no tick yet: nothing collective has run
        try:
            g.adaptive_advance(1)
            raise SystemExit("the failing collective went unnoticed")
        except mhx.MhxError as ex:
            assert ex.code == mhx.capi.ECOMM, ex
            assert "nccl" in str(ex), ex
        try:                                         # the run is over: host and device
            g.adaptive_advance(1)                    # bookkeeping may have come apart
            raise SystemExit("advance after a failed tick")
        except mhx.MhxError as ex:
            assert ex.code == mhx.capi.ESTATE, ex
        st = g.state()
        assert (st["age"] == 201).all()              # the 200 iterations themselves were taken
        g.adaptive_begin(30000, 10.0, 1)             # ... and a new run may begin
        g.adaptive_advance(10)
        g.close()
        print("ok")
    """, tmp_path, fail=where)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
    i = calls.index("GroupStart depth=1")
    # every ncclGroupStart is paired with an ncclGroupEnd, also when a call in between failed
    assert any(c.startswith("GroupEnd depth=0") for c in calls[i + 1:i + 4]), calls[i:i + 5]


def test_comm_init_all_failure_is_reported_by_group_create(tmp_path):
    out, calls = run_child("""
        s = pb.two_peak(n=1500, seed=5)
        try:
            mhx.Group(32, s.d, s.K, devices=[0, 0], seed=3, adapt_mode=mhx.capi.ADAPT_POOLED)
            raise SystemExit("group created without a communicator")
        except mhx.MhxError as ex:
            assert ex.code == mhx.capi.ECOMM and "ncclCommInitAll" in str(ex), ex
        # the per-walker mode needs no communicator at all
        g = mhx.Group(32, s.d, s.K, devices=[0, 0], seed=3)
        g.close()
        print("ok")
    """, tmp_path, fail="initall")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
    assert calls == ["CommInitAll ndev=2 FAIL(injected)"]


def test_one_rank_communicator_on_the_stub(tmp_path):
    """mhx_comm_init_rank's path (one process per GPU): ungrouped ncclAllReduce on the engine's
    stream between the statistics kernels and the factorisation"""
    out, calls = run_child("""
        s = pb.two_peak(n=2000, seed=6)
        C_ = 32
        th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=8)
        mode = mhx.capi.ADAPT_POOLED
        plain = s.engine(mhx, C_, seed=12, adapt_mode=mode)
        comm = s.engine(mhx, C_, seed=12, adapt_mode=mode)
        comm.comm_init_rank(mhx.comm_unique_id(), 0, 1)
        a, b = walk(plain, th0, 700), walk(comm, th0, 700)
        for k in ("theta", "logpost", "age"):
            assert np.array_equal(a[k], b[k]), k
        assert plain.pooled()["refreshes"] == comm.pooled()["refreshes"] == 3
        plain.close(); comm.close()
        print("ok")
    """, tmp_path, force=False)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
    assert calls[0].startswith("GetUniqueId") and calls[1].startswith("CommInitRank")
    assert sum(c.startswith("AllReduce rank=0") and c.endswith("in_group=0") for c in calls) == 3
    assert calls[-1].startswith("CommDestroy")


def test_a_librccl_of_another_hip_runtime_is_refused(tmp_path):
    """MHX_RCCL_LIBRARY pointing at something that is not an RCCL: the loader says why instead of
    crashing later (and a library bound to a second HIP runtime is refused the same way, by
    comparing the address both sides resolve hipStreamSynchronize to)"""
    out, _ = run_child("""
        try:
            mhx.comm_unique_id()
            raise SystemExit("no error")
        except mhx.MhxError as ex:
            assert ex.code == mhx.capi.ECOMM and "RCCL unavailable" in str(ex), ex
        print("ok")
    """, tmp_path, stub=False, force=False,
                       extra_env={"MHX_RCCL_LIBRARY": os.path.join(ROOT, "oracle", "libmhx_oracle.so")})
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]


def test_process_with_rccl_and_hiprtc_exits_cleanly(tmp_path):
    """round 2's exit abort, pinned: torch imported, a run-time compiled expression model (hiprtc +
    comgr loaded), the REAL librccl in a 1-rank communicator, engines left for the interpreter's
    exit to collect.  While librccl was opened RTLD_GLOBAL such processes ended in `double free
    or corruption (!prev)` after their work was done; opened RTLD_LOCAL they exit 0."""
    out, _ = run_child("""
        import torch
        s = pb.two_peak(n=2000, seed=6)
        C_ = 32
        th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=8)
        mode = mhx.capi.ADAPT_POOLED
        rc = s.engine(mhx, C_, seed=12, adapt_mode=mode)
        rc.comm_init_rank(mhx.comm_unique_id(), 0, 1)     # real ncclCommInitRank
        plain = s.engine(mhx, C_, seed=12, adapt_mode=mode)
        a, b = walk(plain, th0, 450), walk(rc, th0, 450)  # two ticks through ncclAllReduce
        assert np.array_equal(a["theta"], b["theta"])
        ex = s.engine(mhx, 8, seed=1)
        keys, cexpr = mhx.sexpr.lambda_to_expr(
            "(lambda (x &key b0 b1 a1 mu1 w1 a2 mu2 w2 &allow-other-keys)"
            " (+ (+ b0 (* b1 x)) (* a1 (exp (- (expt (/ (- x mu1) w1) 2))))"
            "    (* a2 (exp (- (expt (/ (- x mu2) w2) 2))))))")
        ex.set_expr_recognition(False)                          # as written: hiprtc
        ex.set_function_expr(0, cexpr, keys, list(range(8)))
        ex.init_chains(th0[:8])
        assert "rtc" in ex.kernel_name()
        assert torch.cuda.is_available()
        print("ok", flush=True)
        # no close(): what is alive here is torn down by the interpreter and the libraries' own
        # exit handlers, as at the end of a pytest process
    """, tmp_path, stub=False, force=False)
    assert "ok" in out.stdout, out.stderr[-3000:]
    assert out.returncode == 0, (out.returncode, out.stderr[-3000:])


def test_one_device_group_through_the_real_rccl(tmp_path):
    """the first N > 1 run is the driver's, and a wrong hand-restated prototype
    of ncclCommInitAll / ncclGroupStart / ncclGroupEnd (csrc/mhx_engine.cpp) would show only
    there.  A forced group of ONE device takes the communicator branch against the REAL librccl:
    ncclCommInitAll(comms, 1, {0}), and per pooled tick ncclGroupStart, one ncclAllReduce of 73
    doubles on the engine's stream, ncclGroupEnd.  450 iterations (two ticks) = the plain pooled
    engine bit for bit; exit status 0 with everything left to the interpreter's exit."""
    out, _ = run_child("""
        s = pb.two_peak(n=2000, seed=5)
        C_ = 64
        th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=7)
        mode = mhx.capi.ADAPT_POOLED
        assert mhx.capi.lib().mhx_build_id().decode().endswith("+hooks")
        g = mhx.Group(C_, s.d, s.K, devices=[0], seed=11, adapt_mode=mode)
        s.apply(g)
        plain = s.engine(mhx, C_, seed=11, adapt_mode=mode)
        a, b = walk(plain, th0, 450), walk(g, th0, 450)
        for k in ("theta", "logpost", "best_theta", "age", "length"):
            assert np.array_equal(a[k], b[k]), k
        pa, pg = plain.pooled(), g.engines[0].pooled()
        assert pg["refreshes"] == 2 and pa["refreshes"] == 2
        assert np.array_equal(pa["stats"], pg["stats"]) and np.array_equal(pa["L"], pg["L"])
        assert pg["valid"] == pa["valid"]
        print("ok", flush=True)
    """, tmp_path, stub=False, force=True)
    assert "ok" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])
    assert out.returncode == 0, (out.returncode, out.stderr[-3000:])


def test_bench_gpus_2_in_one_process_rehearsed_on_one_gpu(tmp_path):
    """`python3 bench.py --gpus 2` without a launcher: ONE process, mhx.Group over two engines,
    pooled adaptation with the tick's collective inside the timed region - rehearsed here with
    both engines on device 0 and the stand-in librccl.  One JSON line, whole-job rate, the
    contract's fields."""
    import json
    env = dict(os.environ, MHX_RCCL_LIBRARY=STUB, MHX_GROUP_FORCE_RCCL="1", MHX_LIBRARY=HOOKS,
               MHX_STUB_RCCL_LOG=str(tmp_path / "calls.log"))
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env.pop("MHX_SPLIT", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0",
                          "--chains", "512", "--steps", "200", "--warmup", "200", "--no-cpu",
                          "--spin-ms", "5"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 200 and d["warmup"] == 200 and d["scaling"] == "weak"
    assert d["config"]["chains_per_gpu"] == 512 and "ncclCommInitAll" in d["config"]["collective"]
    assert "one host process" in d["config"]["parallelism"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 / (2 * 512) - 1.0) < 1e-9
    assert len(d["roofline"]["kernel_ms_per_gpu"]) == 2
    calls = (tmp_path / "calls.log").read_text().splitlines()
    assert calls[0] == "CommInitAll ndev=2 devices=0,0"
    # warm-up ends on the tick at iteration 200, the timed region on the one at 400: one tick each
    assert sum(c.startswith("GroupStart") for c in calls) == 2


def test_bench_under_the_launcher_rehearsed_on_one_gpu(tmp_path):
    """The driver's N > 1 form: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus
    2 ...`, one rank per GPU.  Rehearsed with MHX_BENCH_REHEARSE_LAUNCHER=1: both ranks on device
    0, torch.distributed on gloo, the tick's sum through the torch hook - every other line of the
    per-rank path (rendezvous, chain ranges by rank, barriers, max-over-ranks time, summed steps,
    rank 0's one JSON line) is what runs on a node.  Two ticks fall into the run (warm-up ends on
    iteration 200, the timed region on 400)."""
    import json
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MHX_BENCH_REHEARSE_LAUNCHER="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MHX_SPLIT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                          str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--chains", "512",
                          "--steps", "200", "--warmup", "200", "--no-cpu", "--spin-ms", "5"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 200 and d["warmup"] == 200 and d["scaling"] == "weak"
    assert d["config"]["chains_per_gpu"] == 512 and "one process per GPU" in d["config"]["parallelism"]
    assert "torch.distributed all_reduce hook" in d["config"]["collective"]
    # whole-job rate: both ranks' chain-steps over the slower rank's time
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 / (2 * 512) - 1.0) < 1e-9
    assert "cpu_baseline" not in d and "value_direct_form" not in d   # (rank 0 at N = 1 only)


@pytest.mark.skipif("__import__('torch').cuda.device_count() < 2")
def test_two_real_gpus_walk_like_two_engines_on_one():
    """(runs only where two GPUs are visible) Group(devices=[0, 1]) through the REAL librccl
    against Group(devices=[0, 0]) through libmhx's host-staged sum: the same chains, pooled
    statistics equal up to the order of the two additions."""
    import numpy as np
    import lisp_mcmc_amd as mhx
    import problems as pb
    s = pb.two_peak(n=2000, seed=5)
    C_ = 64
    th0 = pb.perturbed(s.theta_star, C_, 0.01, seed=7)
    l0 = np.diag(0.01 * np.abs(s.theta_star))
    res = []
    for devs in ([0, 1], [0, 0]):
        g = mhx.Group(C_, s.d, s.K, devices=devs, seed=11, adapt_mode=mhx.capi.ADAPT_POOLED)
        s.apply(g)
        g.init_chains(th0)
        g.adaptive_begin(30000, 10.0, 1, l_matrix=l0)
        for _ in range(4):
            g.adaptive_advance(150)
        res.append((g.state(), g.engines[0].pooled(), g.engines[1].pooled()))
        g.close()
    (sa, a0, a1), (sb, b0, b1) = res
    assert np.array_equal(a0["stats"], a1["stats"]) and a0["refreshes"] == b0["refreshes"] == 3
    assert np.allclose(a0["stats"], b0["stats"], rtol=1e-12, atol=1e-18)
    assert (sa["age"] == sb["age"]).all() and np.isfinite(sa["logpost"]).all()


def test_config5_in_its_eight_engine_form_on_one_gpu(tmp_path):
    """BASELINE config 5 as the node will run it - 524288 chains, 8 engines x 65536, pooled
    adaptive covariance, one all-reduce of 73 doubles per 200 iterations over 8 communicators from
    ONE ncclCommInitAll - with the eight engines on the one GPU a test box has and the stand-in
    librccl as the transport: the partition, the tick at full size (its statistics the sum over
    all 524288 chains, identical on every engine, its factor the Cholesky factor of the covariance
    they describe), global chain ids up to 524287 in the Philox counters (chains equal 8-chain
    engines that own the same ids), every chain moved."""
    out, calls = run_child("""
        import bench
        spec, chains, _, _ = bench.synth_workload("c5")
        assert chains == 65536 and spec.d == 8
        N = 8 * chains
        g = mhx.Group(N, spec.d, spec.K, devices=[0] * 8, seed=21, adapt_mode=mhx.capi.ADAPT_POOLED)
        assert g.ranges == [(i * chains, chains) for i in range(8)]
        spec.apply(g)
        rng = np.random.Generator(np.random.Philox(key=99))
        th0 = spec.theta_star[None, :] * (1.0 + 0.01 * rng.standard_normal((N, spec.d)))
        l0 = np.diag(0.01 * np.abs(spec.theta_star))
        g.init_chains(th0)
        g.adaptive_begin(30000, 10.0, 1, l_matrix=l0)
        n_it = 212
        assert g.adaptive_advance(n_it) == N
        st = g.state()
        assert (st["age"] == n_it + 1).all() and g.counters()[0] == N * n_it
        assert np.isfinite(st["logpost"]).all()
        pools = [e.pooled() for e in g.engines]
        d = spec.d
        for p in pools:
            assert p["refreshes"] == 1 and p["valid"]
            assert np.array_equal(p["stats"], pools[0]["stats"]) and np.array_equal(p["L"], pools[0]["L"])
        n = pools[0]["stats"][0]
        assert n == np.floor(n) and n > N
        mean = pools[0]["stats"][1:1 + d] / n
        cov = pools[0]["stats"][1 + d:].reshape(d, d) / n - np.outer(mean, mean)
        assert np.allclose(pools[0]["L"], (2.38 ** 2 / d) * np.linalg.cholesky(cov), rtol=1e-9, atol=1e-18)
        # one engine's share on its own sees one eighth of the displacements, give or take
        one = spec.engine(mhx, chains, seed=21, chain_offset=3 * chains, adapt_mode=mhx.capi.ADAPT_POOLED)
        one.init_chains(th0[3 * chains:4 * chains])
        one.adaptive_begin(30000, 10.0, 1, l_matrix=l0)
        one.adaptive_advance(200)
        assert 0.11 < one.pooled()["stats"][0] / n < 0.14
        one.close()
        for lo in (0, 3 * chains + 40000, N - 8):
            small = spec.engine(mhx, 8, seed=21, chain_offset=lo, adapt_mode=mhx.capi.ADAPT_POOLED)
            small.init_chains(th0[lo:lo + 8])
            small.adaptive_begin(30000, 10.0, 1, l_matrix=l0)
            small.adaptive_advance(n_it)
            ss = small.state()
            assert np.array_equal(st["theta"][lo:lo + 8], ss["theta"]), lo
            assert np.array_equal(st["logpost"][lo:lo + 8], ss["logpost"]), lo
            small.close()
        g.close()
        print("ok")
    """, tmp_path)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
    assert calls[0] == "CommInitAll ndev=8 devices=0,0,0,0,0,0,0,0"
    i = calls.index("GroupStart depth=1")
    assert [c.split()[1] for c in calls[i + 1:i + 9]] == ["rank=%d" % r for r in range(8)]
    assert calls[i + 9] == "GroupEnd depth=0 pending=8" and calls[i + 10] == "flush clique_of=8 count=73 rc=0"
    assert sum(c.startswith("CommDestroy") for c in calls) == 8


@pytest.mark.parametrize("form,chains", [("split", 2), ("tsplit", 20)])
def test_persistent_launch_whose_sweepers_never_come_is_taken_back(tmp_path, form, chains):
    """k_persist and k_persist_ts count on all their workgroups being on the GPU at once.  When they are not
    (another kernel holds the GPU for longer than a master's patience) the master takes its
    proposal back, the launch ends, mhx_adaptive_advance answers MHX_EDEVICE with no chain
    touched, and the engine goes back to two launches per iteration - after a new begin the walk is
    the one a two-launch engine walks.  The test library's MHX_TEST_LOSE_SWEEPERS=1 makes the
    sweep workgroups leave at once."""
    env = dict(os.environ, MHX_LIBRARY=HOOKS, MHX_TEST_LOSE_SWEEPERS="1")
    for k in ("MHX_SPLIT", "MHX_TSPLIT", "MHX_NO_PERSIST"):
        env.pop(k, None)
    if form == "split":
        env["MHX_TSPLIT"] = "0"  # (the per-chain persistent form)
    body = PRELUDE + textwrap.dedent("""
        FORM, CHAINS = "@FORM@", @CHAINS@
        s = pb.two_peak(n=30000, seed=3)
        th0 = pb.perturbed(s.theta_star, CHAINS, 0.01, seed=2)
        e = s.engine(mhx, CHAINS, seed=9)
        e.init_chains(th0)
        assert "persistent " + FORM + " x" in e.kernel_name(), e.kernel_name()
        before = e.state()
        e.adaptive_begin(900, 10.0, 1)
        try:
            e.adaptive_advance(50)
            raise SystemExit("the launch should have failed")
        except mhx.MhxError as ex:
            assert ex.code == mhx.capi.EDEVICE and "persistent launch" in str(ex), ex
        after = e.state()
        for k in ("theta", "logpost", "age", "length"):
            assert np.array_equal(before[k], after[k]), k      # no chain was touched
        assert (e.chain_status()[0] == mhx.capi.CHAIN_RUNNING).all()
        try:
            e.adaptive_advance(10)
            raise SystemExit("the run should be over")
        except mhx.MhxError as ex:
            assert ex.code == mhx.capi.ESTATE
        assert "persistent" not in e.kernel_name() and " " + FORM + " x" in e.kernel_name(), e.kernel_name()
        e.adaptive_begin(900, 10.0, 1)
        e.adaptive_advance(1 << 40)
        os.environ["MHX_NO_PERSIST"] = "1"
        ref = s.engine(mhx, CHAINS, seed=9)
        ref.init_chains(th0)
        ref.adaptive_begin(900, 10.0, 1)
        ref.adaptive_advance(1 << 40)
        a, b = e.state(), ref.state()
        for k in ("theta", "logpost", "age", "length"):
            assert np.array_equal(a[k], b[k]), k
        print("ok", flush=True)
    """).replace("@FORM@", form).replace("@CHAINS@", str(chains))
    out = subprocess.run([sys.executable, "-c", body], capture_output=True, text=True, env=env, timeout=600)
    assert "ok" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])
    assert out.returncode == 0
