"""world_size-2 gloo rehearsal of the N > 1 path on CPU: chain sharding by global id and the
pooled-statistics all-reduce exactly as the engine invokes it (through the C callback type of
include/mhx.h), with the oracle standing in for the per-rank chains."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def pooled_stats(orc, walkers, d, take=500):
    """(n, sum delta, sum delta delta^T) of the forward-step displacements (older - newer) of
    each walker's newest `take` steps, summed over walkers: what k_pool_stats + k_pool_reduce do"""
    out = np.zeros(1 + d + d * d)
    for w in walkers:
        prob, th = w.trace(take)
        fwd = [i for i in range(len(prob) - 1) if prob[i] > prob[i + 1]]
        if len(fwd) < 2:
            continue
        v = np.array([th[fwd[k + 1]] - th[fwd[k]] for k in range(len(fwd) - 1)])
        out[0] += len(v)
        out[1:1 + d] += v.sum(0)
        out[1 + d:] += (v.T @ v).ravel()
    return out


def make_walkers(orc, spec, ids, steps=300):
    import problems as pb
    op = spec.oracle(orc)
    th0 = pb.perturbed(spec.theta_star, 64, 0.01, seed=3)   # row = GLOBAL chain id
    L = np.diag(0.01 * np.abs(spec.theta_star))
    ws = []
    for g in ids:
        w = orc.Walker(op, th0[g])
        w.many_steps(steps, L, seed=42, chain_id=g)
        ws.append(w)
    return op, ws


def _rank_main(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import lisp_mcmc_amd as mhx
    from lisp_mcmc_amd import distributed as mdist  # noqa
    import oraclelib as orc
    import problems as pb
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total = 13
        off, cnt = mdist.shard(total, world, rank)
        spec = pb.two_peak(n=150, seed=8)
        op, ws = make_walkers(orc, spec, range(off, off + cnt))
        local = pooled_stats(orc, ws, spec.d)
        # the engine's call: C function pointer, host buffer, n doubles, device_buffer = 0
        hook = mdist.torch_allreduce_hook(dist)
        cb = mhx.capi.ALLREDUCE_FN(lambda ctx, buf, n, dev: hook(buf, n, dev))
        buf = (C.c_double * local.size)(*local)
        rc = cb(None, C.cast(buf, mhx.capi.f64p), local.size, 0)
        q.put((rank, off, cnt, rc, np.array(buf[:]), [w.last()[0] for w in ws]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_partition():
    sys.path.insert(0, ROOT)
    from lisp_mcmc_amd import distributed as mdist
    for total in (1, 7, 8, 13, 4096, 524288):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                off, cnt = mdist.shard(total, world, r)
                seen += list(range(off, off + cnt))
            assert seen == list(range(total))


def test_world2_gloo_pooled_allreduce(orc):
    import torch.multiprocessing as mp
    import problems as pb
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference over all 13 chains
    spec = pb.two_peak(n=150, seed=8)
    op, ws = make_walkers(orc, spec, range(13))
    full = pooled_stats(orc, ws, spec.d)
    assert [(r[1], r[2]) for r in res] == [(0, 7), (7, 6)]
    for rank, off, cnt, rc, summed, thetas in res:
        assert rc == 0
        assert np.allclose(summed, full, rtol=1e-13, atol=1e-300)
        # chains are functions of their GLOBAL id only: identical to the unsharded run
        for i in range(cnt):
            assert np.array_equal(thetas[i], ws[off + i].last()[0])
    assert full[0] > 100


def test_c_abi_partition_and_group_without_a_device():
    """mhx_group_partition (host logic, no device needed) is the sharding rule of the Python
    plumbing; mhx_group_create fails loudly without a GPU (no CPU path)"""
    sys.path.insert(0, ROOT)
    import lisp_mcmc_amd as mhx
    from lisp_mcmc_amd import distributed as mdist
    for total in (1, 7, 8, 13, 4096, 524288):
        for world in (1, 2, 3, 8):
            if total < world:
                continue
            assert [mhx.partition(total, world, r) for r in range(world)] == \
                   [mdist.shard(total, world, r) for r in range(world)]
    with pytest.raises(mhx.MhxError):
        mhx.partition(10, 0, 0)
    n = C.c_int(-1)
    if mhx.capi.lib().mhx_device_count(C.byref(n)) == mhx.capi.OK and n.value > 0:
        return  # a GPU is present: tests/test_gpu_group.py covers the rest
    with pytest.raises(mhx.MhxError) as ei:
        mhx.Group(16, 2, devices=[0, 1])
    assert ei.value.code == mhx.capi.EDEVICE
