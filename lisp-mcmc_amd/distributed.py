"""Multi-GPU plumbing: one process per GPU, chains sharded by contiguous global id ranges.

The path shards trivially (the reference maps a list of walkers sequentially,
mcmc-fitting.lisp:1029-1033): datasets are replicated, rank r owns chains
[offset, offset+count), and the Philox counter uses GLOBAL chain ids so results do not depend
on the partition.  The reference's per-walker adaptation needs no collective.  The pooled
adaptive covariance (MHX_ADAPT_POOLED) needs ONE sum of 1+d+d*d doubles every 200 iterations;
`torch_allreduce_hook` provides it through torch.distributed (backend "nccl" = RCCL over
xGMI on the engine's own device buffer; "gloo" on a host copy for CPU rehearsals).
"""
import ctypes as C

import numpy as np


def shard(total_chains, world, rank):
    """(offset, count) of rank's contiguous chain range; the first total % world ranks get one more"""
    q, r = divmod(int(total_chains), int(world))
    count = q + (1 if rank < r else 0)
    offset = rank * q + min(rank, r)
    return offset, count


class _DeviceArray:
    """minimal __cuda_array_interface__ view of `n` doubles at device address `ptr`"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}


def torch_allreduce_hook(dist, group=None):
    """Returns fn(buf, n, device_buffer) for Engine.set_allreduce: in-place SUM over ranks."""
    import torch

    def hook(buf, n, device_buffer):
        addr = C.cast(buf, C.c_void_p).value
        if device_buffer:
            t = torch.as_tensor(_DeviceArray(addr, n), device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.synchronize()
        else:
            a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(int(n),))
            t = torch.from_numpy(a)
            if dist.get_backend(group) == "nccl":
                g = t.cuda()
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
                t.copy_(g.cpu())
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return 0
    return hook
