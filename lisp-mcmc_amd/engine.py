"""Engine: the batched C ABI (include/mhx.h) as a Python object.  Plumbing only -- all
arithmetic happens in libmhx.so's gfx950 kernels."""
import ctypes as C

import numpy as np

from . import _capi as capi


class Engine:
    def __init__(self, n_chains, n_params, n_functions=1, device=0, seed=0, chain_offset=0,
                 adapt_mode=capi.ADAPT_FAITHFUL, history_capacity=0, poisson_logfact_double=False):
        cfg = capi.Config()
        cfg.n_chains, cfg.n_params, cfg.n_functions = int(n_chains), int(n_params), int(n_functions)
        cfg.device, cfg.adapt_mode, cfg.seed = int(device), int(adapt_mode), int(seed)
        cfg.chain_offset, cfg.history_capacity = int(chain_offset), int(history_capacity)
        cfg.poisson_logfact_double = int(bool(poisson_logfact_double))
        self.n_chains, self.d, self.K = int(n_chains), int(n_params), int(n_functions)
        self._h = C.c_void_p()
        self._cb = None
        capi.check(capi.lib().mhx_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            capi.lib().mhx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- problem definition (walker-create) -------------------------------
    def set_function(self, k, model, shape=(), param_index=()):
        sh, shp = capi.as_i32(list(shape) if len(shape) else [0])
        ix, ixp = capi.as_i32(list(param_index))
        capi.check(capi.lib().mhx_set_function(self._h, k, model, shp, len(shape), ixp, len(ix)))

    @staticmethod
    def _names(names):
        arr = (C.c_char_p * max(len(names), 1))(*[n.encode() for n in names])
        return arr

    def set_function_expr(self, k, expr, names, param_index):
        """function k as a C-syntax expression over x and `names` (see include/mhx.h)"""
        ix, ixp = capi.as_i32(list(param_index))
        capi.check(capi.lib().mhx_set_function_expr(self._h, k, expr.encode(), self._names(names),
                                                    ixp, len(ix)))

    def set_expr_recognition(self, on):
        """on=False: every expression of this engine is compiled exactly as written (libmhx
        otherwise serves polynomial + Gaussian / Lorentzian peak bodies with the enumerated
        models' kernels: include/mhx.h, mhx_set_function_expr)"""
        capi.check(capi.lib().mhx_set_expr_recognition(self._h, 1 if on else 0))

    def set_prior_expr(self, k, expr, names, index):
        """body of function k's prior over bounds_total and `names` (global parameter indices)"""
        ix, ixp = capi.as_i32(list(index))
        capi.check(capi.lib().mhx_set_prior_expr(self._h, k, (expr or "").encode(),
                                                 self._names(names), ixp, len(ix)))

    def set_likelihood_expr(self, k, expr):
        """per-point log-likelihood term of function k over y, model, error (dataset k must use
        LIK_EXPR, function k an expression model)"""
        capi.check(capi.lib().mhx_set_likelihood_expr(self._h, k, expr.encode()))

    def set_dataset(self, k, x, y, sigma=None, likelihood=capi.LIK_NORMAL):
        """x: [n], or [n][2] - a vector-valued x, one row per point, as the reference's x list holds
        it (mcmc-fitting.lisp:1136-1137): mhx_set_dataset_cols"""
        xa, xp = capi.as_f64(x)
        ya, yp = capi.as_f64(y)
        if xa.ndim == 2 and xa.shape[0] == ya.shape[0] and ya.ndim == 1:
            cols = [np.ascontiguousarray(xa[:, j]) for j in range(xa.shape[1])]
            ptrs = (capi.f64p * len(cols))(*[c.ctypes.data_as(capi.f64p) for c in cols])
            sp = None
            if sigma is not None:
                sa, sp = capi.as_f64(np.broadcast_to(np.asarray(sigma, dtype=np.float64), ya.shape))
            capi.check(capi.lib().mhx_set_dataset_cols(self._h, k, ptrs, len(cols), yp, sp, ya.size,
                                                       likelihood))
            return
        if xa.shape != ya.shape or xa.ndim != 1:
            raise ValueError("x and y must be 1-d and of equal length (or x [n][2])")
        if sigma is None:
            sp = None
        else:
            sa, sp = capi.as_f64(np.broadcast_to(np.asarray(sigma, dtype=np.float64), xa.shape))
        capi.check(capi.lib().mhx_set_dataset(self._h, k, xp, yp, sp, xa.size, likelihood))

    def set_bounds(self, k, idx, lo, hi):
        ix, ixp = capi.as_i32(list(idx))
        la, lp = capi.as_f64(lo)
        ha, hp = capi.as_f64(hi)
        capi.check(capi.lib().mhx_set_bounds(self._h, k, ixp, lp, hp, len(ix)))

    def init_chains(self, theta0):
        th = np.ascontiguousarray(theta0, dtype=np.float64)
        if th.shape == (self.d,):
            bc = 1
        elif th.shape == (self.n_chains, self.d):
            bc = 0
        else:
            raise ValueError("theta0 must be [d] or [n_chains, d]")
        capi.check(capi.lib().mhx_init_chains(self._h, th.ctypes.data_as(capi.f64p), bc))

    # ---- evaluation / parity hooks ------------------------------------------
    def logpost(self, theta, parts=False):
        th = np.ascontiguousarray(theta, dtype=np.float64).reshape(-1, self.d)
        out = np.zeros(th.shape[0])
        pr = np.zeros((th.shape[0], 2))
        capi.check(capi.lib().mhx_logpost(self._h, th.ctypes.data_as(capi.f64p), th.shape[0],
                                          out.ctypes.data_as(capi.f64p),
                                          pr.ctypes.data_as(capi.f64p)))
        return (out, pr) if parts else out

    def step_injected(self, L, z, u, T=None):
        La = np.ascontiguousarray(L, dtype=np.float64)
        per_chain = 1 if La.ndim == 3 else 0
        za = np.ascontiguousarray(z, dtype=np.float64).reshape(self.n_chains, self.d)
        ua = np.ascontiguousarray(u, dtype=np.float64).reshape(self.n_chains)
        Ta = np.ascontiguousarray(np.ones(self.n_chains) if T is None else
                                  np.broadcast_to(np.asarray(T, dtype=np.float64), (self.n_chains,)))
        acc = np.zeros(self.n_chains, dtype=np.uint8)
        capi.check(capi.lib().mhx_step_injected(
            self._h, La.ctypes.data_as(capi.f64p), per_chain, za.ctypes.data_as(capi.f64p),
            ua.ctypes.data_as(capi.f64p), Ta.ctypes.data_as(capi.f64p),
            acc.ctypes.data_as(capi.u8p)))
        return acc

    # ---- controller -----------------------------------------------------------
    def _opts(self, n, temperature, auto, max_walker_length, l_matrix):
        o = capi.RunOpts()
        capi.lib().mhx_run_opts_default(C.byref(o))
        o.n, o.temperature, o.auto_mode = int(n), float(temperature), int(auto)
        o.max_walker_length = int(max_walker_length or 0)
        if l_matrix is not None:
            self._L_keep = np.ascontiguousarray(l_matrix, dtype=np.float64)
            o.l_matrix = self._L_keep.ctypes.data_as(capi.f64p)
            o.l_matrix_per_chain = 1 if self._L_keep.ndim == 3 else 0
        return o

    def adaptive_begin(self, n=100000, temperature=1e3, auto=1, max_walker_length=0,
                       l_matrix=None):
        o = self._opts(n, temperature, auto, max_walker_length, l_matrix)
        capi.check(capi.lib().mhx_adaptive_begin(self._h, C.byref(o)))

    def adaptive_advance(self, max_iters, count=True):
        n = C.c_int64(-1)
        capi.check(capi.lib().mhx_adaptive_advance(self._h, int(max_iters),
                                                   C.byref(n) if count else None))
        return n.value

    def adaptive_steps_full(self, n=100000, temperature=1e3, auto=1, max_walker_length=0,
                            l_matrix=None):
        o = self._opts(n, temperature, auto, max_walker_length, l_matrix)
        capi.check(capi.lib().mhx_adaptive_steps_full(self._h, C.byref(o)))

    def adaptive_steps(self, n=30000):
        capi.check(capi.lib().mhx_adaptive_steps(self._h, int(n)))

    def many_steps(self, n, L):
        La = np.ascontiguousarray(L, dtype=np.float64)
        capi.check(capi.lib().mhx_many_steps(self._h, int(n), La.ctypes.data_as(capi.f64p),
                                             1 if La.ndim == 3 else 0))

    def take_step(self, L, temperature=1.0):
        """(walker-take-step w :l-matrix L :temperature T) for every chain, device randomness"""
        La = np.ascontiguousarray(L, dtype=np.float64)
        capi.check(capi.lib().mhx_take_step(self._h, La.ctypes.data_as(capi.f64p),
                                            1 if La.ndim == 3 else 0, float(temperature)))

    def request_stop(self):
        capi.check(capi.lib().mhx_request_stop(self._h))

    def comm_init_rank(self, unique_id, rank, n_ranks):
        """join the RCCL communicator of a one-process-per-GPU job (collective); unique_id: the
        128 bytes rank 0 got from comm_unique_id()"""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        capi.check(capi.lib().mhx_comm_init_rank(self._h, buf, int(rank), int(n_ranks)))

    def set_allreduce(self, fn, device_buffer=False):
        """fn(ptr, n, device_buffer) -> 0; sums n doubles at ptr over all ranks in place."""
        if fn is None:
            self._cb = None
            capi.check(capi.lib().mhx_set_allreduce(self._h, capi.ALLREDUCE_FN(), None, 0))
            return

        def tramp(ctx, buf, n, dev):
            try:
                return int(fn(buf, n, dev) or 0)
            except Exception:  # never let an exception cross the C ABI
                import traceback
                traceback.print_exc()
                return 1
        self._cb = capi.ALLREDUCE_FN(tramp)
        capi.check(capi.lib().mhx_set_allreduce(self._h, self._cb, None, int(device_buffer)))

    # ---- read-back (walker-get) ----------------------------------------------------
    def state(self):
        C_, d = self.n_chains, self.d
        th, bt = np.zeros((C_, d)), np.zeros((C_, d))
        lp, bl = np.zeros(C_), np.zeros(C_)
        ln, ag = np.zeros(C_, dtype=np.int64), np.zeros(C_, dtype=np.int64)
        capi.check(capi.lib().mhx_get_state(
            self._h, th.ctypes.data_as(capi.f64p), lp.ctypes.data_as(capi.f64p),
            bt.ctypes.data_as(capi.f64p), bl.ctypes.data_as(capi.f64p),
            ln.ctypes.data_as(capi.i64p), ag.ctypes.data_as(capi.i64p)))
        return dict(theta=th, logpost=lp, best_theta=bt, best_logpost=bl, length=ln, age=ag)

    def chain(self, c):
        """one chain's state (mhx_get_chain): what the accessors of one walker read"""
        d = self.d
        th, bt = np.zeros(d), np.zeros(d)
        lp, bl = C.c_double(0), C.c_double(0)
        ln, ag = C.c_int64(0), C.c_int64(0)
        capi.check(capi.lib().mhx_get_chain(
            self._h, int(c), th.ctypes.data_as(capi.f64p), C.byref(lp),
            bt.ctypes.data_as(capi.f64p), C.byref(bl), C.byref(ln), C.byref(ag)))
        return dict(theta=th, logpost=lp.value, best_theta=bt, best_logpost=bl.value,
                    length=ln.value, age=ag.value)

    def chain_status(self):
        st = np.zeros(self.n_chains, dtype=np.int32)
        li = np.zeros(self.n_chains, dtype=np.int64)
        capi.check(capi.lib().mhx_get_chain_status(self._h, st.ctypes.data_as(capi.i32p),
                                                   li.ctypes.data_as(capi.i64p)))
        return st, li

    def lmatrix(self):
        L = np.zeros((self.n_chains, self.d, self.d))
        capi.check(capi.lib().mhx_get_lmatrix(self._h, L.ctypes.data_as(capi.f64p)))
        return L

    def temperature(self):
        T = np.zeros(self.n_chains)
        capi.check(capi.lib().mhx_get_temperature(self._h, T.ctypes.data_as(capi.f64p)))
        return T

    def acceptance(self, take):
        out = np.zeros(self.n_chains)
        capi.check(capi.lib().mhx_get_acceptance(self._h, int(take), out.ctypes.data_as(capi.f64p)))
        return out

    def trace(self, chain, take):
        take = int(take)
        prob = np.zeros(max(take, 1))
        th = np.zeros((max(take, 1), self.d))
        n = C.c_int(0)
        capi.check(capi.lib().mhx_get_trace(self._h, int(chain), take,
                                            prob.ctypes.data_as(capi.f64p),
                                            th.ctypes.data_as(capi.f64p), C.byref(n)))
        return prob[:n.value], th[:n.value]

    def proposal_factor(self, chain, take):
        L = np.zeros((self.d, self.d))
        st, nf = C.c_int(0), C.c_int(0)
        capi.check(capi.lib().mhx_get_proposal_factor(self._h, int(chain), int(take),
                                                      L.ctypes.data_as(capi.f64p), C.byref(st),
                                                      C.byref(nf)))
        return st.value, L, nf.value

    def set_history(self, chain, prob, theta):
        """restore a walk, newest first (walker-load)"""
        pa, pp = capi.as_f64(prob)
        ta, tp = capi.as_f64(np.asarray(theta, dtype=np.float64).reshape(len(pa), self.d))
        capi.check(capi.lib().mhx_set_history(self._h, int(chain), pp, tp, len(pa)))

    MODIFY = {"burn-walks": 0, "keep-walks": 1, "reset": 2, "reset-to-most-likely": 3}

    def modify(self, action, n=0):
        """walker-modify's :burn-walks / :keep-walks / :reset / :reset-to-most-likely (M:566-578)"""
        capi.check(capi.lib().mhx_walker_modify(self._h, self.MODIFY[action], int(n)))

    def pooled(self):
        d = self.d
        stats, L = np.zeros(1 + d + d * d), np.zeros((d, d))
        valid, n = C.c_int32(0), C.c_uint64(0)
        capi.check(capi.lib().mhx_get_pooled(self._h, stats.ctypes.data_as(capi.f64p),
                                             L.ctypes.data_as(capi.f64p), C.byref(valid),
                                             C.byref(n)))
        return dict(stats=stats, L=L, valid=bool(valid.value), refreshes=n.value)

    def counters(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        capi.check(capi.lib().mhx_get_counters(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kernel_name(self):
        """which kernels serve the problem: 'w16/gauss22_normal', 'w8/rtc[expr:normal]', ..."""
        r = capi.lib().mhx_kernel_name(self._h)
        if r is None:
            raise capi.MhxError(capi.ESTATE, capi.lib().mhx_last_error().decode("utf-8", "replace"))
        return r.decode()

    def kernel_timing(self, reset=False):
        avg, tot, n = C.c_double(0), C.c_double(0), C.c_uint64(0)
        capi.check(capi.lib().mhx_kernel_timing(self._h, int(reset), C.byref(avg), C.byref(n),
                                                C.byref(tot)))
        return dict(avg_ms=avg.value, launches=n.value, total_ms=tot.value)


def comm_unique_id():
    """128 bytes identifying a new RCCL communicator (rank 0 calls this and passes them on)"""
    buf = (C.c_uint8 * 128)()
    capi.check(capi.lib().mhx_comm_get_unique_id(buf))
    return bytes(buf)


def partition(n_chains, n_parts, part):
    """(first, count) of part's contiguous chain range (mhx_group_partition; no device needed)"""
    a, b = C.c_int64(0), C.c_int64(0)
    capi.check(capi.lib().mhx_group_partition(int(n_chains), int(n_parts), int(part),
                                              C.byref(a), C.byref(b)))
    return a.value, b.value


class _GroupEngine(Engine):
    """engine i of a group, seen through the per-engine entry points (owned by the group)"""

    def __init__(self, handle, n_chains, d, K):  # noqa: super().__init__ creates; this borrows
        self._h = C.c_void_p(handle)
        self._cb = None
        self.n_chains, self.d, self.K = int(n_chains), int(d), int(K)

    def close(self):
        self._h = C.c_void_p()


class Group:
    """ONE host process, several GPUs (include/mhx.h, mhx_group_*): n_chains walkers in all,
    contiguous global id ranges per device, launches enqueued on every device before any is
    waited for, the pooled tick's all-reduce through RCCL."""

    def __init__(self, n_chains, n_params, n_functions=1, devices=(0,), seed=0, chain_offset=0,
                 adapt_mode=capi.ADAPT_FAITHFUL, history_capacity=0, poisson_logfact_double=False):
        cfg = capi.Config()
        cfg.n_chains, cfg.n_params, cfg.n_functions = int(n_chains), int(n_params), int(n_functions)
        cfg.adapt_mode, cfg.seed = int(adapt_mode), int(seed)
        cfg.chain_offset, cfg.history_capacity = int(chain_offset), int(history_capacity)
        cfg.poisson_logfact_double = int(bool(poisson_logfact_double))
        self.n_chains, self.d, self.K = int(n_chains), int(n_params), int(n_functions)
        dev, devp = capi.as_i32(list(devices))
        self._h = C.c_void_p()
        capi.check(capi.lib().mhx_group_create(C.byref(cfg), devp, len(dev), C.byref(self._h)))
        self.ranges = []
        self.engines = []
        for i in range(capi.lib().mhx_group_size(self._h)):
            a, b = C.c_int64(0), C.c_int64(0)
            capi.check(capi.lib().mhx_group_chain_range(self._h, i, C.byref(a), C.byref(b)))
            self.ranges.append((a.value, b.value))
            self.engines.append(_GroupEngine(capi.lib().mhx_group_engine(self._h, i), b.value,
                                             self.d, self.K))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            for e in self.engines:
                e.close()
            capi.lib().mhx_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # problem definition: the same calls as Engine, applied to every device
    def set_function(self, k, model, shape=(), param_index=()):
        sh, shp = capi.as_i32(list(shape) if len(shape) else [0])
        ix, ixp = capi.as_i32(list(param_index))
        capi.check(capi.lib().mhx_group_set_function(self._h, k, model, shp, len(shape), ixp, len(ix)))

    def set_dataset(self, k, x, y, sigma=None, likelihood=capi.LIK_NORMAL):
        xa, xp = capi.as_f64(x)
        ya, yp = capi.as_f64(y)
        sp = None
        if sigma is not None:
            sa, sp = capi.as_f64(np.broadcast_to(np.asarray(sigma, dtype=np.float64), xa.shape))
        capi.check(capi.lib().mhx_group_set_dataset(self._h, k, xp, yp, sp, xa.size, likelihood))

    def set_bounds(self, k, idx, lo, hi):
        ix, ixp = capi.as_i32(list(idx))
        la, lp = capi.as_f64(lo)
        ha, hp = capi.as_f64(hi)
        capi.check(capi.lib().mhx_group_set_bounds(self._h, k, ixp, lp, hp, len(ix)))

    def init_chains(self, theta0):
        th = np.ascontiguousarray(theta0, dtype=np.float64)
        bc = 1 if th.ndim == 1 else 0
        capi.check(capi.lib().mhx_group_init_chains(self._h, th.ctypes.data_as(capi.f64p), bc))

    def adaptive_begin(self, n=100000, temperature=1e3, auto=1, max_walker_length=0, l_matrix=None):
        o = Engine._opts(self, n, temperature, auto, max_walker_length, l_matrix)
        capi.check(capi.lib().mhx_group_adaptive_begin(self._h, C.byref(o)))

    def adaptive_advance(self, max_iters, count=True):
        n = C.c_int64(-1)
        capi.check(capi.lib().mhx_group_adaptive_advance(self._h, int(max_iters),
                                                         C.byref(n) if count else None))
        return n.value

    def adaptive_steps_full(self, n=100000, temperature=1e3, auto=1, max_walker_length=0,
                            l_matrix=None):
        o = Engine._opts(self, n, temperature, auto, max_walker_length, l_matrix)
        capi.check(capi.lib().mhx_group_adaptive_steps_full(self._h, C.byref(o)))

    def request_stop(self):
        capi.check(capi.lib().mhx_group_request_stop(self._h))

    def state(self):
        C_, d = self.n_chains, self.d
        th, bt = np.zeros((C_, d)), np.zeros((C_, d))
        lp, bl = np.zeros(C_), np.zeros(C_)
        ln, ag = np.zeros(C_, dtype=np.int64), np.zeros(C_, dtype=np.int64)
        capi.check(capi.lib().mhx_group_get_state(
            self._h, th.ctypes.data_as(capi.f64p), lp.ctypes.data_as(capi.f64p),
            bt.ctypes.data_as(capi.f64p), bl.ctypes.data_as(capi.f64p),
            ln.ctypes.data_as(capi.i64p), ag.ctypes.data_as(capi.i64p)))
        return dict(theta=th, logpost=lp, best_theta=bt, best_logpost=bl, length=ln, age=ag)

    def counters(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        capi.check(capi.lib().mhx_group_get_counters(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value
