"""ctypes binding of oracle/libmhx_oracle.so (the CPU restatement of the reference path).

Test infrastructure: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg only.  The product package never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
# MHX_ORACLE_LIBRARY: another build of the same sources (oracle/Makefile: asan)
_LIB_PATH = os.environ.get("MHX_ORACLE_LIBRARY") or os.path.join(ORACLE_DIR, "libmhx_oracle.so")

f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)

L_OK, L_CAUGHT, L_INVALID, L_EMPTY = 0, 1, 2, 3
RUNNING, DONE, FP_TRAP, STOPPED = 0, 1, 2, 3


class RunOpts(C.Structure):
    _fields_ = [("n", C.c_int64), ("temperature", C.c_double), ("auto_mode", C.c_int32),
                ("max_walker_length", C.c_int64), ("l_matrix", f64p)]


def build(force=False):
    src = os.path.join(ORACLE_DIR, "mhx_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    d, i, vp = C.c_double, C.c_int, C.c_void_p
    sig = {
        "orc_log_normal": (d, [d, d, d]),
        "orc_log_factorial": (d, [d, i]),
        "orc_log_poisson": (d, [d, d, i]),
        "orc_bound_penalty": (d, [d, d, d]),
        "orc_model_eval": (d, [i, i32p, f64p, i, d]),
        "orc_lplist_covariance": (i, [f64p, i, i, f64p]),
        "orc_cholesky": (i, [f64p, i, f64p]),
        "orc_covariant_sample": (None, [f64p, f64p, f64p, i, f64p]),
        "orc_philox4x32_10": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32)]),
        "orc_det_log": (d, [d]),
        "orc_det_cos2pi": (d, [d]),
        "orc_rng_normal": (d, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]),
        "orc_rng_uniform": (d, [C.c_uint64, C.c_uint64, C.c_uint64]),
        "orc_temperature_schedule": (C.c_int64, [C.c_int64, i, d, f64p, C.c_int64]),
        "orc_problem_create": (vp, [i, i]),
        "orc_problem_destroy": (None, [vp]),
        "orc_problem_set_function": (i, [vp, i, i, i32p, i, i32p, i]),
        "orc_problem_set_dataset": (i, [vp, i, f64p, f64p, f64p, C.c_size_t, i]),
        "orc_problem_set_bounds": (i, [vp, i, i32p, f64p, f64p, i]),
        "orc_problem_set_logfact_double": (None, [vp, i]),
        "orc_logpost": (d, [vp, f64p, f64p]),
        "orc_logpost_abs_terms": (d, [vp, f64p]),
        "orc_walker_create": (vp, [vp, f64p]),
        "orc_walker_create2": (vp, [vp, f64p, i]),
        "orc_logpost_mirror": (d, [vp, f64p, f64p]),
        "orc_mirror_set_recurrence": (None, [i]),
        "orc_mirror_set_window_grids": (None, [i]),
        "orc_walker_destroy": (None, [vp]),
        "orc_walker_take_step_injected": (i, [vp, f64p, f64p, d, d]),
        "orc_walker_modify": (i, [vp, i, C.c_int64]),
        "orc_walker_length": (C.c_int64, [vp]),
        "orc_walker_age": (C.c_int64, [vp]),
        "orc_walker_last": (None, [vp, f64p, f64p]),
        "orc_walker_best": (None, [vp, f64p, f64p]),
        "orc_walker_trace": (i, [vp, i, f64p, f64p]),
        "orc_walker_acceptance": (None, [vp, i, i64p, i64p]),
        "orc_walker_forward_count": (i, [vp, i]),
        "orc_walker_l_matrix": (i, [vp, i, f64p, C.POINTER(C.c_int)]),
        "orc_walker_adaptive_begin": (i, [vp, C.POINTER(RunOpts), C.c_uint64, C.c_uint64]),
        "orc_walker_adaptive_advance": (i, [vp, C.c_int64]),
        "orc_walker_status": (i, [vp]),
        "orc_walker_loop_index": (C.c_int64, [vp]),
        "orc_walker_temperature": (d, [vp]),
        "orc_walker_current_l": (None, [vp, f64p]),
        "orc_walker_request_stop": (None, [vp]),
        "orc_walker_many_steps": (i, [vp, C.c_int64, f64p, C.c_uint64, C.c_uint64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(f64p)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(i32p)


class Problem:
    """functions + datasets + bounds priors: the inputs of walker-create (M:1132-1163)."""

    def __init__(self, d, K=1):
        self.d, self.K = d, K
        self.h = lib().orc_problem_create(d, K)
        if not self.h:
            raise ValueError("orc_problem_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_problem_destroy(self.h)
            self.h = None

    def set_function(self, k, model, shape=(), idx=None):
        sh, shp = _i32(list(shape) if len(shape) else [0])
        ix, ixp = _i32(idx)
        rc = lib().orc_problem_set_function(self.h, k, model, shp, len(shape), ixp, len(ix))
        assert rc == 0, "orc_problem_set_function"

    def set_dataset(self, k, x, y, sigma=None, lik=0):
        xa, xp = _f64(x)
        ya, yp = _f64(y)
        if sigma is None:
            sp = None
        else:
            sa, sp = _f64(np.broadcast_to(np.asarray(sigma, dtype=np.float64), xa.shape))
        rc = lib().orc_problem_set_dataset(self.h, k, xp, yp, sp, xa.size, lik)
        assert rc == 0

    def set_bounds(self, k, idx, lo, hi):
        ix, ixp = _i32(idx)
        la, lp = _f64(lo)
        ha, hp = _f64(hi)
        rc = lib().orc_problem_set_bounds(self.h, k, ixp, lp, hp, len(ix))
        assert rc == 0

    def set_logfact_double(self, flag):
        lib().orc_problem_set_logfact_double(self.h, int(flag))

    def logpost(self, theta, parts=False):
        th, tp = _f64(theta)
        pr = np.zeros(2)
        v = lib().orc_logpost(self.h, tp, pr.ctypes.data_as(f64p))
        return (v, pr) if parts else v

    def logpost_mirror(self, theta, parts=False):
        """the GPU kernel's arithmetic on the CPU (NaN when the problem is not mirrored)"""
        th, tp = _f64(theta)
        pr = np.zeros(2)
        v = lib().orc_logpost_mirror(self.h, tp, pr.ctypes.data_as(f64p))
        return (v, pr) if parts else v

    def logpost_many(self, thetas):
        thetas = np.ascontiguousarray(thetas, dtype=np.float64).reshape(-1, self.d)
        return np.array([self.logpost(t) for t in thetas])

    def abs_terms(self, theta):
        th, tp = _f64(theta)
        return lib().orc_logpost_abs_terms(self.h, tp)


class Walker:
    def __init__(self, problem, theta0, mirror=False):
        self.p = problem
        self.d = problem.d
        th, tp = _f64(theta0)
        self.h = lib().orc_walker_create2(problem.h, tp, int(bool(mirror)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_walker_destroy(self.h)
            self.h = None

    def take_step_injected(self, L, z, u, T=1.0):
        La, Lp = _f64(L)
        za, zp = _f64(z)
        return lib().orc_walker_take_step_injected(self.h, Lp, zp, float(u), float(T))

    MODIFY = {"burn-walks": 0, "keep-walks": 1, "reset": 2, "reset-to-most-likely": 3}

    def modify(self, action, n=0):
        return lib().orc_walker_modify(self.h, self.MODIFY[action], int(n))

    @property
    def length(self):
        return lib().orc_walker_length(self.h)

    @property
    def age(self):
        return lib().orc_walker_age(self.h)

    def last(self):
        th = np.zeros(self.d)
        pr = C.c_double()
        lib().orc_walker_last(self.h, th.ctypes.data_as(f64p), C.byref(pr))
        return th, pr.value

    def best(self):
        th = np.zeros(self.d)
        pr = C.c_double()
        lib().orc_walker_best(self.h, th.ctypes.data_as(f64p), C.byref(pr))
        return th, pr.value

    def trace(self, take):
        n = min(int(take), int(self.length))
        prob = np.zeros(n)
        th = np.zeros((n, self.d))
        got = lib().orc_walker_trace(self.h, n, prob.ctypes.data_as(f64p),
                                     th.ctypes.data_as(f64p))
        return prob[:got], th[:got]

    def acceptance(self, take):
        num, den = C.c_int64(), C.c_int64()
        lib().orc_walker_acceptance(self.h, take, C.byref(num), C.byref(den))
        return num.value, den.value

    def forward_count(self, take):
        return lib().orc_walker_forward_count(self.h, take)

    def l_matrix(self, take):
        L = np.zeros((self.d, self.d))
        nf = C.c_int()
        st = lib().orc_walker_l_matrix(self.h, take, L.ctypes.data_as(f64p), C.byref(nf))
        return st, L, nf.value

    def adaptive_begin(self, n, temperature=1e3, auto=0, max_walker_length=0, l_matrix=None,
                       seed=0, chain_id=0):
        o = RunOpts()
        o.n, o.temperature, o.auto_mode = int(n), float(temperature), int(auto)
        o.max_walker_length = int(max_walker_length)
        if l_matrix is not None:
            self._L, o.l_matrix = _f64(l_matrix)
        return lib().orc_walker_adaptive_begin(self.h, C.byref(o), seed, chain_id)

    def adaptive_advance(self, iters):
        return lib().orc_walker_adaptive_advance(self.h, int(iters))

    def adaptive_steps(self, n=30000, seed=0, chain_id=0):
        """(walker-adaptive-steps w n), M:946-947"""
        self.adaptive_begin(n, 10.0, 1, seed=seed, chain_id=chain_id)
        return self.adaptive_advance(1 << 62)

    def many_steps(self, n, L, seed=0, chain_id=0):
        La, Lp = _f64(L)
        return lib().orc_walker_many_steps(self.h, int(n), Lp, seed, chain_id)

    @property
    def status(self):
        return lib().orc_walker_status(self.h)

    @property
    def loop_index(self):
        return lib().orc_walker_loop_index(self.h)

    @property
    def temperature(self):
        return lib().orc_walker_temperature(self.h)

    def current_l(self):
        L = np.zeros((self.d, self.d))
        lib().orc_walker_current_l(self.h, L.ctypes.data_as(f64p))
        return L

    def request_stop(self):
        lib().orc_walker_request_stop(self.h)


def mirror_set_recurrence(on):
    """mirror mode: restate the kernel with (True, default) or without (MHX_NO_RECURRENCE=1) the
    uniform-grid recurrence of the Gaussian peaks"""
    lib().orc_mirror_set_recurrence(int(bool(on)))


def mirror_set_window_grids(on):
    """mirror mode: per-window grids (default) or MHX_NO_WINDOW_GRIDS=1's rule (one grid or none)"""
    lib().orc_mirror_set_window_grids(int(bool(on)))


def temperature_schedule(n, d, temperature):
    sts = 10 * max(50, d)
    ts = max(int(n), 10 * sts)
    out = np.zeros(ts)
    got = lib().orc_temperature_schedule(int(n), d, float(temperature),
                                         out.ctypes.data_as(f64p), ts)
    assert got == ts
    return out


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)
