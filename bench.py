#!/usr/bin/env python3
"""bench.py -- chain-steps/sec of the walker-adaptive-steps path on MI355X.

One "step" = one iteration of the reference's do loop (mcmc-fitting.lisp:902-942) for EVERY
chain of the batch: propose, evaluate the full log-posterior over all data points, accept /
reject, history push, controller bookkeeping (incl. the 200-step proposal adaptation when it
falls inside the timed region).  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5|poly7|c1|c2expr|c2written|g23]

N > 1 runs either way, chains sharded by contiguous global id ranges (weak scaling: --chains per
GPU), datasets replicated, no data-path collective in the reference's per-walker adaptation mode;
the pooled-covariance mode (default for N > 1) adds ONE RCCL all-reduce of 1+d+d^2 doubles per 200
iterations, issued by libmhx itself on the engines' streams:
  * `python bench.py --gpus N` by itself: ONE host process drives the N GPUs through mhx_group_*
    (ncclCommInitAll, the tick's all-reduces inside ncclGroupStart/End) - the reference's own host
    model, a list of walkers mapped in one image (mcmc-fitting.lisp:1029-1033);
  * under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: one rank per
    GPU, each with its engine in a communicator of libmhx's own (mhx_comm_init_rank); torch carries
    the 128-byte id and the timing barriers.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# fp64 vector peak: 256 CUs x 4 SIMDs, one wave64 fp64 instruction per SIMD every 4 cycles,
# 2.4 GHz max clock, an fma = 2 flop x 64 lanes -> 78.6 TFLOP/s (MI355X_MICROARCH.md: half the
# 157.3 TFLOP/s fp32 vector peak).  The roofline is stated in what is actually counted - VALU
# wave-instructions issued per second against 1024 x 2.4e9 / 4 = 614.4 G/s - not in flop: only
# some of the instructions are fmas.
N_SIMD, CLOCK_HZ, CYCLES_PER_F64_INSTR = 1024, 2.4e9, 4
FP64_VALU_PEAK_TFLOPS = N_SIMD * CLOCK_HZ / CYCLES_PER_F64_INSTR * 128 / 1e12
FP64_ISSUE_PEAK_G = N_SIMD * CLOCK_HZ / CYCLES_PER_F64_INSTR / 1e9  # G wave-instructions / s
ROUND_TAG = "r04"  # profiles/<tag>_<workload>_s<steps>_w<warmup>_summary.json


def synth_workload(name, rng_key=0x5EED0001):
    """Synthetic inputs of BASELINE.json's configs."""
    import problems as pb
    rng = np.random.Generator(np.random.Philox(key=rng_key))
    if name == "c5":
        # BASELINE config 5's share of one GPU: config 2's problem, 524288 / 8 chains, pooled
        # adaptive covariance (main() switches the adaptation mode)
        s, _, b_pt, _ = synth_workload("c2", rng_key)
        return s, 65536, b_pt, ("65536 chains (config 5's share of one of 8 GPUs) x config 2's problem, "
                                "pooled adaptive covariance")
    if name in ("c2", "poly7"):
        n = 100000
        x = np.linspace(0.0, 1.0, n)
        sig = rng.uniform(0.05, 0.15, n)
        if name == "c2":
            th = np.array([0.5, 0.3, 1.0, 0.3, 0.05, 0.7, 0.7, 0.08])
            y = pb.model_eval_np(pb.GAUSS, (2, 2), th, x) + sig * rng.standard_normal(n)
            s = pb.Spec(8)
            lo, hi = np.minimum(th * 0.5, th * 1.5), np.maximum(th * 0.5, th * 1.5)
            s.add(pb.GAUSS, (2, 2), range(8), x, y, sig, pb.NORMAL, (list(range(8)), lo, hi))
            desc = "4096 chains x 8-param two-Gaussian-peak + linear bg, weighted normal log-lik, 1e5 points, fp64"
        else:
            th = np.array([0.5, 0.3, -0.2, 0.1, 0.05, -0.03, 0.02, 0.01])
            y = pb.model_eval_np(pb.POLY, (), th, x) + sig * rng.standard_normal(n)
            s = pb.Spec(8)
            s.add(pb.POLY, (), range(8), x, y, sig, pb.NORMAL,
                  (list(range(8)), th - 1.0, th + 1.0))
            desc = "4096 chains x degree-7 polynomial (cheapest 8-param model), weighted normal, 1e5 points"
        s.theta_star = th
        return s, 4096, 24, desc
    if name == "c3":
        n, npk = 1000000, 5
        th = [20.0]
        for k in range(npk):
            th += [float(rng.uniform(40, 150)), (k + 0.5) / npk, float(rng.uniform(0.02, 0.05))]
        th = np.array(th)
        x = np.linspace(0.0, 1.0, n)
        y = rng.poisson(pb.model_eval_np(pb.GAUSS, (1, npk), th, x)).astype(float)
        s = pb.Spec(16)
        s.add(pb.GAUSS, (1, npk), range(16), x, y, None, pb.POISSON,
              (list(range(16)), th * 0.5, th * 1.5))
        s.theta_star = th
        return s, 65536, 16, "65536 chains x 16-param 5-peak Poisson log-lik, 1e6 points"
    if name == "g23":
        # a shape with no ahead-of-time specialisation: linear background + 3 Gaussian peaks,
        # 11 parameters -> the engine compiles it at run time (hiprtc) like the ones above
        n = 100000
        x = np.linspace(0.0, 1.0, n)
        sig = rng.uniform(0.05, 0.15, n)
        th = np.array([0.5, 0.3, 1.0, 0.25, 0.04, 0.7, 0.5, 0.06, 0.9, 0.8, 0.05])
        y = pb.model_eval_np(pb.GAUSS, (2, 3), th, x) + sig * rng.standard_normal(n)
        s = pb.Spec(11)
        lo, hi = np.minimum(th * 0.5, th * 1.5), np.maximum(th * 0.5, th * 1.5)
        s.add(pb.GAUSS, (2, 3), range(11), x, y, sig, pb.NORMAL, (list(range(11)), lo, hi))
        s.theta_star = th
        return s, 4096, 24, "4096 chains x 11-param three-Gaussian-peak + linear bg, weighted normal, 1e5 points"
    if name == "c4":
        s = pb.global_fit(n_each=12500, n_sets=8, seed=3)
        return s, 4096, 24, "4096 chains, 8 datasets x 8 fns sharing 32 params, 12500 points each"
    if name == "c1":
        s = pb.lorder()
        return s, 1, 24, "test.lisp shape: 1 chain, 334 points, 6 params"
    raise SystemExit("unknown workload %r" % name)


def b_alg(spec, b_pt):
    """algorithmic bytes per chain-step, SURVEY 8d / BASELINE.md section 3"""
    d = spec.d
    return sum(len(x) for (x, _, _, _) in spec.data) * b_pt + 8 * d * d + 16 * d + 16


def cpu_baseline(spec, theta0s, budget_s, n_adapt, threads, l0=None):
    """the oracle (CPU restatement of the reference, faithful serial sums, libm) on `threads`
    host cores, same workload, bounded sample: thread i walks chain i (the reference's own way
    of running many walkers is a list of independent ones, mcmc-fitting.lisp:1029-1033); the
    ctypes calls release the GIL, so the threads run in parallel.  Returns (chain-steps/s summed
    over the threads, steps, seconds)."""
    import threading
    import oraclelib as orc
    op = spec.oracle(orc)
    done = [0] * threads
    t_end = [0.0] * threads
    start = threading.Barrier(threads + 1)

    def walk(i):
        w = orc.Walker(op, theta0s[i % len(theta0s)])
        w.adaptive_begin(n_adapt, 10.0, 1, l_matrix=l0, seed=0x5EED0003, chain_id=i)
        start.wait()
        t0 = time.perf_counter()
        chunk = 1
        while True:
            w.adaptive_advance(chunk)
            done[i] += chunk
            if time.perf_counter() - t0 >= budget_s or w.status != orc.RUNNING:
                break
            chunk = max(1, min(64, chunk * 2))
        t_end[i] = time.perf_counter()

    ts = [threading.Thread(target=walk, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in ts:
        t.join()
    el = max(t_end) - t0
    return sum(done) / el, sum(done), el


def load_profile(workload, steps, warmup):
    """the committed rocprofv3 summary of this command (tools/run_profile.sh), or the nearest
    one of the same workload; (summary, file name, exact match?)"""
    import glob
    exact = os.path.join(ROOT, "profiles", "%s_%s_s%d_w%d_summary.json" % (ROUND_TAG, workload, steps, warmup))
    cands = [exact] if os.path.exists(exact) else sorted(
        glob.glob(os.path.join(ROOT, "profiles", "%s_%s_s*_summary.json" % (ROUND_TAG, workload))))
    for f in cands:
        try:
            return json.load(open(f)), os.path.basename(f), f == exact
        except Exception:  # a malformed summary must not break the benchmark
            continue
    return None, None, False


def effective_cores():
    """(threads to use, what they were derived from): the affinity mask says how many CPUs the
    process may run on, the cgroup's cpu.max how much CPU time it gets - a GPU box hands a
    one-GPU job a share of a 256-thread host.  MHX_BENCH_CPU_THREADS overrides."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    info = {"affinity": aff, "cgroup_cpu_max": None}
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                info["cgroup_cpu_max"] = " ".join(txt)
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                info["cgroup_cpu_max"] = "%d %d" % (q, per)
                if q > 0:
                    quota = q / per
            break
        except (OSError, ValueError, IndexError):
            continue
    n = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    env = os.environ.get("MHX_BENCH_CPU_THREADS")
    if env:
        n = max(1, int(env))
    info["threads"] = n
    return n, info


def kernel_label(name):
    """which stepping kernel serves an engine of that mhx_kernel_name"""
    if "persistent tsplit" in name:
        return "k_persist_ts @ " + name
    if "persistent split" in name:
        return "k_persist @ " + name
    if "split x" in name:
        return "k_split_step + sweep @ " + name
    return "k_adaptive @ " + name


class Fleet:
    """the engines of one benchmark process behind one set of calls: an Engine (one GPU) or a
    Group (mhx_group_*: one host process, several GPUs)"""

    def __init__(self, mhx, spec, chains_per_gpu, devices, pooled, seed, chain_offset=0):
        self.mhx, self.devices, self.chains = mhx, list(devices), chains_per_gpu * len(devices)
        mode = mhx.capi.ADAPT_POOLED if pooled else mhx.capi.ADAPT_FAITHFUL
        if len(self.devices) == 1:
            self.obj = spec.engine(mhx, chains_per_gpu, device=self.devices[0], seed=seed,
                                   chain_offset=chain_offset, adapt_mode=mode)
            self.engines = [self.obj]
        else:
            self.obj = mhx.Group(self.chains, spec.d, spec.K, devices=self.devices, seed=seed,
                                 chain_offset=chain_offset, adapt_mode=mode)
            spec.apply(self.obj)
            self.engines = self.obj.engines

    def start(self, th0, n_adapt, l0):
        self.obj.init_chains(th0)
        self.obj.adaptive_begin(n_adapt, 10.0, 1, l_matrix=l0)

    def advance(self, iters):
        self.obj.adaptive_advance(iters, count=False)

    def timing(self, reset=False):
        return [e.kernel_timing(reset=reset) for e in self.engines]

    def steps(self):
        return self.obj.counters()[0]

    def trapped(self):
        return any((e.chain_status()[0] == self.mhx.capi.CHAIN_FP_TRAP).any() for e in self.engines)

    def kernel_name(self):
        return self.engines[0].kernel_name()

    def close(self):
        self.obj.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    # 200 = up to and including the first proposal-adaptation tick of the walk (M:929): the first
    # 200 of a run's 30000 iterations (annealing at T = 10 with the untuned initial L matrix)
    # propose wide, often out-of-bounds steps and run ~12 % slower than everything after them
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (default: the workload's)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--adapt", default="auto", choices=["auto", "faithful", "pooled"],
                    help="auto: the reference's per-walker rule on 1 GPU, pooled covariance "
                         "(one RCCL all-reduce per 200 iterations) on several")
    ap.add_argument("--spin-ms", type=float, default=60.0,
                    help="GPU time spent on a throw-away engine of the same kernel before the walk's "
                         "warm-up launch, so that the timed launch runs at the clock the chip holds "
                         "under this load (0: off)")
    ap.add_argument("--devices", default="",
                    help="one-process runs: the device of each of the --gpus engines, comma separated "
                         "(default 0..N-1).  Naming a device twice rehearses the N-GPU code path on "
                         "fewer GPUs (with MHX_GROUP_FORCE_RCCL=1 and a stand-in librccl: tests/stub_rccl)")
    ap.add_argument("--no-direct", action="store_true",
                    help="skip the second, short measurement with the uniform-grid recurrence off")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = max(args.gpus, 1)
    # one process per GPU only when a launcher started us that way; `--gpus N` by itself is ONE
    # process driving N GPUs through mhx_group_*
    per_rank = world > 1
    if per_rank and world != n_gpus:
        raise SystemExit("--gpus %d under a launcher with WORLD_SIZE=%d" % (n_gpus, world))
    import torch
    dist = None
    # MHX_BENCH_REHEARSE_LAUNCHER=1: the launcher path on a box with fewer GPUs than ranks (the
    # builder's has one): ranks share devices, torch.distributed runs on gloo, and the tick's sum
    # goes through the torch hook on a host buffer (RCCL does not put two ranks on one device).
    # Everything else - rendezvous, chain ranges, barriers, the reductions of the timing, the line
    # rank 0 prints - is the code the driver's N-GPU run executes.
    rehearse = per_rank and os.environ.get("MHX_BENCH_REHEARSE_LAUNCHER", "") not in ("", "0")
    red_dev = "cpu" if rehearse else "cuda"  # where the small tensors of dist.all_reduce live
    if per_rank:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import lisp_mcmc_amd as mhx

    devices = [local_rank] if per_rank else list(range(n_gpus))
    if not per_rank:
        if args.devices:
            devices = [int(v) for v in args.devices.split(",")]
            if len(devices) != n_gpus:
                raise SystemExit("--devices names %d devices for --gpus %d" % (len(devices), n_gpus))
        have = torch.cuda.device_count()
        if have <= max(devices):
            raise SystemExit("device %d asked for but this process sees %d GPU(s)" % (max(devices), have))

    # config 2 with the model given as the TEXT of its Lisp closure: c2expr - what a Lisp host hands
    # over; libmhx recognises the peaks (csrc/mhx_expr.cpp) and runs config 2's kernel; c2written -
    # the same text compiled exactly as written (mhx_set_expr_recognition(0): rounds 1-3's c2expr)
    as_expr = args.workload in ("c2expr", "c2written")
    spec, chains, b_pt, desc = synth_workload("c2" if as_expr else args.workload)
    if args.chains:
        chains = args.chains
    n_adapt = 30000  # (walker-adaptive-steps w) default n, mcmc-fitting.lisp:946
    pooled = args.adapt == "pooled" or (args.adapt == "auto" and (n_gpus > 1 or args.workload == "c5"))
    n_local = chains * len(devices)        # chains this process drives
    first_id = rank * chains if per_rank else 0

    def make(pooled=pooled):
        f = Fleet(mhx, spec, chains, devices, pooled, seed=0x5EED0003, chain_offset=first_id)
        if as_expr:
            keys, cexpr = mhx.sexpr.lambda_to_expr(
                "(lambda (x &key b0 b1 a1 mu1 w1 a2 mu2 w2 &allow-other-keys)"
                " (+ (+ b0 (* b1 x)) (* a1 (exp (- (expt (/ (- x mu1) w1) 2))))"
                "    (* a2 (exp (- (expt (/ (- x mu2) w2) 2))))))")
            for e in f.engines:
                if args.workload == "c2written":
                    e.set_expr_recognition(False)
                e.set_function_expr(0, cexpr, keys, list(range(8)))
        return f

    fleet = make()
    if as_expr:
        desc += (" [model given as its Lisp closure text, compiled at run time exactly as written]"
                 if args.workload == "c2written" else
                 " [model given as its Lisp closure text: recognised below the C ABI, config 2's kernel]")
    collective = None
    if pooled and n_gpus > 1 and not per_rank:
        collective = "ncclCommInitAll + ncclAllReduce in ncclGroupStart/End (one host process, mhx_group_*)"
    if pooled and per_rank:
        # the one exchange step of the path: 1+d+d^2 doubles summed over ranks every 200
        # iterations by libmhx's OWN RCCL communicator (mhx_comm_init_rank): ncclAllReduce on the
        # engine's stream between the statistics kernels and the factorisation, no host
        # synchronisation, no Python in the data path.  torch.distributed only carries the 128-byte
        # communicator id from rank 0 to the others (and the barriers / max of the contract).
        collective = "libmhx RCCL communicator (mhx_comm_init_rank; ncclAllReduce on the engine's stream)"
        e = fleet.engines[0]
        # (every rank takes the same branch: rank 0's failure to make an id travels in the
        # broadcast, a failure to join the communicator through an all-reduced flag)
        uid, why = [None], ""
        if rank == 0:
            try:
                uid = [("failed", "rehearsal: ranks share a device")] if rehearse else [mhx.comm_unique_id()]
            except Exception as ex:
                uid = [("failed", str(ex))]
        dist.broadcast_object_list(uid, src=0)
        ok = isinstance(uid[0], (bytes, bytearray))
        if ok:
            try:
                e.comm_init_rank(uid[0], rank, world)
            except Exception as ex:
                ok, why = False, str(ex)
        else:
            why = uid[0][1]
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:  # librccl not usable by libmhx somewhere: the torch.distributed hook
            from lisp_mcmc_amd import distributed as mdist
            e.set_allreduce(mdist.torch_allreduce_hook(dist), device_buffer=not rehearse)
            collective = "torch.distributed all_reduce hook (libmhx RCCL unavailable: %s)" % (why or "on another rank")
    # per-chain start: theta* (1 + 0.01 N(0,1)), keyed by GLOBAL chain id (rank r of a launcher
    # run and device r of a one-process run get the same chains)
    th0 = np.concatenate([
        spec.theta_star[None, :] * (1.0 + 0.01 * np.random.Generator(
            np.random.Philox(key=0x5EED0002 + (first_id // chains) + i)).standard_normal((chains, spec.d)))
        for i in range(len(devices))])
    # c3: a Poisson rate must stay positive (log-poisson of a negative rate is an error in the
    # reference as well), so the run starts from a small :l-matrix instead of diag(theta) (M:899)
    l0 = np.diag(0.002 * np.abs(spec.theta_star)) if args.workload == "c3" else None

    def sync():
        for dv in sorted(set(devices)):
            torch.cuda.synchronize(dv)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # The warm-up is ONE launch of `warmup` fused iterations and the timed region ONE launch of
    # `steps` iterations per GPU (a pooled run ends its launches on the 200-iteration cadence, so
    # there it is one launch per tick interval).  Only a launch that would run for many seconds is
    # cut: config 3 at its full 65536 chains x 1e6 points takes 0.14 s per iteration, so launches
    # keep chains x points x iterations under 1e12.
    work = chains * float(sum(len(d[0]) for d in spec.data))

    def launches_of(n):
        per = n
        while per > 1 and work * per > 1e12:
            per = max(q for q in range(1, per) if per % q == 0)
        return per, (n // per if per else 0)

    # Clock: a GPU that has idled through the host-side set-up starts its first kernels well under
    # the clock it holds under load (round 2: GRBM_GUI_ACTIVE / duration = 1.95 GHz in a 3.8 ms
    # timed launch that followed a 1 ms warm-up, 2.2-2.3 GHz in a 29 ms one).  A throw-away fleet
    # of the same problem runs the same kernel for --spin-ms first; the walk that is measured -
    # its warm-up iterations and its timed iterations - is untouched by it.
    # (a fixed number of iterations, so that the sequence of dispatches is the same in every
    # profiling pass: about --spin-ms at the rate of the headline workload)
    spin_ms, n_disp = 0.0, 0  # (n_disp: k_adaptive dispatches per GPU so far - tools/profile_summary.py)
    fleet.start(th0, n_adapt, l0)
    spin = None
    variant = os.environ.get("MHX_BENCH_SPIN_VARIANT", "2")  # (A/B of the order below; 2 = default)

    def spin_setup():
        nonlocal spin
        spin = make(pooled=False)  # (no communicator of its own: the kernel is the same)
        spin.start(th0, n_adapt, l0)

    def do_spin():
        nonlocal spin, spin_ms, n_disp
        if spin is None:
            spin_setup()
        spin_iters = max(8, min(2000, int(args.spin_ms / 60.0 * 1.6e11 / work)))
        s_per, s_n = launches_of(spin_iters)
        for _ in range(s_n):
            spin.advance(s_per)
        t = spin.timing(reset=True)
        spin_ms = max(x["total_ms"] for x in t)
        n_disp += t[0]["launches"]
        if variant == "0":
            spin.close()
            spin = None

    w_per, w_n = launches_of(args.warmup) if args.warmup > 0 else (0, 0)

    def do_warmup():
        nonlocal n_disp
        for _ in range(w_n):
            fleet.advance(w_per)
        n_disp += fleet.timing(reset=True)[0]["launches"]

    # The throw-away fleet is set up (allocations, copies, its first steps, a hiprtc compile for
    # c2expr) BEFORE the walk's warm-up iterations and only ADVANCED after them, right in front
    # of the timed region, and it is closed only after the measurement: freeing its buffers (a
    # dozen hipFree) between spin and walk left the GPU idle long enough to lose 3-7 % of the
    # clock again (same box, 3.31 against 3.17 ms).  (Rounds 1-2 had no spin and round 3 put it
    # in front of the warm-up first: --spin-ms 0 gives the like-for-like figure.)
    if variant == "2":
        if args.spin_ms > 0:
            spin_setup()
        do_warmup()
        if args.spin_ms > 0:
            do_spin()
    else:
        if args.spin_ms > 0:
            do_spin()
        do_warmup()
    per_launch, n_launch = launches_of(args.steps)
    steps0 = fleet.steps()
    sync()
    t0 = time.perf_counter()
    for _ in range(n_launch):
        fleet.advance(per_launch)  # one launch (per GPU) = per_launch fused iterations
    sync()
    t1 = time.perf_counter()
    el = t1 - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    kts = fleet.timing()
    timed_disp = (n_disp, n_disp + kts[0]["launches"])  # [first, past-the-last) of the timed region
    n_disp += kts[0]["launches"]
    chain_steps = fleet.steps() - steps0
    if dist is not None:
        t = torch.tensor([float(chain_steps)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_steps = float(t.item())
    else:
        total_steps = float(chain_steps)
    assert not fleet.trapped(), "a chain trapped during the benchmark"
    assert chain_steps == n_local * args.steps, (chain_steps, n_local, args.steps)

    n_points = int(sum(len(d[0]) for d in spec.data))
    bytes_step = b_alg(spec, b_pt)
    # HIP events on each engine's own stream around its launch(es); the slowest GPU counts
    kernel_s = max(k["total_ms"] for k in kts) * 1e-3
    kt = max(kts, key=lambda k: k["total_ms"])
    alg_gbs = chain_steps * bytes_step / kernel_s / 1e9
    build_id = mhx.capi.lib().mhx_build_id().decode()
    # The roof that binds is fp64 VALU issue, not HBM: the chains of a workgroup share every data
    # tile through LDS and the dataset sits in L2, so HBM traffic is a fraction of a per cent of
    # the algorithmic bytes.  achieved = VALU wave-instructions per second, with the instructions
    # per data point MEASURED (SQ_INSTS_VALU of the timed launch of this same command, rocprofv3
    # --pmc, committed under profiles/); peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 fp64
    # instruction = 614.4 G wave-instructions/s per GPU (were every one an fma: 78.6 TFLOP/s).
    prof, prof_name, prof_exact = load_profile(args.workload, args.steps, args.warmup)
    roof = {"bound": "fp64_valu_issue", "achieved": None, "peak": FP64_ISSUE_PEAK_G,
            "unit": "G wave-instr/s", "frac": None, "traffic": None,
            "kernel_ms_per_launch": kt["avg_ms"], "launches": kt["launches"],
            "iterations_per_launch": per_launch if not pooled else None,
            "timed_dispatches": list(timed_disp),
            "algorithmic_bytes_per_chain_step": bytes_step, "algorithmic_gbs": alg_gbs,
            "note": "per GPU; fp64 VALU ISSUE bound (wave64 fp64 instructions issued / s against 1024 SIMDs x "
                    "2.4 GHz / 4 cycles; not a flop rate): 16 chains share each LDS tile and the dataset is "
                    "L2 resident, so HBM (hbm_gbs, hbm_frac) does not bind; algorithmic_gbs = the "
                    "SURVEY 8d bytes per chain-step x chain-steps / kernel time, for reference only"}
    if n_gpus > 1 and not per_rank:
        roof["kernel_ms_per_gpu"] = [k["total_ms"] for k in kts]
    if prof is not None:
        try:
            pm, bs = prof["pmc_timed_launch"], prof["bench_stats"]
            prof_steps = bs["steps"] * bs["config"]["chains_per_gpu"]
            prof_points = prof_steps * float(bs["config"]["n_points"])
            ipp = pm["SQ_INSTS_VALU"] * 64.0 / prof_points  # wave-instructions x 64 lanes / points
            steps_gpu = chains * args.steps                  # chain-steps of ONE GPU's timed region
            instr = ipp * steps_gpu * n_points / 64.0        # its wave-instructions
            prof_build = (bs.get("build") or {}).get("id")
            stale = prof_build != build_id
            roof["instr_per_point"] = ipp
            roof["instr_source"] = ("profiles/%s: SQ_INSTS_VALU of the timed launch%s%s"
                                    % (prof_name, "" if prof_exact else
                                       " (NOT the same --steps/--warmup: the count per point depends "
                                       "on where in the walk the launch sits)",
                                       "" if not stale else
                                       " -- STALE: measured on build %s, this is %s" % (prof_build, build_id)))
            roof["instr_source_stale"] = stale
            roof["achieved"] = instr / kernel_s / 1e9
            roof["frac"] = roof["achieved"] / FP64_ISSUE_PEAK_G
            roof["tflops_if_every_instruction_were_an_fma"] = instr * 128.0 / kernel_s / 1e12
            hbm_per_step = (prof["hbm_read_bytes"] + prof["hbm_write_bytes"]) / prof_steps
            # HBM bytes per launch: FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024
            # of the profiled launch, scaled by chain-steps
            roof["traffic"] = hbm_per_step * per_launch * chains
            roof["hbm_gbs"] = hbm_per_step * steps_gpu / kernel_s / 1e9
            roof["hbm_frac"] = roof["hbm_gbs"] / HBM_PEAK_GBS
        except Exception:  # a malformed summary must not break the benchmark
            pass
    out = {
        "metric": "chain-steps/sec (whole node), 1e5-pt Gaussian log-lik, 8 params"
                  if args.workload == "c2" else "chain-steps/sec (whole node), workload %s" % args.workload,
        "value": total_steps / el,
        "unit": "chain-steps/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": desc, "chains_per_gpu": chains, "n_points": n_points,
                   "n_params": spec.d,
                   "adaptation": ("pooled covariance, all-reduce of %d doubles / 200 iterations" % (1 + spec.d + spec.d ** 2))
                   if pooled else "faithful per-walker (no collective)",
                   "parallelism": ("chains sharded over %d GPU(s), " % n_gpus) +
                                  ("one process per GPU" if per_rank else "one host process") +
                                  ("" if len(set(devices)) == len(devices) else
                                   " [REHEARSAL: %d engines on %d physical device(s)]"
                                   % (len(devices), len(set(devices)))),
                   **({"collective": collective} if collective else {}),
                   "clock_spin": ("%.0f ms of the same kernel on a throw-away engine %s" % (
                       spin_ms, "between the walk's warm-up launch and its timed launch"
                       if variant == "2" else "before the walk's warm-up launch")) if spin_ms else "none",
                   "kernel": kernel_label(fleet.kernel_name())},
        "build": {"id": build_id},
        "roofline": roof,
    }
    # The headline's Gaussians advance by the uniform-grid recurrence (x is a linspace: 3
    # instructions per point and peak instead of 14).  The same walk on the same data with the
    # recurrence switched off - what a dataset that is NOT a grid gets - measured the same way
    # beside it, so that nobody has to guess the factor.
    if (rank == 0 and not per_rank and n_gpus == 1 and not args.no_direct
            and args.workload in ("c2", "c5", "c3", "g23")):
        os.environ["MHX_NO_RECURRENCE"] = "1"  # (read when the dataset is set)
        try:
            direct = make()
        finally:
            os.environ.pop("MHX_NO_RECURRENCE", None)
        direct.start(th0, n_adapt, l0)
        for _ in range(w_n):
            direct.advance(w_per)
        n_disp += direct.timing(reset=True)[0]["launches"]
        d0 = direct.steps()
        sync()
        td0 = time.perf_counter()
        for _ in range(n_launch):
            direct.advance(per_launch)
        sync()
        td = time.perf_counter() - td0
        dk = direct.timing()[0]
        out["value_direct_form"] = (direct.steps() - d0) / td
        out["direct_form"] = {
            "what": "the same walk and data with the uniform-grid recurrence off (MHX_NO_RECURRENCE=1): "
                    "every Gaussian by the table-driven exp, as on data whose x is not a grid",
            "kernel_ms_per_launch": dk["avg_ms"],
            "dispatches": [n_disp, n_disp + dk["launches"]],
            "ratio_to_value": out["value"] / out["value_direct_form"]}
        n_disp += dk["launches"]
        pmd = (prof or {}).get("pmc_direct_launch")
        if isinstance(pmd, dict) and "SQ_INSTS_VALU" in pmd and "instr_per_point" in roof:
            bs = prof["bench_stats"]
            ippd = pmd["SQ_INSTS_VALU"] * 64.0 / (bs["steps"] * bs["config"]["chains_per_gpu"]
                                                  * float(bs["config"]["n_points"]))
            out["direct_form"]["instr_per_point_direct"] = ippd
            out["direct_form"]["frac"] = (ippd * chains * args.steps * n_points / 64.0 * 128.0
                                          / (dk["total_ms"] * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS)
        direct.close()
    # ... and the same walk on an x that is a grid only PIECEWISE (tests/problems.py, piecewise_x:
    # three scans of different steps laid end to end, one jittered window, the first step again -
    # what an instrument file often looks like): since round 4 every 2048-point window on a grid
    # takes the recurrence with its own step and only the junction / jittered windows the direct
    # form; until then the whole dataset took the direct form (value_direct_form).
    if (rank == 0 and not per_rank and n_gpus == 1 and not args.no_direct and args.workload == "c2"):
        import copy
        import problems as pb
        pspec = copy.copy(spec)
        x0, y0, sg0, lk0 = spec.data[0]
        xp = pb.piecewise_x(len(x0), seed=11)
        rngp = np.random.Generator(np.random.Philox(key=0x5EED0004))
        yp = pb.model_eval_np(pb.GAUSS, (2, 2), spec.theta_star, xp) + sg0 * rngp.standard_normal(len(x0))
        pspec.data = [(xp, yp, sg0, lk0)]
        pw = Fleet(mhx, pspec, chains, devices, pooled, seed=0x5EED0003, chain_offset=first_id)
        pw.start(th0, n_adapt, l0)
        for _ in range(w_n):
            pw.advance(w_per)
        pw_warm = pw.timing(reset=True)[0]["launches"]   # (its kernel is compiled at run time:
        p0 = pw.steps()                                   # mhx_user_adaptive, counted on its own)
        sync()
        tp0 = time.perf_counter()
        for _ in range(n_launch):
            pw.advance(per_launch)
        sync()
        tpw = time.perf_counter() - tp0
        pk = pw.timing()[0]
        out["value_piecewise"] = (pw.steps() - p0) / tpw
        out["piecewise"] = {
            "what": "the same walk on an x that is three scans of different steps laid end to end, one "
                    "jittered window, the first step again: the recurrence window by window (FnDesc::tgh), "
                    "the direct form only in the junction and jittered windows",
            "kernel_ms_per_launch": pk["avg_ms"],
            "kernel": "mhx_user_adaptive @ " + pw.kernel_name(),
            "dispatches": [pw_warm, pw_warm + pk["launches"]],   # among mhx_user_adaptive's
            "ratio_to_value": out["value"] / out["value_piecewise"]}
        pw.close()
    # ... and the same walk with EXACT EARLY REJECTION (MHX_EARLY_REJECT=1; csrc/mhx_kernels.hpp,
    # sweep()): a sweep ends where the growing sum of squares has already lost the accept test.  The
    # chains are the same chains bit for bit (tests/test_gpu_early_reject.py); what changes is the
    # time of iterations that reject - all of a walk's first dozens (T = 10, diag(theta) steps),
    # hardly any of a settled walk's.  Opt-in and reported BESIDE the headline, whose sweeps are
    # complete: this figure counts rejections, not sweeps.
    if (rank == 0 and not per_rank and n_gpus == 1 and not args.no_direct and args.workload == "c2"):
        os.environ["MHX_EARLY_REJECT"] = "1"
        try:
            er = make()
            er_name = er.kernel_name()  # (finalised - compiled - under the setting)
        finally:
            os.environ.pop("MHX_EARLY_REJECT", None)
        er.start(th0, n_adapt, l0)
        for _ in range(w_n):
            er.advance(w_per)
        er.timing(reset=True)
        e0 = er.steps()
        sync()
        te0 = time.perf_counter()
        for _ in range(n_launch):
            er.advance(per_launch)
        sync()
        te = time.perf_counter() - te0
        ek = er.timing()[0]
        out["value_early_reject"] = (er.steps() - e0) / te
        out["early_reject"] = {
            "what": "the same walk, MHX_EARLY_REJECT=1: sweeps end where the partial sum of squares has lost "
                    "the accept test (exact: same chains bit for bit; a run-time compiled kernel)",
            "kernel_ms_per_launch": ek["avg_ms"], "kernel": "mhx_user_adaptive @ " + er_name,
            "ratio_to_value": out["value_early_reject"] / out["value"]}
        er.close()
    # ... and the reference's own way of working: ONE walker (and a small batch of 64) on the same
    # data.  Too few chains for the batch kernel: the likelihood sum is spread over sweep
    # workgroups and a whole portion of iterations runs as one persistent launch (k_persist_ts;
    # two launches per iteration until round 3).  Microseconds per iteration of a 1024-iteration
    # stretch after 256 of warm-up, host clock around synchronised calls.
    if (rank == 0 and not per_rank and n_gpus == 1 and not args.no_direct and args.workload == "c2"):
        sb = {"what": "the same data walked by 1 and by 64 walkers (walker-adaptive-steps from theta*(1 + 1 %), "
                      "T = 10): us per iteration over 1024 iterations after 256"}
        for c in (1, 64):
            small = Fleet(mhx, spec, c, devices, False, seed=0x5EED0005, chain_offset=first_id)
            small.start(th0[:c], n_adapt, l0)
            small.advance(256)
            s0 = small.steps()
            sync()
            ts0 = time.perf_counter()
            small.advance(1024)
            sync()
            tsb = time.perf_counter() - ts0
            done = small.steps() - s0
            sb["walkers_%d" % c] = {"us_per_iteration": tsb / 1024 * 1e6, "chain_steps_per_s": done / tsb,
                                    "kernel": small.kernel_name(), "trapped": bool(small.trapped())}
            small.close()
        out["small_batches"] = sb
    if rank == 0 and not per_rank and n_gpus == 1 and not args.no_cpu:
        ncpu, cinfo = effective_cores()
        one = cpu_baseline(spec, th0, args.cpu_seconds * 0.4, n_adapt, 1, l0)
        allc = cpu_baseline(spec, th0, args.cpu_seconds * 0.6, n_adapt, ncpu, l0) if ncpu > 1 else one
        out["cpu_baseline"] = {
            "value": allc[0], "unit": "chain-steps/s", "cores": ncpu, "kind": "port",
            "cores_from": cinfo,
            "parallel_speedup": allc[0] / one[0] if one[0] else None,
            "sample": "%d chains (one thread each) x %d steps in all of the same workload (%.1f s), "
                      "oracle/ faithful serial order, glibc libm" % (ncpu, allc[1], allc[2]),
            "single_core": {"value": one[0], "cores": 1,
                            "sample": "1 chain x %d steps (%.1f s)" % (one[1], one[2])}}
    if rank == 0:
        print(json.dumps(out))
    fleet.close()
    if spin is not None:
        spin.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
