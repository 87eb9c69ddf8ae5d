#!/usr/bin/env python3
"""Round 2's exit abort, looked at once more with the old dlopen flags (diagnostics; run by hand
on a GPU box: `python3 tools/debug/rccl_exit_abort.py > gpurun_out/rccl_exit_abort.txt 2>&1`).

Four full-suite pytest processes of round 2 ended in `double free or corruption (!prev)` /
`free(): invalid pointer` AFTER pytest had printed its summary, while libmhx opened librccl with
RTLD_GLOBAL (| RTLD_NODELETE); none since it opens it RTLD_LOCAL.  This script runs the same
ingredients in child processes - hiprtc-compiled model, real librccl in a 1-rank communicator,
engines left to the interpreter's exit - with MHX_RCCL_DLOPEN_GLOBAL=1 (the old flags) and
without, under an LD_PRELOADed SIGABRT handler that prints the C backtrace and the mapped
HIP/RCCL/compiler libraries.  It prints each child's exit status; it asserts nothing."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.dirname(os.path.abspath(__file__))

CHILD = r"""
import os, sys
sys.path[:0] = [%r, %r]
import numpy as np
order = os.environ["ORDER"]
import lisp_mcmc_amd as mhx
import problems as pb
s = pb.two_peak(n=2000, seed=6)
th0 = pb.perturbed(s.theta_star, 32, 0.01, seed=8)
keep = []
def rccl():
    e = s.engine(mhx, 32, seed=12, adapt_mode=mhx.capi.ADAPT_POOLED)
    e.comm_init_rank(mhx.comm_unique_id(), 0, 1)
    e.init_chains(th0); e.adaptive_begin(30000, 10.0, 1); e.adaptive_advance(450)
    keep.append(e)
def rtc():
    e = s.engine(mhx, 8, seed=1)
    keys, cexpr = mhx.sexpr.lambda_to_expr(
        "(lambda (x &key b0 b1 a1 mu1 w1 a2 mu2 w2 &allow-other-keys)"
        " (+ (+ b0 (* b1 x)) (* a1 (exp (- (expt (/ (- x mu1) w1) 2))))"
        "    (* a2 (exp (- (expt (/ (- x mu2) w2) 2))))))")
    e.set_expr_recognition(False)
    e.set_function_expr(0, cexpr, keys, list(range(8)))
    e.init_chains(th0[:8]); e.kernel_name()
    keep.append(e)
def torch_():
    import torch
    torch.zeros(4, device="cuda").sum().item()
for step in order.split(","):
    {"rccl": rccl, "rtc": rtc, "torch": torch_}[step]()
print("work done:", order, flush=True)
""" % (ROOT, os.path.join(ROOT, "tests"))


def main():
    so = os.path.join(HERE, "abort_bt.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "abort_bt.c")])
    for glob in ("1", "0"):
        # (round 3 dropped the order with a torch step: "hung under the preload", unlogged.  The
        # preload's handler was not async-signal-safe - tools/debug/abort_bt.c says what that does
        # to a process that aborts inside free() - and a timeout here threw away what the child
        # had printed.  Both fixed: the order is back, a child that does not end is reported.)
        for order in ("rccl", "rtc,rccl", "rccl,rtc", "rccl,rtc,torch"):
            env = dict(os.environ, ORDER=order, MHX_RCCL_DLOPEN_GLOBAL=glob, LD_PRELOAD=so,
                       MHX_SPLIT="0", MHX_LIBRARY=os.path.join(ROOT, "tests", "hooks", "libmhx_hooks.so"))
            try:
                r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True,
                                   timeout=240)
            except subprocess.TimeoutExpired as t:
                print("==== RTLD_%s order=%s -> NOT FINISHED after 240 s" %
                      ("GLOBAL|NODELETE" if glob == "1" else "LOCAL", order))
                print((t.stdout or b"")[-500:])
                print((t.stderr or b"")[-6000:])
                sys.stdout.flush()
                continue
            print("==== RTLD_%s order=%s -> exit status %d" % ("GLOBAL|NODELETE" if glob == "1" else "LOCAL",
                                                             order, r.returncode))
            if r.returncode != 0:
                print(r.stdout[-500:])
                print(r.stderr[-6000:])
            sys.stdout.flush()


def bindings():
    """Round 4: the abort reproduces - old flags, order rccl,rtc,torch: `double free or corruption
    (!prev)` after the work is done; RTLD_LOCAL, same order: exit status 0.  This pass runs that
    one configuration under LD_DEBUG=bindings and prints which libraries had symbols of theirs
    bound INTO the librccl that libmhx opened (what RTLD_GLOBAL makes possible and RTLD_LOCAL
    does not): the mechanism, not only the correlation."""
    import collections
    import glob
    import re
    import tempfile
    out = tempfile.mkdtemp(prefix="mhx_ldd_")
    for glob_flag in ("1", "0"):
        base = os.path.join(out, "b%s" % glob_flag)
        env = dict(os.environ, ORDER="rccl,rtc,torch", MHX_RCCL_DLOPEN_GLOBAL=glob_flag, MHX_SPLIT="0",
                   MHX_LIBRARY=os.path.join(ROOT, "tests", "hooks", "libmhx_hooks.so"),
                   LD_DEBUG="bindings", LD_DEBUG_OUTPUT=base)
        try:
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
            status = r.returncode
        except subprocess.TimeoutExpired:
            status = "timeout"
        into = collections.Counter()
        syms = collections.Counter()
        pat = re.compile(r"binding file (\S+) \[\d+\] to (\S+) \[\d+\]: normal symbol `([^']+)'")
        for f in glob.glob(base + ".*"):
            for line in open(f, errors="replace"):
                m = pat.search(line)
                if m and "librccl" in m.group(2) and "librccl" not in m.group(1) and "libmhx" not in m.group(1):
                    into[os.path.basename(m.group(1))] += 1
                    syms[m.group(3)] += 1
        print("==== RTLD_%s order=rccl,rtc,torch under LD_DEBUG=bindings -> exit status %s"
              % ("GLOBAL|NODELETE" if glob_flag == "1" else "LOCAL", status))
        print("   symbols of OTHER libraries bound into libmhx's librccl: %d, from %s"
              % (sum(into.values()), dict(into.most_common(8))))
        print("   the most frequent: %s" % [s for s, _ in syms.most_common(12)])
        sys.stdout.flush()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--bindings":
        bindings()
    else:
        main()
