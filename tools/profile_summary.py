#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/run_profile.sh into one JSON summary (what gets
copied into profiles/).  HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are in
KiB-ish units of 1024 B... (rocprofv3 reports KB); on gfx950 FETCH_SIZE counts 64 B per 128-B
request for wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is."""
import csv
import glob
import json
import os
import sys


def dispatch_ids(root, kern):
    """Dispatch ids of the k_adaptive launches in launch order (kernel trace of the stats pass):
    bench.py names the timed region and the direct-form launch by their position in it."""
    ids = []
    for f in glob.glob(os.path.join(root, "stats", "**", "*_kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        ids = rows
    return ids


def counters(d, kernel_substr, first, last):
    """counter sums over the k_adaptive dispatches [first, last) of the pass under d (every pass
    is the same command: the same sequence of dispatches)"""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                per.setdefault(int(r["Dispatch_Id"]), {})
                c = per[int(r["Dispatch_Id"])]
                c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        order = sorted(per)
        for disp in order[first:last]:
            for name, v in per[disp].items():
                out[name] = out.get(name, 0.0) + v
        out["_dispatches_in_pass"] = len(order)
    return out


def main():
    root = sys.argv[1]
    kern = "k_adaptive"
    res = {"dir": os.path.basename(root)}
    bench = {}
    for name in ("bench_stats.json", "bench_unprofiled.json"):
        p = os.path.join(root, name)
        if os.path.exists(p) and os.path.getsize(p):
            try:
                bench[name[:-5]] = json.loads(open(p).read().strip().splitlines()[-1])
            except Exception as ex:  # noqa
                bench[name[:-5]] = "unparsed: %s" % ex
    bs = bench.get("bench_stats") if isinstance(bench.get("bench_stats"), dict) else {}
    trace = dispatch_ids(root, kern)
    # which dispatches are the timed region / the direct-form launch: bench.py says
    # (older lines: the last dispatch)
    timed = (bs.get("roofline") or {}).get("timed_dispatches") or [len(trace) - 1, len(trace)]
    direct = (bs.get("direct_form") or {}).get("dispatches")
    res["k_adaptive_dispatches"] = len(trace)
    for f in glob.glob(os.path.join(root, "stats", "**", "*_kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        res["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")}
                               for r in rows[:6]]

    def launch_info(span):
        rows = trace[span[0]:span[1]]
        if not rows:
            return None
        r = rows[-1]
        return {"dispatches": list(span),
                "duration_ns": sum(int(q["End_Timestamp"]) - int(q["Start_Timestamp"]) for q in rows),
                "vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"),
                "lds": r.get("LDS_Block_Size"), "scratch": r.get("Scratch_Size"),
                "grid": r.get("Grid_Size_X"), "workgroup": r.get("Workgroup_Size_X")}
    res["timed_launch"] = launch_info(timed)
    c = {}
    for sub in ("fetch", "write", "ic", "sq", "lds"):
        c.update(counters(os.path.join(root, sub), kern, timed[0], timed[1]))
    res["pmc_timed_launch"] = c
    if direct:
        res["direct_launch"] = launch_info(direct)
        cd = {}
        for sub in ("ic", "sq", "lds"):
            cd.update(counters(os.path.join(root, sub), kern, direct[0], direct[1]))
        res["pmc_direct_launch"] = cd
    piece = (bs.get("piecewise") or {}).get("dispatches")
    if piece:  # (a run-time compiled kernel: its dispatches are counted among their own)
        kern_pw = "mhx_user_adaptive"
        trace_pw = dispatch_ids(root, kern_pw)
        rows = trace_pw[piece[0]:piece[1]]
        if rows:
            res["piecewise_launch"] = {
                "dispatches": list(piece),
                "duration_ns": sum(int(q["End_Timestamp"]) - int(q["Start_Timestamp"]) for q in rows),
                "vgpr": rows[-1].get("VGPR_Count"), "scratch": rows[-1].get("Scratch_Size")}
        cp = {}
        for sub in ("sq", "lds"):
            cp.update(counters(os.path.join(root, sub), kern_pw, piece[0], piece[1]))
        res["pmc_piecewise_launch"] = cp
    if "FETCH_SIZE" in c:
        res["hbm_read_bytes"] = c["FETCH_SIZE"] * 1024 * 2  # gfx950 correction (guide, HBM section)
    if "WRITE_SIZE" in c:
        res["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    res.update(bench)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
