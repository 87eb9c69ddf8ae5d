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


def counters(d, kernel_substr):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                key = (int(r["Dispatch_Id"]), r["Counter_Name"])
                per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
        if not per:
            continue
        last = max(k[0] for k in per)  # the timed launch is the last dispatch of the kernel
        for (disp, name), v in per.items():
            if disp == last:
                out[name] = v
    return out


def main():
    root = sys.argv[1]
    kern = "k_adaptive"
    res = {"dir": os.path.basename(root)}
    for f in glob.glob(os.path.join(root, "stats", "**", "*_kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        res["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")}
                               for r in rows[:6]]
    for f in glob.glob(os.path.join(root, "stats", "**", "*_kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
        if rows:
            r = rows[-1]
            res["timed_launch"] = {"duration_ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                   "vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"),
                                   "lds": r.get("LDS_Block_Size"), "scratch": r.get("Scratch_Size"),
                                   "grid": r.get("Grid_Size_X"), "workgroup": r.get("Workgroup_Size_X")}
    c = {}
    for sub in ("fetch", "write", "sq", "lds"):
        c.update(counters(os.path.join(root, sub), kern))
    res["pmc_timed_launch"] = c
    if "FETCH_SIZE" in c:
        res["hbm_read_bytes"] = c["FETCH_SIZE"] * 1024 * 2  # gfx950 correction (guide, HBM section)
    if "WRITE_SIZE" in c:
        res["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    for name in ("bench_stats.json", "bench_unprofiled.json"):
        p = os.path.join(root, name)
        if os.path.exists(p) and os.path.getsize(p):
            try:
                res[name[:-5]] = json.loads(open(p).read().strip().splitlines()[-1])
            except Exception as ex:  # noqa
                res[name[:-5]] = "unparsed: %s" % ex
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
