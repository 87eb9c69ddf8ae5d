#!/bin/bash
# the shader clock during the persistent kernels: GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / duration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_persist_clock
rm -rf $R && mkdir -p $R
rocprofv3 --kernel-trace --output-format csv -d $R/t -- python3 /root/repo/tools/debug/persist_ts_timing.py > $R/out.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/c -- python3 /root/repo/tools/debug/persist_ts_timing.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list)
for f in glob.glob("$R/t/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_persist" in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:40]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cyc = collections.defaultdict(list)
for f in glob.glob("$R/c/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_persist" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k in dur:
    d, c = dur[k][-1], cyc[k][-1] if cyc[k] else 0
    print("%-42s last launch %.2f ms, GRBM_GUI_ACTIVE %.4g -> %.2f GHz" % (k, d / 1e6, c, c / 8 / d))
PY
