#!/bin/bash
# kernel trace of the tile-sliced persistent mode (64 walkers): where a portion's time goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_persist_ts
rm -rf $R && mkdir -p $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R -- python3 /root/repo/tools/debug/persist_ts_timing.py > $R/out.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$R/**/*_kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    last_end = None
    for r in rows[-40:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - last_end) / 1e3 if last_end else 0
        print("%-60s start %10.1f us  dur %9.1f us  gap %8.1f us" % (r["Kernel_Name"][:60], (s - t0) / 1e3, (e - s) / 1e3, gap))
        last_end = e
PY
tail -3 $R/out.txt
