#!/usr/bin/env python3
"""Prints, for the committed profiles/<tag>_*_summary.json, the figures DESIGN.md section 3.2 and
profiles/README.md tabulate (timed region, rate, instructions per point, issue fraction, clock,
waits, HBM bytes per chain-step, L2 hit rate, I-cache misses).  python tools/profile_rows.py [tag]"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
for path in sorted(glob.glob(os.path.join(ROOT, "profiles", tag + "_*_summary.json"))):
    d = json.load(open(path))
    t, p, b = d["timed_launch"], d["pmc_timed_launch"], d["bench_stats"]
    dur = t["duration_ns"] * 1e-9
    cs = b["steps"] * b["config"]["chains_per_gpu"]
    pts = cs * b["config"]["n_points"]
    line = [os.path.basename(path)[:-13], (b.get("build") or {}).get("id"),
            "%.2f ms" % (dur * 1e3), "%.4g chain-steps/s" % b["value"],
            "%.2f instr/pt" % (p["SQ_INSTS_VALU"] * 64 / pts),
            "frac %.3f" % (p["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9 * dur)),
            "clock %.2f GHz" % (p["GRBM_GUI_ACTIVE"] / 8 / dur / 1e9),
            "slots@clock %.3f" % (p["SQ_INSTS_VALU"] * 4 / (1024 * p["GRBM_GUI_ACTIVE"] / 8)),
            "wait_any %.3f" % (p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"]),
            "salu/valu %.2f" % (p["SQ_INSTS_SALU"] / p["SQ_INSTS_VALU"]),
            "scratch %s" % t.get("scratch"),
            "hbm %.0f + %.0f B" % (d["hbm_read_bytes"] / cs, d["hbm_write_bytes"] / cs),
            "l2 %.3f" % (p["TCC_HIT_sum"] / (p["TCC_HIT_sum"] + p["TCC_MISS_sum"])),
            "icache misses %s" % p.get("SQC_ICACHE_MISSES")]
    dl, pd = d.get("direct_launch"), d.get("pmc_direct_launch") or {}
    if dl and pd.get("SQ_INSTS_VALU"):
        line.append("direct: %.2f ms, %.2f instr/pt, %.4g" % (dl["duration_ns"] * 1e-6,
                    pd["SQ_INSTS_VALU"] * 64 / pts, b.get("value_direct_form") or 0))
    print(" | ".join(str(x) for x in line))
