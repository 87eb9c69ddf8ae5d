#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats and, in SEPARATE passes, the PMC
# counters of bench.py's timed launch, condensed into gpurun_out/prof_<name>/summary.json with
# <name> = <round>_<workload>_s<steps>_w<warmup> (what bench.py looks for under profiles/).
# Usage: tools/run_profile.sh <round tag> <workload> <steps> <warmup> [more bench args...]
set -e
TAG=${1:-r04}; WL=${2:-c2}; ST=${3:-200}; WU=${4:-200}; shift 4 || true
NAME=${TAG}_${WL}_s${ST}_w${WU}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_$NAME
rm -rf $R && mkdir -p $R
ARGS="--no-cpu --workload $WL --steps $ST --warmup $WU $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- python3 /root/repo/bench.py $ARGS > $R/bench_stats.json 2> $R/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/fetch -- python3 /root/repo/bench.py $ARGS > /dev/null 2> $R/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/write -- python3 /root/repo/bench.py $ARGS > /dev/null 2> $R/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/sq -- python3 /root/repo/bench.py $ARGS > /dev/null 2> $R/sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/lds -- python3 /root/repo/bench.py $ARGS > /dev/null 2> $R/lds.err || true
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/ic -- python3 /root/repo/bench.py $ARGS > /dev/null 2> $R/ic.err || true
python3 /root/repo/bench.py $ARGS > $R/bench_unprofiled.json 2>/dev/null
python3 /root/repo/tools/profile_summary.py $R > $R/summary.json
cp $R/summary.json $GRAFT_REPO_ROOT/gpurun_out/${NAME}_summary.json
find $R/stats -name "*_kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${NAME}_kernel_stats.csv \;
python3 - <<PY
import json
d = json.load(open("$R/summary.json"))
t, p = d.get("timed_launch") or {}, d.get("pmc_timed_launch", {})
b = d.get("bench_stats", {})
pts = b["steps"] * b["config"]["chains_per_gpu"] * b["config"]["n_points"]
ipp = p.get("SQ_INSTS_VALU", 0) * 64.0 / pts
dur = t.get("duration_ns", 0) * 1e-9
print("$NAME [%s]: timed region %.3f ms (dispatches %s of %s), vgpr %s scratch %s, %.2f VALU instr/point, "
      "VALU issue frac @2.4GHz %.3f, SALU/VALU %.3f, value %.4g"
      % (b.get("build", {}).get("id"), dur * 1e3, t.get("dispatches"), d.get("k_adaptive_dispatches"),
         t.get("vgpr"), t.get("scratch"), ipp,
         p.get("SQ_INSTS_VALU", 0) * 4 / (1024 * 2.4e9 * dur) if dur else 0,
         p.get("SQ_INSTS_SALU", 0) / max(p.get("SQ_INSTS_VALU", 1), 1), b.get("value", 0)))
pl, pp = d.get("piecewise_launch"), d.get("pmc_piecewise_launch", {})
if pl:
    print("  piecewise grids: %.3f ms, %.2f VALU instr/point, value_piecewise %.4g"
          % (pl["duration_ns"] * 1e-6, pp.get("SQ_INSTS_VALU", 0) * 64.0 / pts, b.get("value_piecewise", 0)))
dl, pd = d.get("direct_launch"), d.get("pmc_direct_launch", {})
if dl:
    print("  direct form: %.3f ms, %.2f VALU instr/point, value_direct_form %.4g"
          % (dl["duration_ns"] * 1e-6, pd.get("SQ_INSTS_VALU", 0) * 64.0 / pts, b.get("value_direct_form", 0)))
PY
