#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats and, in SEPARATE passes, the PMC
# counters of bench.py's timed launch.  Usage: tools/run_profile.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $R && mkdir -p $R
ARGS="--no-cpu $*"   # bench.py defaults: 200 warm-up + 200 timed iterations, one launch each
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- python3 /root/repo/bench.py $ARGS > $R/bench_stats.json 2> $R/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/fetch -- python3 /root/repo/bench.py $ARGS > /dev/null 2> $R/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/write -- python3 /root/repo/bench.py $ARGS > /dev/null 2> $R/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/sq -- python3 /root/repo/bench.py $ARGS > /dev/null 2> $R/sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/lds -- python3 /root/repo/bench.py $ARGS > /dev/null 2> $R/lds.err || true
python3 /root/repo/bench.py $ARGS > $R/bench_unprofiled.json 2>/dev/null
python3 /root/repo/tools/profile_summary.py $R > $R/summary.json
cat $R/summary.json
