#!/bin/bash
# Batch kernels against split mode over the number of chains (choose_split's thresholds):
# tools/debug/split_sweep.sh <workload> "<chain counts>" [bench args]
WL=$1; shift; NS=$1; shift
cd $GRAFT_REPO_ROOT
for n in $NS; do
  for v in "MHX_SPLIT=0" "MHX_SPLIT=2" "MHX_SPLIT=4" "MHX_SPLIT=8" "MHX_SPLIT=24" "X=default"; do
    env $v python3 bench.py --no-cpu --workload $WL --chains $n "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$WL chains %6d %-14s %.4g  %s' % ($n, '$v', d['value'], d['config']['kernel']))"
  done
done
