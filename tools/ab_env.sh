#!/bin/bash
# Same-box A/B of run-time switches: tools/ab_env.sh "ENV1=a ENV2=b" "ENV1=c" -- [bench args...]
# (each quoted group is one variant's environment; an empty string "" is the default)
VARS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do VARS+=("$1"); shift; done
shift
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in "${VARS[@]}"; do
    env $v python3 bench.py --no-cpu "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[%s]' % '$v', '$*', 'value %.4g' % d['value'], 'kernel_ms %.3f' % d['roofline']['kernel_ms_per_launch'], d['config']['kernel'])"
  done
done
