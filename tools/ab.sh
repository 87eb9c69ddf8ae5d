#!/bin/bash
# Same-box A/B of prebuilt libraries and run-time switches (box-to-box clocks differ by several
# per cent: variants are only comparable inside ONE gpurun call).
# Usage: tools/ab.sh "label|lib.so|ENV=1 ENV2=x" "label2|lib2.so|" ... -- [bench args...]
# (lib relative to lisp-mcmc_amd/; an empty lib means libmhx.so; ROUNDS rounds - default 2 - interleaved)
VARS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do VARS+=("$1"); shift; done
shift
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in "${VARS[@]}"; do
    IFS='|' read -r label lib envs <<< "$v"
    lib=${lib:-libmhx.so}
    env MHX_LIBRARY=$PWD/lisp-mcmc_amd/$lib $envs python3 bench.py --no-cpu --no-direct "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-14s' % '$label', '$*', 'value %.4g' % d['value'], 'kernel_ms %.3f' % d['roofline']['kernel_ms_per_launch'], d['config']['kernel'])"
  done
done
