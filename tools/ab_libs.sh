#!/bin/bash
# Same-box A/B of prebuilt libraries (box-to-box clocks differ by several per cent, so variants
# are only comparable inside ONE gpurun call).  Usage: tools/ab_libs.sh "libA.so libB.so ..." [bench args...]
LIBS="$1"; shift
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for lib in $LIBS; do
    MHX_LIBRARY=$PWD/lisp-mcmc_amd/$lib python3 bench.py --no-cpu "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', '$*', 'value %.4g' % d['value'], 'kernel_ms %.3f' % d['roofline']['kernel_ms_per_launch'], d['config']['kernel'])"
  done
done
