#!/bin/bash
# Same-box A/B of build-time tuning knobs (box-to-box clocks differ by several per cent, so two
# variants are only comparable when measured in ONE gpurun call).
# Usage: tools/ab_build.sh "<defines A>" "<defines B>" [bench args...]
set -e
A="$1"; B="$2"; shift 2
cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
mkdir -p gpurun_out
make -s -B -C lisp-mcmc_amd/csrc OUT=../libmhx_A.so CXXFLAGS="$BASE $A" > gpurun_out/ab_build.log 2>&1
make -s -B -C lisp-mcmc_amd/csrc OUT=../libmhx_B.so CXXFLAGS="$BASE $B" >> gpurun_out/ab_build.log 2>&1
for r in 1 2; do
  for v in A B; do
    MHX_LIBRARY=$PWD/lisp-mcmc_amd/libmhx_$v.so python bench.py --no-cpu "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '%.4g' % d['value'], d['roofline']['kernel_ms_per_launch'])"
  done
done
rm -f lisp-mcmc_amd/libmhx_A.so lisp-mcmc_amd/libmhx_B.so
