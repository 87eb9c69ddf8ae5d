#!/bin/bash
# tools/ab_build.sh over several workloads: tools/ab_multi.sh "<defines A>" "<defines B>"
set -e
A="$1"; B="$2"; IFS=";"; WORKLOADS=${3:-"--workload c2;--workload c3 --chains 4096;--workload c4;--workload poly7;--workload c1;--workload c1 --chains 4096"}
cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
mkdir -p gpurun_out
make -s -B -C lisp-mcmc_amd/csrc OUT=../libmhx_A.so CXXFLAGS="$BASE $A" > gpurun_out/ab_build.log 2>&1
make -s -B -C lisp-mcmc_amd/csrc OUT=../libmhx_B.so CXXFLAGS="$BASE $B" >> gpurun_out/ab_build.log 2>&1
for wl in $WORKLOADS; do
  for v in A B; do
    MHX_LIBRARY=$PWD/lisp-mcmc_amd/libmhx_$v.so bash -c "python bench.py --no-cpu $wl" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$wl', '%.4g' % d['value'], '%.4g' % d['roofline']['kernel_ms_per_launch'])"
  done
done
rm -f lisp-mcmc_amd/libmhx_A.so lisp-mcmc_amd/libmhx_B.so
