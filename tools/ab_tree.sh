#!/bin/bash
# Same-box A/B of an OLDER SOURCE TREE against the current one.  Before the gpurun call:
#   mkdir -p tools/tmp_old && git archive <commit> lisp-mcmc_amd/csrc include tools/embed_src.py | tar -x -C tools/tmp_old
# (tools/tmp_old is git-ignored).  Usage on the box: tools/ab_tree.sh "<bench args>;<bench args>;..."
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -j2 -C tools/tmp_old/lisp-mcmc_amd/csrc OUT=$PWD/lisp-mcmc_amd/libmhx_A.so > gpurun_out/ab_build.log 2>&1
cp lisp-mcmc_amd/libmhx.so lisp-mcmc_amd/libmhx_B.so
IFS=";"
for wl in $1; do
  for v in A B A B; do
    MHX_LIBRARY=$PWD/lisp-mcmc_amd/libmhx_$v.so bash -c "python bench.py --no-cpu $wl" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$wl', '%.4g' % d['value'], '%.4g' % d['roofline']['kernel_ms_per_launch'])"
  done
done
rm -f lisp-mcmc_amd/libmhx_A.so lisp-mcmc_amd/libmhx_B.so
