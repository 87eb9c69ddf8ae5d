#!/bin/bash
# same-box A/B: the persistent kernels with and without their s_setprio calls (a library built
# with -DMHX_PERSIST_NOPRIO as lisp-mcmc_amd/libmhx_noprio.so)
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for lib in libmhx.so libmhx_noprio.so; do
    echo "== $lib"
    MHX_LIBRARY=$PWD/lisp-mcmc_amd/$lib python3 tools/debug/persist_ts_sizes.py 2>/dev/null | head -6
  done
done
