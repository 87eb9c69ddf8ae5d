#!/bin/bash
# the five round-4 profiles, one after another (tools/run_profile.sh); stops at the first failure
set -e
cd $GRAFT_REPO_ROOT
tools/run_profile.sh r04 c2 20 5
tools/run_profile.sh r04 c2 200 200
tools/run_profile.sh r04 c3 20 20 --chains 4096
tools/run_profile.sh r04 c4 200 200
tools/run_profile.sh r04 c5 200 200
