#!/bin/bash
# debug: which runtime setting makes the run-time compiled g23 kernel run at the profiler's speed?
cd $GRAFT_REPO_ROOT
g() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], '%.4g' % d['value'], '%.2f' % d['roofline']['kernel_ms_per_launch'])" "$1"; }
python3 bench.py --no-cpu --workload g23 2>/dev/null | g "baseline"
for v in "HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0" "HSA_SCRATCH_SINGLE_LIMIT=4000000000" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "GPU_MAX_HW_QUEUES=1" "AMD_DIRECT_DISPATCH=0" "HIP_LAUNCH_BLOCKING=1" "HSA_ENABLE_INTERRUPT=0" "HSA_NO_SCRATCH_RECLAIM=1" "HIP_ENABLE_DEFERRED_LOADING=0" "HSA_XNACK=0 HSA_ENABLE_SDMA=0" "AMD_SERIALIZE_KERNEL=3" "HSA_OVERRIDE_CPU_AFFINITY_DEBUG=0"; do
  env $v python3 bench.py --no-cpu --workload g23 2>/dev/null | g "$v"
done
