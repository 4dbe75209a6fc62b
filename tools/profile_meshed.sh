#!/bin/bash
# Round-3 profile of the meshed workloads: kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes.
#   TAG=r03v1_meshed_loops26 WORKLOAD=meshed_loops26_b8192 STEPS=20 tools/profile_meshed.sh
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r03_meshed}
WORKLOAD=${WORKLOAD:-meshed_loops26_b8192}
STEPS=${STEPS:-20}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
COMMON="--workload $WORKLOAD --no-cpu-baseline --no-also --no-secondary"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --steps $STEPS --warmup 2 --repeats 2 $COMMON > $R/gpurun_out/${TAG}_trace.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --repeats 1 $COMMON > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1 || exit 1
done
cd $R && timeout -k 10 400 python3 bench.py --steps $STEPS --warmup 2 --repeats 3 --workload $WORKLOAD --no-also --no-secondary > gpurun_out/${TAG}_bench.log 2>&1
