#!/bin/bash
# Round profile: kernel-trace stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE) for one bench config.
#   TAG=r01_nr BENCH_ARGS="--solver nr" tools/profile.sh
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r01}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --steps 30 --warmup 5 --repeats 2 --no-cpu-baseline --no-also --no-secondary ${BENCH_ARGS} > $R/gpurun_out/${TAG}_trace.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_pmc_$c -- python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-also --no-secondary ${BENCH_ARGS} > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1 || exit 1
done
cd $R && timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 --no-also ${BENCH_ARGS} > gpurun_out/${TAG}_bench.log 2>&1
