#!/bin/bash
# HBM traffic of the step kernels: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/pmc_$c.log 2>&1 || exit 1
done
find $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head
