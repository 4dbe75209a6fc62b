#!/bin/bash
# W sweep + solver/workload variants; writes one JSON line per run to gpurun_out/sweep.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for wl in ieee123_b8192 ieee13_b4096; do
  for solver in nr fbs; do
    for w in 1 2 4 8 16; do
      echo "## $wl $solver W=$w" >> gpurun_out/sweep.log
      timeout -k 10 120 python bench.py --steps 30 --warmup 5 --workload $wl --solver $solver --waves $w --no-cpu-baseline >> gpurun_out/sweep.log 2>&1 || exit 1
    done
  done
done
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/sweep.log 2>&1
