#!/bin/bash
# All rocprofv3 summaries of a round's final build: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes per workload.
#   ROUND=r03v3 tools/profile_round.sh            (on the GPU box, through gpurun; then tools/collect_round.sh here)
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r04}
run() {   # tag, bench args, steps
  TAG=${ROUND}_$1 BENCH_ARGS="$2" tools/profile.sh || echo "profile $1 failed"
}
cd $R
run fbs "" 
run nr "--solver nr"
run c2 "--workload ieee13_b4096"
run c5 "--workload ieee8500_3ph_b1024"
TAG=${ROUND}_meshed_loops26 WORKLOAD=meshed_loops26_b8192 STEPS=20 tools/profile_meshed.sh || echo "profile loops26 failed"
TAG=${ROUND}_meshed_scalable WORKLOAD=meshed_scalable_b8192 STEPS=5 tools/profile_meshed.sh || echo "profile scalable failed"
