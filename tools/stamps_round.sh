#!/bin/bash
# Phase stamps of every step kernel of a round's final build (one wave of workgroup 0; diagnostic clocks, tools/phase_stamps.py):
#   ROUND=r04v2 tools/stamps_round.sh      (on the GPU box; the files land in gpurun_out/, copy them to profiles/)
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r04}
cd $R
one() {   # tag, args, waves
  out=gpurun_out/${ROUND}_$1_phase_stamps.txt; : > $out
  for w in $3; do GS_STAMP_WAVE=$w timeout -k 10 200 python tools/phase_stamps.py $2 >> $out 2>&1 || return 1; done
}
one fbs "--solver fbs --feeder ieee123" "0 3 7" && one nr "--solver nr --feeder ieee123" "0 3 7" && one c2 "--solver nr --feeder ieee13 --batch 4096" "0 1 2 3" && \
one meshed_loops26 "--solver nr --feeder loops26 --steps 20" "0 1 2 3" && one meshed_scalable "--solver nr --feeder scalable --steps 5" "0"
