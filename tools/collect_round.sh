#!/bin/bash
# After tools/profile_round.sh: copy the summaries into profiles/ and refresh profiles/hbm_traffic.json.
ROUND=${ROUND:-r04}
cd "$(dirname "$0")/.."
python tools/collect_profile.py ${ROUND}_fbs ${ROUND}_fbs --traffic-key ieee123_b8192:fbs --kernel gs_k_step_fbs_flow2h --dispatches-per-step 2
python tools/collect_profile.py ${ROUND}_nr ${ROUND}_nr --traffic-key ieee123_b8192:nr --kernel gs_k_step_nr_flow2
python tools/collect_profile.py ${ROUND}_c2 ${ROUND}_c2 --traffic-key ieee13_b4096:nr --kernel gs_k_step_nr_flow2s --dispatches-per-step 2
python tools/collect_profile.py ${ROUND}_c5 ${ROUND}_c5 --traffic-key ieee8500_3ph_b1024:fbs3 --kernel gs3_k_resident
python tools/collect_profile.py ${ROUND}_meshed_loops26 ${ROUND}_meshed_loops26 --traffic-key meshed_loops26_b8192:nr --kernel gs_k_step_nr_mesh2 --dispatches-per-step 2
python tools/collect_profile.py ${ROUND}_meshed_scalable ${ROUND}_meshed_scalable --traffic-key meshed_scalable_b8192:nr --kernel gs_k_nr_dense_mfma2
