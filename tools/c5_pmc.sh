#!/bin/bash
# SQ counters of the resident three-phase kernel (one counter set per run): THREADS=512 tools/c5_pmc.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GS3_RESIDENT_THREADS=${THREADS:-512}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/c5pmc_$i -- python3 $R/tools/c5_resident_check.py --only-time --b256 > $R/gpurun_out/c5pmc_$i.log 2>&1 || { tail -3 $R/gpurun_out/c5pmc_$i.log; }
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob("$R/gpurun_out/c5pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in tot:
    if "resident" in k or "gs3_k_solve" in k:
        print(k)
        for c, v in sorted(tot[k].items()): print(f"   {c:28s} {v:.4g}")
PY
