#!/bin/bash
# Runs ON THE GPU BOX: the same PMC sets for the one-wave-per-SIMD stepper (FLIGHTBATCH_DUO=0) and the wave-specialised one (=1),
# on tools/profile_workload.py; prints the per-launch means of the stepping kernel's counters side by side.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_ab
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for duo in 0 1; do
  export FLIGHTBATCH_DUO=$duo
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU" \
             "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SMEM" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/duo${duo}_$i -- python3 $ROOT/tools/profile_workload.py 50 3 > $OUT/duo${duo}_$i.log 2>&1
    echo "duo=$duo set $i done"
  done
done
python3 - <<PY
import csv, glob, collections
for duo in (0, 1):
    c = collections.defaultdict(list); d = []
    for f in glob.glob("$OUT/duo%d_*/*/*_counter_collection.csv" % duo):
        for r in csv.DictReader(open(f)):
            if "k_step_duo" in r["Kernel_Name"] or "k_step_air<0, false, false, false>" in r["Kernel_Name"]:
                c[r["Counter_Name"]].append(float(r["Counter_Value"])); d.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    print("duo=%d  mean kernel ns under pmc %.0f" % (duo, sum(d) / max(len(d), 1)))
    for k in sorted(c): print("   %-24s %.4g" % (k, sum(c[k]) / len(c[k])))
PY
