#!/bin/bash
# Runs ON THE GPU BOX: instruction-cache and wait counters of the Cessna172Xv2 airborne stepper on tools/bench_x2.py
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_x2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/bench_x2.py > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
c = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_step_air<0, true, false, false>" in r["Kernel_Name"]: c[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(c): print("%-24s %.4g  (x %d launches)" % (k, sum(c[k]) / len(c[k]), len(c[k])))
PY
