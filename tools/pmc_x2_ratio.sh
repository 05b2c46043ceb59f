#!/bin/bash
# Runs ON THE GPU BOX: instruction counters of the Cessna172Xv2 airborne stepper at control ratio 1 and 50 (tools/bench_x2_ratio.py):
# the difference, over 49 more updates per 50-step launch, is what ONE in-kernel control update issues and waits for.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_x2r
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ratio in 1 50; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_BUSY_CYCLES" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/r${ratio}_p$i -- python3 $ROOT/tools/bench_x2_ratio.py $ratio > $OUT/r${ratio}_p$i.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
res = {}
for ratio in (1, 50):
    c = collections.defaultdict(list)
    for f in glob.glob("$OUT/r%d_p*/*/*_counter_collection.csv" % ratio):
        for r in csv.DictReader(open(f)):
            if "k_step_air<0, true, false, false>" in r["Kernel_Name"]: c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[ratio] = {k: sum(v) / len(v) for k, v in c.items()}
waves = res[1].get("SQ_WAVES", 8192.0)
print("%-22s %14s %14s %18s" % ("counter", "ratio 1", "ratio 50", "per update per wave"))
for k in sorted(res[1]):
    a, b = res[1][k], res[50].get(k, float("nan"))
    print("%-22s %14.5g %14.5g %18.1f" % (k, a, b, (a - b) / 49.0 / waves))
PY
