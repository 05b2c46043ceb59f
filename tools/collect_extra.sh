#!/bin/bash
# extra PMC passes for the stepping kernel (instruction / scalar-data cache behaviour, SMEM latency): tools/collect_extra.sh TAG
set -e
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL" "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH"; do
  tag=$(echo $set | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmcx_$tag -- python3 $ROOT/tools/profile_workload.py 50 3 > $OUT/pmcx_$tag.log 2>&1
  echo "pmc $tag done"
done
python3 - <<PY
import csv, glob, collections
c = collections.defaultdict(list)
for f in glob.glob("$OUT/pmcx_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_step_air" in r["Kernel_Name"]: c[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(c.items()): print(k, sum(v) / len(v), "per wave-step", sum(v) / len(v) / 819200)
PY
