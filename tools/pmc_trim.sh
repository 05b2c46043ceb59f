#!/bin/bash
# ON THE GPU BOX: what k_trim waits for — instruction cache, vector memory (scratch), issue:   tools/pmc_trim.sh -> gpurun_out/pmc_trim.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_trim
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "SQ_IFETCH SQ_WAIT_IFETCH SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -- python3 $ROOT/tools/bench_trim.py > $OUT/s$i.log 2>&1
done
python3 - <<PY > $ROOT/gpurun_out/pmc_trim.txt
import csv, glob, collections
c = collections.defaultdict(list); d = []
for f in glob.glob("$OUT/s*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_trim(" in r["Kernel_Name"]:
            c[r["Counter_Name"]].append(float(r["Counter_Value"])); d.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("k_trim %.3f ms under PMC (mean over %d launches: lattice x2, ramp x2)" % (sum(d) / max(len(d), 1) / 1e6, len(d)))
for k, v in sorted(c.items()): print("   %-22s %.4e per launch" % (k, sum(v) / len(v)))
PY
cat $ROOT/gpurun_out/pmc_trim.txt
