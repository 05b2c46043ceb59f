#!/bin/bash
# ON THE GPU BOX: instruction-cache behaviour of the Cessna172Xv2 steppers (k_step_duo<KIN, true> and, with FLIGHTBATCH_DUO=0, k_step_air<KIN, true>):
#   tools/pmc_icache_x2.sh -> gpurun_out/icache_x2.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_icache_x2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for duo in 1 0; do
  export FLIGHTBATCH_DUO=$duo
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $OUT/duo$duo -- python3 $ROOT/tools/profile_workload_x2.py 50 3 > $OUT/duo$duo.log 2>&1
done
python3 - <<PY > $ROOT/gpurun_out/icache_x2.txt
import csv, glob, collections
for duo in (1, 0):
    c = collections.defaultdict(list); d = []
    for f in glob.glob("$OUT/duo%d/*/*_counter_collection.csv" % duo):
        for r in csv.DictReader(open(f)):
            if ("k_step_duo" in r["Kernel_Name"]) or ("k_step_air" in r["Kernel_Name"] and "true, false" in r["Kernel_Name"]):
                c[r["Counter_Name"]].append(float(r["Counter_Value"])); d.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    print("FLIGHTBATCH_DUO=%d: kernel %.3f ms under PMC" % (duo, sum(d) / max(len(d), 1) / 1e6))
    for k, v in sorted(c.items()): print("   %-20s %.4e per launch" % (k, sum(v) / len(v)))
PY
cat $ROOT/gpurun_out/icache_x2.txt
