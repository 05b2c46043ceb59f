#!/bin/bash
# Runs ON THE GPU BOX, the tail of tools/collect_all.sh (callable on its own): dispersed fleet, mixed fleet, ground batch -> gpurun_out/all_$TAG/
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/all_$TAG
mkdir -p $OUT
cd $ROOT
# the headline batch placed at one point / over a 10 x 10 degree box / over the sphere (tools/bench_dispersed.py): timings, then per placement
# rocprof --stats and the L2 counters of the stepping kernel
python3 tools/bench_dispersed.py 10 > $OUT/${TAG}_dispersed.txt 2>&1
for pl in point box sphere; do
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_disp_$pl -- python3 $ROOT/tools/bench_dispersed.py 10 $pl > $OUT/disp_${pl}_under_rocprof.txt 2> $OUT/stats_disp_$pl.log)
  # (one counter group per pass: together they exceed what the hardware collects at once, and rocprofv3 then aborts; every pass under a timeout)
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tg=$(echo $set | cut -d" " -f1)
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_disp_${pl}_$tg -- python3 $ROOT/tools/bench_dispersed.py 4 $pl > $OUT/disp_${pl}_under_pmc_$tg.txt 2> $OUT/pmc_disp_${pl}_$tg.log) || echo "pmc $pl $tg failed"
  done
  echo "dispersed $pl done"
done
python3 - >> $OUT/${TAG}_dispersed.txt <<PY
import csv, glob, collections
print("\n# per placement: rocprofv3 --kernel-trace --stats (k_step_duo<0, false, false>: calls, average / min / max ns) and the L2 counters per launch of that kernel")
for pl in ("point", "box", "sphere"):
    for f in glob.glob("$OUT/stats_disp_%s/*/*_kernel_stats.csv" % pl):
        for r in csv.DictReader(open(f)):
            if "k_step_duo<0, false, false>" in r["Name"]:
                print("%-7s stats: calls %s avg %.0f ns min %s max %s" % (pl, r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
    c = collections.defaultdict(list)
    for f in glob.glob("$OUT/pmc_disp_%s_*/*/*_counter_collection.csv" % pl):
        for r in csv.DictReader(open(f)):
            if "k_step_duo<0, false, false>" in r["Kernel_Name"]: c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if c: print("%-7s pmc per launch: " % pl + ", ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(c.items())))
PY
echo "dispersed done"
python3 tools/bench_fleet.py > $OUT/${TAG}_fleet_bench.json 2> $OUT/bench_fleet.err; echo "fleet done"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_gnd -- python3 $ROOT/tools/bench_ground.py > $OUT/${TAG}_ground_bench.txt 2> $OUT/stats_gnd.log)
cp $OUT/stats_gnd/*/*_kernel_stats.csv $OUT/${TAG}_ground_kernel_stats.csv; echo "ground done"
ls $OUT
