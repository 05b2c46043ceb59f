#!/bin/bash
# Runs ON THE GPU BOX: every measurement profiles/ holds for one state of the code (tag = $1).
# Order: the PMC passes FIRST, their summaries copied into the box's profiles/ under the names bench.py looks for, the
# bench lines LAST — so that every recorded bench line carries roofline.traffic / roofline_valu from counters of the very tree it ran on
# (bench.py compares the source hash and withholds them otherwise).
#   fp64 rocprof stats + PMC (collect_profile.sh), Xv2 PMC, fp32 PMC, then: fp64 bench line, fp32 bench, Xv2 bench + stats (identical and
#   divergent batches), dispersed fleet (stats + L2 counters), mixed fleet, ground batch
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/all_$TAG
mkdir -p $OUT
cd $ROOT
bash tools/collect_profile.sh $TAG > $OUT/collect.log 2>&1; echo "profile f64 done"
cp gpurun_out/prof_$TAG/summary/* $OUT/
cp gpurun_out/prof_$TAG/summary/${TAG}_counters.json $ROOT/profiles/
bash tools/collect_profile_x2.sh $TAG > $OUT/collect_x2.log 2>&1; cp gpurun_out/prof_x2_$TAG/${TAG}_x2_counters.json $OUT/; cp gpurun_out/prof_x2_$TAG/${TAG}_x2_counters.json $ROOT/profiles/; echo "x2 pmc done"
bash tools/collect_profile_f32.sh > $OUT/collect_f32.log 2>&1; cp gpurun_out/prof_f32pmc/r01_f32_counters.json $OUT/${TAG}_f32_counters.json; echo "profile f32 done"
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err; echo "bench f64 done"
python3 bench.py --dtype f32 > $OUT/${TAG}_f32_bench.json 2> $OUT/bench_f32.err; echo "bench f32 done"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32 -- python3 $ROOT/bench.py --dtype f32 --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/f32_under_rocprof.json 2> $OUT/stats_f32.log)
cp $OUT/stats_f32/*/*_kernel_stats.csv $OUT/${TAG}_f32_kernel_stats.csv
python3 tools/bench_x2.py > $OUT/${TAG}_x2_bench.json 2> $OUT/bench_x2.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_x2 -- python3 $ROOT/tools/bench_x2.py > $OUT/x2_under_rocprof.json 2> $OUT/stats_x2.log)
cp $OUT/stats_x2/*/*_kernel_stats.csv $OUT/${TAG}_x2_kernel_stats.csv; echo "x2 done"
# configs[3] on batches that diverge (tools/bench_x2_divergence.py): the four batches, then rocprof --stats of the fully divergent one
python3 tools/bench_x2_divergence.py > $OUT/${TAG}_x2_divergence.txt 2>&1
X2_RATIO=50 python3 tools/bench_x2_divergence.py identical trim both >> $OUT/${TAG}_x2_divergence.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_x2_lattice -- python3 $ROOT/tools/bench_x2_divergence.py both > $OUT/x2_lattice_under_rocprof.txt 2> $OUT/stats_x2_lattice.log)
cp $OUT/stats_x2_lattice/*/*_kernel_stats.csv $OUT/${TAG}_x2_lattice_kernel_stats.csv; echo "x2 divergence done"
# the headline batch placed at one point / over a 10 x 10 degree box / over the sphere (tools/bench_dispersed.py): timings, then per placement
# rocprof --stats and the L2 counters of the stepping kernel
python3 tools/bench_dispersed.py 10 > $OUT/${TAG}_dispersed.txt 2>&1
for pl in point box sphere; do
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_disp_$pl -- python3 $ROOT/tools/bench_dispersed.py 10 $pl > $OUT/disp_${pl}_under_rocprof.txt 2> $OUT/stats_disp_$pl.log)
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_disp_$pl -- python3 $ROOT/tools/bench_dispersed.py 4 $pl > $OUT/disp_${pl}_under_pmc.txt 2> $OUT/pmc_disp_$pl.log) || echo "pmc $pl failed"
done
python3 - >> $OUT/${TAG}_dispersed.txt <<PY
import csv, glob, collections
print("\n# per placement: rocprofv3 --kernel-trace --stats (k_step_duo<0, false>: calls, average / min / max ns) and the L2 counters per launch of that kernel")
for pl in ("point", "box", "sphere"):
    for f in glob.glob("$OUT/stats_disp_%s/*/*_kernel_stats.csv" % pl):
        for r in csv.DictReader(open(f)):
            if "k_step_duo<0, false>" in r["Name"]:
                print("%-7s stats: calls %s avg %.0f ns min %s max %s" % (pl, r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
    c = collections.defaultdict(list)
    for f in glob.glob("$OUT/pmc_disp_%s/*/*_counter_collection.csv" % pl):
        for r in csv.DictReader(open(f)):
            if "k_step_duo<0, false>" in r["Kernel_Name"]: c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if c: print("%-7s pmc per launch: " % pl + ", ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(c.items())))
PY
echo "dispersed done"
python3 tools/bench_fleet.py > $OUT/${TAG}_fleet_bench.json 2> $OUT/bench_fleet.err; echo "fleet done"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_gnd -- python3 $ROOT/tools/bench_ground.py > $OUT/${TAG}_ground_bench.txt 2> $OUT/stats_gnd.log)
cp $OUT/stats_gnd/*/*_kernel_stats.csv $OUT/${TAG}_ground_kernel_stats.csv; echo "ground done"
ls $OUT
