#!/bin/bash
# Runs ON THE GPU BOX: every measurement profiles/ holds for one state of the code (tag = $1).
#   fp64 bench line + rocprof stats + PMC (collect_profile.sh), fp32 bench + stats + PMC, Xv2 bench + stats, mixed-fleet bench
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/all_$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err; echo "bench f64 done"
bash tools/collect_profile.sh $TAG > $OUT/collect.log 2>&1; echo "profile f64 done"
cp gpurun_out/prof_$TAG/summary/* $OUT/
python3 bench.py --dtype f32 > $OUT/${TAG}_f32_bench.json 2> $OUT/bench_f32.err; echo "bench f32 done"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32 -- python3 $ROOT/bench.py --dtype f32 --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/f32_under_rocprof.json 2> $OUT/stats_f32.log)
cp $OUT/stats_f32/*/*_kernel_stats.csv $OUT/${TAG}_f32_kernel_stats.csv
bash tools/collect_profile_f32.sh > $OUT/collect_f32.log 2>&1; cp gpurun_out/prof_f32pmc/r01_f32_counters.json $OUT/${TAG}_f32_counters.json; echo "profile f32 done"
python3 tools/bench_x2.py > $OUT/${TAG}_x2_bench.json 2> $OUT/bench_x2.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_x2 -- python3 $ROOT/tools/bench_x2.py > $OUT/x2_under_rocprof.json 2> $OUT/stats_x2.log)
cp $OUT/stats_x2/*/*_kernel_stats.csv $OUT/${TAG}_x2_kernel_stats.csv; echo "x2 done"
bash tools/collect_profile_x2.sh $TAG > $OUT/collect_x2.log 2>&1; cp gpurun_out/prof_x2_$TAG/${TAG}_x2_counters.json $OUT/; echo "x2 pmc done"
python3 tools/bench_fleet.py > $OUT/${TAG}_fleet_bench.json 2> $OUT/bench_fleet.err; echo "fleet done"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_gnd -- python3 $ROOT/tools/bench_ground.py > $OUT/${TAG}_ground_bench.txt 2> $OUT/stats_gnd.log)
cp $OUT/stats_gnd/*/*_kernel_stats.csv $OUT/${TAG}_ground_kernel_stats.csv; echo "ground done"
ls $OUT
