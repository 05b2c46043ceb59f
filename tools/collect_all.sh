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
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_x2 -- python3 $ROOT/tools/bench_x2.py --no-parity > $OUT/x2_under_rocprof.json 2> $OUT/stats_x2.log)
cp $OUT/stats_x2/*/*_kernel_stats.csv $OUT/${TAG}_x2_kernel_stats.csv; echo "x2 done"
# configs[3] on batches that diverge (tools/bench_x2_divergence.py): the four batches, then rocprof --stats of the fully divergent one
python3 tools/bench_x2_divergence.py > $OUT/${TAG}_x2_divergence.txt 2>&1
X2_RATIO=50 python3 tools/bench_x2_divergence.py identical trim both >> $OUT/${TAG}_x2_divergence.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_x2_lattice -- python3 $ROOT/tools/bench_x2_divergence.py both > $OUT/x2_lattice_under_rocprof.txt 2> $OUT/stats_x2_lattice.log)
cp $OUT/stats_x2_lattice/*/*_kernel_stats.csv $OUT/${TAG}_x2_lattice_kernel_stats.csv; echo "x2 divergence done"
[ -n "$SKIP_TAIL" ] || bash tools/collect_tail.sh $TAG
