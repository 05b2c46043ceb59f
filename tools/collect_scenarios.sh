#!/bin/bash
# Runs ON THE GPU BOX: the scenario studies profiles/ holds (tag = $1): the million-landing study under rocprofv3 --kernel-trace --stats, the same study
# with the table every 1 / 5 / 25 steps, the traffic pattern at 262 144 and 1 048 576 aircraft, the ground-capable Cessna172Xv2 pass on a parked batch.
set -e
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/scn_$TAG
mkdir -p $OUT
cd $ROOT
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_landing -- python3 $ROOT/examples/crosswind_landing.py 1048576 disperse device > $OUT/landing_under_rocprof.txt 2> $OUT/stats_landing.log)
cp $OUT/stats_landing/*/*_kernel_stats.csv $OUT/${TAG}_scenario_1m_kernel_stats.csv; echo "landing under rocprof done"
for e in 1 5 25; do timeout -k 10 200 python3 examples/crosswind_landing.py 1048576 disperse device every=$e > $OUT/landing_every$e.txt 2>&1; done; echo "landing every=1/5/25 done"
timeout -k 10 200 python3 examples/traffic_pattern.py 262144 device > $OUT/pattern_256k.txt 2>&1
timeout -k 10 400 python3 examples/traffic_pattern.py 1048576 device > $OUT/pattern_1m.txt 2>&1; echo "traffic pattern done"
timeout -k 10 300 python3 tools/bench_ground_x2.py > $OUT/${TAG}_ground_x2.txt 2>&1; echo "ground x2 done"
