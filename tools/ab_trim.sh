#!/bin/bash
# same-box alternating A/B of k_trim (the lane's workspace slot opaque at every access): the library before the change kept as libflightbatch_pretrim.so
mkdir -p gpurun_out/ab_trim
for r in 1 2; do
  for v in base pretrim; do
    if [ $v = base ]; then unset FLIGHTBATCH_LIB; else export FLIGHTBATCH_LIB=$GRAFT_REPO_ROOT/flight.jl_amd/libflightbatch_$v.so; fi
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab_trim/$v.$r -- python3 $GRAFT_REPO_ROOT/tools/bench_trim.py > $GRAFT_REPO_ROOT/gpurun_out/ab_trim/$v.$r.txt 2> $GRAFT_REPO_ROOT/gpurun_out/ab_trim/$v.$r.log) || exit 1
    grep -h "k_trim" gpurun_out/ab_trim/$v.$r/*/*_kernel_stats.csv | awk -F, -v v=$v -v r=$r '{printf "%-8s run %s: k_trim calls %s, average %.2f ms, min %.2f ms\n", v, r, $(NF-6), $(NF-4)/1e6, $(NF-2)/1e6}'
    grep -h "wide\|fb_trim" gpurun_out/ab_trim/$v.$r.txt | tail -2
    timeout -k 10 200 python3 tools/bench_trim_wide.py 262144 2>&1 | tail -1 | sed "s/^/$v: /"
  done
done
