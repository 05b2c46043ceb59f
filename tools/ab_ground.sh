#!/bin/bash
# A/B timing of library variants on a batch that sits on the ground (tools/bench_ground.py) ON THE GPU BOX: tools/ab_ground.sh tag1 tag2 ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out
for tag in "$@"; do
  lib=flight.jl_amd/libflightbatch_$tag.so; [ "$tag" = main ] && lib=flight.jl_amd/libflightbatch.so
  echo "== $tag"; FLIGHTBATCH_LIB=$ROOT/$lib timeout -k 10 200 python tools/bench_ground.py 2>&1 | tail -2
done
