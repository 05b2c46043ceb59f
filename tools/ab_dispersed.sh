#!/bin/bash
# A/B of library variants on the dispersed-fleet placements ON THE GPU BOX: tools/ab_dispersed.sh tag1 tag2 ... ("main" = the shipped library)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for tag in "$@"; do
  lib=flight.jl_amd/libflightbatch_$tag.so; [ "$tag" = main ] && lib=flight.jl_amd/libflightbatch.so
  echo "== $tag"
  FLIGHTBATCH_LIB=$ROOT/$lib timeout -k 10 300 python tools/bench_dispersed.py 10 point sphere 2>&1 | tail -2
done
