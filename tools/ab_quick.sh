#!/bin/bash
# A/B of library variants at 1 / 10 / 50 steps per launch ON THE GPU BOX (tools/quickbench.py): tools/ab_quick.sh tag1 tag2 ... ("main" = shipped)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for tag in "$@"; do
  lib=flight.jl_amd/libflightbatch_$tag.so; [ "$tag" = main ] && lib=flight.jl_amd/libflightbatch.so
  echo "== $tag"
  FLIGHTBATCH_LIB=$ROOT/$lib timeout -k 10 300 python tools/quickbench.py 2>&1 | grep "^k="
done
