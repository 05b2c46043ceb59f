#!/bin/bash
# A/B timing of library variants ON THE GPU BOX: tools/ab.sh tag1 tag2 ... (flight.jl_amd/libflightbatch_<tag>.so; "main" = the shipped library).
# Prints aircraft-steps/s and the stepping kernel's average launch time for the headline workload (10 launches of 50 steps, N = 1 M).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out
for tag in "$@"; do
  lib=flight.jl_amd/libflightbatch_$tag.so; [ "$tag" = main ] && lib=flight.jl_amd/libflightbatch.so
  out=$(FLIGHTBATCH_LIB=$ROOT/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 10 2> gpurun_out/ab_$tag.err) || { echo "$tag FAILED"; tail -3 gpurun_out/ab_$tag.err; continue; }
  echo "$out" | python -c "import json,sys; d=json.load(sys.stdin); print('%-12s %.4e aircraft-steps/s  kernel %.3f ms' % ('$tag', d['value'], d['roofline']['kernel_ms']))"
done
