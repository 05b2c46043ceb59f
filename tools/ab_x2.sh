#!/bin/bash
# A/B timing of library variants for the Cessna172Xv2 leg ON THE GPU BOX: tools/ab_x2.sh tag1 tag2 ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out
for tag in "$@"; do
  lib=flight.jl_amd/libflightbatch_$tag.so; [ "$tag" = main ] && lib=flight.jl_amd/libflightbatch.so
  out=$(FLIGHTBATCH_LIB=$ROOT/$lib timeout -k 10 200 python tools/bench_x2.py 50 2> gpurun_out/abx_$tag.err) || { echo "$tag FAILED"; tail -3 gpurun_out/abx_$tag.err; continue; }
  echo "$out" | python -c "import json,sys; d=json.load(sys.stdin); print('%-12s %.4e aircraft-steps/s  kernel %.3f ms  err %.2e' % ('$tag', d['value'], d['kernel_ms'], d['rel_err_vs_cpu']['max_scaled_error']))"
done
