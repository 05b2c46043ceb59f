#!/bin/bash
# A/B builds of libflightbatch: tools/build_variant.sh <tag> [extra hipcc flags...] -> flight.jl_amd/libflightbatch_<tag>.so
# (use with FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_<tag>.so; variant libraries are git-ignored and travel with gpurun).
# Flags and the spill-placement check are the shipped build's: both live in __graft_entry__.py. A failed compile or a failed check
# fails this script (and compile_library() has removed any earlier library of that name), so a stale variant can never be timed.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 "$ROOT/__graft_entry__.py" --variant "$@" 2>&1 | { grep -v "argument unused" || true; }
ls -la "$ROOT/flight.jl_amd/libflightbatch_$1.so"
