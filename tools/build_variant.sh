#!/bin/bash
# A/B builds of libflightbatch: tools/build_variant.sh <tag> [extra hipcc flags...] -> flight.jl_amd/libflightbatch_<tag>.so
# (use with FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_<tag>.so; variant libraries are git-ignored and travel with gpurun)
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/flight.jl_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -freciprocal-math -fapprox-func \
  -fno-hip-fp32-correctly-rounded-divide-sqrt -mllvm -disable-machine-licm -fPIC -shared "$@" -o $ROOT/flight.jl_amd/libflightbatch_$TAG.so fb_capi.hip 2>&1 | grep -v "argument unused" || true
ls -la $ROOT/flight.jl_amd/libflightbatch_$TAG.so
