#!/bin/bash
# same-box alternating A/B of the first-stage actuator position (FB_ACT_STAGE_FORM, c172_kernels.hpp): python __graft_entry__.py --variant inexact0 -DFB_ACT_STAGE_FORM=0
mkdir -p gpurun_out/ab_stage0
for r in 1 2 3; do
  for v in base inexact0; do
    if [ $v = base ]; then unset FLIGHTBATCH_LIB; else export FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_$v.so; fi
    timeout -k 10 200 python3 tools/bench_x2.py --no-parity > gpurun_out/ab_stage0/$v.$r.txt 2>&1 || exit 1
    python3 - gpurun_out/ab_stage0/$v.$r.txt $v $r <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
print(f"{sys.argv[2]:9s} run {sys.argv[3]}: kernel median {d['kernel_ms']:.3f} ms (min {d['kernel_ms_min']:.3f}, max {d['kernel_ms_max']:.3f}), {d['value']:.4e} aircraft-steps/s")
PY
  done
done
