#!/bin/bash
# same-box alternating A/B of the staggered start of one-step Cessna172Xv2 launches (FB_X2_STAGGER, csrc/c172_kernels.hpp):
#   python __graft_entry__.py --variant stg4 -DFB_X2_STAGGER=4 ; python __graft_entry__.py --variant stg9 -DFB_X2_STAGGER=9
mkdir -p gpurun_out/ab_stagger_x2
for r in 1 2; do
  for v in base stg4 stg9; do
    if [ $v = base ]; then unset FLIGHTBATCH_LIB; else export FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_$v.so; fi
    timeout -k 10 200 python3 tools/bench_x2.py 1 --no-parity --blocks 60 > gpurun_out/ab_stagger_x2/$v.$r.txt 2>&1 || exit 1
    python3 - gpurun_out/ab_stagger_x2/$v.$r.txt $v $r <<'PY'
import json, sys
import numpy as np
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
ms = np.array(d['kernel_ms_per_launch'])
print(f"{sys.argv[2]:8s} run {sys.argv[3]}: 524 288 Xv2, 1 step per launch, update every 2nd: launches without an update {np.median(ms[ms < np.median(ms)])*1e3:.1f} us, with {np.median(ms[ms > np.median(ms)])*1e3:.1f} us; {d['value']:.4e} aircraft-steps/s")
PY
    timeout -k 10 200 python3 examples/crosswind_landing.py 1048576 disperse device 2>&1 | grep "aircraft x" | sed "s/^/$v run $r: /"
  done
done
