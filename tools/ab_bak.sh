#!/bin/bash
# same-box alternating A/B of the launch-start copy of the control-law record in k_step_duo<KIN, true> (FB_X2_BAK, FB_X2_BAK_G_CS / _CU, csrc/c172_kernels.hpp):
#   python __graft_entry__.py --diagnostic-variant nobak -DFB_X2_BAK=0 ; python __graft_entry__.py --variant bak33 -DFB_X2_BAK_G_CS=33 -DFB_X2_BAK_G_CU=28
mkdir -p gpurun_out/ab_bak
for k in 8 50; do
for r in 1 2; do
  for v in base nobak bak33; do
    if [ $v = base ]; then unset FLIGHTBATCH_LIB; else export FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_$v.so; fi
    timeout -k 10 200 python3 tools/bench_x2.py $k --no-parity > gpurun_out/ab_bak/$v.$k.$r.txt 2>&1 || exit 1
    python3 - gpurun_out/ab_bak/$v.$k.$r.txt $v $r $k <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
print(f"{sys.argv[2]:6s} k={sys.argv[4]:2s} run {sys.argv[3]}: kernel median {d['kernel_ms']:.3f} ms (min {d['kernel_ms_min']:.3f}, max {d['kernel_ms_max']:.3f}), {d['value']:.4e} aircraft-steps/s")
PY
  done
done
done
