#!/bin/bash
# Runs ON THE GPU BOX: third part of the configs[3] reproduction — does a monitoring query (rocm-smi, as a driver-side GPU-busy sampler would issue)
# stall the stepping launches? The x2 leg with 150 timed launches (1.5 s), once alone and once with rocm-smi queried in a loop beside it.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/x2_repro
mkdir -p $OUT
cd $ROOT
R=$OUT/r06_x2_repro4.txt
show() { python3 -c "
import json,sys
import numpy as np
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=np.array(d['kernel_ms_per_launch'])
print('  median %.3f min %.3f max %.3f mean %.3f | launches above 1.02 x median: %d of %d:' % (np.median(a), a.min(), a.max(), a.mean(), (a>1.02*np.median(a)).sum(), a.size), [round(float(v),2) for v in a[a>1.02*np.median(a)]])"; }
echo "== 150 timed launches, nothing beside" > $R
python3 tools/bench_x2.py --no-parity --blocks 150 2>&1 | show >> $R
echo "== 150 timed launches, 'rocm-smi --showuse --showpower --showclocks' in a loop beside (every ~0.1 s)" >> $R
( for k in $(seq 1 40); do rocm-smi --showuse --showpower --showclocks > $OUT/smi_last.txt 2>&1; sleep 0.05; done ) &
SMI=$!
python3 tools/bench_x2.py --no-parity --blocks 150 2>&1 | show >> $R
wait $SMI
tail -25 $OUT/smi_last.txt >> $R
echo done
