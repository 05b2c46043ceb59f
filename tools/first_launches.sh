#!/bin/bash
# ON THE GPU BOX: kernel trace of tools/first_launches.py -> gpurun_out/first_launches.txt (copy to profiles/r04_first_launches.txt)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/first_launches
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/first_launches.py run > $OUT/run.log 2>&1
f=$(ls $OUT/*/*_kernel_trace.csv | head -1)
python3 $ROOT/tools/first_launches.py report $f > $ROOT/gpurun_out/first_launches.txt
tail -100 $ROOT/gpurun_out/first_launches.txt
