#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $ROOT/gpurun_out /tmp/cnd && cd /tmp/cnd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $ROOT/tools/microbench/cnd.hip -o cnd && ./cnd | tee $ROOT/gpurun_out/r05_cnd_microbench.txt
