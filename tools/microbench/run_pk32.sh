#!/bin/bash
# ON THE GPU BOX: builds and runs tools/microbench/pk32.hip -> gpurun_out/r05_pk32_microbench.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $ROOT/gpurun_out /tmp/pk32 && cd /tmp/pk32
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $ROOT/tools/microbench/pk32.hip -o pk32 && ./pk32 | tee $ROOT/gpurun_out/r05_pk32_microbench.txt
