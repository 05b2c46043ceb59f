#!/bin/bash
# Runs ON THE GPU BOX (through gpurun). Collects, for the bench configuration:
#   1. rocprofv3 --kernel-trace --stats of bench.py            -> gpurun_out/prof_$TAG/stats
#   2. PMC passes (own runs, kernel-trace only) of tools/profile_workload.py -> gpurun_out/prof_$TAG/pmc_*
# then tools/summarize_profile.py turns them into profiles/${TAG}_*.{csv,json} (copied back by hand).
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$tag -- python3 $ROOT/tools/profile_workload.py 50 3 > $OUT/pmc_$tag.log 2>&1
  echo "pmc $tag done"
done
python3 $ROOT/tools/summarize_profile.py $OUT $TAG
