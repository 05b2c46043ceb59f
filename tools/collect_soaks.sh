#!/bin/bash
# Runs ON THE GPU BOX: the eight 10 000-step soaks against the CPU oracle that profiles/ holds per round (tag = $1).
set -e
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/soak_$TAG
mkdir -p $OUT
cd $ROOT
for kin in WA ECEF NED; do
  timeout -k 10 300 python3 tools/soak_duo.py 1024 65536 $kin > $OUT/${TAG}_soak_sv0_$kin.txt 2>&1; echo "sv0 $kin done"
  timeout -k 10 300 python3 tools/soak_x2.py 1024 $kin > $OUT/${TAG}_soak_x2_$kin.txt 2>&1; echo "x2 $kin done"
done
timeout -k 10 400 python3 tools/soak_duo.py 16384 1048576 > $OUT/${TAG}_soak_sv0_wide.txt 2>&1; echo "sv0 wide done"
timeout -k 10 400 python3 tools/soak_x2.py 8192 > $OUT/${TAG}_soak_x2_wide.txt 2>&1; echo "x2 wide done"
