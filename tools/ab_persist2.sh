#!/bin/bash
# ON THE GPU BOX: base (the tree before the persistent-pairs restructuring: libflightbatch_base.so) against main with FLIGHTBATCH_PERSIST=0 / 1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
run() {  # $1 label, $2 lib, $3 persist
  echo "== $1"
  FLIGHTBATCH_LIB=$2 FLIGHTBATCH_PERSIST=$3 python3 tools/quickbench.py 2>&1 | grep "^k="
  for k in 1 50; do
    FLIGHTBATCH_LIB=$2 FLIGHTBATCH_PERSIST=$3 python3 tools/bench_x2.py $k --no-parity --blocks $((k == 50 ? 12 : 100)) 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   Cessna172Xv2 524 288, k=$k: median %.3f ms per launch (min %.3f max %.3f) -> %.3e aircraft-steps/s' % (d['kernel_ms'], d['kernel_ms_min'], d['kernel_ms_max'], d['value']))"
  done
}
for rep in 1 2; do
run "base (before persistent pairs)" $ROOT/flight.jl_amd/libflightbatch_base.so 0
run "main, one workgroup per 256 aircraft" $ROOT/flight.jl_amd/libflightbatch.so 0
run "main, persistent" $ROOT/flight.jl_amd/libflightbatch.so 1
done
