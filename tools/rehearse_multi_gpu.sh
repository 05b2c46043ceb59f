#!/bin/bash
# Runs ON A ONE-GPU BOX: the N = 2 code path of bench.py end to end — two ranks sharing GPU 0, gloo process group, collectives on host
# copies (FLIGHTBATCH_BENCH_REHEARSAL=1). The timings are meaningless (two ranks on one GPU); the point is that every line the driver's
# multi-GPU run executes — strong + weak passes, per-rank timing reduction, state gather, configs[3] on all ranks, CPU baseline and parity
# sample on rank 0 after the process group is gone — has run once before the 8-GPU node sees it.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT; mkdir -p gpurun_out
FLIGHTBATCH_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/rehearsal_n2.json 2> gpurun_out/rehearsal_n2.err
rc=$?
echo "rc=$rc"; tail -c 1500 gpurun_out/rehearsal_n2.json; tail -5 gpurun_out/rehearsal_n2.err
exit $rc
