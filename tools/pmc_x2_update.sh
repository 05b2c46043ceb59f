#!/bin/bash
# ON THE GPU BOX: memory / fetch latency counters of the Cessna172Xv2 stepper with a control update after EVERY step (Δt = dt: the launch is
# dominated by the updates), for the shipped library and the two one-half-only diagnostic variants -> gpurun_out/pmc_x2_update.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_x2_update
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > $OUT/wl.py <<PY
import os, sys
import numpy as np
sys.path.insert(0, os.path.join("$ROOT", "flight.jl_amd")); sys.path.insert(0, "$ROOT")
import flightbatch as fb
n = 1 << 19
w = fb.Cessna172Xv2World(n)
w.set_params(wind_ned=(1.0, 0.5, 0.0))
sim = fb.Simulation(w, dt=0.01, Δt=0.01, save_on=False, steps_per_launch=50)
fb.init(sim, fb.TrimParameters())
w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 2.0
w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = float(np.deg2rad(30.0))
for _ in range(3): fb.step(sim, 0.5)
w.sync()
PY
for t in main skiplat skiplon; do
  lib=$ROOT/flight.jl_amd/libflightbatch_$t.so; [ $t = main ] && lib=$ROOT/flight.jl_amd/libflightbatch.so
  export FLIGHTBATCH_LIB=$lib
  for set in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $set | cut -d" " -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${t}_$tag -- python3 $OUT/wl.py > $OUT/${t}_$tag.log 2>&1
  done
done
python3 - <<PY > $ROOT/gpurun_out/pmc_x2_update.txt
import csv, glob, collections
for t in ("main", "skiplat", "skiplon"):
    c = collections.defaultdict(list); d = []
    for f in glob.glob("$OUT/%s_*/*/*_counter_collection.csv" % t):
        for r in csv.DictReader(open(f)):
            if "k_step_duo" in r["Kernel_Name"]:
                c[r["Counter_Name"]].append(float(r["Counter_Value"])); d.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    m = {k: sum(v) / len(v) for k, v in c.items()}
    print("%s: kernel %.3f ms under PMC" % (t, sum(d) / max(len(d), 1) / 1e6))
    for k, v in sorted(m.items()): print("   %-22s %.4e" % (k, v))
    if m.get("SQ_INSTS_VMEM_RD"): print("   mean VMEM latency ~ LEVEL_VMEM / (VMEM_RD + VMEM_WR) = %.0f cycles; SMEM %.0f; IFETCH %.0f" % (
        m["SQ_INST_LEVEL_VMEM"] / (m["SQ_INSTS_VMEM_RD"] + m["SQ_INSTS_VMEM_WR"]), m["SQ_INST_LEVEL_SMEM"] / max(m["SQ_INSTS_SMEM"], 1), m["SQ_IFETCH_LEVEL"] / max(m["SQ_IFETCH"], 1)))
PY
cat $ROOT/gpurun_out/pmc_x2_update.txt
