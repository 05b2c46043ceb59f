#!/bin/bash
# Runs ON THE GPU BOX: second half of the configs[3] reproduction (profiles/r06_x2_repro.txt). Neither an idle GPU (x2_repro.sh) nor what the
# scratch rows hold in a fresh process slows the leg; what is left is round 5's bench.py FLOW: the x2 world created behind the headline world
# and the CPU legs, in the same process. (f) that flow, scratch rows left as allocated, under a kernel trace (launch-by-launch durations and
# gaps); (g) the same flow untraced, rows zeroed (the shipped behaviour).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/x2_repro
mkdir -p $OUT
cd $ROOT
R=$OUT/r06_x2_repro3.txt
summ() { python3 - "$1" <<PY
import json, sys
d = json.JSONDecoder().raw_decode(open(sys.argv[1]).read().strip().splitlines()[-1])[0]
e = d["extra"]
print("  headline ms_per_step %.3f | x2 kernel_ms %.3f  x2_lattice %.3f  vs_identical %.3f | fleet %.3f" % (d["ms_per_step"], e["x2"]["kernel_ms"], e["x2_lattice"]["kernel_ms"], e["x2_lattice"]["vs_identical_aircraft"], e["fleet"]["kernel_ms"]))
PY
}
echo "== (f) round 5's bench.py flow, scratch rows LEFT AS ALLOCATED, untraced" > $R
FLIGHTBATCH_SCRATCH_FILL=none python3 /tmp/bench_r05.py  # = git show 1270ce9:bench.py with ROOT pointing at the repository --gpus 1 --steps 20 --warmup 5 > $OUT/f.json 2> $OUT/f.err; summ $OUT/f.json >> $R
echo "== (g) the same flow, scratch rows zeroed at create (shipped)" >> $R
python3 /tmp/bench_r05.py  # = git show 1270ce9:bench.py with ROOT pointing at the repository --gpus 1 --steps 20 --warmup 5 > $OUT/g.json 2> $OUT/g.err; summ $OUT/g.json >> $R
echo "== (f') as (f) under rocprofv3 --kernel-trace: every k_step_duo<0, true, false> launch, start (ms since the first) and duration" >> $R
(cd /tmp && export TMPDIR=/tmp && FLIGHTBATCH_SCRATCH_FILL=none rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_f -- python3 $ROOT//tmp/bench_r05.py  # = git show 1270ce9:bench.py with ROOT pointing at the repository --gpus 1 --steps 20 --warmup 5 > $OUT/f2.json 2> $OUT/f2.err); summ $OUT/f2.json >> $R
python3 - >> $R <<PY
import csv, glob
f = glob.glob("$OUT/trace_f/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_step_duo" in r["Kernel_Name"] and "Lb1ELb0E" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%10.3f ms  +%8.3f ms  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size"))))
PY
echo done
