#!/bin/bash
# Runs ON THE GPU BOX: why did the driver's bench run time configs[3] at 15.09 ms per launch where 9.85 is the kernel's time?
# Per-launch HIP-event durations of the Xv2 leg (a) back to back, (b) behind 25 s of idle GPU with round 5's two warm-up launches,
# (c) the same with the new warm-up-until-stable, then (d) the driver's exact command on the new bench.py, and (e) a per-launch kernel trace.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/x2_repro
mkdir -p $OUT
cd $ROOT
python3 -c "import torch" 2>/dev/null
echo "== (a) no idle, 2 warm-up launches, 30 timed" > $OUT/r06_x2_repro.txt
python3 tools/bench_x2.py --no-parity --warm-cap 2 --blocks 30 >> $OUT/r06_x2_repro.txt 2>&1
echo "== (b) 25 s idle in front, 2 warm-up launches (round 5's bench.py), 30 timed" >> $OUT/r06_x2_repro.txt
python3 tools/bench_x2.py --no-parity --pre-sleep 25 --warm-cap 2 --blocks 30 >> $OUT/r06_x2_repro.txt 2>&1
echo "== (c) 25 s idle in front, warm-up until three launches agree within 1 % (cap 30), 12 timed" >> $OUT/r06_x2_repro.txt
python3 tools/bench_x2.py --no-parity --pre-sleep 25 >> $OUT/r06_x2_repro.txt 2>&1
echo "== (d) the driver's command" >> $OUT/r06_x2_repro.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
python3 - >> $OUT/r06_x2_repro.txt <<PY
import json
d = json.loads(open("$OUT/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("headline ms_per_step %.3f kernel_ms %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]), d["roofline"].get("kernel_ms_min"), d["roofline"].get("kernel_ms_max"))
for k in ("x2", "x2_lattice", "fleet"):
    e = d["extra"][k]
    print(k, "median %.3f min %.3f max %.3f" % (e["kernel_ms"], e["kernel_ms_min"], e["kernel_ms_max"]), "warm-up", e.get("warmup_ms_per_launch"), "timed", e["kernel_ms_per_launch"])
print("vs_identical_aircraft", d["extra"]["x2_lattice"]["vs_identical_aircraft"])
PY
echo "== (e) rocprofv3 --kernel-trace of (b): launch-by-launch durations" >> $OUT/r06_x2_repro.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_b -- python3 $ROOT/tools/bench_x2.py --no-parity --pre-sleep 25 --warm-cap 2 --blocks 30 > $OUT/trace_b.json 2> $OUT/trace_b.log)
python3 - >> $OUT/r06_x2_repro.txt <<PY
import csv, glob
f = glob.glob("$OUT/trace_b/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_step_duo" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%10.3f ms  +%8.3f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
echo done
