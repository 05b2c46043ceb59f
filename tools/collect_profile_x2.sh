#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): PMC passes (own runs, kernel-trace only) of the Cessna172Xv2 airborne stepper
# (k_step_duo<0, true>; with FLIGHTBATCH_DUO=0: k_step_air<0, true, false>) on bench.py's extra.x2 configuration -> gpurun_out/prof_x2_$TAG/${TAG}_x2_counters.json
# (copied into profiles/ by hand; bench.py quotes extra.x2.roofline_valu from it when its source hash matches the tree).
set -e
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_x2_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"; do
  tag=$(echo $set | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$tag -- python3 $ROOT/tools/profile_workload_x2.py 50 4 > $OUT/pmc_$tag.log 2>&1
  echo "x2 pmc $tag done"
done
python3 - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, "$ROOT")
import __graft_entry__ as ge
N, INNER = 1 << 19, 50
c = collections.defaultdict(list); dur = []
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_step_duo<0, true, false>" in r["Kernel_Name"] or "k_step_air<0, true, false, false>" in r["Kernel_Name"]:
            kname = r["Kernel_Name"]
            c[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
m = {k: sum(v) / len(v) for k, v in c.items()}
res = {"kernel": kname, "n": N, "inner": INNER, "ctl_ratio": 2, "tag": "$TAG", "source_hash": ge.source_hash(), "counters_mean_per_launch": m}
flops = 64 * (m["SQ_INSTS_VALU_ADD_F64"] + m["SQ_INSTS_VALU_MUL_F64"] + 2 * m["SQ_INSTS_VALU_FMA_F64"] + m["SQ_INSTS_VALU_TRANS_F64"])
res["fp64_flops_per_launch"] = flops
res["fp64_flops_per_aircraft_step"] = flops / (N * INNER)
res["valu_insts_per_aircraft_step"] = m["SQ_INSTS_VALU"] * 64 / (N * INNER)
res["k_step_mean_ns_under_pmc"] = sum(dur) / len(dur)
res["valu_busy"] = m["SQ_INSTS_VALU"] * 4 / 1024 / (2.4e9 * res["k_step_mean_ns_under_pmc"] * 1e-9)
# FETCH_SIZE / WRITE_SIZE in KB; gfx950: FETCH_SIZE reports half of wide coalesced reads (MI355X_MICROARCH.md, HBM section; the factor
# 2.07 measured on k_f_ode's known byte counts by tools/summarize_profile.py is applied here as 2.0), WRITE_SIZE is exact
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    res["hbm_read_bytes_per_launch"] = m["FETCH_SIZE"] * 1024 * 2.0
    res["hbm_write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024
    res["hbm_bytes_per_launch"] = res["hbm_read_bytes_per_launch"] + res["hbm_write_bytes_per_launch"]
    res["hbm_bytes_per_aircraft_step"] = res["hbm_bytes_per_launch"] / (N * INNER)
if "SQ_WAIT_ANY" in m: res["wait_fraction"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
json.dump(res, open("$OUT/${TAG}_x2_counters.json", "w"), indent=1)
print(json.dumps({k: res.get(k) for k in ("fp64_flops_per_aircraft_step", "valu_insts_per_aircraft_step", "valu_busy", "hbm_bytes_per_aircraft_step", "wait_fraction", "k_step_mean_ns_under_pmc")}))
PY
