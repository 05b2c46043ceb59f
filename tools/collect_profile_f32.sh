#!/bin/bash
# PMC passes for the fp32 stepper (fbf::k_step_f32) on the bench configuration; summary printed as JSON.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_f32pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_WAVES"; do
  tag=$(echo $set | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$tag -- python3 $ROOT/tools/profile_workload.py 50 3 f32 > $OUT/pmc_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
c = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_step_f32" in r["Kernel_Name"]:
            c[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in c.items()}
N = (1 << 20) * 50
m["valu_insts_per_aircraft_step"] = m["SQ_INSTS_VALU"] * 64 / N
m["fp32_flops_per_aircraft_step"] = 64 * (m["SQ_INSTS_VALU_ADD_F32"] + m["SQ_INSTS_VALU_MUL_F32"] + 2 * m["SQ_INSTS_VALU_FMA_F32"] + m["SQ_INSTS_VALU_TRANS_F32"]) / N
m["valu_active_fraction"] = m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"]
m["wait_fraction"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if "SQ_WAIT_ANY" in m else None
json.dump(m, open("$OUT/r01_f32_counters.json", "w"), indent=1)
print(json.dumps({k: m[k] for k in ("valu_insts_per_aircraft_step", "fp32_flops_per_aircraft_step", "valu_active_fraction", "wait_fraction")}))
PY
