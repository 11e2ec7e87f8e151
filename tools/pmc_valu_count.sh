#!/bin/bash
# GPU box: VALU instructions per env step and wave of the headline kernel (config 2, rk45, fused 250-step launches).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_valu; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --steps 2000 --warmup 250 --no-extras --no-cpu-baseline > $OUT/a.log 2>&1 || echo failed
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list); ds = []
for f in glob.glob("gpurun_out/pmc_valu/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_rollout" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_valu/a/**/*kernel_trace.csv", recursive=True):
    ds += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_rollout" in r["Kernel_Name"]]
m = {k: sum(v) / len(v) for k, v in acc.items()}
steps = 250
print("launch_us", sum(ds) / len(ds) / 1e3, m)
print("VALU instructions per wave and env step:", m["SQ_INSTS_VALU"] / m["SQ_WAVES"] / steps,
      " SALU:", m["SQ_INSTS_SALU"] / m["SQ_WAVES"] / steps, " clock GHz:", m["GRBM_GUI_ACTIVE"] / 8 / (sum(ds) / len(ds)))
PY
