#!/bin/bash
# GPU box: memory instructions and wait cycles of k_step in config 4 (per wave and launch).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_step_mem; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --workload config4 --steps 500 --warmup 500 --no-cpu-baseline > $OUT/a.log 2>&1 || echo failed
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_step_mem/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m["SQ_WAVES"]
print({k: round(v / w, 1) for k, v in m.items()})
PY
