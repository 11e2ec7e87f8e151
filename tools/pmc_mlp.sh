#!/bin/bash
# GPU box: SQ counters of k_mlp (config 4: 131072 envs, H=128, T=5).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_mlp; rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --workload config4 --steps 40 --warmup 20 --no-graph --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1 || echo "a failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1 || echo "b failed"
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_mlp/*/")):
    acc = collections.defaultdict(list); dur = {}
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_mlp" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        ds = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_mlp" in r["Kernel_Name"]]
        if ds: dur = dict(avg_us=sum(ds) / len(ds) / 1e3, n=len(ds))
    print(d, dur, {k: sum(v) / len(v) for k, v in acc.items()})
PY
