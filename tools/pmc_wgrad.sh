#!/bin/bash
# GPU box: SQ counters of k_wgrad / k_mlp_bwd at config-5 scale (tools/time_wgrad.py).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_wgrad; rm -rf $OUT; mkdir -p $OUT
ARGS="${1:-tools/time_wgrad.py}"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1 || echo "a failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1 || echo "b failed"
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_wgrad/*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            if "wgrad" in k or "bwd" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            if "wgrad" in k or "bwd" in k:
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in acc:
        print(d, k, "avg_us", round(sum(dur[k]) / max(1, len(dur[k])) / 1e3, 1), {c: round(sum(v) / len(v)) for c, v in acc[k].items()})
PY
