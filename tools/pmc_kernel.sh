#!/bin/bash
# GPU box: SQ counters of the kernels whose name contains PATTERN, over `python3 <script> <args>` (two passes: the counters do not
# fit one).  tools/pmc_kernel.sh NAME PATTERN script.py [args...]  ->  gpurun_out/pmc_NAME/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NAME=$1; PAT=$2; shift 2
OUT=gpurun_out/pmc_$NAME; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 "$@" > $OUT/a.log 2>&1 || echo "a failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d $OUT/b -- python3 "$@" > $OUT/b.log 2>&1 || echo "b failed"
PAT="$PAT" OUT="$OUT" python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, os
pat, out = os.environ["PAT"], os.environ["OUT"]
for d in sorted(glob.glob(out + "/*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in acc:
        ds = dur.get(k, [])
        print(d, k, dict(avg_us=round(sum(ds) / max(len(ds), 1) / 1e3, 2), n=len(ds)), {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()})
PY
