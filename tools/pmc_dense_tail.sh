#!/bin/bash
# GPU box: counters of k_dense_tail_v3 (tools/time_dense_tail.py): SQ (two passes), memory side (FETCH_SIZE / WRITE_SIZE in their own
# passes; FETCH_SIZE x 2 on gfx950) and the vector-memory pipe.  -> gpurun_out/pmc_dense_tail/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc_kernel.sh dense_tail k_dense_tail_v3 tools/time_dense_tail.py 1 > /dev/null 2>&1
OUT=gpurun_out/pmc_dense_tail
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 tools/time_dense_tail.py 1 > $OUT/$C.log 2>&1 || echo "$C failed"
done
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/c -- python3 tools/time_dense_tail.py 1 > $OUT/c.log 2>&1 || echo "c failed"
PAT=k_dense_tail_v3 OUT=$OUT python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, os
pat, out = os.environ["PAT"], os.environ["OUT"]
for d in sorted(glob.glob(out + "/*/")):
    acc = collections.defaultdict(list); dur = []
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"] and r.get("Grid_Size", "") != "":
                acc[(r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                dur.append((r.get("Grid_Size", ""), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    by = collections.defaultdict(list)
    for g, t in dur: by[g].append(t)
    print(d, {g: round(sum(v) / len(v) / 1e3, 1) for g, v in by.items()}, {f"{c}@{g}": round(sum(v) / len(v), 1) for (c, g), v in sorted(acc.items())})
PY
