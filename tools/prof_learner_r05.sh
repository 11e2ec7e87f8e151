#!/bin/bash
# GPU box: the learner's update epoch at config 5's horizon (tools/time_update.py: 65 536 envs x 128 steps = 8.4 M samples) under
# rocprofv3: kernel trace + stats, HBM-side bytes (FETCH_SIZE and WRITE_SIZE in separate passes, corrected as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE x 2 on gfx950) and the SQ counters of k_mlp_grad_v5.
#   tools/prof_learner_r05.sh TAG [ENV=VALUE ...]      (QENV_GRAD_ONE_LAUNCH=0: rounds 1-4's path, for the comparison)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r05}; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/learner_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/time_update.py > $OUT/trace.log 2>&1 || echo "trace failed"
cp "$(find $OUT/trace -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 tools/time_update.py > $OUT/$C.log 2>&1 || echo "$C failed"
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq_a -- python3 tools/time_update.py > $OUT/sq_a.log 2>&1 || echo "sq_a failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/sq_b -- python3 tools/time_update.py > $OUT/sq_b.log 2>&1 || echo "sq_b failed"
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
KEYS = ("k_mlp_grad", "k_mlp_bwd", "k_wgrad", "k_adam", "k_sum_")
res = {}
for d in ("FETCH_SIZE", "WRITE_SIZE", "sq_a", "sq_b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
            if any(x in k for x in KEYS):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{out}/{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
            if any(x in k for x in KEYS):
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in acc:
        e = res.setdefault(k, {"launches_per_pass": len(dur[k])})
        e.setdefault("avg_us", {})[d] = round(sum(dur[k]) / max(1, len(dur[k])) / 1e3, 1)
        for c, v in acc[k].items():
            e[c] = sum(v) / len(v)
# bytes per launch: FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1 KB? rocprofv3 reports them in KB (1024 B) on gfx9:
for k, e in res.items():
    if "FETCH_SIZE" in e: e["hbm_read_GB_per_launch"] = e["FETCH_SIZE"] * 1024 * 2 / 1e9    # gfx950: x 2 (the guide's correction)
    if "WRITE_SIZE" in e: e["hbm_write_GB_per_launch"] = e["WRITE_SIZE"] * 1024 / 1e9
print(json.dumps(res, indent=1))
open(out + "/summary.json", "w").write(json.dumps(res, indent=1))
PY
grep update_epoch $OUT/trace.log
head -12 $OUT/kernel_stats.csv | cut -c1-160
