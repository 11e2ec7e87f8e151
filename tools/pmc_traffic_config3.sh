#!/bin/bash
# GPU box: HBM-side bytes per launch of the headline kernel (config 3, k_rollout<lean, rk45, lqr>, the driver's 20-step launches and
# 250-step launches): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_traffic_config3; rm -rf $OUT; mkdir -p $OUT
for K in 20 250; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${C}_K$K -- python3 bench.py --workload config3 --steps $K --warmup $K --no-extras --no-subconfigs --no-cpu-baseline --no-sustained --min-seconds 0.05 > $OUT/${C}_K$K.log 2>&1 || echo "$C K$K failed"
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]; res = {}
for K in (20, 250):
    e = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(f"{out}/{C}_K{K}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_rollout" in r["Kernel_Name"] and r["Counter_Name"] == C:
                    vals.append(float(r["Counter_Value"]))
        vals = vals[len(vals) // 4:]   # steady state
        e[C + "_KB"] = sum(vals) / max(1, len(vals)); e["launches"] = len(vals)
    e["hbm_bytes_per_launch"] = e["FETCH_SIZE_KB"] * 1024 * 2 + e["WRITE_SIZE_KB"] * 1024
    e["algorithmic_bytes_per_launch"] = 224 * 262144 * K
    e["rows_once_bytes"] = 262144 * 8 * (13 + 13 + 40 + 40)   # fp64 state rows read + written, ~40 bookkeeping rows each way
    res[f"K{K}"] = e
print(json.dumps(res, indent=1))
open(out + "/summary.json", "w").write(json.dumps(res, indent=1))
PY
