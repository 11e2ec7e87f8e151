#!/bin/bash
# GPU box: VALU / SALU instructions per env step and wave of the fused rollout kernel of a bench.py run.
#   tools/pmc_valu_count.sh TAG [bench.py arguments]     (default: the headline, config 2 rk45 250-step launches)
#   PMC_KERNEL=k_step PMC_STEPS_PER_LAUNCH=1 tools/pmc_valu_count.sh TAG --workload config4     (another kernel of the run)
#   PMC_CMD="tools/rollout_driver.py lqr 2 rk45" tools/pmc_valu_count.sh TAG      (any other driver of 250-step launches)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-headline}; shift
OUT=gpurun_out/pmc_valu_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 ${PMC_CMD:-bench.py --workload ${WL:-config2} --steps 2000 --warmup 250 --min-seconds 0 --no-extras --no-cpu-baseline --no-subconfigs} "$@" > $OUT/a.log 2>&1 || echo failed
python3 - "$OUT" "${PMC_KERNEL:-k_rollout}" "${PMC_STEPS_PER_LAUNCH:-250}" <<'PY'
import csv, glob, collections, sys, json
out, kern, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
acc = collections.defaultdict(list); ds = []; names = set()
for f in glob.glob(out + "/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); names.add(r["Kernel_Name"][:90])
for f in glob.glob(out + "/a/**/*kernel_trace.csv", recursive=True):
    ds += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
m = {k: sum(v) / len(v) for k, v in acc.items()}
us = sum(ds) / len(ds) / 1e3
res = {"kernel": sorted(names), "launch_us": us, "launches": len(ds), "counters": m,
       "valu_per_wave_step": m["SQ_INSTS_VALU"] / m["SQ_WAVES"] / steps, "salu_per_wave_step": m["SQ_INSTS_SALU"] / m["SQ_WAVES"] / steps,
       "active_valu_cycles_per_inst": 4 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"],
       "wave_cycles_per_wave_step": 4 * m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"] / steps,
       "clock_ghz": m["GRBM_GUI_ACTIVE"] / 8 / (us * 1e3), "steps_per_launch": steps}
print(json.dumps(res))
open(out + "/summary.json", "w").write(json.dumps(res, indent=1))
PY
