#!/bin/bash
# GPU box: SQ counters of the fused rollout kernel (1M envs) per integrator.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_sq; rm -rf $OUT; mkdir -p $OUT
for I in rk45 dp5 rk4; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/$I -- python3 bench.py --workload ${WL:-config2} --steps 100 --warmup 100 --chunk 100 --launch fused --envs 1048576 --integrator $I --no-cpu-baseline > $OUT/$I.log 2>&1 || echo "$I failed"
  rocprofv3 --pmc SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INST_CYCLES_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${I}_b -- python3 bench.py --workload ${WL:-config2} --steps 100 --warmup 100 --chunk 100 --launch fused --envs 1048576 --integrator $I --no-cpu-baseline > $OUT/${I}_b.log 2>&1 || echo "$I b failed"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_sq/*/")):
    acc = collections.defaultdict(list); dur = {}
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rollout" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        ds = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_rollout" in r["Kernel_Name"]]
        if ds: dur = dict(avg_us=sum(ds) / len(ds) / 1e3, n=len(ds))
    print(d, dur, {k: sum(v) / len(v) for k, v in acc.items()})
PY
