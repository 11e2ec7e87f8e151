#!/bin/bash
# GPU box: HBM traffic of the dominant kernels from rocprofv3 PMC counters, one counter per pass
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# plus the calibration passes on a known byte count in the same access shape.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_${1:-r01}
rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib_$C -- python3 tools/calib_copy.py > $OUT/calib_$C.log 2>&1 || echo "calib $C failed"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/step_$C -- python3 bench.py --workload ${WL:-config2} --steps 200 --warmup 100 --launch step --no-graph --no-extras --no-subconfigs --no-cpu-baseline --min-seconds 0 > $OUT/step_$C.log 2>&1 || echo "step $C failed"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/fused_$C -- python3 bench.py --workload ${WL:-config2} --steps 200 --warmup 100 --launch fused --no-extras --no-subconfigs --no-cpu-baseline --min-seconds 0 > $OUT/fused_$C.log 2>&1 || echo "fused $C failed"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/big_$C -- python3 bench.py --workload ${WL:-config2} --steps 20 --warmup 10 --envs 16777216 --launch step --no-graph --no-extras --no-subconfigs --no-cpu-baseline --min-seconds 0 > $OUT/big_$C.log 2>&1 || echo "big $C failed"
done
find $OUT -name "*.csv" | head -40
python3 tools/parse_pmc.py $OUT | tee $OUT/summary.txt
