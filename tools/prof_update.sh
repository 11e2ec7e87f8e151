#!/bin/bash
# GPU box: rocprofv3 kernel-trace of one PPO update epoch at config-5 size (tools/time_update.py); prints the learner kernels.
# tools/prof_update.sh NAME [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NAME=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/prof_update_$NAME; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/time_update.py > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
echo "== $NAME $*"; grep -E "k_mlp_bwd|k_mlp_grad|k_wgrad|k_adam|k_sum" $OUT/kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
