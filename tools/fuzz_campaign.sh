#!/bin/bash
# GPU box: a long differential-fuzz campaign for the record (profiles/): SEEDS seeds x 300 env cases on the lean and the IEEE
# handle, 100 controller cases and 40 PPO-worker cases per seed on the lean handle; one summary line per run.
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/fuzz_campaign; rm -rf $OUT; mkdir -p $OUT
FIRST=${1:-10}; LAST=${2:-29}
for S in $(seq $FIRST $LAST); do
  for H in lean f64; do
    QENV_FUZZ_HANDLE=$H python3 tests/tools/fuzz_parity.py $S 300 2>&1 | tail -n 1 | sed "s/^/seed $S fuzz_parity /" >> $OUT/summary.jsonl
  done
  QENV_FUZZ_HANDLE=lean python3 tests/tools/fuzz_controllers.py $S 100 2>&1 | tail -n 1 | sed "s/^/seed $S fuzz_controllers /" >> $OUT/summary.jsonl
  QENV_FUZZ_HANDLE=lean python3 tests/tools/fuzz_policy_rollout.py $S 40 2>&1 | tail -n 1 | sed "s/^/seed $S fuzz_policy_rollout /" >> $OUT/summary.jsonl
  tail -n 4 $OUT/summary.jsonl | cut -c1-160
done
