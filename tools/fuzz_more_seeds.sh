#!/bin/bash
# GPU box: further seeds of the env / controller / rollout fuzzers on the lean (parity-qualified) handle.
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/fuzz_lean_more; rm -rf $OUT; mkdir -p $OUT
for S in 1 2 3 4 5 6; do
  QENV_FUZZ_HANDLE=lean python3 tests/tools/fuzz_parity.py $S 300 > $OUT/fuzz_parity_lean_seed$S.jsonl 2>&1 || echo "fuzz_parity seed $S: mismatches"
  tail -n 1 $OUT/fuzz_parity_lean_seed$S.jsonl
done
for S in 1 2 3; do
  QENV_FUZZ_HANDLE=lean python3 tests/tools/fuzz_controllers.py $S 100 > $OUT/fuzz_controllers_lean_seed$S.jsonl 2>&1 || echo "fuzz_controllers seed $S: mismatches"
  tail -n 1 $OUT/fuzz_controllers_lean_seed$S.jsonl
  QENV_FUZZ_HANDLE=lean python3 tests/tools/fuzz_policy_rollout.py $S 40 > $OUT/fuzz_policy_rollout_lean_seed$S.jsonl 2>&1 || echo "fuzz_policy_rollout seed $S: mismatches"
  tail -n 1 $OUT/fuzz_policy_rollout_lean_seed$S.jsonl
done
