#!/bin/bash
# GPU box: the parity record of a round (profiles/parity_rNN.txt and the fuzz logs): every handle against the reference-
# generated fixtures, the full-size config 2 / 3 slices, and the differential fuzzers on the IEEE-fp64 and the lean handle.
cd "$GRAFT_REPO_ROOT"; R=${1:-r02}; OUT=gpurun_out/parity_$R; rm -rf $OUT; mkdir -p $OUT
python3 tools/calibrate_parity.py f32,mix,lean,f64 rk45 > $OUT/calibrate_rk45.txt 2>&1 || echo "calibrate failed"
python3 tools/calibrate_parity.py lean,f64 dp5,rk4 > $OUT/calibrate_fixed_step.txt 2>&1 || echo "calibrate(fixed) failed"
python3 tests/tools/config2_parity.py > $OUT/config2_parity.txt 2>&1 || echo "config2 failed"
python3 tests/tools/config3_parity.py > $OUT/config3_parity.txt 2>&1 || echo "config3 failed"
for H in f64 lean; do
  for T in fuzz_parity:300 fuzz_controllers:100 fuzz_policy_rollout:40 fuzz_outer_loops:40; do
    t=${T%%:*}; n=${T##*:}
    QENV_FUZZ_HANDLE=$H python3 tests/tools/$t.py 0 $n > $OUT/${t}_${H}_seed0.jsonl 2>&1 || echo "$t $H: mismatches"
    tail -n 1 $OUT/${t}_${H}_seed0.jsonl
  done
done
for H in f32 mix; do
  QENV_FUZZ_HANDLE=$H python3 tests/tools/fuzz_parity_f32.py 0 300 0.01 > $OUT/fuzz_parity_f32_${H}_seed0.jsonl 2>&1 || echo "fuzz_parity_f32 $H: violations"
  tail -n 1 $OUT/fuzz_parity_f32_${H}_seed0.jsonl
done
for t in fuzz_sensor fuzz_robust fuzz_mlp fuzz_gae fuzz_learner fuzz_visual_reward fuzz_quad_dropin; do
  python3 tests/tools/$t.py 0 40 > $OUT/${t}_seed0.jsonl 2>&1 || echo "$t: mismatches"
  tail -n 1 $OUT/${t}_seed0.jsonl
done
