#!/bin/bash
# GPU box: the one-off timings DESIGN.md quotes besides bench.py (micro-benchmarks, A/B tools), collected into one directory.
cd "$GRAFT_REPO_ROOT"; R=${1:-r02}; OUT=gpurun_out/measure_$R; rm -rf $OUT; mkdir -p $OUT
run() { local name=$1; shift; echo "== $name: $*"; "$@" > $OUT/$name.txt 2>&1 || echo "$name failed"; tail -n 4 $OUT/$name.txt | cut -c1-400; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/ubench/valu_rate.hip && run ubench_valu_rate /tmp/valu_rate
QENV_PPO_OVERLAP=0 run ab_ppo_sequential python3 tests/tools/ab_ppo_overlap.py 65536 96 lean
QENV_PPO_OVERLAP=1 run ab_ppo_overlap python3 tests/tools/ab_ppo_overlap.py 65536 96 lean
run rollout_shards python3 tools/time_rollout_shards.py 65536
run rollout_host python3 tools/time_rollout_host.py 65536
run learner_small python3 tools/time_learner_small.py 256
run mlp_sizes_128 python3 tools/time_mlp_sizes.py 128
run mlp_sizes_256 python3 tools/time_mlp_sizes.py 256
run time_update python3 tools/time_update.py
