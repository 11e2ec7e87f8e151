#!/bin/bash
# GPU box: the driver's K = 20 headline region (one 20-step launch + launch/sync latency) under host-runtime environment
# knobs that change how the host waits for / submits to the queue.  tools/ab_launch_env.sh > gpurun_out/ab_launch_env.txt
B="python bench.py --workload ${WL:-config2} --steps 20 --warmup 5 --no-subconfigs --no-extras --no-sustained --no-cpu-baseline"
run() { echo "== $1"; env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value']/1e9, d['ms_per_step']*20*1e3, 'us per region')"; }
run "QENV_NOP=1"
run "HSA_ENABLE_INTERRUPT=0"
run "ROC_ACTIVE_WAIT_TIMEOUT=500"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "HSA_ENABLE_INTERRUPT=0 HIP_FORCE_DEV_KERNARG=1"
run "AMD_DIRECT_DISPATCH=0"
run "GPU_MAX_HW_QUEUES=1"
