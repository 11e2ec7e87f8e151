#!/bin/bash
# GPU box: what one K = 20 region (one 20-step launch + torch.cuda.synchronize) costs under the runtime's wait modes.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B="python3 bench.py --workload ${WL:-config2} --steps 20 --warmup 5 --no-cpu-baseline --no-subconfigs --no-extras"
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value']/1e9, d['ms_per_step']*20*1e3, 'us/region', d['roofline']['kernel_us'], 'us kernel', d.get('sustained',{}).get('value',0)/1e9)"; }
$B | pick default
HSA_ENABLE_INTERRUPT=0 $B | pick hsa_interrupt_0
HIP_FORCE_DEV_KERNARG=1 $B | pick dev_kernarg
HSA_ENABLE_INTERRUPT=0 HIP_FORCE_DEV_KERNARG=1 $B | pick both
GPU_MAX_HW_QUEUES=2 $B | pick hwq2
