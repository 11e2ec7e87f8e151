#!/bin/bash
# GPU box: k_step launch time with and without the tile loop (QENV_STEP_GRID_LANES / QENV_STEP_TILES) at three batch sizes, both
# row types, then the step / rollout / in-band tests (tests/tools/time_step_one.py).
set -e
for n in 131072 262144 524288; do
  python tests/tools/time_step_one.py $n
  QENV_STEP_GRID_LANES=100000000 python tests/tools/time_step_one.py $n
done
python tests/tools/time_step_one.py 131072 f32
QENV_STEP_GRID_LANES=100000000 python tests/tools/time_step_one.py 131072 f32
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=line -x -k "fused or inband or branch or step or rollout" > gpurun_out/r3d_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3d_tests.log; tail -5 gpurun_out/r3d_tests.log
