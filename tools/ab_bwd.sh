#!/bin/bash
# GPU box: one PPO update epoch (tools/time_update.py) under the ablation builds of k_mlp_bwd_v3 (build_variants/libqenv_bwdexp*.so,
# tools/build_variant.sh bwdexpN "-DQENV_BWD3_EXP=N" qenv_train.hip): where the backward kernel's time goes.
python tools/time_update.py | cut -c60-200
for e in "$@"; do echo "== QENV_BWD3_EXP=$e"; QENV_LIB=build_variants/libqenv_bwdexp$e.so python tools/time_update.py | cut -c60-200; done
