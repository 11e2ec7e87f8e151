#!/bin/bash
# GPU box: A/B of learner builds (k_mlp_grad_v5 variants) on ONE box, alternating (tools/time_update.py: the epoch at config 5's horizon).
#   tools/ab_learner.sh product g4pre g4bal ...   (names of build_variants/libqenv_NAME.so; "product" = the shipped library)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for n in "$@"; do
    if [ "$n" = product ]; then L=""; else L="build_variants/libqenv_$n.so"; fi
    r=$(QENV_LIB=$L python3 tools/time_update.py 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['update_epoch_s']*1e3)")
    echo "$n $r ms"
  done
done
