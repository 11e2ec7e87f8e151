#!/bin/bash
# GPU box: several library builds on the same box, alternating:  tools/ab_libs.sh "lib1 lib2 ..." "workloads"   ("-" = the product)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for rep in 1 2; do
  for lib in $1; do
    [ "$lib" = "-" ] && L="" || L=$lib
    for wl in ${2:-config2 config3}; do
      case $wl in config2) args="--steps 2000 --warmup 250";; config5) args="--workload config5 --steps 256 --warmup 128";; *) args="--workload $wl --steps 500 --warmup 250";; esac
      QENV_LIB=$L python3 bench.py $args --no-cpu-baseline --no-subconfigs --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$wl', round(d['value']/1e9,3), 'G env-steps/s')"
    done
  done
done
