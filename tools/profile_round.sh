#!/bin/bash
# GPU box: the rocprofv3 kernel-trace summaries of the bench commands (one per workload) for profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r01}
OUT=gpurun_out/prof_$R; rm -rf $OUT; mkdir -p $OUT
run() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 bench.py "$@" --no-cpu-baseline --no-subconfigs --no-extras > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"
  local f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -6 "$f" | cut -c1-170 > $OUT/${name}_kernel_stats_top.csv && cp "$f" $OUT/${name}_kernel_stats.csv
}
run config2_default --workload config2 --steps 2000 --warmup 250
run config2_driver --workload config2 --steps 20 --warmup 5 --no-sustained
run config2_step --workload config2 --steps 2000 --warmup 200 --launch step
run config2_dp5 --workload config2 --steps 2000 --warmup 250 --integrator dp5
run config2_f32 --workload config2 --steps 2000 --warmup 250 --dtype f32
run config3 --workload config3 --steps 500 --warmup 250
run config3_driver --workload config3 --steps 20 --warmup 5 --no-sustained
run config4 --workload config4 --steps 500 --warmup 100
run config5 --workload config5 --steps 256 --warmup 128
for n in config2_default config2_driver config2_step config2_dp5 config2_f32 config3 config3_driver config4 config5; do echo "== $n"; cat $OUT/$n.json | cut -c1-1500; echo; head -4 $OUT/${n}_kernel_stats_top.csv; done
