#!/bin/bash
# Here (after the gpurun call that ran tools/profile_round.sh / pmc_valu_count.sh / pmc_traffic.sh): copy the summaries that are to be
# judged from gpurun_out/ (scratch) into profiles/ (tracked).   tools/collect_profiles.sh r04
set -e
R=${1:-r04}; cd "$(dirname "$0")/.."
mkdir -p profiles/$R
for n in config2_default config2_driver config2_step config2_dp5 config2_f32 config3 config4 config5; do
  [ -f gpurun_out/prof_$R/$n.json ] && cp gpurun_out/prof_$R/$n.json profiles/$R/${n}_bench.json
  [ -f gpurun_out/prof_$R/${n}_kernel_stats.csv ] && cp gpurun_out/prof_$R/${n}_kernel_stats.csv profiles/$R/${n}_kernel_stats.csv
done
[ -f gpurun_out/pmc_valu_c2lean/summary.json ] && cp gpurun_out/pmc_valu_c2lean/summary.json profiles/${R}_pmc_valu_config2_lean.json
[ -f gpurun_out/pmc_valu_c3lean/summary.json ] && cp gpurun_out/pmc_valu_c3lean/summary.json profiles/${R}_pmc_valu_config3_lean.json
[ -f gpurun_out/pmc_$R/summary.txt ] && cp gpurun_out/pmc_$R/summary.txt profiles/${R}_pmc_traffic_summary.txt
ls profiles/$R profiles/${R}_* 2>/dev/null
