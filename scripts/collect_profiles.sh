#!/bin/bash
# Copy the summaries of scripts/profile_round.sh (tags <round>, <round>_cfg2_bf16) from gpurun_out/ into profiles/ under
# their committed names.   usage: bash scripts/collect_profiles.sh r03
set -e
R=${1:-r03}
G=gpurun_out
for t in $R ${R}_cfg2_bf16; do
  p=profiles/$t
  b=$p; [ $t = $R ] && b=profiles/${R}_bench
  cp $G/$t/kernel_stats.csv ${b}_kernel_stats.csv
  grep "^{" $G/$t/bench_line.json | tail -1 > ${p}_bench_line.json
  grep "^{" $G/$t/bench_line_under_rocprof.json | tail -1 > ${p}_bench_line_under_rocprof.json
  cp $G/$t/rocprof_family_summary.json ${p}_rocprof_family_summary.json
  cp $G/$t/traffic.json ${p}_traffic.json
done
cp $G/$R/pmc_sq_conv3x3.json profiles/${R}_pmc_sq_conv3x3.json
cp $G/${R}_cfg2_bf16/pmc_sq_conv3x3.json profiles/${R}_cfg2_bf16_pmc_sq.json
ls profiles | grep "^$R" 
