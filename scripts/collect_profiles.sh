#!/bin/bash
# Copy the summaries of scripts/bench_lines.sh and scripts/profile_round.sh (tags r02, r02_cfg2_bf16) from
# gpurun_out/ into profiles/ under their committed names.
set -e
G=gpurun_out
for t in r02 r02_cfg2_bf16; do
  p=profiles/$t
  b=$p; [ $t = r02 ] && b=profiles/r02_bench
  cp $G/$t/kernel_stats.csv ${b}_kernel_stats.csv
  grep "^{" $G/$t/bench_line_under_rocprof.json | tail -1 > ${p}_bench_line_under_rocprof.json
  cp $G/$t/rocprof_family_summary.json ${p}_rocprof_family_summary.json
  cp $G/$t/traffic.json ${p}_traffic.json
done
cp $G/r02/pmc_sq_conv3x3.json profiles/r02_pmc_sq_conv3x3.json
cp $G/r02_cfg2_bf16/pmc_sq_conv3x3.json profiles/r02_cfg2_bf16_pmc_sq.json
for f in r02_bench_line r02_cfg2_bf16_bench_line r02_cfg3_8192_frames_one_gpu_bench_line r02_cfg3_two_rank_gloo_one_gpu_bench_line r02_cfg4_mixed_bench_line; do
  grep "^{" $G/lines/$f.json | tail -1 > profiles/$f.json
done
