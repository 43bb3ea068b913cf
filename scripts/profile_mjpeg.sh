#!/bin/bash
# Measurement set of the Motion-JPEG decode path for profiles/ (run on the GPU box from the repo root):
#   decode rate at three qualities with / without restart markers (scripts/mjpeg_rate.py, in-kernel cycle counters included),
#   rocprofv3 --kernel-trace --stats of the quality-95 run, verify passes needed per encoding (scripts/mjpeg_sync_probe.py).
set -u
O=${1:-gpurun_out/r3_mjpeg}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for q in 95 90 75; do
  timeout -k 10 300 python3 scripts/mjpeg_rate.py --variants none,rows1,blk8 --quality $q 2>&1 | grep -v amdgpu.ids
done > $O/decode_rates.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o mj -- python3 scripts/mjpeg_rate.py --variants none --reps 5 > $O/rate_under_rocprof.txt 2>&1
python3 scripts/mj_trace.py $O/stats/mj_kernel_trace.csv > $O/kernel_timeline_q95.txt
timeout -k 10 300 python3 scripts/mjpeg_sync_probe.py 2>&1 | grep -v amdgpu.ids > $O/sync_rounds.txt
ls $O
