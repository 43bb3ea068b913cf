#!/bin/bash
# Measurement set of the Motion-JPEG decode path for profiles/ (run on the GPU box from the repo root):
#   decode rate of one decoder at three qualities with / without restart markers, on the headline's clip and on the
#   camera-like one (scripts/mjpeg_rate.py, in-kernel cycle counters included); three decoders at once; groups per call;
#   rocprofv3 --kernel-trace --stats of the quality-95 run; verify passes needed per encoding (scripts/mjpeg_sync_probe.py).
set -u
O=${1:-gpurun_out/r3_mjpeg}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
  for q in 95 90 75; do
    timeout -k 10 300 python3 scripts/mjpeg_rate.py --variants none,rows1,blk8 --quality $q 2>&1 | grep -v amdgpu.ids
  done
  echo "--- camera-like clip (three bits of noise per sample instead of five)"
  for q in 95 75; do
    timeout -k 10 300 python3 scripts/mjpeg_rate.py --variants none,blk8 --quality $q --noise-mask 7 2>&1 | grep -v amdgpu.ids | grep -v "^   "
  done
} > $O/decode_rates.txt
{
  echo "--- decoders at once, each on a stream of its own, one group per call (PA_MJPEG_GROUPS=1)"
  for st in 2 3 4; do
    PA_MJPEG_GROUPS=1 timeout -k 10 300 python3 scripts/mjpeg_rate.py --variants none --streams $st --reps 6 2>&1 | grep "decoders on"
    PA_MJPEG_GROUPS=1 timeout -k 10 300 python3 scripts/mjpeg_rate.py --variants none --streams $st --reps 6 --noise-mask 7 2>&1 | grep "decoders on" | sed 's/^none /quiet/'
  done
  echo "--- one decoder, frame groups per call"
  for gr in 1 2 3 4; do
    echo -n "groups $gr: "; PA_MJPEG_GROUPS=$gr timeout -k 10 300 python3 scripts/mjpeg_rate.py --variants none 2>&1 | grep "frames/s"
  done
} > $O/concurrency.txt
PA_MJPEG_GROUPS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o mj -- python3 scripts/mjpeg_rate.py --variants none --reps 5 > $O/rate_under_rocprof.txt 2>&1
python3 scripts/mj_trace.py $O/stats/mj_kernel_trace.csv > $O/kernel_timeline_q95.txt
timeout -k 10 300 python3 scripts/mjpeg_sync_probe.py 2>&1 | grep -v amdgpu.ids > $O/sync_rounds.txt
ls $O
