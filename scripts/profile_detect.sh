#!/bin/bash
# Measurement set of the detection stage (YOLOv5s layer table) for profiles/: bench line + rocprofv3 kernel stats + share-of-time summary.
set -u
# usage: scripts/profile_detect.sh [out dir] [extra bench flags, e.g. --dtype emulated_f32]
O=${1:-gpurun_out/r4_detect}
shift || true
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py --workload detect --steps 10 --warmup 2 "$@" > $O/bench_line.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o det -- python3 bench.py --workload detect --steps 3 --warmup 1 "$@" > $O/bench_line_under_rocprof.json 2> $O/rocprof.err
{
  echo "detection stage $*, 64 x 1080p frames per step (bench.py --workload detect --steps 3 --warmup 1 under rocprofv3 --kernel-trace --stats), share of the stage's kernel time"
  python3 - "$O/stats/det_kernel_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pa::" in r["Name"]]
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
    print(f'{int(r["TotalDurationNs"]) / tot * 100:5.1f} %  {int(r["Calls"]):6d} calls  {float(r["AverageNs"]) / 1e3:9.1f} us avg  {r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]}')
PY
} > $O/kernel_summary.txt
cat $O/bench_line.json | cut -c1-300; cat $O/kernel_summary.txt
