#!/bin/bash
# Measurement set of the alternative temporal models for profiles/: bench lines + rocprofv3 kernel stats.
set -u
O=${1:-gpurun_out/r4_f4}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in rnn resformer; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 20 --warmup 3 > $O/${w}_bench_line.json 2> $O/${w}.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${w}_stats -o f4 -- python3 bench.py --workload $w --steps 5 --warmup 2 > $O/${w}_bench_line_under_rocprof.json 2> $O/${w}_rocprof.err
  python3 - "$O/${w}_stats/f4_kernel_stats.csv" > $O/${w}_kernel_summary.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pa::" in r["Name"]]
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
    print(f'{int(r["TotalDurationNs"]) / tot * 100:5.1f} %  {int(r["Calls"]):6d} calls  {float(r["AverageNs"]) / 1e3:9.1f} us avg  {r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]}')
PY
done
cat $O/*_bench_line.json; head -12 $O/*_kernel_summary.txt
