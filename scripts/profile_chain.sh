#!/bin/bash
# Kernel share-of-time summary of the whole chain (bench.py --workload chain under rocprofv3 --kernel-trace --stats) for profiles/.
set -u
O=${1:-gpurun_out/chain}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o chain -- python3 bench.py --workload chain --steps 3 --warmup 1 > $O/bench_line_under_rocprof.json 2> $O/rocprof.err
{
  echo "the chain (bench.py --workload chain --steps 3 --warmup 1 under rocprofv3 --kernel-trace --stats): share of all kernel time, per symbol"
  python3 - "$O/stats/chain_kernel_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:45]:
    print(f'{int(r["TotalDurationNs"]) / tot * 100:5.1f} %  {int(r["Calls"]):6d} calls  {float(r["AverageNs"]) / 1e3:9.1f} us avg  {r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]}')
PY
} > $O/kernel_summary.txt
cat $O/kernel_summary.txt
rm -rf $O/stats
