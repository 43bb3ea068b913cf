#!/bin/bash
# Hardware counters of the detection network's kernels (rocprofv3 --pmc passes, each with --kernel-trace only, as
# MI355X_MICROARCH.md prescribes): HBM bytes per launch (FETCH_SIZE x2 on gfx950 for wide reads + WRITE_SIZE), MFMA-pipe busy,
# LDS bank conflicts; per kernel and grid by scripts/pmc_kernels.py. A pass that fails or times out stops the script.
set -u
O=${1:-gpurun_out/r4_detect_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --workload detect --steps 2 --warmup 1"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1 && echo p1 done || { echo "pass p1 failed or timed out (see $O/p1.log)"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p2 -- $CMD > $O/p2.log 2>&1 && echo p2 done || { echo "pass p2 failed or timed out (see $O/p2.log)"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p3 -- $CMD > $O/p3.log 2>&1 && echo p3 done || { echo "pass p3 failed or timed out (see $O/p3.log)"; exit 1; }
for p in p1 p2 p3; do echo "== pass $p"; python3 scripts/pmc_kernels.py $O/$p "pa::"; done > $O/summary.txt 2>&1
wc -l $O/summary.txt
