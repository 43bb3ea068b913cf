#!/bin/bash
# Hardware counters of the Winograd 3x3 kernel alone (scripts/wino_times.py's shapes): SQ passes with --kernel-trace only, as
# MI355X_MICROARCH.md prescribes. Per kernel and grid by scripts/pmc_kernels.py. A pass that fails or times out stops the script.
#   usage: scripts/pmc_wino.sh [out dir]      PA_WINO_ABL=<n> in the environment profiles an ablated form (timing experiments)
set -u
O=${1:-gpurun_out/r5_wino_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 scripts/wino_times.py"
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1 && echo p1 done || { echo "pass p1 failed or timed out (see $O/p1.log)"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/p2 -- $CMD > $O/p2.log 2>&1 && echo p2 done || { echo "pass p2 failed or timed out (see $O/p2.log)"; exit 1; }
for p in p1 p2; do echo "== pass $p"; python3 scripts/pmc_kernels.py $O/$p "wino"; done > $O/summary.txt 2>&1
wc -l $O/summary.txt
