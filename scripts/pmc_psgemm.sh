#!/bin/bash
# Hardware counters of the emulated-fp32 convolution kernel (psgemm.hip) beside the exact one on a few layer shapes
# (scripts/conv_layers_ab.py --only <substring>): SQ passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
#   usage: scripts/pmc_psgemm.sh [out dir] [layer substring]
set -u
O=${1:-gpurun_out/r6_psgemm_pmc}
L=${2:-256->512}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 scripts/conv_layers_ab.py --rounds 1 --only $L"
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1 && echo p1 done || { echo "pass p1 failed or timed out (see $O/p1.log)"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/p2 -- $CMD > $O/p2.log 2>&1 && echo p2 done || { echo "pass p2 failed or timed out (see $O/p2.log)"; exit 1; }
# HBM traffic (TCC; FETCH_SIZE costs 3 of the 4 slots, WRITE_SIZE 2: a pass each). FETCH_SIZE reads half the bytes of wide streaming reads on gfx950
# (MI355X_MICROARCH.md): double it before comparing; both are in KiB.
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p3 -- $CMD > $O/p3.log 2>&1 && echo p3 done || { echo "pass p3 failed or timed out (see $O/p3.log)"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p4 -- $CMD > $O/p4.log 2>&1 && echo p4 done || { echo "pass p4 failed or timed out (see $O/p4.log)"; exit 1; }
for p in p1 p2 p3 p4; do echo "== pass $p"; python3 scripts/pmc_kernels.py $O/$p "gemm"; done > $O/summary.txt 2>&1
wc -l $O/summary.txt
