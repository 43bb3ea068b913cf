#!/bin/bash
# Hardware counters of the crop stage (crop_plan_kernel, crop_fused_kernel) and the stem at configs[1]: rocprofv3 --pmc passes,
# each with --kernel-trace only, over a short one-stream bench run; per-kernel means by scripts/pmc_kernels.py.
set -u
O=${1:-gpurun_out/r3_crop_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 2 --warmup 1 --inner-repeat 2 --no-cpu-baseline --no-pcie --no-profile --no-pipeline"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1 && echo p1 done || { echo "pass p1 failed or timed out: the remaining passes are not started (see $O/p1.log)"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p2 -- $CMD > $O/p2.log 2>&1 && echo p2 done || { echo "pass p2 failed or timed out: the remaining passes are not started (see $O/p2.log)"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p3 -- $CMD > $O/p3.log 2>&1 && echo p3 done || { echo "pass p3 failed or timed out: the remaining passes are not started (see $O/p3.log)"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p4 -- $CMD > $O/p4.log 2>&1 && echo p4 done || { echo "pass p4 failed or timed out: the remaining passes are not started (see $O/p4.log)"; exit 1; }
for p in p1 p2 p3 p4; do echo "== pass $p"; python3 scripts/pmc_kernels.py $O/$p crop; python3 scripts/pmc_kernels.py $O/$p stem_pool; done > $O/summary.txt 2>&1
wc -l $O/summary.txt
