#!/bin/bash
# One measurement set for profiles/: plain bench line, rocprofv3 --kernel-trace --stats of the same
# workload on one stream, and three separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ), each with
# --kernel-trace only. Run on the GPU box through gpurun:
#   gpurun -- 'bash scripts/profile_round.sh r01e' ;  ... 'bash scripts/profile_round.sh r01e_bf16 --dtype bf16 --frames 256 --height 720 --width 1280'
set -e -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" > $O/bench.log 2>&1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-pipeline "$@" > $O/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-profile --no-pipeline "$@" > $O/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-profile --no-pipeline "$@" > $O/pmc_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-profile --no-pipeline "$@" > $O/pmc_sq.log 2>&1
echo "sq done"
find $O -name "*.csv" | head -20
