#!/bin/bash
# One measurement set for profiles/: plain bench line, rocprofv3 --kernel-trace --stats of the same
# workload on one stream, and three separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ), each with
# --kernel-trace only, then the summaries (scripts/rocprof_families.py, pmc_traffic.py, pmc_sq.py).
# Run on the GPU box through gpurun:
#   gpurun -- 'bash scripts/profile_round.sh r02 f32'
#   gpurun -- 'bash scripts/profile_round.sh r02_cfg2_bf16 bf16 --dtype bf16 --frames 256 --height 720 --width 1280'
set -e -o pipefail
TAG=$1; DT=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" > $O/bench_line.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-pipeline "$@" > $O/bench_line_under_rocprof.json 2> $O/stats.err
echo "stats done"
PMC="--steps 5 --warmup 1 --inner-repeat 1 --no-cpu-baseline --no-pcie --no-profile --no-pipeline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $PMC "$@" > $O/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py $PMC "$@" > $O/pmc_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py $PMC "$@" > $O/pmc_sq.log 2>&1
echo "sq done"
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 $R/scripts/rocprof_families.py $(ls $O/stats/*/*kernel_trace.csv | head -1) $O/rocprof_family_summary.json $DT > $O/families.txt
python3 $R/scripts/pmc_traffic.py $O $O/traffic.json $DT $WORKLOAD   # WORKLOAD="frames height width" of a backbone batch when not a default shape
python3 $R/scripts/pmc_sq.py $O/pmc_sq $O/pmc_sq_conv3x3.json $DT
# raw traces are large: keep the summaries only
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
ls -la $O
