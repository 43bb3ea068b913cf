#!/bin/bash
# One SQ pass over a short one-stream bench run: matrix-pipe busy cycles and clock PER KERNEL (scripts/pmc_kernels.py), to set
# the stem beside the convolution kernels. busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x 256 x GRBM_GUI_ACTIVE / 8).
set -u
O=${1:-gpurun_out/r4_stem_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 2 --warmup 1 --inner-repeat 2 --no-cpu-baseline --no-pcie --no-profile --no-pipeline"
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1 && echo p1 done || { echo "pass p1 failed or timed out (see $O/p1.log)"; exit 1; }
python3 scripts/pmc_kernels.py $O/p1 > $O/summary.txt 2>&1
python3 - $O/summary.txt <<'PY'
import re, sys
name = None
vals = {}
for line in open(sys.argv[1]):
    if not line.startswith("    "):
        if name and "GRBM_GUI_ACTIVE" in vals:
            gui = vals["GRBM_GUI_ACTIVE"] / 8
            print(f"{name[:60]:60s} dur_us={dur:7.1f} mfma_busy={vals.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * gui):.3f} clock_GHz={gui / (dur * 1e3):.2f}")
        name = line.split(" grid=")[0]
        dur = float(re.search(r"dur_us=([\d.]+)", line).group(1))
        vals = {}
    else:
        k, v = line.split()
        vals[k] = float(v)
PY
