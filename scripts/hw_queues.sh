#!/bin/bash
# does raising the HIP runtime's hardware-queue count (GPU_MAX_HW_QUEUES, default 4) remove the
# stream-to-queue lottery of the two lanes? Headline shape, side measurements off; prints value + the calibration's rates.
# Usage (GPU box): bash scripts/hw_queues.sh > gpurun_out/r06_hw_queues.txt
set -o pipefail
run() {   # label, env assignment ("" for none), extra flags
    local label="$1" envs="$2"; shift 2
    for i in 1 2; do
        local out
        out=$(env $envs python bench.py --no-pcie --no-cpu-baseline --no-profile --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1)
        python - "$label" "$i" "$out" <<'PY'
import json, sys
label, i, line = sys.argv[1:4]
d = json.loads(line)
det = json.load(open("bench_details.json"))
cal = det["config"].get("lane_stream_calibration") or {}
rates = cal.get("rates", {})
spread = (max(rates.values()) / min(rates.values()) - 1) * 100 if rates else float("nan")
print(f"{label:44s} run {i}: value {d['value']:9.1f} frames/s  calibration rates {rates}  spread {spread:.1f} %", flush=True)
PY
    done
}
run "default queues, calibrated" "" --calibrate
run "default queues, no calibration" "" --no-calibrate
run "GPU_MAX_HW_QUEUES=8, calibrated" "GPU_MAX_HW_QUEUES=8" --calibrate
run "GPU_MAX_HW_QUEUES=8, no calibration" "GPU_MAX_HW_QUEUES=8" --no-calibrate
run "GPU_MAX_HW_QUEUES=16, no calibration" "GPU_MAX_HW_QUEUES=16" --no-calibrate
