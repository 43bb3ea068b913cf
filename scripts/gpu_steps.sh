#!/bin/bash
# Run GPU steps one after the other on a gpurun box; a step that TIMES OUT or is KILLED (rc 124 / 137 / 143) ends the call -- no
# further GPU step after a hang -- while an ordinary failure (a red test, rc 1) is recorded and the next step still runs.
#   source scripts/gpu_steps.sh; step <seconds> <log> <command...>
mkdir -p gpurun_out
step() {
    local limit="$1" log="$2"; shift 2
    echo "== $(date +%T) step (limit ${limit}s): $*" | tee -a gpurun_out/steps.log
    timeout -k 10 "$limit" "$@" > "$log" 2> "${log}.err"
    local rc=$?
    echo "== rc=$rc $(date +%T)" | tee -a gpurun_out/steps.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 143 ]; then
        echo "== step timed out or was killed: stopping the call here" | tee -a gpurun_out/steps.log
        exit $rc
    fi
    return 0
}
