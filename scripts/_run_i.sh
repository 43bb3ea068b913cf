source scripts/gpu_steps.sh
step 1100 gpurun_out/r6i_tests.log python -m pytest tests -m gpu -q -k "emulated"
grep -a "^FAILED\|^ERROR\|passed\|failed" gpurun_out/r6i_tests.log | tail -40
