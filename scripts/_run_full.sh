source scripts/gpu_steps.sh
step 1150 gpurun_out/r6s_gpu_tests.log python -m pytest tests -m gpu -q
tail -c 1500 gpurun_out/r6s_gpu_tests.log
