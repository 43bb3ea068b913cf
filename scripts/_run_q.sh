source scripts/gpu_steps.sh
step 200 gpurun_out/r6q_probe.txt python scripts/psgemm_overhead_probe.py
cat gpurun_out/r6q_probe.txt
