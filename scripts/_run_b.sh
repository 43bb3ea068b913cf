source scripts/gpu_steps.sh
step 300 gpurun_out/r6b_tests.log python -m pytest tests/test_psgemm.py -m gpu -x -q
tail -c 1500 gpurun_out/r6b_tests.log
grep -q "passed" gpurun_out/r6b_tests.log && ! grep -q "failed" gpurun_out/r6b_tests.log || exit 1
step 500 gpurun_out/r06_pgemm_split_layers.txt python scripts/conv_layers_ab.py --rounds 5
cat gpurun_out/r06_pgemm_split_layers.txt
