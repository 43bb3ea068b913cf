source scripts/gpu_steps.sh
step 300 gpurun_out/r6e_tests.log python -m pytest tests/test_psgemm.py -m gpu -x -q
tail -c 600 gpurun_out/r6e_tests.log
grep -q "passed" gpurun_out/r6e_tests.log && ! grep -q "failed" gpurun_out/r6e_tests.log || exit 1
step 500 gpurun_out/r06_pgemm_split_layers_v2.txt python scripts/conv_layers_ab.py --rounds 5
cut -c1-150 gpurun_out/r06_pgemm_split_layers_v2.txt
