source scripts/gpu_steps.sh
step 120 gpurun_out/r06_mfma_f32_mix.txt ./build/mfma_mix
tail -8 gpurun_out/r06_mfma_f32_mix.txt
for r in 1 2; do
step 200 gpurun_out/r6k_wino_packed_$r.txt python scripts/wino_times.py
PA_LIB_PATH=$PWD/build/libplayaid_wnscalar.so step 200 gpurun_out/r6k_wino_scalar_$r.txt python scripts/wino_times.py
done
paste -d'|' <(cut -c1-58,68-80 gpurun_out/r6k_wino_packed_1.txt) <(cut -c68-80 gpurun_out/r6k_wino_scalar_1.txt) <(cut -c68-80 gpurun_out/r6k_wino_packed_2.txt) <(cut -c68-80 gpurun_out/r6k_wino_scalar_2.txt)
PA_LIB_PATH=$PWD/build/libplayaid_wnscalar.so step 300 gpurun_out/r6k_wino_scalar_tests.log python -m pytest tests/test_wino.py -m gpu -q -x
tail -2 gpurun_out/r6k_wino_scalar_tests.log
