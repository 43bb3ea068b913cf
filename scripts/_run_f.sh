source scripts/gpu_steps.sh
step 900 gpurun_out/r6f_tests.log python -m pytest tests/test_yolov5.py tests/test_psgemm.py -m gpu -x -q -s
grep -a "passed\|failed\|boxes\|device-f64" gpurun_out/r6f_tests.log | tail -40
DTYPE=f32 step 200 gpurun_out/r6f_layers_f32.txt python scripts/detect_layer_times.py
DTYPE=emulated_f32 step 200 gpurun_out/r6f_layers_emu.txt python scripts/detect_layer_times.py
DTYPE=emulated_f32 PA_DET_EMU_S1=1 step 200 gpurun_out/r6f_layers_emu_s1.txt python scripts/detect_layer_times.py
python scripts/cmp_layers.py gpurun_out/r6f_layers_f32.txt gpurun_out/r6f_layers_emu.txt
tail -2 gpurun_out/r6f_layers_emu_s1.txt
step 200 gpurun_out/r6f_detect_f32.json python bench.py --workload detect
step 200 gpurun_out/r6f_detect_emu.json python bench.py --workload detect --dtype emulated_f32
cat gpurun_out/r6f_detect_f32.json gpurun_out/r6f_detect_emu.json
