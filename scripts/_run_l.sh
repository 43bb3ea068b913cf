source scripts/gpu_steps.sh
step 600 gpurun_out/r6l_tests.log python -m pytest tests/test_yolov5.py -m gpu -x -q -k "emulated" -s
grep -a "passed\|failed\|emulated_f32.*boxes" gpurun_out/r6l_tests.log | tail -14
grep -q "passed" gpurun_out/r6l_tests.log && ! grep -q "failed" gpurun_out/r6l_tests.log || exit 1
DTYPE=emulated_f32 step 200 gpurun_out/r6l_layers_emu.txt python scripts/detect_layer_times.py
DTYPE=emulated_f32 PA_DET_EMU_STEM=0 step 200 gpurun_out/r6l_layers_emu_nostem.txt python scripts/detect_layer_times.py
python scripts/cmp_layers.py gpurun_out/r6l_layers_emu_nostem.txt gpurun_out/r6l_layers_emu.txt | head -4; tail -1 gpurun_out/r6l_layers_emu.txt
