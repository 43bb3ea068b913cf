source scripts/gpu_steps.sh
step 600 gpurun_out/r6h_tests.log python -m pytest tests/test_psgemm.py tests/test_yolov5.py -m gpu -x -q -k "emulated or psgemm"
grep -a "passed\|failed" gpurun_out/r6h_tests.log | tail -3
grep -q "passed" gpurun_out/r6h_tests.log && ! grep -q "failed" gpurun_out/r6h_tests.log || exit 1
DTYPE=emulated_f32 step 200 gpurun_out/r6h_layers_emu.txt python scripts/detect_layer_times.py
DTYPE=emulated_f32 PA_DET_EMU_S1=1 step 200 gpurun_out/r6h_layers_emu_s1.txt python scripts/detect_layer_times.py
DTYPE=emulated_f32 PA_DET_EMU_S1=1 PA_PS_RES128=0 step 200 gpurun_out/r6h_layers_emu_s1_r64.txt python scripts/detect_layer_times.py
python scripts/cmp_layers.py gpurun_out/r6h_layers_emu.txt gpurun_out/r6h_layers_emu_s1.txt | grep "k3 s1\|total\|up2"
python scripts/cmp_layers.py gpurun_out/r6h_layers_emu_s1_r64.txt gpurun_out/r6h_layers_emu_s1.txt | grep "k3 s1\|total"
step 300 gpurun_out/r6h_resnet_layers.txt python scripts/conv_layers_ab.py --rounds 5 --only "res "
cut -c1-150 gpurun_out/r6h_resnet_layers.txt
