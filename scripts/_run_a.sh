source scripts/gpu_steps.sh
step 600 gpurun_out/r6a_tests.log python -m pytest tests/test_gpu_contract.py::test_rccl_world_size_one_runs_the_device_collectives tests/test_wino.py tests/test_rnn_detector.py tests/test_yolov5.py tests/test_bench_launcher.py -m gpu -k "not emulated" -x -q
step 240 gpurun_out/r06_yolov5_parity.txt python scripts/yolov5_parity.py
step 420 gpurun_out/r06_hw_queues.txt bash scripts/hw_queues.sh
step 300 gpurun_out/r6a_bench.json python bench.py
cp bench_details.json gpurun_out/r6a_bench_details.json
tail -c 400 gpurun_out/r6a_tests.log
