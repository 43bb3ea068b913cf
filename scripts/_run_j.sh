source scripts/gpu_steps.sh
step 400 gpurun_out/r6j_tests.log python -m pytest tests/test_clean_detections.py -m gpu -q -x
tail -3 gpurun_out/r6j_tests.log
step 300 gpurun_out/r06_yolov5_parity.txt python scripts/yolov5_parity.py
head -12 gpurun_out/r06_yolov5_parity.txt
step 500 gpurun_out/r6j_bench.json python bench.py
cat gpurun_out/r6j_bench.json; tail -3 gpurun_out/r6j_bench.json.err | cut -c1-600
cp bench_details.json gpurun_out/r6j_bench_details.json
