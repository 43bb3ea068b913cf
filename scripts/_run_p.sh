source scripts/gpu_steps.sh
step 600 gpurun_out/r6p_tests.log python -m pytest tests/test_resformer_detector.py tests/test_rnn_detector.py -m gpu -q -x
tail -3 gpurun_out/r6p_tests.log
for w in resformer rnn; do for d in f32 emulated_f32; do
step 200 gpurun_out/r6p_${w}_$d.json python bench.py --workload $w --dtype $d
echo "$w $d: $(python -c "import json;d=json.loads(open('gpurun_out/r6p_${w}_$d.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'])")"
done; done
