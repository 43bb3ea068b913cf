source scripts/gpu_steps.sh
PA_DET_LANES=2 step 600 gpurun_out/r6t_tests.log python -m pytest tests/test_yolov5.py tests/test_chain.py -m gpu -q -x
tail -2 gpurun_out/r6t_tests.log
for r in 1 2; do for d in f32 emulated_f32; do for l in 1 2; do
PA_DET_LANES=$l step 100 gpurun_out/r6t_det_${d}_l${l}_$r.json python bench.py --workload detect --dtype $d
echo "detect $d lanes $l run $r: $(python -c "import json;d=json.loads(open('gpurun_out/r6t_det_${d}_l${l}_$r.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'])")"
done; done; done
for d in f32 emulated_f32; do for l in 1 2; do
PA_DET_LANES=$l step 200 gpurun_out/r6t_chain_${d}_l${l}.json python bench.py --workload chain --dtype $d --steps 12
echo "chain $d lanes $l: $(python -c "import json;d=json.loads(open('gpurun_out/r6t_chain_${d}_l${l}.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['chain']['stage_ms_per_clip_alone'])")"
done; done
