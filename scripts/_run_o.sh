source scripts/gpu_steps.sh
for r in 1 2 3; do
for v in 0 1; do
PA_DS_SIDE=$v step 200 gpurun_out/r6o_ds${v}_$r.json python bench.py --no-pcie --no-cpu-baseline --no-profile --steps 20 --warmup 3
done
done
for r in 1 2 3; do for v in 0 1; do echo "PA_DS_SIDE=$v run $r: $(python -c "import json;print(json.loads(open('gpurun_out/r6o_ds${v}_$r.json').read().strip().splitlines()[-1])['value'])")"; done; done
PA_DS_SIDE=1 step 400 gpurun_out/r6o_tests.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_contract.py -m gpu -q -x -k "f32 and not emulated"
tail -2 gpurun_out/r6o_tests.log
