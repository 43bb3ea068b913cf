source scripts/gpu_steps.sh
for r in 1 2; do for q in 4 8; do for d in f32 emulated_f32; do
GPU_MAX_HW_QUEUES=$q step 200 gpurun_out/r6u_chain_${d}_q${q}_$r.json python bench.py --workload chain --dtype $d --steps 12
echo "chain $d GPU_MAX_HW_QUEUES=$q run $r: $(python -c "import json;d=json.loads(open('gpurun_out/r6u_chain_${d}_q${q}_$r.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'])")"
done; done; done
