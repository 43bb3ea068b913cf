source scripts/gpu_steps.sh
PA_PS_STAGES=2 step 300 gpurun_out/r6m_tests.log python -m pytest tests/test_psgemm.py -m gpu -x -q
tail -2 gpurun_out/r6m_tests.log
for r in 1 2; do
step 200 gpurun_out/r6m_chain_emu_$r.json python bench.py --workload chain --dtype emulated_f32 --steps 12
PA_PS_STAGES=2 step 200 gpurun_out/r6m_chain_emu_s2_$r.json python bench.py --workload chain --dtype emulated_f32 --steps 12
done
step 200 gpurun_out/r6m_chain_f32.json python bench.py --workload chain --steps 12
step 100 gpurun_out/r6m_det_emu.json python bench.py --workload detect --dtype emulated_f32
PA_PS_STAGES=2 step 100 gpurun_out/r6m_det_emu_s2.json python bench.py --workload detect --dtype emulated_f32
python - <<'PY'
import json
for f in ("chain_f32","chain_emu_1","chain_emu_s2_1","chain_emu_2","chain_emu_s2_2","det_emu","det_emu_s2"):
    d=json.loads(open(f"gpurun_out/r6m_{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("chain",{}).get("stage_ms_per_clip_alone"))
PY
