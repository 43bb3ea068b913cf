source scripts/gpu_steps.sh
step 300 gpurun_out/r6n_tests.log python -m pytest tests/test_psgemm.py -m gpu -x -q
tail -2 gpurun_out/r6n_tests.log
step 200 gpurun_out/r6n_smoke.log python -c "import __graft_entry__ as g; g.smoke()"
tail -1 gpurun_out/r6n_smoke.log
step 900 gpurun_out/r6n_profile_round.log bash scripts/profile_round.sh r06 f32
tail -5 gpurun_out/r6n_profile_round.log; ls gpurun_out/r06
