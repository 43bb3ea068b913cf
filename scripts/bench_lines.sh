set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/lines; mkdir -p $O
cd $R
python3 bench.py > $O/r02_bench_line.json 2> $O/bench.err
echo "plain done"
python3 bench.py --dtype bf16 --frames 256 --height 720 --width 1280 > $O/r02_cfg2_bf16_bench_line.json 2> $O/bf16.err
echo "bf16 done"
python3 bench.py --clip-frames 8192 --steps 3 --warmup 1 > $O/r02_cfg3_8192_frames_one_gpu_bench_line.json 2> $O/cfg3.err
echo "cfg3 done"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo > $O/r02_cfg3_two_rank_gloo_one_gpu_bench_line.json 2> $O/cfg3_2r.err
echo "cfg3 2 ranks done"
python3 bench.py --workload mixed > $O/r02_cfg4_mixed_bench_line.json 2> $O/cfg4.err
echo "cfg4 done"
