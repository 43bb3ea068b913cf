set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_final; mkdir -p $O; cd $R
timeout -k 10 600 python3 bench.py > $O/bench_line.json 2> $O/bench.err || exit 1
echo plain done
timeout -k 10 300 python3 bench.py --lanes 1 --no-cpu-baseline --no-pcie --no-pipeline > $O/bench_line_lanes1.json 2> $O/lanes1.err || exit 1
echo lanes1 done
bash scripts/profile_detect.sh $O/detect > $O/detect.log 2>&1 || exit 1
echo detect done
timeout -k 10 200 python3 scripts/detect_layer_times.py > $O/detect_layer_times.txt 2>&1 || exit 1
timeout -k 10 200 python3 scripts/layer_times.py > $O/resnet_layer_times.txt 2>&1 || exit 1
timeout -k 10 300 python3 scripts/yolov5_parity.py > $O/yolov5_parity.txt 2>&1 || exit 1
echo parity done
timeout -k 10 300 python3 bench.py --dtype bf16 --frames 256 --height 720 --width 1280 > $O/cfg2_bf16_bench_line.json 2> $O/bf16.err || exit 1
timeout -k 10 300 python3 bench.py --clip-frames 8192 --steps 3 --warmup 1 > $O/cfg3_8192_frames_one_gpu_bench_line.json 2> $O/cfg3.err || exit 1
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo > $O/cfg3_two_rank_gloo_one_gpu_bench_line.json 2> $O/cfg3_2r.err || exit 1
timeout -k 10 300 python3 bench.py --workload mixed > $O/cfg4_mixed_bench_line.json 2> $O/cfg4.err || exit 1
echo lines done
