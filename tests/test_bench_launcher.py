"""`python bench.py --gpus N` (no torch.distributed.run around it) starts its own ranks as a child process."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_launch_builds_the_torchrun_command_and_returns_its_code():
    bench = _bench()
    seen = {}

    def stub(cmd, env):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    rc = bench.self_launch(["--gpus", "4", "--steps", "3", "--warmup", "1"], 4, run=stub)
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_launches_children_before_any_gpu_call(monkeypatch):
    """--gpus 2 without WORLD_SIZE: main() must hand over to self_launch and exit with the child's code; reaching
    torch.cuda.set_device instead would raise here (no GPU) -- and would be the forbidden order on a GPU box."""
    bench = _bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    called = {}
    monkeypatch.setattr(bench, "self_launch", lambda argv, gpus: called.setdefault("a", (argv, gpus)) and 0)
    import torch

    monkeypatch.setattr(torch.cuda, "set_device", lambda *_: (_ for _ in ()).throw(AssertionError("GPU touched before the launch")))
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    else:
        raise AssertionError("main() returned instead of exiting with the child's code")
    assert called["a"] == (["--gpus", "2", "--steps", "1", "--warmup", "0"], 2)


import pytest  # noqa: E402


def test_compact_line_carries_the_judged_scalars_and_fits_the_drivers_tail():
    """The driver keeps 2000 characters of a run's tail: the last stdout line must hold the headline, the roofline (frac = the
    EXECUTED fraction), the CPU baseline and the chain's scalars in under 1800, however long the notes of the full record are."""
    import json

    bench = _bench()
    note = "x" * 4000
    full = {
        "metric": "1080p frames/sec end-to-end (decode->labels; 'decode' here = ingest of raw BGR frames already resident in HBM)",
        "value": 63445.1, "unit": "frames/s", "n_gpus": 1, "steps": 20, "warmup": 3, "ms_per_step": 1.0087, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: 64 x 1080x1920 BGR frames per GPU per clip, 2 fighters/frame, " + note, "parallelism": "single GPU",
                   "lanes": 2, "lane_stream_calibration": {"rates": [1.0] * 6, "note": note}},
        "roofline": {"kernel": "igemm_conv3x3", "bound": "mfma", "achieved": 85.8, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.5455,
                     "achieved_is": note, "algorithmic_tflops": 161.4, "algorithmic_frac": 1.0259, "executed_frac": 0.5455, "traffic": 71300000,
                     "algorithmic_bytes_per_launch": 39300000, "avg_launch_ms": 0.04623, "traffic_source": note},
        "cpu_baseline": {"value": 3.58, "unit": "frames/s", "cores": 128, "kind": "port", "sample": "40 synthetic 1080x1920 frames (78 windows, " + note,
                         "batched_value": 17.0},
        "chain_inclusive": {"value": 7181.0, "stage_ms_per_clip_alone": {"mjpeg_decode": 3.46, "detector_network": 5.58, "nms_and_label_repair": 0.2,
                                                                           "detector_crops_jpeg_runner_inputs_cnn_head": 1.47}, "method": note},
        "chain_inclusive_camera_like": {"value": 9000.0, "method": note},
        "decode_inclusive": {"value": 18000.0, "method": note}, "decode_inclusive_camera_like": {"error": "RuntimeError: " + note[:50]},
        "emulated_fp32": {"frames_per_s": 1.0, "detector_ms": 3.7, "chain_frames_per_s": 9600.0, "max_dlogp_vs_oracle": 2e-6, "note": note},
        "kernels": {f"k{i}": {"share": 0.1, "note": note} for i in range(20)},
    }
    line = bench.compact_line(full)
    assert len(line) < 1800 and "\n" not in line
    d = json.loads(line)
    assert d["value"] == 63445.1 and d["ms_per_step"] == 1.0087 and d["dtype"] == "f32" and d["metric"].startswith("1080p frames/sec")
    assert d["config"]["workload"].startswith("configs[1]: 64 x 1080x1920")
    assert d["roofline"]["frac"] == 0.5455 and d["roofline"]["algorithmic_frac"] == 1.0259 and d["roofline"]["traffic"] == 71300000
    assert d["cpu_baseline"]["cores"] == 128 and d["cpu_baseline"]["kind"] == "port"
    assert d["chain_frames_per_s"] == 7181.0 and d["chain_stage_ms"] == {"decode": 3.46, "detector": 5.58, "nms_repair": 0.2, "crops_cnn_head": 1.47}
    assert d["chain_camera_like_frames_per_s"] == 9000.0 and d["decode_inclusive_frames_per_s"] == 18000.0
    assert d["emulated_fp32"]["detector_ms"] == 3.7


@pytest.mark.gpu
def test_driver_scale_command_shape_on_two_ranks():
    """The command the driver's SCALE tier runs -- ``python3 bench.py --gpus N ...`` with no launcher around it -- as a FRESH child
    process (never an exec of a process that touched the GPU): bench.py starts its own two ranks, both on the box's one GPU
    over gloo (RCCL needs one GPU per rank), shards one 256-frame clip, exchanges the halo features and prints ONE JSON line. The
    nearest thing to an 8-GPU rehearsal this pool allows; no scaling claim -- none can be measured on one GPU."""
    import json
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--clip-frames", "256", "--steps", "1",
           "--warmup", "0", "--no-cpu-baseline", "--no-pcie", "--no-decode"]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert r.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 1800, "the LAST stdout line is the compact record the driver's tail keeps"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 0 and d["higher_is_better"] is True
    assert d["metric"].startswith("1080p frames/sec") and d["unit"] == "frames/s" and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["exchange"]["halo_rows_sent_rank0"] > 0, "two ranks of one clip must exchange halo features"
    assert d["config"]["workload"] and d["scaling"] in ("strong", "weak")
    # (rank 0 asserts that every record of the clip arrived and every log-probability is finite before it prints the line:
    # bench.py, "sanity: results are finite and complete" -- a non-finite label fails the child, not this parse)
