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
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 0 and d["higher_is_better"] is True
    assert d["metric"].startswith("1080p frames/sec") and d["unit"] == "frames/s" and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["exchange"]["halo_rows_sent_rank0"] > 0, "two ranks of one clip must exchange halo features"
    assert d["config"]["workload"] and d["scaling"] in ("strong", "weak")
    # (rank 0 asserts that every record of the clip arrived and every log-probability is finite before it prints the line:
    # bench.py, "sanity: results are finite and complete" -- a non-finite label fails the child, not this parse)
