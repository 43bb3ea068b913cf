"""PCIe-inclusive rate of the headline workload: frames start in pinned
host memory, each step uploads its 64 x 1080p frames (398 MB) and boxes, then runs the path.
Serial (copy, then compute) and double-buffered (copy of step k+1 on a side stream under the
compute of step k). Not the bench value: bench.py times with frames resident in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from playaid_core_amd import synth
from playaid_core_amd.engine import Engine

n, h, w = 64, 1080, 1920
sd = synth.make_state_dict(seed=1234)
eng = Engine(sd, max_batch_frames=n, max_clip_frames=64)
host = torch.from_numpy(synth.make_frames(8, h, w)).repeat(8, 1, 1, 1).contiguous().pin_memory()
boxes = torch.from_numpy(synth.make_boxes(n, h, w)).cuda()
dev = [torch.empty_like(host, device="cuda") for _ in range(2)]
rec = eng.alloc_records(n - 1)
K = 10
for _ in range(2):
    dev[0].copy_(host, non_blocking=True)
    eng.infer_clip_device(dev[0], boxes, rec)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    dev[0].copy_(host, non_blocking=True)
    eng.infer_clip_device(dev[0], boxes, rec)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"serial   : {dt * 1e3:.2f} ms/step  {n / dt:.0f} frames/s  (H2D {host.numel() / 1e6:.0f} MB/step)")
t0 = time.perf_counter()
dev[0].copy_(host, non_blocking=True)
torch.cuda.synchronize()
print(f"copy only: {(time.perf_counter() - t0) * 1e3:.2f} ms -> {host.numel() / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
ready = [torch.cuda.Event() for _ in range(2)]
free = [torch.cuda.Event() for _ in range(2)]
for e in free:
    e.record(main)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K + 1):
    if k < K:
        with torch.cuda.stream(side):
            side.wait_event(free[k & 1])
            dev[k & 1].copy_(host, non_blocking=True)
            ready[k & 1].record(side)
    if k > 0:
        j = (k - 1) & 1
        main.wait_event(ready[j])
        eng.infer_clip_device(dev[j], boxes, rec)
        free[j].record(main)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"overlap  : {dt * 1e3:.2f} ms/step  {n / dt:.0f} frames/s")
