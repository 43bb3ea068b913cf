"""Per-layer times of the Winograd 3x3 kernel at the shapes of the headline's ResNet-18 (128 crops) and of the detector
(64 x 1080p frames): HIP events on the current stream around 20 back-to-back launches, median of 5 after 5 warm-ups. Prints algorithmic (direct-form)
GFLOP, time, effective TFLOP/s (algorithmic / time) and executed TFLOP/s (4 / 9 of that)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from playaid_core_amd import wino

SHAPES = [("resnet layer1", 128, 32, 32, 64, 64), ("resnet layer2", 128, 16, 16, 128, 128), ("resnet layer3", 128, 8, 8, 256, 256),
          ("resnet layer4", 128, 4, 4, 512, 512), ("yolo 96x160 c32", 64, 96, 160, 32, 32), ("yolo 48x80 c64", 64, 48, 80, 64, 64),
          ("yolo 24x40 c128", 64, 24, 40, 128, 128), ("yolo 12x20 c256", 64, 12, 20, 256, 256)]
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for name, n, h, w, cin, cout in SHAPES:
    xp = torch.zeros((n, h + 2, w + 2, cin), device=dev)
    xp[:, 1:-1, 1:-1] = torch.randn((n, h, w, cin), device=dev)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    ug = torch.from_numpy(wino.transform_weights(wt)).to(dev)
    out = torch.zeros((n, h + 2, w + 2, cout), device=dev)
    res = torch.randn((n, h + 2, w + 2, cout), device=dev) if os.environ.get("RES") else None   # RES=1: with a residual, as the networks' layers run
    for _ in range(5):
        wino.conv3x3(xp, ug, cin, cout, out=out, act=1, residual=res)
    # 20 launches back to back between two events, five times: a launch's own time in a stream that is kept busy (a single
    # launch between two events with the host in between reads 5-10 us longer: the queue runs dry in front of it)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            wino.conv3x3(xp, ug, cin, cout, out=out, act=1, residual=res)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    us = float(np.median(ts))
    gf = 2.0 * n * h * w * cin * cout * 9 / 1e9
    print(f"{name:18s} n={n:3d} {h:3d}x{w:3d} cin {cin:3d} cout {cout:3d}  {gf:6.2f} GF  {us:7.1f} us  {gf / us * 1e3:6.1f} TF effective  {gf / us * 1e3 * 4 / 9:6.1f} TF executed", flush=True)
