"""Timeline of one launch of the persistent GEMM (pigemm.hip) inside the detection network, from its per-wave s_memtime stamps:
where a workgroup's life goes -- prologue, per tile: wait for the first stage, k loop, epilogue.
  PA_PG_STAMP_FILE=/tmp/pg.bin PA_PG_STAMP_SHAPE=245760,128,128 python scripts/pgemm_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from playaid_core_amd import synth  # noqa: E402
from playaid_core_amd.yolov5 import YoloV5Detector  # noqa: E402

n, H, W = 64, 1080, 1920
dev = torch.device("cuda:0")
sd = synth.make_yolov5s_state_dict()
det = YoloV5Detector(sd, 6, (384, 640), max_images=n, device="cuda:0")
frames = torch.from_numpy(synth.make_frames(4, H, W)).to(dev).repeat(n // 4, 1, 1, 1).contiguous()
pred = torch.empty((n, det.rows, 11), dtype=torch.float32, device=dev)
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(6):
    rc = det._lib.pa_detector_forward(det._h, C.c_void_p(frames.data_ptr()), n, H, W, C.c_void_p(pred.data_ptr()), stream)
    assert rc == 0
torch.cuda.synchronize()
raw = open(os.environ["PA_PG_STAMP_FILE"], "rb").read()
grid, nk, tm, tn = np.frombuffer(raw[:16], dtype=np.int32)
st = np.frombuffer(raw[16:], dtype=np.uint64).reshape(grid, 4, 64).astype(np.int64)
T = 100.0   # ticks per printed unit: s_memtime runs at 100 MHz on this part -> 1 unit = 1 us
t0 = st[..., 0][st[..., 0] > 0].min()
alive = st[..., 63] > 0
print(f"shape {os.environ.get('PA_PG_STAMP_SHAPE')}: grid {grid} ({alive[:, 0].sum()} workgroups with tiles), {nk} k-steps per tile, {tm} x {tn} tiles")
life = (st[..., 63] - st[..., 0])[alive]
print(f"launch span {(st[..., 63].max() - t0) / T:.2f}; wave lifetime median {np.median(life) / T:.2f}, max {life.max() / T:.2f}  (units: s_memtime ticks / {T:.0f})")
start = ((st[..., 0] - t0) / T)[alive]
print(f"wave start: p10 {np.percentile(start, 10):.2f} median {np.median(start):.2f} p90 {np.percentile(start, 90):.2f} max {start.max():.2f}")
pro = ((st[..., 1] - st[..., 0]) / T)[alive]
print(f"prologue (entry -> first two stages requested): median {np.median(pro):.2f}, p90 {np.percentile(pro, 90):.2f}")
for a, b, name in ((0, 60, "entry -> tile range known (kernel arguments arrived)"), (60, 61, "-> first tile's row offsets"), (61, 62, "-> first stage requested"),
                   (62, 1, "-> second stage requested, bias in registers")):
    d = ((st[..., b] - st[..., a]) / T)[alive]
    print(f"  prologue {name}: median {np.median(d):.2f}, p90 {np.percentile(d, 90):.2f}")
for t in range(15):
    a, b, c, d = (st[..., 2 + 4 * t + j] for j in range(4))
    ok = alive & (d > 0)
    if not ok.any():
        break
    print(f"tile {t:2d} ({ok[:, 0].sum():4d} wgs): wait for first stage {np.median((b - a)[ok]) / T:6.2f}  k loop {np.median((c - b)[ok]) / T:6.2f}  "
          f"epilogue {np.median((d - c)[ok]) / T:6.2f}  (p90 {np.percentile((b - a)[ok], 90) / T:.2f} {np.percentile((c - b)[ok], 90) / T:.2f} {np.percentile((d - c)[ok], 90) / T:.2f})"
          + (f"  gap to next tile {np.median((st[..., 6 + 4 * t] - d)[ok & (st[..., 6 + 4 * t] > 0)]) / T:.2f}" if t < 14 and (ok & (st[..., 6 + 4 * t] > 0)).any() else ""))
