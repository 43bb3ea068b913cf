"""Measured error of the detection network (pa_detector_forward) against its oracle, with a float64 run of the same oracle as
the arbiter: max |device - oracle32|, |device - oracle64|, |oracle32 - oracle64| over the head rows of the test's cases.
Output kept under profiles/ (round 4: r04_yolov5_parity.txt)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import yolov5 as oy  # noqa: E402  (checker only)
from playaid_core_amd import synth  # noqa: E402
from playaid_core_amd.yolov5 import YoloV5Detector  # noqa: E402

NC, NET = 6, (384, 640)
sd = synth.make_yolov5s_state_dict()
sd64 = {k: torch.from_numpy(np.asarray(v)).double() for k, v in sd.items()}
dets = {dt: YoloV5Detector(sd, NC, NET, max_images=5, compute_dtype=dt) for dt in ("f32", "emulated_f32")}
print("compute_dtype  case                     rows   | boxes (px): dev-o32   dev-o64   o32-o64 | scores: dev-o32    dev-o64    o32-o64  | max score")
worst = {dt: 0.0 for dt in dets}
for h, w, n, seed in ((720, 1280, 3, 5), (1080, 1920, 2, 5), (270, 480, 5, 9)):
    frames = synth.make_frames(n, h, w, seed=seed)
    x = torch.from_numpy(np.stack([oy.letterbox(f, NET) for f in frames]))
    w32 = oy.forward(x, sd, NC).numpy().astype(np.float64)
    w64 = oy.forward(x.double(), sd64, NC).numpy()
    for dt, det in dets.items():
        got = det(frames)
        torch.cuda.synchronize()
        got = got.cpu().numpy().astype(np.float64)
        e = lambda a, b, sl: float(np.abs(a[..., sl] - b[..., sl]).max())
        B, S = slice(0, 4), slice(4, None)
        print(f"{dt:13s}  {n} x {h:4d}x{w:4d} seed {seed}  {got.shape[1]:6d} |          {e(got, w32, B):9.2e} {e(got, w64, B):9.2e} {e(w32, w64, B):9.2e} |"
              f"       {e(got, w32, S):9.2e}  {e(got, w64, S):9.2e}  {e(w32, w64, S):9.2e} | {w64[..., 4].max():.3f}")
        worst[dt] = max(worst[dt], e(got, w32, S))
for dt, det in dets.items():
    print(f"{dt}: worst score error against the fp32 oracle {worst[dt]:.3e} (test bar: see tests/test_yolov5.py)")
    det.close()

# Which Winograd layers move the boxes: the 1080p case with NO layer in Winograd form, with each of the
# eligible stride-1 3x3 convolutions alone in it (PA_DET_WINO_MASK, read at pa_detector_create), and with all of them.
from playaid_core_amd.yolov5 import build_yolov5s_table  # noqa: E402

layers = build_yolov5s_table(sd, NET, NC)[0]
elig = [l for l in layers if l.kind == 0 and l.ksize == 3 and l.stride == 1 and l.in_h % 4 == 0 and l.in_w % 4 == 0]
frames = synth.make_frames(2, 1080, 1920, seed=5)
x = torch.from_numpy(np.stack([oy.letterbox(f, NET) for f in frames]))
w64 = oy.forward(x.double(), sd64, NC).numpy()
w32 = oy.forward(x, sd, NC).numpy().astype(np.float64)
print(f"\n2 x 1080x1920 seed 5, box error (px) against the float64 run; the fp32 oracle itself: {np.abs(w32[..., :4] - w64[..., :4]).max():.2e}")


def run(mask):
    os.environ["PA_DET_WINO_MASK"] = hex(mask)
    d = YoloV5Detector(sd, NC, NET, max_images=4)
    g = d(frames)
    torch.cuda.synchronize()
    g = g.cpu().numpy().astype(np.float64)
    d.close()
    return float(np.abs(g[..., :4] - w64[..., :4]).max()), float(np.abs(g[..., 4:] - w64[..., 4:]).max())


b, sc = run(0)
print(f"  no layer as Winograd (direct patch kernel)        boxes {b:.2e}  scores {sc:.2e}")
for k, l in enumerate(elig):
    b, sc = run(1 << k)
    print(f"  only #{k:2d} {l.in_h:3d}x{l.in_w:3d} cin {l.cin:3d} cout {l.cout:3d} as Winograd   boxes {b:.2e}  scores {sc:.2e}")
b, sc = run((1 << len(elig)) - 1)
print(f"  all {len(elig)} eligible layers as Winograd                boxes {b:.2e}  scores {sc:.2e}")
os.environ.pop("PA_DET_WINO_MASK")
