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
det = YoloV5Detector(sd, NC, NET, max_images=4)
print("case                     rows   | boxes (px): dev-o32   dev-o64   o32-o64 | scores: dev-o32    dev-o64    o32-o64  | max score")
worst = 0.0
for h, w, n, seed in ((720, 1280, 3, 5), (1080, 1920, 2, 5), (270, 480, 5, 9)):
    frames = synth.make_frames(n, h, w, seed=seed)
    got = det(frames)
    torch.cuda.synchronize()
    got = got.cpu().numpy().astype(np.float64)
    x = torch.from_numpy(np.stack([oy.letterbox(f, NET) for f in frames]))
    w32 = oy.forward(x, sd, NC).numpy().astype(np.float64)
    w64 = oy.forward(x.double(), sd64, NC).numpy()
    e = lambda a, b, sl: float(np.abs(a[..., sl] - b[..., sl]).max())
    B, S = slice(0, 4), slice(4, None)
    print(f"{n} x {h:4d}x{w:4d} seed {seed}  {got.shape[1]:6d} |          {e(got, w32, B):9.2e} {e(got, w64, B):9.2e} {e(w32, w64, B):9.2e} |"
          f"       {e(got, w32, S):9.2e}  {e(got, w64, S):9.2e}  {e(w32, w64, S):9.2e} | {w64[..., 4].max():.3f}")
    worst = max(worst, e(got, w32, S))
print(f"worst score error against the fp32 oracle: {worst:.3e} (test bar: see tests/test_yolov5.py)")
det.close()
