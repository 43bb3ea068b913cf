"""Child of tests/test_knobs.py: the A/B knobs of the library are environment variables read once per process (or per handle), so
each setting runs in a process of its own. Computes the detection network's rows (emulated-fp32 handle) on four 720p frames and the
action CNN's log-probabilities (exact and emulated engines) on an eight-frame clip and saves them."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from playaid_core_amd import synth
from playaid_core_amd.engine import Engine
from playaid_core_amd.yolov5 import YoloV5Detector


def main():
    out_path = sys.argv[1]
    res = {}
    frames = synth.make_frames(16, 720, 1280, seed=5)
    for dt in ("f32", "emulated_f32"):
        det = YoloV5Detector(synth.make_yolov5s_state_dict(), 6, (384, 640), max_images=16, compute_dtype=dt)
        try:
            rows = det(frames)
            torch.cuda.synchronize()
            res[f"rows_{dt}"] = rows.cpu().numpy()
        finally:
            det.close()
    sd = synth.make_state_dict(seed=1234)
    f8, b8 = synth.make_frames(8, 720, 1280), synth.make_boxes(8, 720, 1280)
    for dt in ("f32", "emulated_f32"):
        eng = Engine(sd, max_batch_frames=8, max_clip_frames=64, max_frame_height=720, max_frame_width=1280, compute_dtype=dt)
        try:
            res[f"logp_{dt}"] = eng.infer_clip(f8, b8)["logp"]
        finally:
            eng.close()
    np.savez(out_path, **res)


if __name__ == "__main__":
    main()
