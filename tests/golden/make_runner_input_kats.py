#!/usr/bin/env python3
"""Golden vectors of the runner-input branch (``playaid/ai_runner.py:446-459``): crop images of
assorted sizes -> 128 x 128 RGB model inputs, computed by the CPU oracle
(``oracle/yolo_crop.runner_input_from_crop``; its Pillow half is pinned against live Pillow, its
INTER_AREA half is a restatement -- "parity unpinned", see oracle/resample.py). Inputs are
regenerated from seeds; the fixture holds shapes, seeds and expected outputs only.

    python tests/golden/make_runner_input_kats.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import yolo_crop  # noqa: E402
from playaid_core_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# (height, width): what each exercises
CASES = [
    (128, 128),   # identity: imutils.resize is a copy, no pad
    (100, 128),   # 128 wide, short: copy, letterbox top / bottom
    (200, 128),   # 128 wide, tall: copy, BICUBIC shrink both axes in the pad step, letterbox left / right
    (315, 275),   # a typical YOLO crop: fractional INTER_AREA -> 146 rows -> pad
    (256, 256),   # 2 x 2 fast path
    (384, 384),   # integer scale 3
    (320, 640),   # integer scale 5 -> 64 rows -> letterbox
    (100, 100),   # smaller than 128: the bilinear emulation, both axes
    (50, 100),    # enlarging, 64 rows
    (100, 200),   # fractional 1.5625 on both axes, 64 rows
    (128, 129),   # 127 rows (int(128 * 128 / 129)): the one-row letterbox
    (120, 300),   # wide: 51 rows
    (331, 97),    # enlarging in x, 436 rows: near the 3.5x filter limit of the pad step
    (1, 1),       # a single pixel
    (7, 300),     # int(7 * 128 / 300) = 2 rows
]


def crop_image(k: int, h: int, w: int) -> np.ndarray:
    """Deterministic BGR test image: a window of a synthetic frame (texture + gradient + noise)."""
    frame = synth.make_frame(3 + k, 720, 1280, seed=21)
    y0, x0 = (37 * k) % (720 - h + 1), (91 * k) % (1280 - w + 1)
    return np.ascontiguousarray(frame[y0 : y0 + h, x0 : x0 + w])


def main():
    outs = [yolo_crop.runner_input_from_crop(crop_image(k, h, w)) for k, (h, w) in enumerate(CASES)]
    np.savez_compressed(os.path.join(HERE, "runner_input_kats.npz"), shapes=np.array(CASES, dtype=np.int64), inputs=np.stack(outs))
    print("wrote", len(CASES), "runner-input KATs")


if __name__ == "__main__":
    main()
