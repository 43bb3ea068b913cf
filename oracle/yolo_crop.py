"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the crop geometry of ``playaid/fighter.py:305-381``
(``YoloCrop.yolo_pixels`` / ``YoloCrop.square_crop``) on top of the resampler
restatements in ``oracle/resample.py``.

Pinning: the Pillow half is pinned against live Pillow; the INTER_AREA half is
"parity unpinned" (cv2 absent) -- see ``oracle/resample.py``.
"""
from __future__ import annotations

import numpy as np

from . import resample as R


def yolo_pixels(cx, cy, w, h, image_width, image_height):
    """``fighter.py:305-314``: truncating int() of the normalised box."""
    return (int(cx * image_width), int(cy * image_height), int(w * image_width), int(h * image_height))


def square_crop(image: np.ndarray, box, output_size: int = 128, padding: int = 0):
    """``YoloCrop.square_crop`` (``fighter.py:323-381``) -> (ok, uint8[128,128,3]).

    Follows the reference statement by statement, including numpy's slice
    semantics for a negative stop (an off-screen box above/left of the frame
    wraps the stop index around, ``fighter.py:335-343``)."""
    cx, cy, cw, ch = yolo_pixels(box[0], box[1], box[2], box[3], image.shape[1], image.shape[0])
    square_dim = max(cw, ch)
    square_half = int(square_dim / 2)
    raw_crop = image[
        max(cy - square_half - padding, 0) : min(cy + square_half + padding, image.shape[0]),
        max(cx - square_half - padding, 0) : min(cx + square_half + padding, image.shape[1]),
        :,
    ]
    if raw_crop.shape[0] != square_dim or raw_crop.shape[1] != square_dim:
        try:
            raw_crop = R.pil_pad_black(raw_crop, (square_dim, square_dim))
        except ValueError:
            return False, None
    if raw_crop.shape[0] == 0 or raw_crop.shape[1] == 0:
        return False, None
    crop = R.imutils_resize_width(raw_crop, output_size)
    if crop.shape[0] != output_size or crop.shape[1] != output_size:
        crop = R.pil_pad_black(crop, (output_size, output_size))
    assert crop.shape == (output_size, output_size, 3)
    return True, crop


def runner_input_from_crop(crop_bgr: np.ndarray, output_size: int = 128) -> np.ndarray:
    """``ai_runner.py:446-459`` applied to an in-memory BGR crop (the JPEG
    write/read between ``cv2.imwrite`` :420 and ``cv2.imread`` :446 is outside
    the synthetic configs): BGR->RGB, ``imutils.resize(width=128)`` (identity
    for a 128-wide crop), black pad to 128x128 if needed."""
    frame = crop_bgr[:, :, ::-1]
    frame = R.imutils_resize_width(np.ascontiguousarray(frame), output_size)
    if frame.shape[0] != output_size or frame.shape[1] != output_size:
        frame = R.pil_pad_black(frame, (output_size, output_size))
    assert frame.shape == (128, 128, 3)
    return frame
