"""Damage read-out plumbing: host mirror of ``AIRunner.run_damage_detection``
(``playaid/ai_runner.py:537-590``) and ``damage_crop_to_percent`` (``:109-133``).

The reference cuts two fixed HUD boxes out of every frame (``YoloCrop(center_x=402/1280 | 898/1280,
center_y=637/720, crop_width=133/1280, crop_height=60/720).crop_img(frame)``), enlarges each to 256 px wide
(``imutils.resize(width=256)``) and hands it to PaddleOCR; two text boxes ordered by their x position become
``"<whole>.<decimal>"``. Here the crops and the resize run on the device (``pa_crop_resize_width``, bit-exact against
the CPU oracle); the recogniser is an external model whose arithmetic is not in the reference, so it stays a
callable the caller supplies -- ``ocr(image_bgr_uint8[oh,256,3]) -> (boxes, detected_text, extra)``, the
PaddleOCR call shape of ``:115-117``.
"""
from __future__ import annotations

import re
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .fighter import YoloCrop

# (player id, normalised x centre of its damage read-out), ai_runner.py:552-555
PLAYER_DAMAGE_X: Tuple[Tuple[int, float], ...] = ((0, 402 / 1280), (1, 898 / 1280))
DAMAGE_CROP_WIDTH = 256


def damage_box(damage_x: float) -> YoloCrop:
    """``ai_runner.py:562-567``."""
    return YoloCrop(center_x=damage_x, center_y=637 / 720, crop_width=133 / 1280, crop_height=60 / 720)


def damage_rects(image_width: int, image_height: int) -> List[Tuple[int, int, int, int]]:
    """Pixel rectangles ``(x1, y1, x2, y2)`` of the two read-outs (``YoloCrop.xyxy_pixels``, ``fighter.py:284-294``)."""
    return [damage_box(x).xyxy_pixels(image_width, image_height) for _, x in PLAYER_DAMAGE_X]


def extract_numbers(text: str) -> str:
    """``ai_runner.py:97-99``."""
    return "".join(re.findall(r"\d+", text))


def parse_damage(results):
    """The part of ``damage_crop_to_percent`` behind the recogniser (``ai_runner.py:117-133``): ``results`` =
    ``(boxes, detected_text, extra)`` -> ``(ok, (damage, original_string, confidence, results))``.

    Exactly two text boxes are a reading: the one whose first corner lies further left is the whole part, the
    other the decimal; both must be non-empty. The percent keeps only the digits of each part; the reported
    confidence is the FIRST list entry's (a reference quirk: not the whole part's when the boxes arrive swapped).
    Anything else is a miss: -1, the texts joined by "_", confidence 0."""
    boxes, texts, _extra = results
    if len(texts) == 2:
        left_first = boxes[0][0][0] < boxes[1][0][0]
        whole, decimal = (texts[0][0], texts[1][0]) if left_first else (texts[1][0], texts[0][0])
        if whole and decimal:
            value = float(f"{extract_numbers(whole)}.{extract_numbers(decimal)}")
            return True, (value, f"{whole}.{decimal}", texts[0][1], results)
    return False, (-1, "_".join(t[0] for t in texts), 0.0, results)


def damage_crops(frames, engine) -> List[np.ndarray]:
    """frames uint8[n,H,W,3] (BGR) -> one uint8[n, oh, 256, 3] array per player: what the reference passes to the
    recogniser for every frame (``:568-571`` + ``:114``)."""
    n, h, w, _ = frames.shape
    return engine.crop_resize_width(frames, damage_rects(w, h), DAMAGE_CROP_WIDTH)


def run_damage_detection(frames, engine, ocr: Callable, player_id_to_fighter: Dict[int, str], ai_output_data=None):
    """``AIRunner.run_damage_detection``: per frame and player, ``damage`` (float percent, -1 when the read-out
    could not be parsed) stored at ``ai_output_data[fighter][i].damage`` when a table is given.
    -> (damage float64[n, 2], fraction of confident reads)."""
    crops = damage_crops(frames, engine)
    n = frames.shape[0]
    damage = np.full((n, len(PLAYER_DAMAGE_X)), -1.0)
    confident = 0
    for i in range(n):
        for j, (player_id, _) in enumerate(PLAYER_DAMAGE_X):
            ok, (value, _text, _conf, _res) = parse_damage(ocr(crops[j][i]))
            confident += int(ok)
            damage[i, j] = value
            if ai_output_data is not None:
                ai_output_data[player_id_to_fighter[player_id]][i].damage = value
    return damage, confident / float(max(n, 1) * len(PLAYER_DAMAGE_X))
