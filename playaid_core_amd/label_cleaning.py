"""Tracking-label repair -- row a3 of SURVEY.md section 8a.

In-memory mirror of ``AIRunner.clean_yolo_crops`` / ``clean_yolo_crops_for_fighter``
(``playaid/ai_runner.py:226-289, 306-424``). The reference repairs the detector's
``labels/<video>_<n>.txt`` files and ``crops/<Fighter>/<video>_<n>.jpg`` images in place;
here the same decisions are taken on the label *text* of a ``ClipSource`` and the result is
a table saying, for every (frame, fighter), which box the crop is cut with and from which
decoded frame -- the crop pixels themselves are then produced by the HIP crop stage.

Reference behaviour kept, quirks included:

* duplicate detections of one class in a frame: keep the one whose centre is nearest (L1,
  normalised units) to that class's box in the previous labelled frame (``:314-340``); a
  duplicate of a class never seen before trips the reference's
  ``assert len(crops) == 1`` (``:343``) -- same ``AssertionError`` here;
* gaps: for a fighter missing in frames ``latest+1 .. current-1`` the box is
  ``start.interp(end, (current - j) / (current - latest))`` (``:389-390``) -- the percentage
  is measured from the *end* frame, so the interpolated track runs backwards through the gap;
  the line is appended to label ``j`` (``:393-397``);
* the repaired crop of label frame ``j`` (1-indexed) is cut from ``VideoCapture`` position
  ``j`` (``:405-406``), i.e. the decoded frame one *after* the one the detector labelled
  ``j``; if that read fails (past the end) the previous frame's crop image is copied (``:407-416``);
* a gap before a fighter's first detection asserts (``:375-378``, the reference's ``TODO``),
  unless the first detection is frame 2 (no gap is seen then and frame 1 simply has no crop);
* tail: the fighter whose crops end first gets its last crop image duplicated up to, but not
  including, the other fighter's last frame (``range(last, last + remaining)``, ``:270-289``);
  labels are not touched there, so those frames report ``crop: None``.

Not modelled (file-name artefacts of the external detector, no arithmetic involved): the
``<video>_<n>2.jpg`` names YOLOv5's ``increment_path`` gives duplicate crops (``:247-258``).
"""
from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import constants
from .fighter import YoloCrop


def parse_label(text: str, where: str = "<memory>") -> List[YoloCrop]:
    """``read_yolo_crops`` (``ai_runner.py:74-94``) on the text of one label file."""
    crops = []
    for line in text.splitlines():
        if not line:
            continue
        assert len(line.split(" ")) == 6, f"Too much data for line: {line} in label {where}"
        crops.append(YoloCrop.from_string(line))
    return crops


@dataclass
class CleanedLabels:
    """Result of ``clean_yolo_labels``. Frame axis: list index ``i`` = label frame ``i + 1``."""

    max_frames: int                                  # number of the last non-empty label (ai_runner.py:244-245)
    labels: List[str]                                # repaired label text, ``max_frames`` entries
    label_crop: List[List[Optional[YoloCrop]]]       # [i][p]: what read_fighter_yolo_crop returns after repair
    pixel_frame: np.ndarray                          # int32 [max_frames, F]: decoded-frame index (0-based) the crop is cut from, -1 = no crop
    pixel_box: np.ndarray                            # float64 [max_frames, F, 4]: normalised cx cy w h used for the cut
    crop_kind: np.ndarray = None                     # int32 [max_frames, F]: what crops/<Fighter>/<video>_<n>.jpg holds -- 0 nothing,
                                                     # 1 the detector's save_one_box crop (ai_runner.py:208), 2 a square_crop repair (:417-420)
    crop_row: np.ndarray = None                      # float32 [max_frames, F, 6]: kind 1: the label row (cls cx cy w h conf) whose box
                                                     # save_one_box cut, from decoded frame pixel_frame
    log: List[str] = field(default_factory=list)     # the messages the reference prints

    def identity_source(self) -> bool:
        """True when every crop is cut from its own label's frame (no repaired gaps)."""
        want = np.arange(self.max_frames, dtype=np.int32)[:, None]
        return bool(np.all((self.pixel_frame == want) | (self.pixel_frame < 0)))


def _fighter_crop(crops: List[YoloCrop], class_id: int) -> Optional[YoloCrop]:
    for c in crops:  # first matching line, as read_fighter_yolo_crop (ai_runner.py:53-71)
        if c.class_id == class_id:
            return c
    return None


def clean_yolo_labels(labels: List[str], fighters: List[str], n_decoded_frames: int, name: str = "clip") -> CleanedLabels:
    nonempty = [i for i, t in enumerate(labels) if t.strip()]
    if not nonempty:
        raise ValueError("no detections in any label")
    max_frames = nonempty[-1] + 1
    log: List[str] = []
    frames: List[List[YoloCrop]] = [parse_label(labels[i], f"{name}_{i + 1}.txt") for i in range(max_frames)]
    original = [list(fr) for fr in frames]  # as the detector wrote them (the crop files follow these, not the repair)
    class_ids = [constants.CHAR_LIST.index(f) for f in fighters]

    # crops the detector saved: one per (frame, class) it reported
    has_crop: Dict[int, set] = {c: {i + 1 for i in range(max_frames) if _fighter_crop(frames[i], c)} for c in class_ids}

    # -- duplicates (:314-359); the reference runs this once per fighter, the second pass finds none
    previous: Dict[int, YoloCrop] = {}
    for i in range(max_frames):
        by_class = defaultdict(list)
        for c in frames[i]:
            by_class[c.class_id].append(c)
        found = False
        for cid, crops in by_class.items():
            if len(crops) > 1 and cid in previous:
                found = True
                min_distance, nearest = 10000, None
                for c in crops:
                    d = abs(c.center_x - previous[cid].center_x) + abs(c.center_y - previous[cid].center_y)
                    if d < min_distance:
                        min_distance, nearest = d, c
                by_class[cid] = [nearest]
        new = []
        for cid, crops in by_class.items():
            assert len(crops) == 1, "We should have cleaned out the duplicates at this point"
            new.append(crops[0])
            previous[cid] = crops[0]
        if found:
            log.append(f"Re-writing {name}_{i + 1}.txt")
            frames[i] = new

    F = len(fighters)
    pixel_frame = np.full((max_frames, F), -1, dtype=np.int32)
    pixel_box = np.zeros((max_frames, F, 4), dtype=np.float64)
    crop_kind = np.zeros((max_frames, F), dtype=np.int32)
    crop_row = np.zeros((max_frames, F, 6), dtype=np.float32)
    for p, cid in enumerate(class_ids):
        for f in has_crop[cid]:
            pixel_frame[f - 1, p] = f - 1
            pixel_box[f - 1, p] = _fighter_crop(frames[f - 1], cid).yolo_crop()
            # the crop FILE without a counter in its name is the first detection of the class the detector wrote
            # (detect.py saves in label order; later ones get <n>2.jpg, which the reference deletes, :247-258)
            first = _fighter_crop(original[f - 1], cid)
            crop_kind[f - 1, p] = 1
            crop_row[f - 1, p] = [first.class_id, first.center_x, first.center_y, first.crop_width, first.crop_height, first.confidence]

    # -- gaps (:361-424)
    for p, (fighter, cid) in enumerate(zip(fighters, class_ids)):
        latest = 1  # number of the first label file (:363); empties were created for 1..max_frames-1 (:261-266)
        for current in sorted(has_crop[cid]):
            if current - latest > 1:
                log.append(f"Missing frames {latest + 1}-{current - 1} for {fighter}")
                start = _fighter_crop(frames[latest - 1], cid)
                assert start, f"missing start_yolo_crop {name}_{latest}.txt for {fighter}"
                end = _fighter_crop(frames[current - 1], cid)
                assert end, f"missing end_yolo_crop {name}_{current}.txt for {fighter}"
                for j in range(latest + 1, current):
                    if _fighter_crop(frames[j - 1], cid):
                        continue  # "Already have the intermediate crop" (:383-385)
                    percent = (current - j) / (current - latest)
                    interp = start.interp(end, percent=percent)
                    frames[j - 1].append(interp)
                    if j < n_decoded_frames:  # VideoCapture position j = decoded frame index j (:405-406)
                        pixel_frame[j - 1, p] = j
                        pixel_box[j - 1, p] = interp.yolo_crop()
                        crop_kind[j - 1, p] = 2
                    else:  # read failed: the crop image of frame j-1 is copied (:407-416)
                        log.append(f"Failed to read from frame {j} during interpolation")
                        pixel_frame[j - 1, p] = pixel_frame[j - 2, p]
                        pixel_box[j - 1, p] = pixel_box[j - 2, p]
                        crop_kind[j - 1, p] = crop_kind[j - 2, p]
                        crop_row[j - 1, p] = crop_row[j - 2, p]
            latest = current

    # -- tail (:270-289): duplicate the shorter fighter's last crop image
    last = {p: int(np.nonzero(pixel_frame[:, p] >= 0)[0][-1]) + 1 for p in range(F) if np.any(pixel_frame[:, p] >= 0)}
    if len(last) != F:
        raise ValueError("a fighter has no detection at all")
    mx = max(last.values())
    for p, lf in last.items():
        if mx - lf:
            log.append(f"For {fighters[p]} duplicating last frame {lf} {mx - lf} times")
            for i in range(lf, mx):  # frames lf .. mx-1 (frame mx itself is left without a crop)
                pixel_frame[i - 1, p] = pixel_frame[lf - 1, p]
                pixel_box[i - 1, p] = pixel_box[lf - 1, p]
                crop_kind[i - 1, p] = crop_kind[lf - 1, p]  # a copy of that image FILE, whatever made it
                crop_row[i - 1, p] = crop_row[lf - 1, p]

    out_labels = ["".join(str(c) + "\n" for c in fr) for fr in frames]
    label_crop = [[_fighter_crop(fr, cid) for cid in class_ids] for fr in frames]
    return CleanedLabels(max_frames, out_labels, label_crop, pixel_frame, pixel_box, crop_kind, crop_row, log)
