"""Detector hand-off: from detection-head rows to the label text the
runner reads.

The reference shells out to a YOLOv5 checkout (``playaid/ai_runner.py:191-224``) and gets its detections back
as files: ``labels/<video>_<n>.txt`` with one ``"cls cx cy w h conf"`` line per box (``--save-txt
--save-conf``), parsed by ``read_fighter_yolo_crop`` / ``read_yolo_crops`` (``:53-94``). Here the
post-network half of that subprocess -- confidence gates, class filter ``--classes 2 3``, class-aware NMS,
``--max-det 2``, mapping back to the frame -- runs on the MI355X (``pa_detect_postprocess``), and this module
turns its float32 rows into exactly that text: every field written with ``'%g'`` as ``detect.py`` does.
``labels_for_clip`` gives the per-frame label blocks a ``ClipSource`` (and through it ``clean_yolo_labels``)
takes. The detection network itself has no counterpart in the reference (external weights and code).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def label_lines(rows: np.ndarray) -> str:
    """float32[k, 6] (cls cx cy w h conf) -> label-file text, ``('%g ' * 6).rstrip() % line + '\\n'`` per row."""
    return "".join(("%g " * 6).rstrip() % tuple(float(v) for v in r) + "\n" for r in rows)


def labels_for_clip(engine, pred, net_hw: Tuple[int, int], img_hw: Tuple[int, int], conf_thres: float = 0.25,
                    iou_thres: float = 0.45, classes: Sequence[int] = (2, 3), max_det: int = 2, batch: int = 256) -> List[str]:
    """pred: float32[n_frames, rows, 5 + nc] head rows (host or device) -> one label block per frame (empty
    string where nothing was detected, like a frame for which detect.py writes no file)."""
    import torch

    labels: List[str] = []
    n = pred.shape[0]
    for f0 in range(0, n, batch):
        dets, counts = engine.detect_postprocess(pred[f0 : f0 + batch], net_hw, img_hw, conf_thres, iou_thres, classes, max_det)
        torch.cuda.synchronize(engine.device)
        d, c = dets.cpu().numpy(), counts.cpu().numpy()
        labels.extend(label_lines(d[i, : c[i]]) for i in range(d.shape[0]))
    return labels
