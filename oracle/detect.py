"""ORACLE (test infrastructure, never shipped, never on the product path).

Detection post-processing as the reference runs it through its YOLOv5 subprocess
(``playaid/ai_runner.py:191-224``: ``detect.py --max-det 2 --save-txt --save-conf --classes 2 3``,
every other flag at its default). The arithmetic lives in an un-vendored, unpinned checkout
(``third_party/yolov5`` is absent from /root/reference, ``constants.py:6``), so this file DEFINES the
contract the device stage matches, restating the published ultralytics/yolov5 v7.0 behaviour step by step
in numpy float32: ``utils/general.py::non_max_suppression`` (objectness gate, conf = obj * cls, best class
per row, class filter, class-offset batched ``torchvision.ops.nms``, ``max_det``), ``scale_boxes`` +
``clip_boxes`` + ``.round()``, ``xyxy2xywh / gn`` and the ``'%g '`` label line of ``detect.py`` (written for
``reversed(det)``: lowest confidence first). **Parity unpinned**: nothing in the reference pins any of it;
the consumer side -- 6 space-separated fields per line, class id first -- is what
``read_fighter_yolo_crop`` asserts (``ai_runner.py:53-71``) and is pinned by tests/test_label_cleaning.py.

Input: the detection head's decoded output rows ``pred[rows, 5 + nc]`` = (cx, cy, w, h in network-input
pixels, objectness, class scores), i.e. what the model hands to ``non_max_suppression``. The network itself
has no counterpart in the reference.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

F = np.float32
MAX_WH = F(7680.0)  # class offset of the batched NMS (general.py)
MAX_NMS = 30000


def _nms_order(scores: np.ndarray) -> np.ndarray:
    """Descending by score; equal scores keep their row order (the contract's tie rule: torch's
    sort is not guaranteed stable, so the reference itself leaves ties open)."""
    return np.argsort(-scores.astype(np.float64), kind="stable")


def non_max_suppression(pred: np.ndarray, conf_thres: float = 0.25, iou_thres: float = 0.45,
                        classes: Sequence[int] = (2, 3), max_det: int = 2) -> np.ndarray:
    """-> float32[k, 6] rows (x1, y1, x2, y2, conf, cls), k <= max_det, confidence descending."""
    pred = np.asarray(pred, dtype=F)
    x = pred[pred[:, 4] > F(conf_thres)]
    if x.shape[0] == 0:
        return np.zeros((0, 6), F)
    cls_conf = (x[:, 5:] * x[:, 4:5]).astype(F)
    half_w, half_h = (x[:, 2] / F(2)).astype(F), (x[:, 3] / F(2)).astype(F)
    box = np.stack([x[:, 0] - half_w, x[:, 1] - half_h, x[:, 0] + half_w, x[:, 1] + half_h], axis=1).astype(F)
    j = cls_conf.argmax(axis=1)  # first maximum, like torch.max on the CPU
    conf = cls_conf[np.arange(len(j)), j]
    keep = conf > F(conf_thres)
    box, conf, j = box[keep], conf[keep], j[keep]
    keep = np.isin(j, np.asarray(classes))
    box, conf, j = box[keep], conf[keep], j[keep]
    if box.shape[0] == 0:
        return np.zeros((0, 6), F)
    order = _nms_order(conf)[:MAX_NMS]
    box, conf, j = box[order], conf[order], j[order]
    off = (j.astype(F) * MAX_WH).astype(F)
    b = (box + off[:, None]).astype(F)
    area = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(F)
    suppressed = np.zeros(len(conf), bool)
    kept: List[int] = []
    for i in range(len(conf)):  # torchvision.ops.nms (cpu kernel): rows are already in score order
        if suppressed[i]:
            continue
        kept.append(i)
        if len(kept) == max_det:
            break
        xx1 = np.maximum(b[i, 0], b[i + 1:, 0])
        yy1 = np.maximum(b[i, 1], b[i + 1:, 1])
        xx2 = np.minimum(b[i, 2], b[i + 1:, 2])
        yy2 = np.minimum(b[i, 3], b[i + 1:, 3])
        w = np.maximum(F(0), (xx2 - xx1).astype(F))
        h = np.maximum(F(0), (yy2 - yy1).astype(F))
        inter = (w * h).astype(F)
        ovr = (inter / ((area[i] + area[i + 1:]).astype(F) - inter).astype(F)).astype(F)
        suppressed[i + 1:] |= ovr > F(iou_thres)
    kept = np.asarray(kept, dtype=np.int64)
    return np.concatenate([box[kept], conf[kept, None], j[kept, None].astype(F)], axis=1).astype(F)


def scale_and_normalise(det: np.ndarray, net_hw: Tuple[int, int], img_hw: Tuple[int, int]) -> np.ndarray:
    """``scale_boxes(im.shape[2:], det[:, :4], im0.shape).round()`` then ``xyxy2xywh(...) / gn``
    -> float32[k, 6] rows (cls, cx, cy, w, h, conf) in label-file order (``reversed(det)``)."""
    det = det.astype(F).copy()
    gain = min(net_hw[0] / img_hw[0], net_hw[1] / img_hw[1])
    pad = ((net_hw[1] - img_hw[1] * gain) / 2, (net_hw[0] - img_hw[0] * gain) / 2)
    det[:, [0, 2]] -= F(pad[0])
    det[:, [1, 3]] -= F(pad[1])
    det[:, :4] /= F(gain)
    det[:, [0, 2]] = np.clip(det[:, [0, 2]], F(0), F(img_hw[1]))
    det[:, [1, 3]] = np.clip(det[:, [1, 3]], F(0), F(img_hw[0]))
    det[:, :4] = np.rint(det[:, :4])  # torch.round: half to even
    out = np.zeros((det.shape[0], 6), F)
    gn = np.array([img_hw[1], img_hw[0], img_hw[1], img_hw[0]], F)
    xywh = np.stack([(det[:, 0] + det[:, 2]) / F(2), (det[:, 1] + det[:, 3]) / F(2), det[:, 2] - det[:, 0], det[:, 3] - det[:, 1]],
                    axis=1).astype(F)
    out[:, 0] = det[:, 5]
    out[:, 1:5] = (xywh / gn).astype(F)
    out[:, 5] = det[:, 4]
    return out[::-1].copy()


def label_text(rows: np.ndarray) -> str:
    """``('%g ' * len(line)).rstrip() % line + '\\n'`` per detection; the class is written as ``%g`` of a float
    (``2``), the rest with six significant digits."""
    return "".join(("%g " * 6).rstrip() % tuple(float(v) for v in r) + "\n" for r in rows)


def detect_frame(pred: np.ndarray, net_hw: Tuple[int, int], img_hw: Tuple[int, int], conf_thres: float = 0.25,
                 iou_thres: float = 0.45, classes: Sequence[int] = (2, 3), max_det: int = 2) -> Tuple[np.ndarray, str]:
    rows = scale_and_normalise(non_max_suppression(pred, conf_thres, iou_thres, classes, max_det), net_hw, img_hw)
    return rows, label_text(rows)


def save_one_box_rect(row: np.ndarray, img_hw: Tuple[int, int], gain: float = 1.02, pad: int = 10):
    """The rectangle ``utils/plots.py::save_one_box`` cuts for one detection, as ``detect.py --save-crop`` calls it
    (``ai_runner.py:208``; ``gain=1.02, pad=10, square=False``): from a LABEL row (cls, cx, cy, w, h, conf; normalised)
    back to the rounded pixel box it was written from (``scale_boxes(...).round()``: whole numbers, so the centre is a
    multiple of 0.5 and recovered exactly), then in float32 like torch: ``b = xyxy2xywh(xyxy)``, ``wh = wh * gain + pad``,
    ``xywh2xyxy(b).long()`` (truncation), ``clip_boxes``. -> (x1, y1, x2, y2) ints; the crop is ``im[y1:y2, x1:x2]``."""
    H, W = img_hw
    cx, cy, w, h = (F(v) for v in row[1:5])
    xc = F(np.rint(F(2) * (cx * F(W)))) / F(2)
    yc = F(np.rint(F(2) * (cy * F(H)))) / F(2)
    bw = F(np.rint(w * F(W))) * F(gain) + F(pad)
    bh = F(np.rint(h * F(H))) * F(gain) + F(pad)
    x1, y1 = F(xc - bw / F(2)), F(yc - bh / F(2))
    x2, y2 = F(xc + bw / F(2)), F(yc + bh / F(2))
    t = lambda v: int(np.trunc(v))  # noqa: E731  (.long())
    x1, x2 = min(max(t(x1), 0), W), min(max(t(x2), 0), W)
    y1, y2 = min(max(t(y1), 0), H), min(max(t(y2), 0), H)
    return x1, y1, x2, y2


def save_one_box(row: np.ndarray, im_bgr: np.ndarray, quality: int = 95):
    """-> what ``cv2.imread(crops/<Fighter>/<video>_<n>.jpg)`` returns in the reference (``ai_runner.py:446``): the BGR
    crop after YOLOv5 v7.0's ``Image.fromarray(crop[..., ::-1]).save(f, quality=95, subsampling=0)`` and a JPEG read, or
    None for an empty rectangle. **Parity unpinned** for the rectangle (un-vendored YOLOv5); the JPEG step is the
    libjpeg-pinned ``oracle.jpeg.roundtrip_any``."""
    from . import jpeg

    x1, y1, x2, y2 = save_one_box_rect(row, im_bgr.shape[:2])
    if x2 <= x1 or y2 <= y1:
        return None
    crop = np.ascontiguousarray(im_bgr[y1:y2, x1:x2])
    return np.ascontiguousarray(jpeg.roundtrip_any(crop[..., ::-1], quality, subsampling=0)[..., ::-1])
