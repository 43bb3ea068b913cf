"""frames + detection table -> labels, with no label files, crop files or host pixels in between (rows a2 / a3 / f1).

What the reference does between its YOLOv5 subprocess and the CNN (``playaid/ai_runner.py:191-289, 306-464``) -- write and
re-read ``labels/*.txt``, repair them, write and re-read ``crops/<Fighter>/*.jpg`` -- as one device-side flow over the
table ``pa_detect_postprocess`` wrote:

    pa_clean_detections      duplicate resolution, gap interpolation, tail copies -> per (frame, fighter): box, source
                             frame, kind of crop (detector's own / square_crop repair)
    pa_save_one_box_crops    the detector's crops (box x 1.02 + 10 px) + their 4:4:4 JPEG write / read
    pa_square_crops          the repaired frames' ``square_crop(128, padding=30)`` + cv2.imwrite's 4:2:0 JPEG
    pa_backbone_crop_images  every crop image through the runner's resize / letterbox and the backbone
    pa_head_frames           windows, temporal head, log-softmax, argmax

Only two small things come back to the host before the end: the repair's ``info`` words (last labelled frame, the
reference's assertions as error codes) and the list of repaired entries (to size the square-crop call).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch


class LabelRepairError(AssertionError):
    """The reference's asserts in ``clean_yolo_crops`` (``ai_runner.py:343, 375-378``), raised from the device's error code."""


_REPAIR_ERRORS = {
    1: "We should have cleaned out the duplicates at this point (duplicate detections of a class never seen before)",
    2: "missing start_yolo_crop (a gap before the fighter's first detection)",
    3: "a fighter has no detection at all",
}


def run_detections_to_labels(engine, frames_dev: torch.Tensor, dets: torch.Tensor, counts: torch.Tensor,
                             jpeg_quality: int = 95, want_crops: bool = False) -> Dict:
    """frames uint8[n,H,W,3] (device), dets float32[n,max_det,6] / counts int32[n] (device, ``Engine.detect_postprocess``)
    -> the result dict of ``Engine.infer_clip`` for frames 1 .. max_frames - 1, plus ``cleaned`` (the repair tables, host
    copies) and ``max_frames``."""
    dev = engine.device
    n_decoded = frames_dev.shape[0]
    F = engine.F
    tab = engine.clean_detections(dets, counts, n_decoded)
    info = tab["info"].cpu().numpy()  # synchronises: the clip's length decides every later launch
    if info[1]:
        raise LabelRepairError(f"{_REPAIR_ERRORS.get(int(info[1]), 'label repair failed')} (label {int(info[2])})")
    n = int(info[0])
    if n < 2:
        raise ValueError("no detections in any label")
    kind = tab["crop_kind"][:n]
    src = tab["pixel_frame"][:n]
    slot = torch.arange(F, dtype=torch.int32, device=dev)[None, :].expand(n, F)
    det_index = torch.where(kind == 1, slot, torch.full_like(slot, -1)).contiguous()
    src_own = torch.where(kind == 1, src, torch.zeros_like(src)).contiguous()
    row_counts = torch.full((n,), F, dtype=torch.int32, device=dev)
    step = engine.max_batch_frames
    parts, descs, base = [], [], 0
    for f0 in range(0, n, step):
        cnt = min(step, n - f0)
        images, desc = engine.save_one_box_crops(frames_dev, tab["crop_row"][f0:f0 + cnt].contiguous(), row_counts[f0:f0 + cnt],
                                                 det_index=det_index[f0:f0 + cnt], jpeg_quality=jpeg_quality,
                                                 src_frame=src_own[f0:f0 + cnt])
        used = int((desc[:, 0] + ((desc[:, 1] & 0xFFFFFFFF) * (desc[:, 1] >> 32) * 3 + 15) // 16 * 16).max().item())
        desc = desc.clone()
        desc[:, 0] += base
        parts.append(images[:used])
        descs.append(desc)
        base += used
    engine.check_device_errors()
    desc = torch.cat(descs)
    rep = torch.nonzero(kind == 2)  # [k, 2] = (frame, fighter) of the square_crop repairs
    if rep.shape[0]:
        boxes = tab["pixel_box"][:n][rep[:, 0], rep[:, 1]]              # [k, 4]
        fr = frames_dev[src[rep[:, 0], rep[:, 1]].long()]
        engine.set_crop_jpeg_quality(jpeg_quality)
        try:
            sq, st = engine.square_crops(fr, boxes[:, None, :].expand(-1, F, -1).contiguous(), padding=engine.cfg.crop_padding)
        finally:
            engine.set_crop_jpeg_quality(0)
        if (st[:, 0] != 0).any():
            raise AssertionError(f"Failed to get square crop from frame {int(rep[np.nonzero(st[:, 0])[0][0], 0]) + 1}")  # ai_runner.py:418
        parts.append(torch.from_numpy(np.ascontiguousarray(sq[:, 0])).to(dev).reshape(-1))
        e = rep[:, 0] * F + rep[:, 1]
        desc[e, 0] = base + torch.arange(rep.shape[0], device=dev, dtype=torch.int64) * (128 * 128 * 3)
        desc[e, 1] = (128 << 32) | 128
    images = torch.cat(parts + [torch.zeros(64, dtype=torch.uint8, device=dev)])
    out = engine.infer_clip_from_packed_crop_images(images, desc, n, want_crops=want_crops)
    out["max_frames"] = n
    out["cleaned"] = {k: v[:n].cpu().numpy() if k != "info" else info for k, v in tab.items()}
    return out
