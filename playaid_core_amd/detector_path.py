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

ONE small thing comes back to the host in the middle: five words -- the repair's ``info`` (last labelled frame, the
reference's assertions as error codes) and the number of repaired entries -- because the clip's length decides every later
launch. What the later launches need from the repaired table (which detection each crop is cut from, the repairs' boxes and
source frames in entry order, the crop images' descriptors) is worked out on the device by two small kernels
(``pa_detector_plan``, ``pa_detector_plan_desc``), not by a dozen generic tensor launches. ``begin`` enqueues the repair, the
plan and that copy, ``finish`` waits for the five words and enqueues the rest; pixels,
boxes, crop images and descriptors never leave the device, nothing else synchronises (``finish(..., device_results=True)``
does not even wait for the results), so a caller can put the next clip's detector under this clip's wait.
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


class Ticket:
    """A clip between ``begin`` and ``finish``: the repair tables and the plan (device), the five host words and the event
    behind them."""

    def __init__(self, frames_dev, tab, plan, words, event):
        self.frames_dev, self.tab, self.plan, self.words, self.event = frames_dev, tab, plan, words, event


def begin(engine, frames_dev: torch.Tensor, dets: torch.Tensor, counts: torch.Tensor) -> Ticket:
    """Enqueue the label repair (``pa_clean_detections``), the plan of the crop hand-off (``pa_detector_plan``: only rows below
    the repair's max_frames count -- the tables behind them may be recycled memory) and the copy of the five host words on
    the current stream. Successive ``begin`` calls of one engine belong on one stream (the repair's scratch is the engine's)."""
    n_rows = dets.shape[0]
    tab = engine.clean_detections(dets, counts, frames_dev.shape[0])
    plan = engine.detector_plan(tab, n_rows)
    words = torch.empty(5, dtype=torch.int32).pin_memory()
    words.copy_(plan["words"], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(engine.device))
    return Ticket(frames_dev, tab, plan, words, ev)


def _crop_buffer(engine, nbytes: int) -> torch.Tensor:
    """The packed crop-image buffer, kept on the engine between clips (the worst case is large: every crop a whole frame).
    Successive ``finish`` calls on one engine therefore belong on ONE stream (stream order keeps clip k + 1's writes behind
    clip k's reads; the engine's activation buffers demand the same anyway)."""
    buf = getattr(engine, "_detector_crop_buf", None)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=engine.device)
        engine._detector_crop_buf = buf
    return buf


def finish(engine, t: Ticket, jpeg_quality: int = 95, want_crops: bool = False, device_results: bool = False) -> Dict:
    """Wait for the clip's five words, then enqueue crops -> runner inputs -> backbone -> head on the current stream."""
    t.event.synchronize()  # the only wait in the middle of a clip
    info = t.words.numpy().copy()
    if info[1]:
        raise LabelRepairError(f"{_REPAIR_ERRORS.get(int(info[1]), 'label repair failed')} (label {int(info[2])})")
    n, n_rep = int(info[0]), int(info[4])
    if n < 2:
        raise ValueError("no detections in any label")
    dev, F, tab, plan, frames_dev = engine.device, engine.F, t.tab, t.plan, t.frames_dev
    h, w = frames_dev.shape[1], frames_dev.shape[2]
    step = engine.max_batch_frames
    rc = getattr(engine, "_detector_row_counts", None)
    if rc is None or rc.numel() < n:
        rc = engine._detector_row_counts = torch.full((max(n, engine.max_clip_frames),), F, dtype=torch.int32, device=dev)
    # every chunk of frames packs its crop images into a region of its own (the offsets inside a region are the kernel's
    # prefix sums), so no chunk has to know where the one before it ended
    region = step * F * min(h * w, 1 << 20) * 3 + 64
    n_chunks = (n + step - 1) // step
    k_rep = (n_rep + F - 1) // F * F  # the repairs are cut F at a time (pa_square_crops_src); the plan padded their list
    sq_bytes = k_rep * 128 * 128 * 3
    images = _crop_buffer(engine, n_chunks * region + sq_bytes + 64)
    desc = torch.zeros((n * F, 2), dtype=torch.int64, device=dev)
    for k, f0 in enumerate(range(0, n, step)):
        cnt = min(step, n - f0)
        engine.save_one_box_crops(frames_dev, tab["crop_row"][f0:f0 + cnt], rc[f0:f0 + cnt], det_index=plan["det_index"][f0:f0 + cnt],
                                  jpeg_quality=jpeg_quality, src_frame=plan["src_own"][f0:f0 + cnt], images=images[k * region:(k + 1) * region],
                                  desc=desc[f0 * F:(f0 + cnt) * F])
    sq_status = None
    base = n_chunks * region
    if n_rep:
        # the square_crop repairs: their entries, boxes and source frames came out of the plan in entry order; their pixels are
        # cut on the device into the tail of the same buffer
        engine.set_crop_jpeg_quality(jpeg_quality)
        try:
            sq_status = engine.square_crops_src_device(frames_dev, plan["rep_boxes"], plan["rep_src"], k_rep, images[base:base + sq_bytes],
                                                       padding=engine.cfg.crop_padding)[:n_rep]
        finally:
            engine.set_crop_jpeg_quality(0)
    engine.detector_plan_desc(desc, tab["crop_kind"], n, step, region, plan["rep_entry"], n_rep, base)
    out = engine.infer_clip_from_packed_crop_images(images, desc, n, want_crops=want_crops, device_results=device_results)
    out["max_frames"] = n
    if device_results:
        out["cleaned"], out["info"], out["square_crop_status"] = tab, info, sq_status
        return out
    # (infer_clip_from_packed_crop_images synchronised: the checks below cost nothing more)
    engine.check_device_errors()
    if sq_status is not None:
        st = sq_status.cpu().numpy()
        if (st != 0).any():
            bad = int(plan["rep_entry"][int(np.nonzero(st)[0][0])]) // F
            raise AssertionError(f"Failed to get square crop from frame {bad + 1}")  # ai_runner.py:418
    out["cleaned"] = {k: (v[:n].cpu().numpy() if k != "info" else info[:4]) for k, v in tab.items()}
    return out


def run_detections_to_labels(engine, frames_dev: torch.Tensor, dets: torch.Tensor, counts: torch.Tensor,
                             jpeg_quality: int = 95, want_crops: bool = False) -> Dict:
    """frames uint8[n,H,W,3] (device), dets float32[n,max_det,6] / counts int32[n] (device, ``Engine.detect_postprocess``)
    -> the result dict of ``Engine.infer_clip`` for frames 1 .. max_frames - 1, plus ``cleaned`` (the repair tables, host
    copies) and ``max_frames``."""
    return finish(engine, begin(engine, frames_dev, dets, counts), jpeg_quality=jpeg_quality, want_crops=want_crops)
