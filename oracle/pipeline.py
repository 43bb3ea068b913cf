"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the inference loop of ``playaid/ai_runner.py:426-520`` on
in-memory synthetic frames: the label/crop *files* of the reference become
arrays (boxes[N,2,4], frames[N,H,W,3] BGR), the JPEG round trip is skipped,
everything else keeps the reference's order of operations:

  crop      = YoloCrop.square_crop(frame, 128, padding=30)   ai_runner.py:417-418
  window    = action_sample_from_frame_middle_out(f, 7, 3, max_frames, min_frame=1)  :430-439
  per slot  : BGR->RGB, imutils.resize(width=128), pad        :446-459
  x         = stack.permute(0,3,1,2)[None].float()/255        :461-463
  logp      = model(x) ; argmax ; exp()*100                   :472-477

Frames are 1-indexed like the YOLO files (``ai_runner.py:516``): frame_num f
uses ``frames[f-1]``. Two modes: ``literal`` (batch 1 per (frame, fighter), 7
backbone forwards per window -- the shape of ``run_action_recognition``) and
``cached`` (backbone once per crop, gather 7 cached feature rows per window --
legal because eval-mode BN makes crops independent).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import cnn, window, yolo_crop


def crops_for_clip(frames: np.ndarray, boxes: np.ndarray, padding: int = 30):
    """-> (crops_rgb uint8[N,2,128,128,3], ok bool[N,2]). Failed crops are zero."""
    n = frames.shape[0]
    nf = boxes.shape[1]
    out = np.zeros((n, nf, 128, 128, 3), dtype=np.uint8)
    ok = np.zeros((n, nf), dtype=bool)
    for i in range(n):
        for p in range(nf):
            res, crop = yolo_crop.square_crop(frames[i], boxes[i, p], 128, padding=padding)
            if res:
                out[i, p] = yolo_crop.runner_input_from_crop(crop)
                ok[i, p] = True
    return out, ok


def window_input(crops_rgb: np.ndarray, fighter: int, frame_nums: List[int], dtype=torch.float32) -> torch.Tensor:
    """``ai_runner.py:461-463``: -> [1,S,3,128,128] float /255."""
    frames = [crops_rgb[f - 1, fighter] for f in frame_nums]
    x = torch.tensor(np.array(frames)).permute(0, 3, 1, 2).unsqueeze(0)
    return x.to(dtype) / 255.0


def run_action_recognition(
    frames: np.ndarray,
    boxes: np.ndarray,
    sd: Dict,
    num_frames_per_sample: int = 7,
    frame_delta: int = 3,
    mode: str = "cached",
    dtype=torch.float32,
    frame_nums: Optional[List[int]] = None,
    crops_rgb: Optional[np.ndarray] = None,
):
    """-> dict(logp float[n_out,2,A], action_id int[n_out,2], confidence
    float[n_out,2], frame_nums list, crops_rgb). ``max_frames`` = N, frame_nums
    default to ``range(1, N)`` (``ai_runner.py:508``)."""
    n = frames.shape[0]
    nf = boxes.shape[1]
    max_frames = n
    if crops_rgb is None:
        crops_rgb, ok = crops_for_clip(frames, boxes)
        assert ok.all(), "oracle: a synthetic crop failed (reference asserts at ai_runner.py:418)"
    if frame_nums is None:
        frame_nums = list(range(1, max_frames))
    num_actions = np.asarray(sd["model.classifier.2.bias"]).shape[0]
    logp = np.zeros((len(frame_nums), nf, num_actions), dtype=np.float64)
    feats = None
    if mode == "cached":
        with torch.no_grad():
            x = torch.from_numpy(crops_rgb.reshape(n * nf, 128, 128, 3)).permute(0, 3, 1, 2).to(dtype) / 255.0
            fl = []
            for i in range(0, x.shape[0], 32):
                fl.append(cnn.resnet18_features(x[i : i + 32], sd))
            feats = torch.cat(fl).view(n, nf, -1)
    for oi, f in enumerate(frame_nums):
        idx = window.action_sample_from_frame_middle_out(
            f, num_frames_per_sample, frame_delta, max_frames, min_frame=1
        )
        for p in range(nf):
            with torch.no_grad():
                if mode == "literal":
                    lp = cnn.forward(window_input(crops_rgb, p, idx, dtype), sd)
                else:
                    wf = torch.stack([feats[j - 1, p] for j in idx])[None]
                    lp = F.log_softmax(cnn.head_logits(wf, sd), dim=1)
            logp[oi, p] = lp[0].double().numpy()
    action_id = logp.argmax(axis=2)
    conf = np.exp(np.take_along_axis(logp, action_id[..., None], axis=2))[..., 0] * 100.0
    return {
        "logp": logp,
        "action_id": action_id,
        "confidence": conf,
        "frame_nums": frame_nums,
        "crops_rgb": crops_rgb,
    }
