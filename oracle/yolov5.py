"""ORACLE (test infrastructure, never shipped, never on the product path).

The detection network the reference shells out to (``playaid/ai_runner.py:191-224``: ``python third_party/yolov5/detect.py
--weights <yolov5s checkpoint> --source <video> ...``). The checkout is un-vendored and unpinned (``constants.py:6``), its
weights absent; this file restates the PUBLISHED ultralytics/yolov5 v7.0 graph of ``models/yolov5s.yaml`` (``Conv`` =
conv + BatchNorm(eps 1e-3) + SiLU, ``C3``, ``Bottleneck``, ``SPPF``, ``Upsample``, ``Concat``, ``Detect`` in inference
mode) with the live ``torch.nn.functional`` CPU kernels on a state dict in the checkpoint's key layout, and the
``LoadImages`` front end (``letterbox`` + ``cv2.resize(INTER_LINEAR)`` restated from OpenCV's fixed-point bilinear
resizer, BGR -> RGB, / 255). **Parity unpinned**: nothing in the reference pins any of it -- it DEFINES the contract
the device path (``csrc/yolo.hip``, ``playaid_core_amd/yolov5.py``) matches; seeded synthetic weights.
"""
from __future__ import annotations

from typing import Mapping, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # ultralytics initialises every BatchNorm2d with eps = 1e-3


def _t(a) -> torch.Tensor:
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))


def conv(x, sd: Mapping, prefix: str, k: int, s: int = 1, act: bool = True):
    """``models/common.py::Conv``: Conv2d(k, s, autopad, bias=False) -> BatchNorm2d -> SiLU."""
    y = F.conv2d(x, _t(sd[prefix + ".conv.weight"]), None, s, k // 2)
    y = F.batch_norm(y, _t(sd[prefix + ".bn.running_mean"]), _t(sd[prefix + ".bn.running_var"]), _t(sd[prefix + ".bn.weight"]),
                     _t(sd[prefix + ".bn.bias"]), False, 0.0, BN_EPS)
    return F.silu(y) if act else y


def c3(x, sd, prefix: str, n: int, shortcut: bool):
    """``C3``: cv3(cat(m(cv1(x)), cv2(x))), m = n Bottlenecks (1x1 then 3x3, added to their input when ``shortcut``)."""
    y = conv(x, sd, prefix + ".cv1", 1)
    for j in range(n):
        z = conv(conv(y, sd, f"{prefix}.m.{j}.cv1", 1), sd, f"{prefix}.m.{j}.cv2", 3)
        y = y + z if shortcut else z
    return conv(torch.cat([y, conv(x, sd, prefix + ".cv2", 1)], 1), sd, prefix + ".cv3", 1)


def sppf(x, sd, prefix: str):
    x = conv(x, sd, prefix + ".cv1", 1)
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    return conv(torch.cat([x, y1, y2, F.max_pool2d(y2, 5, 1, 2)], 1), sd, prefix + ".cv2", 1)


STRIDES = (8.0, 16.0, 32.0)


def forward(x: torch.Tensor, sd: Mapping, nc: int) -> torch.Tensor:
    """x float32[n,3,H,W] (letter-boxed, RGB, 0..1; H, W multiples of 32) -> pred float32[n, rows, 5 + nc] as
    ``Detect.forward`` returns it in inference mode (the input of ``non_max_suppression``)."""
    with torch.no_grad():
        # v7.0's stem is Conv(c1, c2, k=6, s=2, p=2): the padding is explicit, not autopad(6) = 3
        y = F.conv2d(x, _t(sd["model.0.conv.weight"]), None, 2, 2)
        y = F.batch_norm(y, _t(sd["model.0.bn.running_mean"]), _t(sd["model.0.bn.running_var"]), _t(sd["model.0.bn.weight"]),
                         _t(sd["model.0.bn.bias"]), False, 0.0, BN_EPS)
        x0 = F.silu(y)
        x1 = conv(x0, sd, "model.1", 3, 2)
        x2 = c3(x1, sd, "model.2", 1, True)
        x3 = conv(x2, sd, "model.3", 3, 2)
        x4 = c3(x3, sd, "model.4", 2, True)
        x5 = conv(x4, sd, "model.5", 3, 2)
        x6 = c3(x5, sd, "model.6", 3, True)
        x7 = conv(x6, sd, "model.7", 3, 2)
        x8 = c3(x7, sd, "model.8", 1, True)
        x9 = sppf(x8, sd, "model.9")
        x10 = conv(x9, sd, "model.10", 1)
        x13 = c3(torch.cat([F.interpolate(x10, scale_factor=2, mode="nearest"), x6], 1), sd, "model.13", 1, False)
        x14 = conv(x13, sd, "model.14", 1)
        x17 = c3(torch.cat([F.interpolate(x14, scale_factor=2, mode="nearest"), x4], 1), sd, "model.17", 1, False)
        x18 = conv(x17, sd, "model.18", 3, 2)
        x20 = c3(torch.cat([x18, x14], 1), sd, "model.20", 1, False)
        x21 = conv(x20, sd, "model.21", 3, 2)
        x23 = c3(torch.cat([x21, x10], 1), sd, "model.23", 1, False)
        no, z = 5 + nc, []
        anchors = _t(sd["model.24.anchors"]).float()  # [3, 3, 2] in units of the scale's stride
        for i, f in enumerate((x17, x20, x23)):
            t = F.conv2d(f, _t(sd[f"model.24.m.{i}.weight"]), _t(sd[f"model.24.m.{i}.bias"]))
            bs, _, ny, nx = t.shape
            t = t.view(bs, 3, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            yv, xv = torch.meshgrid(torch.arange(ny, dtype=torch.float32), torch.arange(nx, dtype=torch.float32), indexing="ij")
            grid = torch.stack((xv, yv), 2).expand(1, 3, ny, nx, 2) - 0.5
            anchor_grid = (anchors[i] * STRIDES[i]).view(1, 3, 1, 1, 2).expand(1, 3, ny, nx, 2)
            xy, wh, conf = t.sigmoid().split((2, 2, nc + 1), 4)
            xy = (xy * 2 + grid) * STRIDES[i]
            wh = (wh * 2) ** 2 * anchor_grid
            z.append(torch.cat((xy, wh, conf), 4).view(bs, 3 * ny * nx, no))
        return torch.cat(z, 1)


def resize_linear_u8(src: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """``cv2.resize(src, (new_w, new_h), interpolation=cv2.INTER_LINEAR)`` on uint8[h, w, c], restated from OpenCV's
    ``resize.cpp`` (the non-IPP path): float source coordinates, 11-bit fixed-point coefficients, ``HResize`` into int,
    ``VResizeLinear``. (An exact 2x reduction is switched to INTER_AREA by OpenCV; both give ``(a + b + c + d + 2) >> 2``.)"""
    h, w, _ = src.shape
    sx = (np.float32(1) * ((np.arange(new_w) + 0.5) * (1.0 / (new_w / w)) - 0.5)).astype(np.float32)
    sy = (np.float32(1) * ((np.arange(new_h) + 0.5) * (1.0 / (new_h / h)) - 0.5)).astype(np.float32)

    def prep(f, size):
        i = np.floor(f).astype(np.int64)
        fr = (f - i.astype(np.float32)).astype(np.float32)
        lo = i < 0
        fr[lo], i[lo] = 0, 0
        hi = i >= size - 1
        fr[hi], i[hi] = 0, size - 1
        c1 = np.clip(np.rint(fr * np.float32(2048)), -32768, 32767).astype(np.int64)
        c0 = np.clip(np.rint((np.float32(1) - fr) * np.float32(2048)), -32768, 32767).astype(np.int64)
        return i, np.minimum(i + 1, size - 1), c0, c1

    x0, x1, a0, a1 = prep(sx, w)
    y0, y1, b0, b1 = prep(sy, h)
    s = src.astype(np.int64)
    rows = s[:, x0] * a0[None, :, None] + s[:, x1] * a1[None, :, None]          # [h, new_w, c]
    out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(frame_bgr: np.ndarray, net_hw: Tuple[int, int]) -> np.ndarray:
    """``LoadImages.__next__`` for one frame with the network input fixed to ``net_hw`` (what ``letterbox(auto=True)`` picks
    for the clip's frame size): resize to the un-padded size, 114-grey border, HWC BGR -> CHW RGB, / 255 -> float32[3,H,W]."""
    h, w = frame_bgr.shape[:2]
    r = min(net_hw[0] / h, net_hw[1] / w)
    new_w, new_h = int(round(w * r)), int(round(h * r))
    im = frame_bgr if (new_w, new_h) == (w, h) else resize_linear_u8(frame_bgr, new_w, new_h)
    dw, dh = (net_hw[1] - new_w) / 2, (net_hw[0] - new_h) / 2
    top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
    out = np.full((net_hw[0], net_hw[1], 3), 114, np.uint8)
    out[top:top + new_h, left:left + new_w] = im
    return np.ascontiguousarray(out[..., ::-1].transpose(2, 0, 1)).astype(np.float32) / np.float32(255)
