"""The detection network on the device: YOLOv5s as a layer table for ``pa_detector_*``.

The reference gets its boxes from ``python third_party/yolov5/detect.py --weights models/yolo/<...>.pt --source <video>
--max-det 2 --classes 2 3 ...`` (``playaid/ai_runner.py:191-224``; the checkout and the weights are not in the reference
tree). This module takes a YOLOv5s v7.0 state dict in the checkpoint's key layout (``model.<i>.conv.weight``,
``model.<i>.bn.*``, ``model.<i>.cv1 ...``, ``model.24.m.<k>.{weight,bias}``, ``model.24.anchors``), folds every BatchNorm
(eps 1e-3) into its convolution in float64, lays the weights out for the implicit-GEMM kernel and wires the graph of
``models/yolov5s.yaml`` as ``pa_net_layer`` rows over zero-bordered NHWC buffers in which every concatenation is a buffer
the producers write their channel slice of:

    frames (device, uint8 BGR) -> pa_detector_forward: letterbox, 6x6/2 stem, 57 convolutions on the matrix cores, SPPF
    max-pools, two nearest up-samplings, Detect decode -> pred[n, rows, 5 + nc] -> Engine.detect_postprocess (NMS ...)

``YoloV5Detector.labels(engine, frames)`` returns the label text the runner reads, ``detections`` the device table that
``detector_path.run_detections_to_labels`` takes: decode -> detect -> repair -> crops -> CNN -> labels without leaving the GPU.
The arithmetic contract is ``oracle/yolov5.py`` (live torch CPU kernels on the same state dict; parity unpinned: no
checkpoint, no YOLOv5 checkout). fp32 throughout.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Mapping, Tuple

import numpy as np
import torch

from . import _lib
from .engine import EngineError

BN_EPS = 1e-3
WIDTHS = (32, 64, 128, 256, 512)      # yolov5s: width_multiple 0.50 of (64, 128, 256, 512, 1024)
STRIDES = (8, 16, 32)


def _np(a) -> np.ndarray:
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def _fold(sd: Mapping, prefix: str) -> Tuple[np.ndarray, np.ndarray]:
    """Conv (no bias) + BatchNorm(eval) -> (weight [c2, c1, k, k], bias [c2]) in float64."""
    w = _np(sd[prefix + ".conv.weight"]).astype(np.float64)
    g, b = _np(sd[prefix + ".bn.weight"]).astype(np.float64), _np(sd[prefix + ".bn.bias"]).astype(np.float64)
    m, v = _np(sd[prefix + ".bn.running_mean"]).astype(np.float64), _np(sd[prefix + ".bn.running_var"]).astype(np.float64)
    s = g / np.sqrt(v + BN_EPS)
    return w * s[:, None, None, None], b - m * s


class _Table:
    """Buffers, weights and layers while the graph is wired."""

    def __init__(self):
        self.layers: List[_lib.pa_net_layer] = []
        self.bufs: List[Tuple[int, int, int, int]] = []   # (h, w, pad, channels)
        self.weights: List[np.ndarray] = []
        self.n_weights = 0

    def buf(self, h, w, pad, c) -> int:
        self.bufs.append((h, w, pad, c))
        return len(self.bufs) - 1

    def put(self, a: np.ndarray) -> int:
        off = self.n_weights
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        pad = (-a.size) % 4   # 16-byte aligned rows for the vector loads
        self.weights.append(a)
        if pad:
            self.weights.append(np.zeros(pad, np.float32))
        self.n_weights += a.size + pad
        return off

    def conv(self, w, b, src, dst, k, s, act=2, res=None, res_after=0):
        """w [cout, cin, k, k] / b [cout] float64 (already padded to the kernel's multiples); src / dst / res = (buf, coff, c)."""
        cout, cin = w.shape[0], w.shape[1]
        assert cin % 32 == 0 and cout % 32 == 0 and src[2] == cin and dst[2] == cout, (w.shape, src, dst)
        ih, iw, ipad, ic = self.bufs[src[0]]
        oh, ow, opad, oc = self.bufs[dst[0]]
        assert (ih // s, iw // s) == (oh, ow) and ipad >= k // 2, (self.bufs[src[0]], self.bufs[dst[0]], k, s)
        L = _lib.pa_net_layer()
        L.kind, L.cin, L.cout, L.ksize, L.stride, L.in_h, L.in_w = 0, cin, cout, k, s, ih, iw
        L.in_buf, L.in_coff, L.in_cstride, L.in_pad = src[0], src[1], ic, ipad
        L.out_buf, L.out_coff, L.out_cstride, L.out_pad = dst[0], dst[1], oc, opad
        L.res_buf, L.res_coff = (res[0], res[1]) if res is not None else (-1, 0)
        if res is not None:
            assert self.bufs[res[0]] == self.bufs[dst[0]] and res[2] == cout
        L.act, L.res_after = act, res_after
        L.w_off = self.put(w.transpose(0, 2, 3, 1))
        L.b_off = self.put(b)
        self.layers.append(L)

    def move(self, kind, src, dst):
        ih, iw, ipad, ic = self.bufs[src[0]]
        oh, ow, opad, oc = self.bufs[dst[0]]
        L = _lib.pa_net_layer()
        L.kind, L.cin, L.cout, L.ksize, L.stride, L.in_h, L.in_w = kind, src[2], src[2], 5 if kind == 4 else 1, 1, ih, iw
        L.in_buf, L.in_coff, L.in_cstride, L.in_pad = src[0], src[1], ic, ipad
        L.out_buf, L.out_coff, L.out_cstride, L.out_pad = dst[0], dst[1], oc, opad
        L.res_buf = -1
        assert src[2] == dst[2] and (oh, ow) == ((ih * 2, iw * 2) if kind == 5 else (ih, iw))
        self.layers.append(L)


def _pad_rows(w, b, cout):
    if w.shape[0] == cout:
        return w, b
    wp = np.zeros((cout,) + w.shape[1:], np.float64)
    wp[: w.shape[0]] = w
    bp = np.zeros(cout, np.float64)
    bp[: b.shape[0]] = b
    return wp, bp


def build_yolov5s_table(sd: Mapping, net_hw: Tuple[int, int], nc: int):
    """-> (layers, buffer sizes in floats per image, weight blob float32, rows of pred per image)."""
    H, W = net_hw
    assert H % 32 == 0 and W % 32 == 0
    T = _Table()
    c1, c2, c3, c4, c5 = WIDTHS
    size = {2: (H // 2, W // 2), 4: (H // 4, W // 4), 8: (H // 8, W // 8), 16: (H // 16, W // 16), 32: (H // 32, W // 32)}

    def B(scale, pad, c):
        return T.buf(size[scale][0], size[scale][1], pad, c)

    def C3(prefix, src, dst, scale, cin, cout, n, shortcut):
        """src / dst slices; writes cv3's output into dst. cv1 and cv2 read the same input, so ONE convolution writes
        X = [cv1 | cv2]; every bottleneck reads X's first half through its 1x1 into R and its 3x3 writes that half back IN
        PLACE (adding it as the residual after the SiLU when the block has shortcuts), so cv3's input [bottlenecks | cv2] is
        X itself: no concatenation, no ping-pong buffers, src read once. (c_ = 32, model.2: the bottleneck's convolutions run
        on 32-channel tiles.)"""
        c_ = cout // 2
        w1, b1 = _fold(sd, prefix + ".cv1")
        w2, b2 = _fold(sd, prefix + ".cv2")
        w3, b3 = _fold(sd, prefix + ".cv3")
        X = B(scale, 0, 2 * c_)
        R = B(scale, 1, c_)
        T.conv(np.concatenate([w1, w2]), np.concatenate([b1, b2]), src, (X, 0, 2 * c_), 1, 1)
        for j in range(n):
            wa, ba = _fold(sd, f"{prefix}.m.{j}.cv1")
            T.conv(wa, ba, (X, 0, c_), (R, 0, c_), 1, 1)
            wb, bb = _fold(sd, f"{prefix}.m.{j}.cv2")
            T.conv(wb, bb, (R, 0, c_), (X, 0, c_), 3, 1, res=(X, 0, c_) if shortcut else None, res_after=1)
        T.conv(w3, b3, (X, 0, 2 * c_), dst, 1, 1)

    def conv(prefix, src, dst, k, s):
        w, b = _fold(sd, prefix)
        T.conv(w, b, src, dst, k, s)

    # concatenation buffers of the head (the backbone writes its skip connections straight into them)
    C12 = B(16, 1, 2 * c4)   # [up(model.10) | model.6]
    C16 = B(8, 1, 2 * c3)    # [up(model.14) | model.4]
    C19 = B(16, 0, 2 * c3)   # [model.18 | model.14]
    C22 = B(32, 0, 2 * c4)   # [model.21 | model.10]
    # backbone
    assert c1 == 32, "the direct stem kernel is built for 32 output channels"
    B0 = B(2, 1, c1)
    w0, b0 = _fold(sd, "model.0")
    # stem weights in the direct kernel's lane layout: lane = (kx half) * 32 + channel holds W[channel][c][ky][3 * half + j]
    # at ky * 9 + j * 3 + c (csrc/yolo.hip::stem6x6_direct_kernel)
    stem = np.zeros((2, 32, 56), np.float64)
    for half in range(2):
        stem[half, :, :54] = w0[:, :, :, 3 * half:3 * half + 3].transpose(0, 2, 3, 1).reshape(32, 54)
    L = _lib.pa_net_layer()
    L.kind, L.cin, L.cout, L.ksize, L.stride, L.in_h, L.in_w = 3, 3, c1, 6, 2, H, W
    L.in_buf, L.res_buf = -1, -1
    L.out_buf, L.out_coff, L.out_cstride, L.out_pad = B0, 0, c1, 1
    L.act = 2
    L.w_off, L.b_off = T.put(stem), T.put(b0)
    T.layers.append(L)
    B1 = B(4, 0, c2)
    conv("model.1", (B0, 0, c1), (B1, 0, c2), 3, 2)
    B2 = B(4, 1, c2)
    C3("model.2", (B1, 0, c2), (B2, 0, c2), 4, c2, c2, 1, True)
    B3 = B(8, 0, c3)
    conv("model.3", (B2, 0, c2), (B3, 0, c3), 3, 2)
    C3("model.4", (B3, 0, c3), (C16, c3, c3), 8, c3, c3, 2, True)
    B5 = B(16, 0, c4)
    conv("model.5", (C16, c3, c3), (B5, 0, c4), 3, 2)
    C3("model.6", (B5, 0, c4), (C12, c4, c4), 16, c4, c4, 3, True)
    B7 = B(32, 0, c5)
    conv("model.7", (C12, c4, c4), (B7, 0, c5), 3, 2)
    B8 = B(32, 0, c5)
    C3("model.8", (B7, 0, c5), (B8, 0, c5), 32, c5, c5, 1, True)
    # SPPF: x | pool(x) | pool(pool(x)) | pool(pool(pool(x))) side by side
    S9 = B(32, 0, 2 * c5)
    conv("model.9.cv1", (B8, 0, c5), (S9, 0, c4), 1, 1)
    for k in range(3):
        T.move(4, (S9, k * c4, c4), (S9, (k + 1) * c4, c4))
    B9 = B(32, 0, c5)
    conv("model.9.cv2", (S9, 0, 2 * c5), (B9, 0, c5), 1, 1)
    # head
    conv("model.10", (B9, 0, c5), (C22, c4, c4), 1, 1)
    T.move(5, (C22, c4, c4), (C12, 0, c4))
    B13 = B(16, 0, c4)
    C3("model.13", (C12, 0, 2 * c4), (B13, 0, c4), 16, 2 * c4, c4, 1, False)
    conv("model.14", (B13, 0, c4), (C19, c3, c3), 1, 1)
    T.move(5, (C19, c3, c3), (C16, 0, c3))
    B17 = B(8, 1, c3)
    C3("model.17", (C16, 0, 2 * c3), (B17, 0, c3), 8, 2 * c3, c3, 1, False)
    conv("model.18", (B17, 0, c3), (C19, 0, c3), 3, 2)
    B20 = B(16, 1, c4)
    C3("model.20", (C19, 0, 2 * c3), (B20, 0, c4), 16, 2 * c3, c4, 1, False)
    conv("model.21", (B20, 0, c4), (C22, 0, c4), 3, 2)
    B23 = B(32, 0, c5)
    C3("model.23", (C22, 0, 2 * c4), (B23, 0, c5), 32, 2 * c4, c5, 1, False)
    # Detect: one 1x1 convolution per scale (3 x (5 + nc) channels, padded to 64), then the decode
    no = 5 + nc
    assert 3 * no <= 64, "more classes than the 64-channel head slice holds"
    anchors = _np(sd["model.24.anchors"]).astype(np.float64)
    rows = 0
    for i, (feat, ch, scale) in enumerate(((B17, c3, 8), (B20, c4, 16), (B23, c5, 32))):
        D = B(scale, 0, 64)
        w = _np(sd[f"model.24.m.{i}.weight"]).astype(np.float64)
        b = _np(sd[f"model.24.m.{i}.bias"]).astype(np.float64)
        T.conv(*_pad_rows(w, b, 64), (feat, 0, ch), (D, 0, 64), 1, 1, act=0)
        L = _lib.pa_net_layer()
        L.kind, L.cin, L.cout, L.ksize, L.stride = 6, 64, 3 * no, 1, 1
        L.in_h, L.in_w = size[scale]
        L.in_buf, L.in_coff, L.in_cstride, L.in_pad = D, 0, 64, 0
        L.out_buf, L.res_buf = -1, -1
        L.aux[0] = float(STRIDES[i])
        for a in range(3):
            L.aux[1 + 2 * a] = float(anchors[i, a, 0] * STRIDES[i])
            L.aux[2 + 2 * a] = float(anchors[i, a, 1] * STRIDES[i])
        T.layers.append(L)
        rows += 3 * size[scale][0] * size[scale][1]
    buf_floats = [(h + 2 * p) * (w + 2 * p) * c for (h, w, p, c) in T.bufs]
    return T.layers, buf_floats, np.concatenate(T.weights), rows


class YoloV5Detector:
    """``pa_detector_*`` handle for a YOLOv5s state dict. ``net_hw``: the network input (what ``letterbox(auto=True)`` picks
    for the clip: 384 x 640 for 16:9 frames at ``--imgsz 640``)."""

    def __init__(self, state_dict: Mapping, nc: int, net_hw: Tuple[int, int] = (384, 640), max_images: int = 64, device: str = "cuda:0",
                 compute_dtype: str = "f32"):
        self._lib = _lib.load()
        if compute_dtype not in ("f32", "emulated_f32"):
            raise ValueError("compute_dtype must be 'f32' or 'emulated_f32'")
        self.compute_dtype = compute_dtype
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("no HIP device visible to PyTorch-ROCm; the detection network has no CPU fallback")
        self.device = torch.device(device)
        self.nc, self.net_hw, self.max_images = nc, tuple(net_hw), max_images
        layers, buf_floats, weights, rows = build_yolov5s_table(state_dict, self.net_hw, nc)
        self.rows = rows
        arr = (_lib.pa_net_layer * len(layers))(*layers)
        bf = (C.c_int64 * len(buf_floats))(*buf_floats)
        h = C.c_void_p()
        torch.cuda.set_device(self.device)
        rc = self._lib.pa_detector_create_dtype(self.device.index or 0, arr, len(layers), bf, len(buf_floats), weights.ctypes.data_as(C.c_void_p),
                                                weights.size, max_images, self.net_hw[0], self.net_hw[1], nc, _lib.DTYPES[compute_dtype], C.byref(h))
        self._h = h
        if rc != 0:
            msg = self._lib.pa_detector_last_error(h).decode() if h else "bad argument"
            self.close()
            raise EngineError(rc, msg)
        assert self._lib.pa_detector_rows(self._h) == rows
        self.n_layers = len(layers)
        # multiply-adds the table executes per image (padding channels included) and the ones the graph defines
        self.flops_per_image = 0.0
        for L in layers:
            oh, ow = L.in_h // max(L.stride, 1), L.in_w // max(L.stride, 1)
            if L.kind == 0:
                self.flops_per_image += 2.0 * oh * ow * L.cout * L.ksize * L.ksize * L.cin
            elif L.kind == 3:
                self.flops_per_image += 2.0 * oh * ow * L.cout * 108

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pa_detector_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, frames) -> torch.Tensor:
        """frames uint8[n,H,W,3] BGR (device tensor or numpy) -> pred float32[n, rows, 5 + nc] on the device."""
        fd = frames if isinstance(frames, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frames))
        fd = fd.to(self.device).contiguous()
        n, h, w, _ = fd.shape
        out = torch.empty((n, self.rows, 5 + self.nc), dtype=torch.float32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        for f0 in range(0, n, self.max_images):
            cnt = min(self.max_images, n - f0)
            rc = self._lib.pa_detector_forward(self._h, C.c_void_p(fd[f0:].data_ptr()), cnt, h, w, C.c_void_p(out[f0:].data_ptr()), stream)
            if rc != 0:
                raise EngineError(rc, self._lib.pa_detector_last_error(self._h).decode())
        return out

    __call__ = forward

    def detections(self, engine, frames, conf_thres: float = 0.25, iou_thres: float = 0.45, classes=(2, 3), max_det: int = 2):
        """-> (dets float32[n, max_det, 6], counts int32[n]) on the device: ``detect.py``'s label rows (``ai_runner.py:209-217``)."""
        fd = frames if isinstance(frames, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frames)).to(self.device)
        pred = self.forward(fd)
        return engine.detect_postprocess(pred, self.net_hw, (fd.shape[1], fd.shape[2]), conf_thres, iou_thres, classes, max_det)

    def labels(self, engine, frames, **kw) -> List[str]:
        from .detect import label_lines

        dets, counts = self.detections(engine, frames, **kw)
        torch.cuda.synchronize(self.device)
        d, c = dets.cpu().numpy(), counts.cpu().numpy()
        return [label_lines(d[i, : c[i]]) for i in range(d.shape[0])]
