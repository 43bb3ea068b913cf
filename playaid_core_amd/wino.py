"""One stride-1 3x3 convolution on the Winograd F(2x2, 3x3) kernel (``csrc/wino.hip``, ``pa_wino_*``).

The engine's ResNet-18 (``playaid/models/cnn_action_detector.py:16,32``: torchvision ``resnet18``'s stride-1 3x3
convolutions) and the detector's Bottlenecks run on this kernel inside the library; this module exposes it as a
single-layer operator for parity tests and measurements. No CPU fallback: it needs the HIP library and a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def channels_per_workgroup(cout: int, sub_blocks: int) -> int:
    """The workgroup width (64 | 32 output channels) the engine picks for a layer launched over ``sub_blocks`` 4x4-pixel
    sub-blocks (``n * H / 4 * W / 4``): part of the filter layout and of the launch."""
    bn = _lib.load().pa_wino_channels_per_workgroup(cout, sub_blocks)
    if bn == 0:
        raise ValueError("cout must be a multiple of 32")
    return bn


def transform_weights(w_oihw: np.ndarray, bn: int = 0) -> np.ndarray:
    """[cout, cin, 3, 3] (BatchNorm already folded) -> the kernel's filter layout (float32, host) for workgroups of ``bn`` output
    channels (0: 64 where the layer has them)."""
    lib = _lib.load()
    w = np.ascontiguousarray(np.asarray(w_oihw, dtype=np.float32).transpose(0, 2, 3, 1))  # [cout][ky][kx][cin]
    cout, cin = w.shape[0], w.shape[3]
    n = lib.pa_wino_weight_floats(cin, cout)
    if n == 0:
        raise ValueError("cin must be a multiple of 8, cout a multiple of 32")
    bn = bn or (64 if cout % 64 == 0 else 32)
    ug = np.empty(n, dtype=np.float32)
    rc = lib.pa_wino_transform_weights(w.ctypes.data_as(C.c_void_p), cin, cout, bn, ug.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError(f"pa_wino_transform_weights: {rc}")
    return ug


class SplitKScratch:
    """Device scratch of ``pa_wino_conv3x3_splitk``: the slab the partial output tiles meet in (16 MB covers every launch) and the
    tickets (zero between launches). Launches that share one belong on one stream."""

    def __init__(self, device="cuda:0", slab_floats: int = 256 * 512 * 32, n_tickets: int = 4096):
        self.slab = torch.empty(slab_floats, dtype=torch.float32, device=device)
        self.tickets = torch.zeros(n_tickets, dtype=torch.int32, device=device)


def conv3x3(x_pad: torch.Tensor, ug: torch.Tensor, cin: int, cout: int, bias=None, residual=None, out=None, out_pad: int = 1,
            act: int = 0, res_after: bool = False, out_px_stride: int = None, bn: int = 0, split_k: SplitKScratch = None) -> torch.Tensor:
    """x_pad float32[n, H + 2, W + 2, C >= cin] (device, zero border) -> out float32[n, H + 2 out_pad, W + 2 out_pad, C'] (interior
    written, border untouched). Enqueues on the current stream. ``split_k``: scratch that lets the launcher split the input
    channels over several workgroups per tile where the tiles alone would not fill the chip."""
    lib = _lib.load()
    if x_pad.dtype != torch.float32 or not x_pad.is_cuda or not x_pad.is_contiguous() or x_pad.dim() != 4:
        raise ValueError("x_pad: contiguous float32[n, H + 2, W + 2, C] on the device")
    n, hp, wp, cs = x_pad.shape
    h, w = hp - 2, wp - 2
    ops = out_px_stride or cout
    if out is None:
        out = torch.zeros((n, h + 2 * out_pad, w + 2 * out_pad, ops), dtype=torch.float32, device=x_pad.device)
    ptr = lambda t_: C.c_void_p(t_.data_ptr()) if t_ is not None else C.c_void_p(0)
    stream = C.c_void_p(torch.cuda.current_stream(x_pad.device).cuda_stream)
    args = (ptr(x_pad), ptr(ug), ptr(bias), ptr(residual), ptr(out), n, h, w, cin, cout, bn or (64 if cout % 64 == 0 else 32), cs, out.shape[3], out_pad,
            int(act), int(bool(res_after)))
    if split_k is None:
        rc = lib.pa_wino_conv3x3(*args, stream)
    else:
        rc = lib.pa_wino_conv3x3_splitk(*args, ptr(split_k.slab), split_k.slab.numel(), ptr(split_k.tickets), split_k.tickets.numel(), stream)
    if rc:
        raise ValueError(f"pa_wino_conv3x3: status {rc}")
    return out
