"""One convolution + bias + activation (+ residual) on the persistent implicit-GEMM kernels (``pa_conv2d``).

``compute_dtype="f32"`` runs ``csrc/pigemm.hip`` (the exact fp32 matrix instruction), ``"emulated_f32"`` runs
``csrc/psgemm.hip``: fp32 in, fp32 out, fp32-accurate sums, the products as six bf16 matrix instructions per fp32 product
(three bf16 slices per operand). These are the kernels the detection network's 1x1 / stride-2 layers
(``playaid/ai_runner.py:191-224``: YOLOv5s) and, under ``emulated_f32``, every 3x3 convolution of the ResNet-18
(``playaid/models/cnn_action_detector.py:16,32``) run on inside the library; this module exposes them as a single-layer
operator for parity tests and measurements. No CPU fallback: it needs the HIP library and a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def pack_weights(w_oihw: np.ndarray, compute_dtype: str = "emulated_f32", has_residual: bool = False) -> np.ndarray:
    """[cout, cin, k, k] fp32 (BatchNorm folded) -> the bytes the kernel of ``compute_dtype`` reads (uint8, host)."""
    lib = _lib.load()
    w = np.ascontiguousarray(np.asarray(w_oihw, dtype=np.float32).transpose(0, 2, 3, 1))   # [cout][ky][kx][cin]
    cout, k, _, cin = w.shape
    dt = _lib.DTYPES[compute_dtype]
    nbytes = lib.pa_conv_weight_bytes(cin, cout, k, dt, int(has_residual))
    if nbytes == 0:
        raise ValueError("cin and cout must be multiples of 32, the kernel 1x1 or 3x3, the dtype 'f32' or 'emulated_f32'")
    out = np.empty(nbytes, dtype=np.uint8)
    rc = lib.pa_conv_pack_weights(w.ctypes.data_as(C.c_void_p), cin, cout, k, dt, int(has_residual), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError(f"pa_conv_pack_weights: status {rc}")
    return out


def conv2d(x_pad: torch.Tensor, w_packed: torch.Tensor, cin: int, cout: int, ksize: int, stride: int = 1, in_pad: int = None, bias=None,
           residual=None, out=None, out_pad: int = 0, act: int = 0, res_after: bool = False, out_px_stride: int = None,
           compute_dtype: str = "emulated_f32") -> torch.Tensor:
    """x_pad float32[n, H + 2 in_pad, W + 2 in_pad, C >= cin] (device, zero border) -> out float32[n, H / stride + 2 out_pad,
    W / stride + 2 out_pad, C' >= cout] (interior written). ``w_packed``: ``pack_weights`` of the same dtype (and the same
    ``has_residual``), on the device. Enqueues on the current stream."""
    lib = _lib.load()
    if x_pad.dtype != torch.float32 or not x_pad.is_cuda or not x_pad.is_contiguous() or x_pad.dim() != 4:
        raise ValueError("x_pad: contiguous float32[n, H + 2 pad, W + 2 pad, C] on the device")
    in_pad = (ksize - 1) // 2 if in_pad is None else in_pad
    n, hp, wp, cs = x_pad.shape
    h, w = hp - 2 * in_pad, wp - 2 * in_pad
    oh, ow = h // stride, w // stride
    ops = out_px_stride or cout
    if out is None:
        out = torch.zeros((n, oh + 2 * out_pad, ow + 2 * out_pad, ops), dtype=torch.float32, device=x_pad.device)
    ptr = lambda t_: C.c_void_p(t_.data_ptr()) if t_ is not None else C.c_void_p(0)
    stream = C.c_void_p(torch.cuda.current_stream(x_pad.device).cuda_stream)
    rc = lib.pa_conv2d(ptr(x_pad), ptr(w_packed), ptr(bias), ptr(residual), ptr(out), n, h, w, cin, cout, ksize, stride, in_pad, cs, out.shape[3],
                       out_pad, int(act), int(bool(res_after)), _lib.DTYPES[compute_dtype], stream)
    if rc:
        raise ValueError(f"pa_conv2d: status {rc}")
    return out
