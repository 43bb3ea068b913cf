"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement, in numpy integer/float arithmetic, of the two third-party
resamplers the reference's crop path calls. Neither library's source lives in
/root/reference; both are restated from their published algorithms.

1. Pillow ``Image.resize(..., BICUBIC)`` for 8-bit images, reached through
   ``ImageOps.pad`` at ``playaid/fighter.py:349-355,371-373`` and
   ``playaid/ai_runner.py:454-456``. Algorithm: Pillow ``src/libImaging/Resample.c``
   (``precompute_coeffs`` / ``normalize_coeffs_8bpc`` / ``ImagingResampleHorizontal_8bpc``
   / ``ImagingResampleVertical_8bpc``): separable two-pass (horizontal first),
   Keys cubic a=-0.5 with support 2*max(scale,1), coefficients normalised in
   double then rounded to 22-bit fixed point, u8 rounding between the passes.
   PINNED: Pillow (12.2.0) is importable in the build container and on the GPU
   box, and tests/test_oracle_resample.py checks this restatement bit-for-bit
   against live ``Image.resize`` / ``ImageOps.pad``.

2. OpenCV ``cv2.resize(..., interpolation=INTER_AREA)`` for 8-bit 3-channel
   images when shrinking, reached through ``imutils.resize`` (imutils 0.5.4
   ``convenience.py``: ``dim = (width, int(h * width / float(w)))``) at
   ``playaid/fighter.py:364`` and ``playaid/ai_runner.py:450``. Algorithm:
   OpenCV 4.5.5 ``modules/imgproc/src/resize.cpp`` (``cv::resize`` dispatch,
   ``ResizeAreaFastVec`` 2x2 path, ``ResizeAreaFast_Invoker`` integer-scale
   path, ``computeResizeAreaTab`` + ``ResizeArea_Invoker`` fractional path in
   fp32). PARITY UNPINNED: cv2 is not installed here, so this restatement is
   checked only through properties (constant images, integer-scale box means,
   weights summing to one) -- see DESIGN.md. The enlarging branch
   (scale < 1, which OpenCV routes to its fixed-point linear resizer with
   area-mode coefficients) is restated in ``_cv_resize_area_enlarge`` below,
   equally unpinned.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2


# ----------------------------------------------------------------------------
# Pillow BICUBIC (8 bits per channel)
# ----------------------------------------------------------------------------

def _bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_coeffs(in_size: int, out_size: int):
    """Return (ksize, bounds[out,2] (xmin, count), kk[out,ksize] int32 fixed-point)
    for the full-image box, exactly as Resample.c computes them (C doubles ==
    Python floats, same operation order)."""
    in0, in1 = 0.0, float(in_size)
    scale = (in1 - in0) / out_size
    filterscale = scale
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)  # C (int) cast truncates toward 0
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        k = [0.0] * ksize
        for x in range(xmax):
            w = _bicubic_filter((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        for x in range(ksize):
            v = k[x]
            if v < 0:
                kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS))
            else:
                kk[xx, x] = int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx, 0] = xmin
        bounds[xx, 1] = xmax
    return ksize, bounds, kk


def _resample_axis1(img: np.ndarray, out_size: int) -> np.ndarray:
    """One 8bpc pass along axis 1 of ``img[rows, in_size, C]``."""
    in_size = img.shape[1]
    ksize, bounds, kk = pil_bicubic_coeffs(in_size, out_size)
    idx = bounds[:, 0:1].astype(np.int64) + np.arange(ksize, dtype=np.int64)[None, :]
    valid = np.arange(ksize)[None, :] < bounds[:, 1:2]
    idx = np.where(valid, idx, 0)
    coef = np.where(valid, kk, 0)  # [out, ksize]
    src = img.astype(np.int64)
    acc = np.full((img.shape[0], out_size, img.shape[2]), 1 << (PRECISION_BITS - 1), dtype=np.int64)
    for t in range(ksize):
        acc += src[:, idx[:, t], :] * coef[None, :, t, None]
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize_bicubic(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``Image.fromarray(img).resize((out_w, out_h), BICUBIC)`` for uint8 HWC."""
    h, w = img.shape[:2]
    if (w, h) == (out_w, out_h):
        return img.copy()
    out = img
    need_h = out_w != w
    need_v = out_h != h
    if need_h:
        if need_v:
            # Pillow only resamples the rows the vertical pass will read; the
            # values of those rows are the same as resampling all of them.
            pass
        out = _resample_axis1(out, out_w)
    if need_v:
        out = _resample_axis1(out.transpose(1, 0, 2), out_h).transpose(1, 0, 2)
    return np.ascontiguousarray(out)


def py_round(x: float) -> int:
    """Python 3 ``round`` (banker's rounding), as ImageOps uses."""
    return int(round(x))


def pil_contain_size(w: int, h: int, size: Tuple[int, int]) -> Tuple[int, int]:
    """Target size chosen by ``ImageOps.contain`` (Pillow ``ImageOps.py``)."""
    im_ratio = w / h
    dest_ratio = size[0] / size[1]
    if im_ratio != dest_ratio:
        if im_ratio > dest_ratio:
            new_height = py_round(h / w * size[0])
            if new_height != size[1]:
                size = (size[0], new_height)
        else:
            new_width = py_round(w / h * size[1])
            if new_width != size[0]:
                size = (new_width, size[1])
    return size


def pil_pad_black(img: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """``np.array(ImageOps.pad(Image.fromarray(img), size, color="black"))``.
    Raises ValueError for an empty input (the reference maps that to
    ``(False, None)`` at ``fighter.py:356-357``; Pillow 12 raises
    ZeroDivisionError for a zero-height slice and ValueError for a zero-width
    one -- both are treated as "no crop" here)."""
    h, w = img.shape[:2]
    if h == 0 or w == 0 or size[0] <= 0 or size[1] <= 0:
        raise ValueError("empty image")
    rw, rh = pil_contain_size(w, h, size)
    if rw <= 0 or rh <= 0:
        raise ValueError("height and width must be > 0")
    resized = pil_resize_bicubic(img, rw, rh)
    if (rw, rh) == tuple(size):
        return resized
    out = np.zeros((size[1], size[0], img.shape[2]), dtype=np.uint8)
    if rw != size[0]:
        x = py_round((size[0] - rw) * 0.5)
        out[:, x : x + rw] = resized
    else:
        y = py_round((size[1] - rh) * 0.5)
        out[y : y + rh, :] = resized
    return out


# ----------------------------------------------------------------------------
# OpenCV INTER_AREA (8UC3, shrinking)
# ----------------------------------------------------------------------------

def cv_area_tab(ssize: int, dsize: int, scale: float):
    """``computeResizeAreaTab``: list of (si, di, alpha float32)."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1 = int(math.ceil(fsx1))
        sx2 = int(math.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((sx1 - 1, dx, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((sx, dx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((sx2, dx, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _cv_round_half_even(x: np.ndarray) -> np.ndarray:
    """``saturate_cast<uchar>(float)``: cvRound (round-half-even) then clamp."""
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def cv_linear_area_coeffs(ssize: int, dsize: int):
    """Offsets and 11-bit fixed-point weights of OpenCV's bilinear resizer in ``area_mode``
    (cv::hal::resize, generic path, ``interpolation == INTER_AREA`` with a scale < 1: "true area
    interpolation is only implemented for scale >= 1, in other cases it is emulated using some
    variant of bilinear"): ``s = floor(d * scale)``, ``f = (d + 1) - (s + 1) * inv_scale``,
    ``f = 0 if f <= 0 else f - floor(f)`` (float32), weights ``cvRound((1 - f) * 2048)``,
    ``cvRound(f * 2048)``. -> (ofs int[dsize], w0 int[dsize], w1 int[dsize], first index whose
    right neighbour would fall outside the source)."""
    inv_scale = dsize / ssize
    scale = 1.0 / inv_scale
    ofs = np.zeros(dsize, dtype=np.int64)
    w0 = np.zeros(dsize, dtype=np.int64)
    w1 = np.zeros(dsize, dtype=np.int64)
    dmax = dsize
    for d in range(dsize):
        s = int(math.floor(d * scale))
        f = np.float32((d + 1) - (s + 1) * inv_scale)
        f = np.float32(0.0) if f <= 0 else np.float32(f - np.float32(math.floor(float(f))))
        if s + 1 >= ssize:
            dmax = min(dmax, d)
            if s >= ssize - 1:
                f = np.float32(0.0)
                s = ssize - 1
        ofs[d] = s
        w0[d] = int(np.rint(np.float32(np.float32(1.0) - f) * np.float32(2048.0)))
        w1[d] = int(np.rint(f * np.float32(2048.0)))
    return ofs, w0, w1, dmax


def _cv_resize_area_enlarge(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """INTER_AREA with a destination larger than the source: the 8-bit fixed-point bilinear
    resizer (HResizeLinear into int rows scaled by 2048, VResizeLinear
    ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``) with the area-mode
    coefficients above. **Parity unpinned**: restated from the OpenCV 4.5.5 sources as
    remembered, no cv2 in the build container to check against."""
    h, w, cn = img.shape
    xofs, a0, a1, xmax = cv_linear_area_coeffs(w, out_w)
    yofs, b0, b1, _ = cv_linear_area_coeffs(h, out_h)
    src = img.astype(np.int64)
    hbuf = np.zeros((h, out_w, cn), dtype=np.int64)
    for dx in range(out_w):
        sx = xofs[dx]
        if dx < xmax:
            hbuf[:, dx] = src[:, sx] * a0[dx] + src[:, sx + 1] * a1[dx]
        else:
            hbuf[:, dx] = src[:, sx] * 2048
    out = np.zeros((out_h, out_w, cn), dtype=np.uint8)
    for dy in range(out_h):
        # the row coefficients are NOT zeroed at the bottom edge (only the row index is clipped)
        inv = out_h / h
        sy = int(math.floor(dy * (1.0 / inv)))
        f = np.float32((dy + 1) - (sy + 1) * inv)
        f = np.float32(0.0) if f <= 0 else np.float32(f - np.float32(math.floor(float(f))))
        c0 = int(np.rint(np.float32(np.float32(1.0) - f) * np.float32(2048.0)))
        c1 = int(np.rint(f * np.float32(2048.0)))
        r0 = min(max(sy, 0), h - 1)
        r1 = min(max(sy + 1, 0), h - 1)
        v = (((c0 * (hbuf[r0] >> 4)) >> 16) + ((c1 * (hbuf[r1] >> 4)) >> 16) + 2) >> 2
        out[dy] = (v & 0xFF).astype(np.uint8)
    return out


def cv_resize_area(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_AREA)`` for a
    uint8 HWC image (shrinking: box / area-weighted paths; enlarging: the bilinear emulation)."""
    h, w, cn = img.shape
    if (w, h) == (out_w, out_h):
        return img.copy()
    inv_sx = out_w / w
    inv_sy = out_h / h
    scale_x = 1.0 / inv_sx
    scale_y = 1.0 / inv_sy
    if scale_x < 1.0 or scale_y < 1.0:
        return _cv_resize_area_enlarge(img, out_w, out_h)
    eps = np.finfo(np.float64).eps
    isx = int(np.rint(scale_x))  # saturate_cast<int>(double) == cvRound
    isy = int(np.rint(scale_y))
    if abs(scale_x - isx) < eps and abs(scale_y - isy) < eps:
        # integer-scale "fast" paths. Destination columns/rows beyond
        # (ssize / iscale) cannot occur for an exact integer ratio.
        blocks = img[: out_h * isy, : out_w * isx].reshape(out_h, isy, out_w, isx, cn).astype(np.int64)
        s = blocks.sum(axis=(1, 3))
        if isx == 2 and isy == 2:
            return ((s + 2) >> 2).astype(np.uint8)
        scale = np.float32(1.0) / np.float32(isx * isy)
        return _cv_round_half_even(s.astype(np.float32) * scale)
    xtab = cv_area_tab(w, out_w, scale_x)
    ytab = cv_area_tab(h, out_h, scale_y)
    src = img.astype(np.float32)
    # horizontal accumulation, in table order, all in fp32 (mul then add)
    buf = np.zeros((h, out_w, cn), dtype=np.float32)
    for si, di, a in xtab:
        buf[:, di, :] = buf[:, di, :] + src[:, si, :] * a
    out = np.zeros((out_h, out_w, cn), dtype=np.uint8)
    prev_dy = ytab[0][1]
    acc = np.zeros((out_w, cn), dtype=np.float32)
    for sy, dy, beta in ytab:
        if dy != prev_dy:
            out[prev_dy] = _cv_round_half_even(acc)
            acc = beta * buf[sy]
            prev_dy = dy
        else:
            acc = acc + beta * buf[sy]
    out[prev_dy] = _cv_round_half_even(acc)
    return out


def imutils_resize_width(img: np.ndarray, width: int) -> np.ndarray:
    """``imutils.resize(img, width=width)``: aspect-preserving INTER_AREA with
    ``dim = (width, int(h * (width / float(w))))`` (imutils 0.5.4)."""
    h, w = img.shape[:2]
    r = width / float(w)
    dim = (width, int(h * r))
    return cv_resize_area(img, dim[0], dim[1])
