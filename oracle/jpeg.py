"""ORACLE (test infrastructure, never shipped, never on the product path).

The pixel arithmetic of a baseline JPEG write + read, i.e. what ``cv2.imwrite(path, crop)`` followed by
``cv2.imread(path)`` does to a crop in the reference (``playaid/ai_runner.py:420`` writes every repaired crop as a JPEG,
``:446`` reads every crop back before the CNN sees it; YOLOv5's ``--save-crop`` writes the others the same way). The
entropy coding in between is lossless, so the round trip is: colour conversion -> 2x2 chroma down-sampling -> 8x8
forward DCT -> quantisation | de-quantisation -> inverse DCT -> "fancy" chroma up-sampling -> colour conversion.

OpenCV is not vendored in /root/reference and not installed here; its JPEG codec is the bundled libjpeg(-turbo) with
OpenCV's defaults (quality 95, 4:2:0 chroma, ``JDCT_ISLOW``, fancy up-sampling). This module restates libjpeg's
published integer algorithms for exactly that configuration: ``jccolor.c`` (RGB->YCbCr tables), ``jcsample.c``
(``h2v2_downsample``), ``jfdctint.c``, ``jcdctmgr.c`` (rounded division by ``quantval << 3``), ``jcparam.c`` (standard
tables, quality scaling), ``jidctint.c``, ``jdsample.c`` (``h2v2_fancy_upsample``), ``jdcolor.c``.

Pinning: **pinned against the live library** -- Pillow 12.2 in this image links libjpeg-turbo 3.1 (whose SIMD paths are
bit-exact with these C algorithms); ``tests/test_oracle_jpeg.py`` compares ``roundtrip`` byte for byte with
``Image.save(quality=95, subsampling=2)`` + ``Image.open`` on fixed and random images. Only sizes that are multiples
of 16 are restated (the 128 x 128 crops of the path); that OpenCV's build of the library behaves like Pillow's is an
assumption (same library, same defaults), cv2 itself being absent.
"""
from __future__ import annotations

import numpy as np

STD_LUMA = np.array([
    16, 11, 10, 16, 24, 40, 51, 61,
    12, 12, 14, 19, 26, 58, 60, 55,
    14, 13, 16, 24, 40, 57, 69, 56,
    14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77,
    24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101,
    72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int64).reshape(8, 8)
STD_CHROMA = np.array([
    17, 18, 24, 47, 99, 99, 99, 99,
    18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99,
    47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99], dtype=np.int64).reshape(8, 8)

CONST_BITS, PASS1_BITS = 13, 2
F_0_298631336, F_0_390180644, F_0_541196100, F_0_765366865 = 2446, 3196, 4433, 6270
F_0_899976223, F_1_175875602, F_1_501321110, F_1_847759065 = 7373, 9633, 12299, 15137
F_1_961570560, F_2_053119869, F_2_562915447, F_3_072711026 = 16069, 16819, 20995, 25172


def quant_tables(quality: int):
    """``jpeg_set_quality(quality, force_baseline=TRUE)`` -> (luma[8,8], chroma[8,8]) in natural order."""
    quality = min(max(int(quality), 1), 100)
    scale = 5000 // quality if quality < 50 else 200 - quality * 2
    out = []
    for basic in (STD_LUMA, STD_CHROMA):
        t = (basic * scale + 50) // 100
        out.append(np.clip(t, 1, 255))
    return out[0], out[1]


def _fix(x: float) -> int:
    return int(x * 65536 + 0.5)


def rgb_to_ycc(rgb: np.ndarray):
    """jccolor.c ``rgb_ycc_convert``: uint8[h,w,3] (R, G, B) -> three int64[h,w] planes."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    half, off = 1 << 15, 128 << 16
    y = (_fix(0.29900) * r + _fix(0.58700) * g + _fix(0.11400) * b + half) >> 16
    cb = (-_fix(0.16874) * r - _fix(0.33126) * g + _fix(0.50000) * b + off + half - 1) >> 16
    cr = (_fix(0.50000) * r + off + half - 1 - _fix(0.41869) * g - _fix(0.08131) * b) >> 16
    return y, cb, cr


def h2v2_downsample(p: np.ndarray) -> np.ndarray:
    """jcsample.c: 2x2 box with the bias alternating 1, 2, 1, 2 along a row."""
    s = p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]
    bias = np.tile(np.array([1, 2], dtype=np.int64), s.shape[1] // 2 + 1)[: s.shape[1]]
    return (s + bias[None, :]) >> 2


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _fdct_1d(d, first_pass: bool):
    d0, d1, d2, d3, d4, d5, d6, d7 = d
    tmp0, tmp7, tmp1, tmp6 = d0 + d7, d0 - d7, d1 + d6, d1 - d6
    tmp2, tmp5, tmp3, tmp4 = d2 + d5, d2 - d5, d3 + d4, d3 - d4
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    if first_pass:
        o0, o4 = (tmp10 + tmp11) << PASS1_BITS, (tmp10 - tmp11) << PASS1_BITS
        n = CONST_BITS - PASS1_BITS
    else:
        o0, o4 = _descale(tmp10 + tmp11, PASS1_BITS), _descale(tmp10 - tmp11, PASS1_BITS)
        n = CONST_BITS + PASS1_BITS
    z1 = (tmp12 + tmp13) * F_0_541196100
    o2 = _descale(z1 + tmp13 * F_0_765366865, n)
    o6 = _descale(z1 + tmp12 * (-F_1_847759065), n)
    z1, z2, z3, z4 = tmp4 + tmp7, tmp5 + tmp6, tmp4 + tmp6, tmp5 + tmp7
    z5 = (z3 + z4) * F_1_175875602
    tmp4, tmp5, tmp6, tmp7 = tmp4 * F_0_298631336, tmp5 * F_2_053119869, tmp6 * F_3_072711026, tmp7 * F_1_501321110
    z1, z2, z3, z4 = z1 * -F_0_899976223, z2 * -F_2_562915447, z3 * -F_1_961570560, z4 * -F_0_390180644
    z3, z4 = z3 + z5, z4 + z5
    o7, o5 = _descale(tmp4 + z1 + z3, n), _descale(tmp5 + z2 + z4, n)
    o3, o1 = _descale(tmp6 + z2 + z3, n), _descale(tmp7 + z1 + z4, n)
    return [o0, o1, o2, o3, o4, o5, o6, o7]


def fdct_islow(blocks: np.ndarray) -> np.ndarray:
    """jfdctint.c on int64[..., 8, 8] blocks of (sample - 128): rows first, then columns; output scaled by 8."""
    rows = np.stack(_fdct_1d([blocks[..., :, k] for k in range(8)], True), axis=-1)
    return np.stack(_fdct_1d([rows[..., k, :] for k in range(8)], False), axis=-2)


def quantize(coef: np.ndarray, q: np.ndarray) -> np.ndarray:
    """jcdctmgr.c: round-half-away division of the scaled coefficients by ``quantval << 3``."""
    d = q << 3
    a = (np.abs(coef) + (d >> 1)) // d
    return np.where(coef < 0, -a, a)


def _idct_1d(v, first_pass: bool):
    i0, i1, i2, i3, i4, i5, i6, i7 = v
    z2, z3 = i2, i6
    z1 = (z2 + z3) * F_0_541196100
    tmp2 = z1 + z3 * (-F_1_847759065)
    tmp3 = z1 + z2 * F_0_765366865
    tmp0, tmp1 = (i0 + i4) << CONST_BITS, (i0 - i4) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = i7, i5, i3, i1
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F_1_175875602
    tmp0, tmp1, tmp2, tmp3 = tmp0 * F_0_298631336, tmp1 * F_2_053119869, tmp2 * F_3_072711026, tmp3 * F_1_501321110
    z1, z2, z3, z4 = z1 * -F_0_899976223, z2 * -F_2_562915447, z3 * -F_1_961570560, z4 * -F_0_390180644
    z3, z4 = z3 + z5, z4 + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    n = CONST_BITS - PASS1_BITS if first_pass else CONST_BITS + PASS1_BITS + 3
    return [_descale(tmp10 + tmp3, n), _descale(tmp11 + tmp2, n), _descale(tmp12 + tmp1, n), _descale(tmp13 + tmp0, n),
            _descale(tmp13 - tmp0, n), _descale(tmp12 - tmp1, n), _descale(tmp11 - tmp2, n), _descale(tmp10 - tmp3, n)]


def idct_islow(qcoef: np.ndarray, q: np.ndarray) -> np.ndarray:
    """jidctint.c with the de-quantisation folded in: columns first, then rows, range-limited to 0..255."""
    c = qcoef * q
    cols = np.stack(_idct_1d([c[..., k, :] for k in range(8)], True), axis=-2)
    rows = np.stack(_idct_1d([cols[..., :, k] for k in range(8)], False), axis=-1)
    return np.clip(rows + 128, 0, 255)


def h2v2_fancy_upsample(p: np.ndarray) -> np.ndarray:
    """jdsample.c: triangle filter (3/4, 1/4) in both directions, rounding 8 / 7 alternately; the rows above the first
    and below the last are the first / last row themselves (jdmainct.c's context rows)."""
    h, w = p.shape
    above = np.vstack([p[:1], p[:-1]])
    below = np.vstack([p[1:], p[-1:]])
    out = np.zeros((2 * h, 2 * w), dtype=np.int64)
    for v, nb in ((0, above), (1, below)):
        col = p * 3 + nb                      # column sums of this output row
        last = np.hstack([col[:, :1], col[:, :-1]])
        nxt = np.hstack([col[:, 1:], col[:, -1:]])
        even = (col * 3 + last + 8) >> 4
        odd = (col * 3 + nxt + 7) >> 4
        even[:, 0] = (col[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (col[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out


def ycc_to_rgb(y: np.ndarray, cb: np.ndarray, cr: np.ndarray) -> np.ndarray:
    """jdcolor.c ``ycc_rgb_convert`` -> uint8[h,w,3] (R, G, B)."""
    half = 1 << 15
    xb, xr = cb - 128, cr - 128
    r = y + ((_fix(1.40200) * xr + half) >> 16)
    g = y + ((-_fix(0.34414) * xb + half - _fix(0.71414) * xr) >> 16)
    b = y + ((_fix(1.77200) * xb + half) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def _blocks(p: np.ndarray) -> np.ndarray:
    h, w = p.shape
    return p.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)


def _unblocks(b: np.ndarray) -> np.ndarray:
    nh, nw = b.shape[:2]
    return b.transpose(0, 2, 1, 3).reshape(nh * 8, nw * 8)


def _plane_roundtrip(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    return _unblocks(idct_islow(quantize(fdct_islow(_blocks(p) - 128), q), q))


def roundtrip(rgb: np.ndarray, quality: int = 95) -> np.ndarray:
    """uint8[h,w,3] in R, G, B order (h, w multiples of 16) -> what a 4:2:0 baseline JPEG of that quality decodes to."""
    h, w, _ = rgb.shape
    if h % 16 or w % 16:
        raise ValueError("only multiples of 16 are restated (the 128 x 128 crops of the path)")
    ql, qc = quant_tables(quality)
    y, cb, cr = rgb_to_ycc(rgb)
    y2 = _plane_roundtrip(y, ql)
    cb2 = h2v2_fancy_upsample(_plane_roundtrip(h2v2_downsample(cb), qc))
    cr2 = h2v2_fancy_upsample(_plane_roundtrip(h2v2_downsample(cr), qc))
    return ycc_to_rgb(y2, cb2, cr2)


def roundtrip_bgr(bgr: np.ndarray, quality: int = 95) -> np.ndarray:
    """The same for OpenCV's channel order: ``cv2.imread(cv2.imwrite(bgr))``."""
    return np.ascontiguousarray(roundtrip(np.ascontiguousarray(bgr[..., ::-1]), quality)[..., ::-1])
