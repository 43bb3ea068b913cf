"""Pin the oracle's resampler restatements (oracle/resample.py).

Pillow half: bit-exact against LIVE Pillow (the reference's own dependency,
fighter.py:349-355) over many sizes. OpenCV INTER_AREA half: cv2 is not
installed, so only properties of the published algorithm are checked
("parity unpinned")."""
import numpy as np
import pytest
from PIL import Image, ImageOps

from oracle import resample as R
from oracle import yolo_crop

RNG = np.random.default_rng(42)


@pytest.mark.parametrize(
    "h,w,ow,oh",
    [(374, 374, 315, 315), (60, 80, 33, 71), (200, 190, 128, 127), (374, 300, 253, 315), (129, 129, 128, 128),
     (500, 640, 640, 500), (100, 100, 250, 250), (7, 5, 3, 2), (315, 315, 315, 200), (636, 636, 576, 576),
     (188, 188, 128, 128), (247, 249, 324, 320)],
)
def test_bicubic_resize_matches_live_pillow(h, w, ow, oh):
    a = RNG.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.array(Image.fromarray(a).resize((ow, oh), Image.BICUBIC))
    assert np.array_equal(R.pil_resize_bicubic(a, ow, oh), ref)


@pytest.mark.parametrize("h,w,s", [(374, 374, 315), (374, 300, 315), (250, 374, 315), (128, 127, 128), (127, 128, 128), (90, 374, 200), (316, 316, 317)])
def test_pad_black_matches_live_pillow(h, w, s):
    a = RNG.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.array(ImageOps.pad(Image.fromarray(a), (s, s), color="black"))
    assert np.array_equal(R.pil_pad_black(a, (s, s)), ref)


def test_pad_black_empty_raises_like_reference_maps_to_false():
    with pytest.raises(ValueError):
        R.pil_pad_black(np.zeros((0, 10, 3), np.uint8), (20, 20))
    with pytest.raises(ValueError):
        R.pil_pad_black(np.zeros((10, 0, 3), np.uint8), (20, 20))


def test_area_constant_image_stays_constant():
    for d, oh in [(315, 128), (200, 127), (256, 128), (384, 128), (129, 128), (577, 128)]:
        a = np.full((d, d, 3), 173, np.uint8)
        out = R.cv_resize_area(a, 128, oh)
        assert out.shape == (oh, 128, 3) and (out == 173).all()


def test_area_integer_scale_is_box_mean():
    a = RNG.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    out = R.cv_resize_area(a, 128, 128)
    ref = (a.reshape(128, 2, 128, 2, 3).astype(np.int64).sum(axis=(1, 3)) + 2) >> 2
    assert np.array_equal(out, ref.astype(np.uint8))
    a = RNG.integers(0, 256, (384, 384, 3), dtype=np.uint8)
    out = R.cv_resize_area(a, 128, 128).astype(np.float64)
    ref = a.reshape(128, 3, 128, 3, 3).astype(np.float64).mean(axis=(1, 3))
    assert np.abs(out - ref).max() <= 0.5 + 1e-4


def test_area_table_weights_sum_to_one():
    for ssize, dsize in [(315, 128), (200, 127), (577, 128), (129, 128)]:
        tab = R.cv_area_tab(ssize, dsize, ssize / dsize)
        sums = np.zeros(dsize)
        for si, di, a in tab:
            assert 0 <= si < ssize
            sums[di] += float(a)
        assert np.abs(sums - 1.0).max() < 1e-5


def test_area_fractional_close_to_exact_area_average():
    a = RNG.integers(0, 256, (315, 315, 3), dtype=np.uint8)
    out = R.cv_resize_area(a, 128, 128).astype(np.float64)
    # exact area-weighted average via integration of the piecewise-constant image
    edges = np.arange(129) * (315 / 128)
    cs = np.concatenate([np.zeros((1, 315, 3)), np.cumsum(a.astype(np.float64), axis=0)])
    def integ(c, e):  # integral of rows up to position e
        i = np.minimum(np.floor(e).astype(int), 314)
        return c[i] + (e - i)[:, None, None] * (c[i + 1] - c[i])
    rows = (integ(cs, edges[1:]) - integ(cs, edges[:-1]))
    cs2 = np.concatenate([np.zeros((128, 1, 3)), np.cumsum(rows, axis=1)], axis=1)
    i = np.minimum(np.floor(edges).astype(int), 314)
    at = lambda e, ii: cs2[:, ii] + (e - ii)[None, :, None] * (cs2[:, ii + 1] - cs2[:, ii])
    ref = (at(edges[1:], i[1:]) - at(edges[:-1], i[:-1])) / (315 / 128) ** 2
    assert np.abs(out - ref).max() <= 0.51


def test_imutils_height_truncation_quirk():
    # int(h * (128 / float(w))) is 127 for some square sizes (fighter.py:366-368 comment)
    assert int(196 * (128 / float(196))) == 127
    a = RNG.integers(0, 256, (196, 196, 3), dtype=np.uint8)
    assert R.imutils_resize_width(a, 128).shape == (127, 128, 3)


def test_square_crop_shapes_and_failures():
    frame = RNG.integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    ok, crop = yolo_crop.square_crop(frame, (0.5, 0.5, 0.15, 0.3), 128, padding=30)
    assert ok and crop.shape == (128, 128, 3) and crop.dtype == np.uint8
    ok, crop = yolo_crop.square_crop(frame, (1.8, 0.5, 0.15, 0.3), 128, padding=30)
    assert not ok and crop is None
    # clipped slice is letterboxed with black bars
    ok, crop = yolo_crop.square_crop(frame, (0.0, 0.5, 0.15, 0.3), 128, padding=30)
    assert ok and (crop[:, :8] == 0).all() and crop[:, 40:90].any()


def test_square_crop_pil_stage_matches_live_pillow():
    """The crop path up to the INTER_AREA step, with Pillow doing the pad live."""
    frame = RNG.integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    cx, cy, cw, ch = yolo_crop.yolo_pixels(0.1, 0.4, 0.15, 0.3, 1280, 720)
    d = max(cw, ch)
    half = int(d / 2)
    raw = frame[max(cy - half - 30, 0) : min(cy + half + 30, 720), max(cx - half - 30, 0) : min(cx + half + 30, 1280)]
    ref = np.array(ImageOps.pad(Image.fromarray(raw), (d, d), color="black"))
    assert np.array_equal(R.pil_pad_black(raw, (d, d)), ref)


def test_cv_area_enlarge_properties():
    """INTER_AREA with a destination larger than the source (parity unpinned restatement of
    cv::resize's bilinear emulation): an exact 2x enlargement duplicates pixels (f = 0 everywhere,
    OpenCV's documented "similar to INTER_NEAREST"), constants stay constant, values stay inside
    the source range, and the last column/row replicate the edge."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    out = R.cv_resize_area(img, 128, 128)
    assert np.array_equal(out, img.repeat(2, 0).repeat(2, 1))
    for d in (24, 50, 97, 127):
        img = rng.integers(0, 256, (d, d, 3), dtype=np.uint8)
        out = R.imutils_resize_width(img, 128)
        assert out.shape == (128, 128, 3)
        assert out.min() >= img.min() and out.max() <= img.max()
        assert np.array_equal(out[0, 0], img[0, 0]) and np.array_equal(out[-1, -1], img[-1, -1])
        flat = np.full((d, d, 3), 201, np.uint8)
        assert (R.imutils_resize_width(flat, 128) == 201).all()
    ofs, w0, w1, xmax = R.cv_linear_area_coeffs(100, 128)
    assert ofs[0] == 0 and w0[0] == 2048 and w1[0] == 0 and (w0 + w1 == 2048).all() and xmax == 127
