"""oracle/jpeg.py (the JPEG write + read of the reference's crops, ai_runner.py:420,446) pinned byte for byte against
the live libjpeg-turbo behind Pillow -- the library OpenCV bundles, with OpenCV's defaults (quality 95, 4:2:0,
integer DCT, fancy up-sampling)."""
import io

import numpy as np
import pytest
from PIL import Image, features

from oracle import jpeg
from playaid_core_amd import synth


def _pillow_roundtrip(rgb, quality):
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, "JPEG", quality=quality, subsampling=2)   # 2 = 4:2:0
    return np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))


pytestmark = pytest.mark.skipif(not features.check_codec("jpg"), reason="Pillow built without libjpeg")


def test_quant_tables_match_the_files_pillow_writes():
    for q in (95, 75, 30):
        buf = io.BytesIO()
        Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(buf, "JPEG", quality=q, subsampling=2)
        tabs = Image.open(io.BytesIO(buf.getvalue())).quantization
        ql, qc = jpeg.quant_tables(q)
        assert sorted(tabs[0]) == sorted(ql.reshape(-1).tolist()) and sorted(tabs[1]) == sorted(qc.reshape(-1).tolist())


@pytest.mark.parametrize("quality", [95, 90, 75, 50, 20])
def test_roundtrip_equals_libjpeg(quality):
    rng = np.random.default_rng(quality)
    frame = synth.make_frames(1, 720, 1280)[0]
    cases = [
        rng.integers(0, 256, (128, 128, 3), dtype=np.uint8),                                   # noise: every coefficient busy
        (np.add.outer(np.arange(128), np.arange(128))[..., None] * np.array([1, 0.5, 0.25])).astype(np.uint8),
        np.ascontiguousarray(frame[300:428, 500:628, ::-1]),                                   # a patch of the synthetic frames
        np.full((16, 16, 3), 77, np.uint8),
        rng.integers(0, 256, (32, 64, 3), dtype=np.uint8),
        (128 + rng.integers(-20, 20, (128, 128, 3))).astype(np.uint8),
        np.zeros((128, 128, 3), np.uint8),
        np.full((48, 16, 3), 255, np.uint8),
    ]
    for img in cases:
        assert np.array_equal(jpeg.roundtrip(img, quality), _pillow_roundtrip(img, quality))


def test_bgr_order_and_size_rule():
    rng = np.random.default_rng(1)
    bgr = rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)
    want = _pillow_roundtrip(np.ascontiguousarray(bgr[..., ::-1]), 95)[..., ::-1]
    assert np.array_equal(jpeg.roundtrip_bgr(bgr), want)
    assert np.abs(jpeg.roundtrip_bgr(bgr).astype(int) - bgr).max() > 0        # lossy
    with pytest.raises(ValueError):
        jpeg.roundtrip(np.zeros((100, 128, 3), np.uint8))
