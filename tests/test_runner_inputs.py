"""The runner's own input branch (``playaid/ai_runner.py:446-459``) -- SURVEY.md section 8 row a5.
CPU: the oracle reproduces its committed vectors and, for the shapes that avoid OpenCV, equals the live
Pillow pipeline literally. GPU: ``pa_runner_inputs`` / ``pa_backbone_crop_images`` are bit-exact
against both."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import yolo_crop

GOLD = os.path.join(os.path.dirname(__file__), "golden")
_spec = importlib.util.spec_from_file_location("make_runner_input_kats", os.path.join(GOLD, "make_runner_input_kats.py"))
kats = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(kats)


def test_oracle_reproduces_the_committed_vectors():
    z = np.load(os.path.join(GOLD, "runner_input_kats.npz"))
    assert [tuple(s) for s in z["shapes"]] == kats.CASES
    for k, (h, w) in enumerate(kats.CASES):
        assert np.array_equal(yolo_crop.runner_input_from_crop(kats.crop_image(k, h, w)), z["inputs"][k]), (h, w)


def test_128_wide_crops_equal_live_pillow():
    """For a 128-wide crop imutils.resize is a copy, so the whole branch is Pillow: run it literally."""
    from PIL import Image, ImageOps

    for k, (h, w) in enumerate(kats.CASES):
        if w != 128:
            continue
        img = kats.crop_image(k, h, w)
        rgb = np.ascontiguousarray(img[:, :, ::-1])
        want = rgb if h == 128 else np.array(ImageOps.pad(Image.fromarray(rgb), (128, 128), color="black"))
        assert np.array_equal(yolo_crop.runner_input_from_crop(img), want), (h, w)


@pytest.mark.gpu
def test_runner_inputs_on_gpu_bit_exact(engine):
    z = np.load(os.path.join(GOLD, "runner_input_kats.npz"))
    imgs = [kats.crop_image(k, h, w) for k, (h, w) in enumerate(kats.CASES)]
    got, status = engine.runner_inputs(imgs)
    assert (status == 0).all(), status
    for k, (h, w) in enumerate(kats.CASES):
        assert np.array_equal(got[k], z["inputs"][k]), (h, w, np.abs(got[k].astype(int) - z["inputs"][k]).max())
    # shapes the fixture does not hold, against the live oracle
    rng = np.random.default_rng(3)
    extra = [rng.integers(0, 256, (int(h), int(w), 3), dtype=np.uint8)
             for h, w in zip(rng.integers(20, 400, 24), rng.integers(40, 500, 24))]
    got, status = engine.runner_inputs(extra)
    for im, g, st in zip(extra, got, status):
        oh = int(im.shape[0] * (128 / float(im.shape[1])))
        if oh > 448:  # beyond the bicubic table of the pad step: reported, not computed
            assert st == 4
            continue
        assert st == 0 and np.array_equal(g, yolo_crop.runner_input_from_crop(im)), im.shape
    # status codes: too tall after the resize; empty destination (cv2.resize raises there)
    got, status = engine.runner_inputs([np.zeros((300, 64, 3), np.uint8), np.zeros((1, 300, 3), np.uint8), imgs[3]])
    assert list(status) == [4, 1, 0] and not got[0].any() and not got[1].any() and np.array_equal(got[2], z["inputs"][3])


@pytest.mark.gpu
def test_clip_from_crop_images_equals_clip_from_frames(engine):
    """A clip handed over as crop images (the reference's crops/<Fighter>/<video>_<n>.jpg after decoding)
    gives the labels of the same clip handed over as frames + boxes: the images are the BGR square
    crops the crop stage itself cuts (128 x 128), for which the runner-input branch is the identity."""
    from playaid_core_amd import synth

    n, h, w = 24, 720, 1280
    frames, boxes = synth.make_frames(n, h, w), synth.make_boxes(n, h, w)
    ref = engine.infer_clip(frames, boxes, want_crops=True)
    bgr = ref["crops_rgb"][..., ::-1]
    images = [[np.ascontiguousarray(bgr[i, p]) for p in range(2)] for i in range(n)]
    got = engine.infer_clip_from_crop_images(images, want_crops=True)
    assert np.array_equal(got["crops_rgb"], ref["crops_rgb"]) and np.array_equal(got["logp"], ref["logp"])
    # and with non-square crop images: the inputs are what the oracle makes of them
    rng = np.random.default_rng(9)
    images = [[rng.integers(0, 256, (int(rng.integers(90, 330)), int(rng.integers(90, 330)), 3), dtype=np.uint8) for _ in range(2)]
              for _ in range(10)]
    got = engine.infer_clip_from_crop_images(images, want_crops=True)
    for i in range(10):
        for p in range(2):
            assert np.array_equal(got["crops_rgb"][i, p], yolo_crop.runner_input_from_crop(images[i][p]))
    assert np.isfinite(got["logp"]).all()
