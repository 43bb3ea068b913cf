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


@pytest.mark.gpu
def test_airunner_on_a_crop_image_clip(tmp_path, state_dict, engine):
    """b2 with the reference's real hand-off: the runner is given the detector's saved crop images (plus the
    label text) instead of decoded frames and produces the same ai_output.yaml."""
    import yaml

    from playaid_core_amd import synth
    from playaid_core_amd.ai_runner import AIRunner, ClipSource
    from playaid_core_amd.anim_ontology import MOVE_TO_CLASS_ID
    from playaid_core_amd.cnn_action_detector import CNNActionDetector

    ckpt = str(tmp_path / "seeded.ckpt")
    synth.save_checkpoint(ckpt, seed=1234)
    model = CNNActionDetector.load_from_checkpoint(ckpt, actions=list(MOVE_TO_CLASS_ID.keys()), max_batch_frames=32,
                                                   max_clip_frames=64, max_frame_height=720, max_frame_width=1280)
    clip = ClipSource.synthetic(20, 720, 1280)
    # crop_jpeg_quality=0: clip2's images below stand for DECODED crop files, so the frame-cut crops must not be re-coded
    by_frames = AIRunner(clip, model=model, output_dir=str(tmp_path / "a"), crop_jpeg_quality=0, crop_mode="square")
    by_frames.run_action_recognition()
    by_frames.write_output()
    # what YOLOv5 --save-crop would have stored: here the 128 x 128 BGR square crops of the same boxes
    bgr, status = engine.square_crops(clip.frames, synth.make_boxes(20, 720, 1280), padding=30, swap_rb=False)
    assert (status == 0).all()
    crops = [[np.ascontiguousarray(bgr[i, p]) for p in range(2)] for i in range(20)]
    clip2 = ClipSource(np.zeros((20, 0, 0, 3), np.uint8), clip.labels, name="crops_only", crop_images=crops)
    by_images = AIRunner(clip2, model=model, output_dir=str(tmp_path / "b"))
    by_images.run_action_recognition()
    by_images.write_output()
    a = yaml.safe_load(open(by_frames.ai_output_file))
    b = yaml.safe_load(open(by_images.ai_output_file))
    assert a == b and len(a["Pikachu"]) == 19
    inp, frames7 = by_images.get_action_recognition_input_for_frame(5, "Joker")
    assert inp.shape == (1, 7, 3, 128, 128) and np.array_equal(frames7[3], by_frames.get_action_recognition_input_for_frame(5, "Joker")[1][3])
