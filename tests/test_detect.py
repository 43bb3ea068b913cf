"""Detection post-processing: the contract in oracle/detect.py (a
restatement of the un-vendored YOLOv5 ``non_max_suppression`` + ``scale_boxes`` + label writer the reference
runs as a subprocess, ``playaid/ai_runner.py:191-224`` -- parity unpinned) and its HIP implementation."""
import numpy as np
import pytest

from oracle import detect as odet
from playaid_core_amd import detect as pdet

NET, IMG = (384, 640), (1080, 1920)  # letterboxed 640 input of a 1080p frame (stride-32 rectangle)


def _row(cx, cy, w, h, obj, cls, p, nc=6):
    r = np.zeros(5 + nc, np.float32)
    r[:5] = (cx, cy, w, h, obj)
    r[5 + cls] = p
    return r


def test_hand_worked_frame():
    pred = np.stack([
        _row(320, 192, 100, 120, 0.9, 2, 0.95),   # Pikachu, conf 0.855
        _row(322, 190, 102, 118, 0.8, 2, 0.90),   # same class, IoU ~0.93 with the first -> suppressed
        _row(100, 100, 50, 60, 0.7, 3, 0.90),     # Joker, conf 0.63: third in line, cut by --max-det 2
        _row(500, 300, 80, 80, 0.95, 0, 0.90),    # class 0: not in --classes 2 3
        _row(320, 192, 100, 120, 0.2, 2, 1.00),   # objectness below 0.25
        _row(321, 191, 100, 120, 0.85, 3, 0.97),  # Joker on top of Pikachu: other class, never suppressed by it
    ])
    rows, text = odet.detect_frame(pred, NET, IMG)
    assert rows.shape == (2, 6) and list(rows[:, 0]) == [3.0, 2.0]  # reversed(det): lowest confidence first
    # the 1080p frame sits at gain 1/3 with 12 px of letterbox above and below: (320, 192) is the frame centre
    assert text == "3 0.501562 0.497222 0.15625 0.333333 0.8245\n2 0.5 0.5 0.15625 0.333333 0.855\n"
    assert pdet.label_lines(rows) == text
    from playaid_core_amd.ai_runner import read_fighter_yolo_crop_text

    c = read_fighter_yolo_crop_text(text, "Pikachu")  # ai_runner.py:53-71 accepts the emitted lines
    assert (c.class_id, c.center_x, c.crop_width, c.confidence) == (2, 0.5, 0.15625, 0.855)
    assert odet.detect_frame(pred[3:5], NET, IMG)[1] == ""
    one, _ = odet.detect_frame(pred, NET, IMG, max_det=1)
    assert one.shape == (1, 6) and one[0, 0] == 2.0


def _random_pred(rng, n, rows, nc=6):
    """Head rows clustered around two fighters + clutter, with exact duplicates (score ties) mixed in."""
    pred = np.zeros((n, rows, 5 + nc), np.float32)
    for f in range(n):
        centres = rng.uniform([80, 60], [560, 320], (2, 2))
        for r in range(rows):
            k = rng.integers(0, 3)
            if k < 2:
                c = centres[k] + rng.normal(0, 6, 2)
                wh = rng.uniform(60, 140, 2)
                cls = 2 + k if rng.random() < 0.9 else rng.integers(0, nc)
                obj = rng.uniform(0.1, 1.0)
            else:
                c = rng.uniform([0, 0], [640, 384])
                wh = rng.uniform(5, 300, 2)
                cls = rng.integers(0, nc)
                obj = rng.uniform(0.0, 0.6)
            pred[f, r, :4] = (c[0], c[1], wh[0], wh[1])
            pred[f, r, 4] = obj
            pred[f, r, 5:] = rng.uniform(0, 0.2, nc)
            pred[f, r, 5 + cls] = rng.uniform(0.3, 1.0)
        dup = rng.integers(0, rows, 8)
        pred[f, dup[4:]] = pred[f, dup[:4]]
    pred[n - 1, :, 4] = 0.01  # a frame without any detection
    return pred


@pytest.mark.gpu
@pytest.mark.parametrize("max_det", [1, 2, 4])
def test_detect_postprocess_on_gpu_bit_exact(engine, max_det):
    rng = np.random.default_rng(11 + max_det)
    pred = _random_pred(rng, 12, 700)
    dets, counts = engine.detect_postprocess(pred, NET, IMG, max_det=max_det)
    dets, counts = dets.cpu().numpy(), counts.cpu().numpy()
    some = 0
    for f in range(pred.shape[0]):
        want, text = odet.detect_frame(pred[f], NET, IMG, max_det=max_det)
        assert counts[f] == want.shape[0], f
        assert np.array_equal(dets[f, : counts[f]].view(np.uint32), want.view(np.uint32)), (f, dets[f], want)
        assert pdet.label_lines(dets[f, : counts[f]]) == text
        some += want.shape[0]
    assert some >= 11 and counts[-1] == 0
    # other thresholds / class sets / image geometry (720p frame in a 384 x 640 input: gain 0.5, pad 12 rows)
    dets, counts = engine.detect_postprocess(pred, NET, (720, 1280), conf_thres=0.4, iou_thres=0.3, classes=(0, 3, 5), max_det=max_det)
    dets, counts = dets.cpu().numpy(), counts.cpu().numpy()
    for f in range(pred.shape[0]):
        want, _ = odet.detect_frame(pred[f], NET, (720, 1280), conf_thres=0.4, iou_thres=0.3, classes=(0, 3, 5), max_det=max_det)
        assert counts[f] == want.shape[0] and np.array_equal(dets[f, : counts[f]].view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_detector_rows_to_labels_to_runner(engine, tmp_path, state_dict):
    """f1 end to end: head rows -> device NMS -> '%g' label text -> label repair -> crops -> actions."""
    from playaid_core_amd import synth
    from playaid_core_amd.ai_runner import AIRunner, ClipSource
    from playaid_core_amd.anim_ontology import MOVE_TO_CLASS_ID
    from playaid_core_amd.cnn_action_detector import CNNActionDetector

    n, h, w = 16, 720, 1280
    boxes = synth.make_boxes(n, h, w)
    rng = np.random.default_rng(2)
    rows = 300
    pred = np.zeros((n, rows, 11), np.float32)
    pred[:, :, 4] = rng.uniform(0, 0.2, (n, rows))  # clutter below the objectness gate
    pred[:, :, :4] = rng.uniform(10, 300, (n, rows, 4))
    for i in range(n):
        for p in range(2):  # network-input pixels: gain 0.5, 12 px letterbox (720p in 384 x 640)
            cx, cy, bw, bh = boxes[i, p] * np.array([w, h, w, h]) * 0.5 + np.array([0, 12, 0, 0])
            for k in range(3):  # three near-duplicates per fighter, the best one first in confidence
                pred[i, 10 * p + k] = _row(cx + k, cy - k, bw, bh, 0.95 - 0.1 * k, 2 + p, 0.9)
    pred[5, 10:13, 4] = 0.0  # the detector misses fighter 1 in frame 6: the label repair interpolates it
    labels = pdet.labels_for_clip(engine, pred, (384, 640), (h, w))
    assert labels[0].count("\n") == 2 and labels[5].count("\n") == 1
    for i in (0, 7):
        want = odet.detect_frame(pred[i], (384, 640), (h, w))[1]
        assert labels[i] == want
    clip = ClipSource(synth.make_frames(n, h, w), labels, name="detected")
    ckpt = str(tmp_path / "seeded.ckpt")
    synth.save_checkpoint(ckpt, seed=1234)
    model = CNNActionDetector.load_from_checkpoint(ckpt, actions=list(MOVE_TO_CLASS_ID.keys()), max_batch_frames=16,
                                                   max_clip_frames=64, max_frame_height=h, max_frame_width=w)
    runner = AIRunner(clip, model=model, output_dir=str(tmp_path / "out"))
    assert runner.fighters == ["Pikachu", "Joker"] and any("Missing frames" in line for line in runner.cleaned.log)
    runner.run_action_recognition()
    rec = runner.ai_output_data["Joker"][5]
    assert rec.action in MOVE_TO_CLASS_ID and rec.crop.startswith("3 ")
