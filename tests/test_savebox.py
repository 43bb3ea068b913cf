"""Rows a2 / f1: the crop hand-off of `detect.py --save-crop` (YOLOv5 save_one_box + its JPEG write, cv2.imread) on the device."""
import io

import numpy as np
import pytest

from playaid_core_amd import synth

F32 = np.float32


def _label_rows(rng, n, h, w, max_det=2):
    """Label rows as detect.py writes them: rounded pixel boxes -> xyxy2xywh / gn in float32; class ids 2 / 3."""
    dets = np.zeros((n, max_det, 6), F32)
    counts = np.zeros(n, np.int32)
    for i in range(n):
        k = int(rng.integers(0, max_det + 1)) if i % 5 == 4 else max_det
        counts[i] = k
        for j in range(k):
            bw, bh = int(rng.integers(8, 420)), int(rng.integers(8, 420))
            x1 = int(rng.integers(-40, w - 4))
            y1 = int(rng.integers(-40, h - 4))
            x1c, y1c = max(x1, 0), max(y1, 0)
            x2c, y2c = min(x1 + bw, w), min(y1 + bh, h)
            xyxy = np.array([x1c, y1c, max(x2c, x1c + 1), max(y2c, y1c + 1)], F32)
            xywh = np.array([(xyxy[0] + xyxy[2]) / F32(2), (xyxy[1] + xyxy[3]) / F32(2), xyxy[2] - xyxy[0], xyxy[3] - xyxy[1]], F32)
            dets[i, j, 0] = 2 + (j % 2) if i % 7 else 3 - (j % 2)  # label order is by confidence, not by class
            dets[i, j, 1:5] = (xywh / np.array([w, h, w, h], F32)).astype(F32)
            dets[i, j, 5] = F32(0.5 + 0.1 * j)
    return dets, counts


def test_oracle_any_size_roundtrip_is_pinned_to_live_libjpeg_turbo():
    """oracle.jpeg.roundtrip_any (4:4:4 as save_one_box writes, 4:2:0 as cv2.imwrite writes; any size) == Pillow's
    save + open, byte for byte."""
    from PIL import Image

    from oracle import jpeg

    fr = np.ascontiguousarray(synth.make_frame(3, 400, 500)[..., ::-1])
    rng = np.random.default_rng(1)
    sizes = [(128, 128), (133, 77), (17, 31), (8, 8), (1, 1), (250, 333), (64, 100), (395, 301), (2, 2), (15, 16), (16, 15)]
    sizes += [tuple(int(v) for v in rng.integers(1, 300, 2)) for _ in range(12)]
    for h, w in sizes:
        for q in (95, 75):
            for ss in (0, 2):
                rgb = fr[:h, :w]
                b = io.BytesIO()
                Image.fromarray(rgb).save(b, "JPEG", quality=q, subsampling=ss)
                want = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))
                assert np.array_equal(jpeg.roundtrip_any(rgb, q, ss), want), (h, w, q, ss)
    c = fr[:128, :128]
    assert np.array_equal(jpeg.roundtrip(c, 95), jpeg.roundtrip_any(c, 95, 2))


def test_oracle_save_one_box_rectangle():
    """Hand-checked rows of the rectangle arithmetic (gain 1.02, pad 10, truncation, clipping)."""
    from oracle import detect as odet

    h, w = 720, 1280
    row = np.array([2, 640 / w, 360 / h, 100 / w, 50 / h, 0.9], F32)   # box 590..690 x 335..385
    # w' = 100 * 1.02 + 10 = 112 -> 584..696 ; h' = 50 * 1.02 + 10 = 61 -> 329.5..390.5 -> 329..390
    assert odet.save_one_box_rect(row, (h, w)) == (584, 329, 696, 390)
    row = np.array([3, 20 / w, 15 / h, 40 / w, 30 / h, 0.9], F32)      # box 0..40 x 0..30 at the corner: clipped at 0
    assert odet.save_one_box_rect(row, (h, w)) == (0, 0, 45, 35)
    row = np.array([3, 1270 / w, 710 / h, 20 / w, 20 / h, 0.9], F32)
    assert odet.save_one_box_rect(row, (h, w)) == (1254, 694, 1280, 720)


@pytest.mark.gpu
def test_save_one_box_crops_on_the_device(engine):
    """pa_save_one_box_crops == oracle save_one_box for every (frame, fighter): rectangle, raw cut, 4:4:4 JPEG write + read."""
    import torch

    from oracle import detect as odet

    n, h, w = 24, 720, 1280
    rng = np.random.default_rng(11)
    frames = synth.make_frames(n, h, w)
    dets, counts = _label_rows(rng, n, h, w)
    fd = torch.from_numpy(frames).cuda()
    dd, cd = torch.from_numpy(dets).cuda(), torch.from_numpy(counts).cuda()
    for quality in (95, 0, 60):
        images, desc = engine.save_one_box_crops(fd, dd, cd, jpeg_quality=quality)
        engine.check_device_errors()
        got = engine.unpack_crop_images(images, desc)
        assert len(got) == n * 2
        seen = 0
        for i in range(n):
            for p, cls in enumerate((2, 3)):
                ks = [k for k in range(counts[i]) if int(dets[i, k, 0]) == cls]
                g = got[i * 2 + p]
                if not ks:
                    assert g is None
                    continue
                row = dets[i, ks[0]]
                x1, y1, x2, y2 = odet.save_one_box_rect(row, (h, w))
                if quality:
                    want = odet.save_one_box(row, frames[i], quality)
                else:
                    want = frames[i][y1:y2, x1:x2] if x2 > x1 and y2 > y1 else None
                if want is None:
                    assert g is None
                else:
                    assert g is not None and g.shape == want.shape, (i, p, None if g is None else g.shape, want.shape)
                    assert np.array_equal(g, want), (quality, i, p, np.abs(g.astype(int) - want).max())
                    seen += 1
        assert seen > n
    # an explicit choice of detections (what the label cleaning decides), and a buffer that is too small
    idx = np.full((n, 2), -1, np.int32)
    idx[:, 0] = np.where(counts > 1, 1, -1)
    images, desc = engine.save_one_box_crops(fd, dd, cd, det_index=idx, jpeg_quality=95)
    got = engine.unpack_crop_images(images, desc)
    for i in range(n):
        assert got[i * 2 + 1] is None
        if counts[i] > 1:
            want = odet.save_one_box(dets[i, 1], frames[i], 95)
            assert (got[i * 2] is None) == (want is None) and (want is None or np.array_equal(got[i * 2], want))
    from playaid_core_amd import _lib
    from playaid_core_amd.engine import EngineError

    small = torch.empty(4096, dtype=torch.uint8, device="cuda")
    engine.save_one_box_crops(fd, dd, cd, images=small)
    with pytest.raises(EngineError) as ei:
        engine.check_device_errors()
    assert ei.value.code == _lib.PA_ERR_CAPACITY and "did not fit" in str(ei.value)


@pytest.mark.gpu
def test_detector_crops_to_labels_without_host_pixels(engine, state_dict):
    """frames + detections -> save_one_box crops (JPEG) -> runner inputs -> backbone -> labels, all on the device, ==
    the oracle pipeline fed with the oracle's crops (the reference's own hand-off, ai_runner.py:208,445-459)."""
    import torch

    from oracle import detect as odet
    from oracle import pipeline, yolo_crop

    n, h, w = 16, 720, 1280
    frames = synth.make_frames(n, h, w)
    boxes = synth.make_boxes(n, h, w)
    dets = np.zeros((n, 2, 6), F32)
    for i in range(n):
        for p in range(2):  # the label rows a detector would have written for the synthetic fighters
            cx, cy, bw, bh = boxes[i, p] * np.array([w, h, w, h])
            x1, y1, x2, y2 = np.rint([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2]).clip(0, [w, h, w, h]).astype(F32)
            xywh = np.array([(x1 + x2) / F32(2), (y1 + y2) / F32(2), x2 - x1, y2 - y1], F32)
            dets[i, p] = np.concatenate([[2 + p], (xywh / np.array([w, h, w, h], F32)).astype(F32), [0.9]]).astype(F32)
    counts = np.full(n, 2, np.int32)
    fd = torch.from_numpy(frames).cuda()
    images, desc = engine.save_one_box_crops(fd, torch.from_numpy(dets).cuda(), torch.from_numpy(counts).cuda(), jpeg_quality=95)
    got = engine.infer_clip_from_packed_crop_images(images, desc, n, want_crops=True)
    engine.check_device_errors()
    crops_ref = np.zeros((n, 2, 128, 128, 3), np.uint8)
    for i in range(n):
        for p in range(2):
            crops_ref[i, p] = yolo_crop.runner_input_from_crop(odet.save_one_box(dets[i, p], frames[i], 95))
    ref = pipeline.run_action_recognition(np.zeros((n, 1, 1, 3), np.uint8), boxes, state_dict, mode="cached", crops_rgb=crops_ref)
    assert (got["crop_status"] == 0).all()
    assert np.array_equal(got["crops_rgb"], crops_ref)
    assert np.abs(got["logp"] - ref["logp"]).max() <= 1e-4
    assert np.array_equal(got["action_id"], ref["action_id"])


@pytest.mark.gpu
def test_runner_default_is_the_references_mix_of_crops(tmp_path, state_dict):
    """AIRunner's default (crop_mode="yolo"): detector frames get YOLOv5's save_one_box crop (4:4:4 JPEG), repaired gap
    frames the reference's own square_crop (4:2:0 JPEG, cut from VideoCapture position j), tail frames a copy of the last
    crop file -- composed here from the oracle pieces, independently of label_cleaning.py's tables."""
    from oracle import detect as odet
    from oracle import jpeg, pipeline, yolo_crop
    from playaid_core_amd.ai_runner import AIRunner, ClipSource
    from playaid_core_amd.anim_ontology import ACTIONS, MOVE_TO_CLASS_ID
    from playaid_core_amd.cnn_action_detector import CNNActionDetector
    from playaid_core_amd.fighter import YoloCrop

    n, h, w = 36, 720, 1280
    clip = ClipSource.synthetic(n, h, w)
    full = [[YoloCrop.from_string(l) for l in t.splitlines()] for t in clip.labels]
    drop = {12, 13, 14, 35, 36}  # Joker (second line) missing in these 1-indexed frames
    clip.labels = ["".join(str(c) + "\n" for k, c in enumerate(cs) if not (k == 1 and i + 1 in drop)) for i, cs in enumerate(full)]
    ckpt = str(tmp_path / "seeded.ckpt")
    synth.save_checkpoint(ckpt, seed=1234)
    model = CNNActionDetector.load_from_checkpoint(ckpt, actions=list(MOVE_TO_CLASS_ID.keys()), max_batch_frames=64,
                                                   max_clip_frames=64, max_frame_height=h, max_frame_width=w)
    runner = AIRunner(clip, model=model, output_dir=str(tmp_path / "ai_cache"))
    assert runner.crop_mode == "yolo" and runner.max_frames == n
    runner.run_action_recognition()

    def row(c):  # the label row as the file holds it ('%g', six digits), read back
        c = YoloCrop.from_string(str(c))
        return np.array([c.class_id, c.center_x, c.center_y, c.crop_width, c.crop_height, c.confidence], F32)

    crops = np.zeros((n, 2, 128, 128, 3), np.uint8)
    for i in range(n):
        for p in range(2):
            j = i + 1
            if p == 1 and j in (12, 13, 14):  # interpolated from the END frame, pixels from VideoCapture position j
                it = full[10][1].interp(full[14][1], (15 - j) / (15 - 11))
                ok, c = yolo_crop.square_crop(clip.frames[j], np.array(it.yolo_crop()), 128, padding=30)
                assert ok
                bgr = jpeg.roundtrip_bgr(np.ascontiguousarray(c), 95)   # cv2.imwrite / imread of the 128 x 128 crop
            elif p == 1 and j == 35:  # tail: a copy of frame 34's crop file
                bgr = odet.save_one_box(row(full[33][1]), clip.frames[33], 95)
            elif p == 1 and j == 36:
                continue  # never needed
            else:
                bgr = odet.save_one_box(row(full[i][p]), clip.frames[i], 95)
            crops[i, p] = yolo_crop.runner_input_from_crop(bgr)
    boxes = np.array([[c.yolo_crop() for c in cs] for cs in full])
    want = pipeline.run_action_recognition(clip.frames, boxes, state_dict, mode="cached", crops_rgb=crops)
    res = runner._results
    assert np.array_equal(res["crops_rgb"][:35], crops[:35]) and np.array_equal(res["crops_rgb"][35, 0], crops[35, 0])
    assert np.abs(res["logp"] - want["logp"]).max() <= 1e-4
    assert np.array_equal(res["action_id"], want["action_id"])
    for f in range(1, n):
        assert runner.ai_output_data["Joker"][f - 1].action == ACTIONS[int(want["action_id"][f - 1, 1])]
    # and it is NOT what the square-crop formulation computes
    sq = AIRunner(clip, model=model, output_dir=str(tmp_path / "sq"), crop_mode="square")
    sq.run_action_recognition()
    assert np.abs(sq._results["logp"] - res["logp"]).max() > 1e-3
