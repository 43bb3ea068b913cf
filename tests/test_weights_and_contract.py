"""Host logic that needs no GPU: weight packing, checkpoint layout, YoloCrop
scalar methods, the ai_output.yaml <-> timeline contract, the manuscript CLI
on the committed 128-frame fixture (BASELINE.json configs[0])."""
import json
import os

import numpy as np
import pytest
import torch
import yaml
from click.testing import CliRunner

from playaid_core_amd import anim_ontology, constants, synth, timeline, weights
from playaid_core_amd.ai_runner import ClipSource, read_fighter_yolo_crop_text
from playaid_core_amd.fighter import YoloCrop
from playaid_core_amd.manuscript import run_manuscript

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def test_label_table():
    assert len(anim_ontology.ACTIONS) == 63
    assert anim_ontology.MOVE_TO_CLASS_ID["Jab"] == 0 and anim_ontology.MOVE_TO_CLASS_ID["Grabbed"] == 62
    assert anim_ontology.MOVE_TO_CLASS_ID["Wait"] == 34
    assert constants.CHAR_LIST.index("Pikachu") == 2 and constants.CHAR_LIST.index("Joker") == 3


def test_state_dict_layout_and_packing(state_dict):
    assert sum(v.size for k, v in state_dict.items() if "running" not in k) == 15347815
    assert state_dict["model.cnn1d.0.weight"].shape == (512, 1000, 7)
    blob = weights.pack_state_dict(state_dict, 7, 63)
    hdr = blob[:32].view(np.int32)
    assert hdr[0] == 0x31574150 and hdr[2] == 7 and hdr[3] == 63
    assert weights.infer_geometry(state_dict) == (7, 63)
    bad = dict(state_dict)
    bad["model.cnn2d.fc.weight"] = np.zeros((1000, 511), np.float32)
    with pytest.raises(ValueError):
        weights.pack_state_dict(bad, 7, 63)
    del bad["model.cnn2d.fc.weight"]
    with pytest.raises(KeyError):
        weights.pack_state_dict(bad, 7, 63)


def test_seeded_inputs_are_reproducible():
    a = synth.make_frame(3, 72, 128, seed=9)
    b = synth.make_frame(3, 72, 128, seed=9)
    assert np.array_equal(a, b) and a.dtype == np.uint8 and a.shape == (72, 128, 3)
    assert not np.array_equal(a, synth.make_frame(4, 72, 128, seed=9))
    w1 = synth.make_state_dict(5)["model.classifier.0.weight"]
    assert np.array_equal(w1, synth.make_state_dict(5)["model.classifier.0.weight"])
    # boxes stay >= 128 px on the long side at 720p and 1080p (INTER_AREA decimation only)
    for h, w in ((720, 1280), (1080, 1920)):
        b = synth.make_boxes(200, h, w)
        assert (np.maximum((b[..., 2] * w).astype(int), (b[..., 3] * h).astype(int)) >= 128).all()


def test_checkpoint_round_trip(tmp_path):
    p = str(tmp_path / "seed.ckpt")
    synth.save_checkpoint(p, seed=77)
    ck = torch.load(p, map_location="cpu", weights_only=False)
    assert ck["hyper_parameters"]["sequence_length"] == 7
    assert "model.cnn2d.layer4.1.bn2.running_var" in ck["state_dict"]
    assert weights.infer_geometry(ck["state_dict"]) == (7, 63)


def test_yolocrop_scalar_methods():
    c = YoloCrop.from_string("2 0.5 0.25 0.1 0.2 0.9")
    assert (c.class_id, c.center_x, c.center_y, c.crop_width, c.crop_height, c.confidence) == (2, 0.5, 0.25, 0.1, 0.2, 0.9)
    assert str(c) == "2 0.5 0.25 0.1 0.2 0.9"
    assert YoloCrop.from_string(str(c)).yolo_crop() == c.yolo_crop()
    assert c.yolo_pixels(1920, 1080) == (960, 270, 192, 216)
    assert c.xyxy_pixels(1920, 1080) == (864, 162, 1056, 378)
    d = YoloCrop(0.7, 0.45, 0.2, 0.4, confidence=0.5, class_id=2)
    m = c.interp(d, 0.25)
    assert m.center_x == pytest.approx(0.55) and m.crop_height == pytest.approx(0.25) and m.class_id == 2
    with pytest.raises(AssertionError):
        c.interp(YoloCrop(0, 0, 0, 0, class_id=3), 0.5)


def test_label_text_reader():
    text = "2 0.5 0.5 0.1 0.2 1.0\n3 0.2 0.3 0.1 0.2 0.8\n"
    assert read_fighter_yolo_crop_text(text, "Joker").center_x == 0.2
    assert read_fighter_yolo_crop_text(text, "Byleth") is None
    with pytest.raises(AssertionError):
        read_fighter_yolo_crop_text("2 0.5 0.5 0.1 0.2\n", "Pikachu")


def test_clip_source_round_trip(tmp_path):
    clip = ClipSource.synthetic(3, 72, 128)
    assert clip.labels[0].count("\n") == 2 and clip.labels[0].startswith("2 ")
    p = str(tmp_path / "c.npz")
    clip.save(p)
    back = ClipSource.load(p)
    assert np.array_equal(back.frames, clip.frames) and back.labels == clip.labels


def test_ai_output_yaml_to_timeline():
    tl = timeline.load_timeline_from_ai_output(os.path.join(GOLD, "ai_output_128.yaml"))
    assert len(tl) == 127 and all(len(f) == 2 for f in tl)
    joker, pika = tl[5]
    assert joker["fighter_name"] == 82 and pika["fighter_name"] == 8 and pika["fighter_id"] == 0
    assert joker["action"] in anim_ontology.MOVE_TO_CLASS_ID and joker["motion_kind"] == 19292652517
    c = YoloCrop.from_string(pika["crop"])
    assert c.class_id == 2 and c.confidence == 1.0


def test_log_loader_gap_fill_and_renumber(tmp_path):
    p = tmp_path / "g.log"
    base = {"damage": 0.0, "stock_count": 3, "motion_kind": 1, "fighter_name": 8}
    lines = []
    for nfl in (100, 99, 96, 95):  # 98, 97 missing -> two frames inserted
        for fid in (4, 0):
            lines.append(json.dumps(dict(base, fighter_id=fid, num_frames_left=nfl)))
    p.write_text("\n".join(lines) + "\n")
    tl = timeline.load_ground_truth_from_path(str(p))
    assert len(tl) == 6 and [f[0]["fighter_id"] for f in tl] == [0] * 6 and [f[1]["fighter_id"] for f in tl] == [1] * 6
    # reference quirk kept: the inserted frames alias the list of the frame being read
    # (timeline.py:249-255 repeats ground_truth[-1], which is the just-appended current frame)
    assert [f[0]["num_frames_left"] for f in tl] == [100, 99, 96, 96, 96, 95]


def test_manuscript_cli_plumbing_config0(tmp_path):
    """BASELINE.json configs[0]: 128 synthetic 1080p frames + stub log through the CLI, CPU only."""
    log = str(tmp_path / "stub.log")
    synth.make_stub_log(log, 128)
    out = str(tmp_path / "summary.json")
    r = CliRunner().invoke(run_manuscript, ["--video-path", "synthetic_128.npz", "--log-path", log, "--skip-graphs", "--summary-json", out])
    assert r.exit_code == 0, r.output
    s = json.load(open(out))
    assert s["frames"] == 128 - 5  # log_offset 5 (manuscript.py:377)
    # a log-only run derives the action from motion_kind (fighter.py:540-547): nothing in the table -> Undefined
    assert [f["actions"] for f in s["fighters"]] == [{"Undefined": 123}] * 2
    labels = tmp_path / "params_labels.csv"
    labels.write_text("0x0000000000,\n0x047dee83e5,wait\n")
    r = CliRunner().invoke(run_manuscript, ["--video-path", "synthetic_128.npz", "--log-path", log, "--params-labels", str(labels), "--summary-json", out])
    assert r.exit_code == 0, r.output
    s = json.load(open(out))
    assert [f["actions"] for f in s["fighters"]] == [{"Wait": 123}] * 2 and [f["moves"] for f in s["fighters"]] == [0, 0]
    assert [f["fighter_name"] for f in s["fighters"]] == ["Pikachu", "Joker"]
    anim_ontology.HEX_TO_ACTION.clear()
    anim_ontology._hex_table_path = None
    r = CliRunner().invoke(
        run_manuscript,
        ["--video-path", "synthetic_128.npz", "--log-path", log, "--ai-output-path", os.path.join(GOLD, "ai_output_128.yaml"), "--summary-json", out],
    )
    assert r.exit_code == 0, r.output
    s = json.load(open(out))
    assert s["frames"] == 127 and len(s["fighters"]) == 2
    gold = yaml.safe_load(open(os.path.join(GOLD, "ai_output_128.yaml")))
    for f in s["fighters"]:
        name = f["fighter_name"]  # the enum of the timeline dict resolved to a name, as Fighter.set_from_json does
        assert sum(f["actions"].values()) == 127
        assert f["last_crop"] == gold[name][126]["crop"]
