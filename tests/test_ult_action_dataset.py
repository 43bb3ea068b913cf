"""Row N4: the dataset-shaped input API (``playaid/ult_action_dataset.py:233-371``) over a clip, fed to the operator."""
import numpy as np
import pytest
import torch

from playaid_core_amd import constants, synth

pytestmark = pytest.mark.gpu


def test_clip_window_dataset_feeds_the_operator_like_the_runner(tmp_path):
    from playaid_core_amd.ai_runner import AIRunner, ClipSource
    from playaid_core_amd.anim_ontology import MOVE_TO_CLASS_ID
    from playaid_core_amd.cnn_action_detector import CNNActionDetector
    from playaid_core_amd.ult_action_dataset import ClipWindowDataset

    n, h, w = 40, 720, 1280
    ckpt = str(tmp_path / "seeded.ckpt")
    synth.save_checkpoint(ckpt, seed=1234)
    actions = list(MOVE_TO_CLASS_ID.keys())
    model = CNNActionDetector.load_from_checkpoint(ckpt, actions=actions, max_batch_frames=64, max_clip_frames=512,
                                                   max_frame_height=h, max_frame_width=w)
    runner = AIRunner(ClipSource.synthetic(n, h, w, seed=17, name="n4"), model=model, output_dir=str(tmp_path / "out"), crop_mode="square")
    gt = [[actions[(f * 7 + 3 * p) % len(actions)] for f in range(1, runner.max_frames)] for p in range(len(runner.fighters))]
    ds = ClipWindowDataset(runner, actions=gt)
    assert len(ds) == (runner.max_frames - 1) * 2
    # one item: the reference's 4-tuple (ult_action_dataset.py:349-371)
    x, char_id, action_ids, meta = ds[5]
    assert x.shape == (7, 3, 128, 128) and x.dtype == torch.float32 and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    assert int(char_id) == constants.CHAR_LIST.index(runner.fighters[0]) and action_ids.shape == (7,)
    assert set(meta) >= {"char", "frames", "frame_paths", "actions", "frame_delta", "preceding_actions", "preceding_actions_tensor"}
    assert [actions[i] for i in action_ids.tolist()] == meta["actions"] and meta["frame_delta"] == runner.frame_delta
    assert np.array_equal((x * 255).round().byte().permute(0, 2, 3, 1).numpy(), np.array(meta["frames"]))
    # its batches through the operator == the runner's own answers, item for item
    logps = []
    for xb, cb, ab, metas in ds.batches(16):
        assert xb.shape[1:] == (7, 3, 128, 128) and ab.shape[1] == 7 and len(metas) == xb.shape[0]
        logps.append(model(xb).cpu())
    logp = torch.cat(logps).numpy()
    n_per = runner.max_frames - 1
    for idx in (0, 3, n_per - 1, n_per, n_per + 11, 2 * n_per - 1):
        p, frame_num = idx // n_per, 1 + idx % n_per
        in_frames, char, pred_id, info = runner.action_recognition(frame_num, runner.fighters[p])
        item = ds[idx]
        assert torch.equal(item[0], in_frames[0]) and int(item[1]) == char
        assert int(np.argmax(logp[idx])) == int(pred_id)
        assert abs(float(np.exp(logp[idx].max())) * 100.0 - info["confidence"]) <= 1e-2
    # the windows the operator saw are the runner's: same log-probabilities as the clip-level pass, to the batching's rounding
    want = np.asarray(runner._run_clip()["logp"])  # [frames, F, A]
    got = logp.reshape(2, n_per, -1).transpose(1, 0, 2)
    assert np.abs(got - want.reshape(-1, 2, got.shape[2])[:n_per]).max() <= 1e-4
    # an action outside the animation list has no "Unknown" to fall back on in the 63-class ontology (the reference raises too)
    bad = ClipWindowDataset(runner, actions=[["NotAMove"] * n_per] * 2)
    with pytest.raises(ValueError):
        bad[0]
