"""The dataset-shaped input API of the action model over a decoded clip (``playaid/ult_action_dataset.py:233-371``).

The reference's ``UltActionRecogDataset.__getitem__`` hands the training / evaluation loop
``(frames[S, 3, H, W].float() / 255, tensor(char_id), tensor(action_ids[S]), meta)`` -- S crops around a middle frame picked by
``action_sample_from_frame_middle_out`` (``dataset_utils.py:109-138``), each ``cv2.imread`` + ``BGR2RGB`` +
``imutils.resize(width=crop_size)`` of a crop file, the action string of each of the S frames mapped through the animation
list, and a dict with the pieces. ``ClipWindowDataset`` yields the same 4-tuple for every (fighter, frame) of a clip the
runner has open, with the crops cut ON THE DEVICE by the engine's crop path (``pa_save_one_box_crops`` / ``pa_square_crops`` /
``pa_runner_inputs`` -- whatever ``AIRunner`` was configured with) instead of read from crop files, and the window indices from
the same sampler with the runner's own range (frames are 1-indexed crop files, ``ai_runner.py:438-439``). Its batches are what
``CNNActionDetector.forward`` takes (``cnn_action_detector.py:86-92``).

Not built, by scope (training only): the random choice of (fighter, action, frame), the frame-delta
choice, ``synth_difficulty`` augmentation, the synthetic-stage compositing of the "simple" split, ``preceding_actions`` (the
reference's own loop over them is empty: ``range(a, a)``, ``ult_action_dataset.py:284-286``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from . import constants
from .anim_ontology import MOVE_TO_CLASS_ID
from .dataset_utils import action_sample_from_frame_middle_out


class ClipWindowDataset:
    """Index i -> fighter ``i // (max_frames - 1)``, frame ``1 + i % (max_frames - 1)`` (the order of
    ``AIRunner.run_action_recognition``'s two loops, ``ai_runner.py:493-520``)."""

    def __init__(self, runner, actions: Optional[Sequence[Sequence[str]]] = None, animations: Optional[List[str]] = None):
        """``runner``: an ``AIRunner`` (its clip, label repair, crop mode and sampler settings are used as they are).
        ``actions[p][f - 1]``: the ground-truth action string of fighter slot p in frame f (what the reference reads from the
        frame's label file, ``ult_action_dataset.py:340-343``), or None: every frame is labelled ``Undefined``.
        ``animations``: the action list the ids index (default: the ontology's 63, ``anim_ontology.py:592-600``)."""
        self.runner = runner
        self.animations = list(animations) if animations is not None else list(MOVE_TO_CLASS_ID.keys())
        self.characters = list(constants.CHAR_LIST)
        self.num_frames_per_sample = runner.num_frames_per_sample
        self.frame_delta = runner.frame_delta
        self.crop_size = 128
        self.actions = actions
        if actions is not None:
            if len(actions) != len(runner.fighters) or any(len(a) < runner.max_frames - 1 for a in actions):
                raise ValueError("actions: one list per fighter with an entry for every frame in [1, max_frames)")

    def __len__(self):
        return (self.runner.max_frames - 1) * len(self.runner.fighters)

    def _action_id(self, action: str) -> int:
        # ult_action_dataset.py:352-357 (an action outside the list maps to "Unknown", which must then be in the list)
        return self.animations.index(action) if action in self.animations else self.animations.index("Unknown")

    def __getitem__(self, idx: int):
        n_per = self.runner.max_frames - 1
        if not 0 <= idx < len(self):
            raise IndexError(idx)
        p, frame_num = idx // n_per, 1 + idx % n_per
        fighter_name = self.runner.fighters[p]
        res = self.runner._run_clip()  # the clip's crops, cut once on the device and cached by the runner
        frame_nums = action_sample_from_frame_middle_out(
            frame_num, num_frames_per_sample=self.num_frames_per_sample, frame_delta=self.frame_delta,
            max_frames=self.runner.max_frames, min_frame=1,
        )
        frames = [res["crops_rgb"][f - 1, p] for f in frame_nums]
        actions = [self.actions[p][f - 1] if self.actions is not None else "Undefined" for f in frame_nums]
        input_frames = torch.tensor(np.array(frames)).permute(0, 3, 1, 2)
        anim_label = [self._action_id(a) for a in actions]
        return (
            input_frames.float() / 255.0,
            torch.tensor(self.characters.index(fighter_name)),
            torch.tensor(anim_label),
            {
                "char": fighter_name,
                "frames": [np.array(f) for f in frames],
                "frame_paths": [f"{self.runner.video_name}_{f}.jpg" for f in frame_nums],  # the crop files' names (ai_runner.py:440-444)
                "actions": actions,
                "frame_delta": self.frame_delta,
                "preceding_actions": [],
                "preceding_actions_tensor": torch.tensor([], dtype=torch.int64),
            },
        )

    def batches(self, batch_size: int):
        """(x[B, S, 3, 128, 128], char_ids[B], action_ids[B, S], metas) in index order -- ``torch.utils.data.DataLoader``'s
        default collation without its worker processes (the crops live in one engine)."""
        for i0 in range(0, len(self), batch_size):
            items = [self[i] for i in range(i0, min(i0 + batch_size, len(self)))]
            yield (torch.stack([it[0] for it in items]), torch.stack([it[1] for it in items]), torch.stack([it[2] for it in items]),
                   [it[3] for it in items])
