"""Output contract: ``ai_output.yaml`` <-> timeline (b3 in SURVEY.md section 8b).

``ai_output.yaml`` layout written by ``AIRunner.write_output``
(``playaid/ai_runner.py:606-608``) and read by ``manuscript.py
--ai-output-path`` through ``load_timeline_from_ai_output``
(``playaid/timeline.py:52-105``):

    {fighter_name: {frame_idx: {crop: "cls cx cy w h conf", action: str,
                                predicted_action_confidence: float, damage: float}}}

A timeline is ``list[frame] -> list[2 dicts]``; the AI keys overlay a fixed
base dict. ``load_ground_truth_from_path`` mirrors the JSON-lines log loader
(``playaid/timeline.py:204-280``): 2 lines per frame, gap fill from
``num_frames_left``, fighter ids renumbered 0/1.
"""
from __future__ import annotations

import json
from typing import Dict, List

import yaml

from .anim_ontology import FIGHTER_NAME_TO_ENUM


def base_fighter_record(fighter: str, player_id: int) -> Dict:
    """The constant part of an AI-derived timeline entry (``timeline.py:69-100``)."""
    return {
        "raw_animation_frame_num": 0,
        "attack_connected": False,
        "camera_fov": 30.0,
        "camera_position": {"x": 0.0002484553260728717, "y": 15.847139358520508, "z": 148.460693359375},
        "camera_target_position": {"x": 0.0002776149194687605, "y": 11.162917137145996, "z": 0.0},
        "can_act": True,
        "damage": 0.0,
        "facing": 1.0,
        "fighter_id": player_id,
        "fighter_name": FIGHTER_NAME_TO_ENUM[fighter],
        "hitstun_left": 0.0,
        "motion_kind": 19292652517,
        "num_frames_left": 54000,
        "pos_x": -50.0,
        "pos_y": 0.21623137593269348,
        "shield_size": 50.0,
        "stage_id": 86,
        "status_kind": 0,
        "stock_count": 20,
    }


def load_timeline_from_ai_output(file_path: str, max_frames: int = None, fighters=("Joker", "Pikachu")) -> List[List[Dict]]:
    """``timeline.py:52-105``. The reference hard-codes 600 frames and the
    Joker/Pikachu pair; both are defaults here (``max_frames=None`` -> every
    frame present for both fighters, capped at the reference's 600)."""
    with open(file_path, "r") as f:
        ai_output = yaml.safe_load(f)
    fighter_to_player_id = {"Pikachu": 0, "Joker": 1}
    if max_frames is None:
        max_frames = min(600, min(len(ai_output[f]) for f in fighters))
    timeline = []
    for i in range(max_frames):
        frame_data = []
        for fighter in fighters:
            base = base_fighter_record(fighter, fighter_to_player_id.get(fighter, len(frame_data)))
            base.update(ai_output[fighter][i])
            frame_data.append(base)
        timeline.append(frame_data)
    return timeline


class _LogAssembler:
    """Folds the half-frame rows of a game log (one JSON line per fighter per frame) into a timeline.

    ``slots`` counts half-frames consumed. A row whose ``num_frames_left`` has dropped by d > 1 since
    the previous row means the logger skipped d - 1 frames: the reference fills them by repeating
    the LAST frame list (``timeline.py:249-255``) -- which at that moment is the list of the frame
    being read, so the d - 1 fillers and the post-gap frame are ONE shared list holding the post-gap
    rows, not copies of the pre-gap frame. Callers rely on nothing else, but the timeline's length
    (and therefore every later frame's index) depends on it, so the behaviour is kept."""

    def __init__(self):
        self.frames: List[List[Dict]] = []
        self.slots = 0
        self.last_left = -1

    def prefill(self, rows: List[Dict], copies: int):
        self.frames = [rows] * copies  # negative log_offset: the reference marks this branch "DOES NOT WORK"
        self.slots = 2 * copies

    def push(self, row: Dict):
        at = self.slots // 2
        if at >= len(self.frames):
            self.frames.append([])
        skipped = self.last_left - row["num_frames_left"] - 1 if self.last_left > 0 else 0
        if skipped > 0:
            self.frames.extend([self.frames[-1]] * skipped)
            self.slots += 2 * skipped
        self.frames[at].append(row)
        self.slots += 1
        self.last_left = row["num_frames_left"]

    def finish(self, validate: bool) -> List[List[Dict]]:
        # fighter ids can be anything in the log (p1 = 0, p2 = 4 ...): order by id, then renumber 0, 1
        timeline = []
        for rows in self.frames:
            ordered = sorted(rows, key=lambda r: r["fighter_id"])
            for slot, r in enumerate(ordered):
                r["fighter_id"] = slot
            timeline.append(ordered)
        if validate:
            for i, rows in enumerate(timeline):
                assert len(rows) == 2, (
                    "there should be the ground truth for 2 players for every frame, found " + f"{len(rows)} for frame #{i}"
                )
        return timeline


def load_ground_truth_from_path(label_path: str, validate: bool = True, log_offset: int = 0, max_lines=0):
    """JSON-lines game log -> timeline (``timeline.py:204-280``). ``log_offset`` > 0 drops that many
    leading frames (2 lines each); ``max_lines`` stops once more than that many half-frame slots are
    filled."""
    asm = _LogAssembler()
    with open(label_path, "r") as f:
        if log_offset < 0:
            first_two = [json.loads(f.readline()), json.loads(f.readline())]
            asm.prefill(first_two, -log_offset)
            f.seek(0)
        for lineno, line in enumerate(f):
            if max_lines and asm.slots > max_lines:
                break
            if lineno < 2 * log_offset:
                continue
            asm.push(json.loads(line))
    return asm.finish(validate)
