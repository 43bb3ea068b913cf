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


def load_ground_truth_from_path(label_path: str, validate: bool = True, log_offset: int = 0, max_lines=0):
    """JSON-lines game log -> timeline (``timeline.py:204-280``)."""
    ground_truth: List[List[Dict]] = []
    prev_num_frames_left = -1
    index = 0
    offset_count = 0
    with open(label_path, "r") as f:
        for line in f:
            if max_lines and index > max_lines:
                break
            if offset_count < (2 * log_offset):
                offset_count += 1
                continue
            json_data = json.loads(line)
            frame_number = index // 2
            if frame_number >= len(ground_truth):
                ground_truth.append([])
            diff = prev_num_frames_left - json_data["num_frames_left"]
            if prev_num_frames_left > 0 and diff > 1:
                ground_truth += [ground_truth[-1]] * (diff - 1)
                index += (diff - 1) * 2
            ground_truth[frame_number].append(json_data)
            index += 1
            prev_num_frames_left = json_data["num_frames_left"]
    for i, frame_data in enumerate(ground_truth):
        frame_data = sorted(frame_data, key=lambda x: x["fighter_id"])
        for j, fighter_data in enumerate(frame_data):
            fighter_data["fighter_id"] = j
        ground_truth[i] = frame_data
    if validate:
        for i, gt in enumerate(ground_truth):
            assert len(gt) == 2, (
                "there should be the ground truth for 2 players for every frame, found " + f"{len(gt)} for frame #{i}"
            )
    return ground_truth
