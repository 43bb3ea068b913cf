"""Boxes from the game log (the reference's current,
non-deprecated source of fighter crops: ``playaid/fighter.py:494-539`` and
``playaid/data_gen_scripts/gen_gt_action_detection.py:26-91``).

``log_rows_from_timeline`` lifts the fields ``Fighter.set_from_json`` reads
(``pos_x``, ``pos_y``, ``camera_position``, ``camera_target_position``,
``stage_id`` -> fov) out of timeline dicts; ``Engine.project_boxes`` runs the
projection for all (frame, fighter) pairs in one HIP launch.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from .anim_ontology import STAGE_ENUM_TO_DATA

# STAGE_ENUM_TO_DATA[...]["fov"] (playaid/anim_ontology.py:497-570): 50 everywhere except
# TOWN_AND_CITY (95) = 30; unknown stages fall back to stage 0 (fighter.py:479-480).
STAGE_FOV = {sid: d["fov"] for sid, d in STAGE_ENUM_TO_DATA.items()}


def log_rows_from_timeline(timeline: List[List[Dict]]) -> np.ndarray:
    """-> float64[N, F, 9]: pos_x, pos_y, camera xyz, target xyz, fov (degrees)."""
    n, f = len(timeline), len(timeline[0])
    out = np.zeros((n, f, 9), dtype=np.float64)
    for i, frame in enumerate(timeline):
        for p, d in enumerate(frame):
            cam, tgt = d["camera_position"], d["camera_target_position"]
            fov = STAGE_FOV.get(d["stage_id"], STAGE_FOV[0])
            out[i, p] = (d["pos_x"], d["pos_y"], cam["x"], cam["y"], cam["z"], tgt["x"], tgt["y"], tgt["z"], fov)
    return out
