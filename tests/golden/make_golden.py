#!/usr/bin/env python3
"""Generate the committed golden fixtures from the CPU oracle.

Run from the repo root: ``python tests/golden/make_golden.py``. Inputs are
regenerated from seeds (playaid_core_amd/synth.py), so the fixtures hold only
seeds/geometry and expected outputs:

* window_kats.json   -- window indices for S in {3,5,7}, delta in {1,2,3}, edge frames
* crop_kats.npz      -- square_crop outputs for centred / clipped / off-screen /
                        integer-scale boxes at 720p and 1080p
* clip720_golden.npz -- 63 log-probs, argmax, confidence for a 12-frame 720p clip
* ai_output_128.yaml -- run_action_recognition output for the 128-frame 1080p
                        plumbing config (BASELINE.json configs[0])

The reference itself cannot run in this container, so
these are outputs of the restatement in oracle/, not of the reference.
"""
import json
import os
import sys

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pipeline, window, yolo_crop  # noqa: E402
from playaid_core_amd import synth  # noqa: E402
from playaid_core_amd.anim_ontology import ACTIONS  # noqa: E402
from playaid_core_amd.fighter import YoloCrop  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
WEIGHT_SEED = 1234

CROP_CASES = [
    # (height, width, frame_seed, frame_idx, box, padding)
    (1080, 1920, 11, 0, (0.5, 0.5, 0.1432, 0.2917), 30),      # centred
    (1080, 1920, 11, 1, (0.03, 0.05, 0.16, 0.30), 30),         # clipped left/top
    (1080, 1920, 11, 2, (0.97, 0.96, 0.15, 0.28), 30),         # clipped right/bottom
    (1080, 1920, 11, 3, (1.6, 0.5, 0.15, 0.3), 30),            # fully off-screen -> (False, None)
    (1080, 1920, 11, 4, (0.5, 0.5, 0.30, 0.20), 30),           # w > h, large
    (1080, 1920, 11, 5, (0.5, 0.5, 256.5 / 1920, 200.5 / 1080), 30),  # d = 256: INTER_AREA 2x2 path
    (1080, 1920, 11, 6, (0.4, 0.6, 128.5 / 1920, 100.5 / 1080), 30),  # d = 128: INTER_AREA copy
    (1080, 1920, 11, 7, (0.5, 0.5, 384.5 / 1920, 300.5 / 1080), 30),  # d = 384: integer-scale path
    (1080, 1920, 11, 8, (0.5, -0.4, 0.15, 0.3), 30),           # above the frame: numpy negative-stop wrap
    (720, 1280, 5, 0, (0.5, 0.5, 0.1432, 0.2917), 30),
    (720, 1280, 5, 1, (0.2, 0.3, 0.1432, 0.2917), 0),          # padding 0, odd d: 1-px upscale in PIL
    (720, 1280, 5, 2, (0.02, 0.5, 0.15, 0.30), 30),
]


def main():
    # 1. window KATs
    kats = []
    for s in (3, 5, 7):
        for delta in (1, 2, 3):
            for max_frames in (8, 64, 600):
                for f in sorted({1, 2, 3, max_frames // 2, max_frames - 3, max_frames - 2, max_frames - 1}):
                    if 1 <= f < max_frames:
                        kats.append(
                            {
                                "middle": f, "S": s, "delta": delta, "max_frames": max_frames, "min_frame": 1,
                                "expect": window.action_sample_from_frame_middle_out(f, s, delta, max_frames, min_frame=1),
                            }
                        )
    with open(os.path.join(HERE, "window_kats.json"), "w") as fh:
        json.dump(kats, fh)

    # 2. crop KATs
    crops, oks = [], []
    for (h, w, seed, idx, box, pad) in CROP_CASES:
        frame = synth.make_frame(idx, h, w, seed)
        ok, crop = yolo_crop.square_crop(frame, box, 128, padding=pad)
        oks.append(ok)
        crops.append(crop if ok else np.zeros((128, 128, 3), np.uint8))
    np.savez_compressed(
        os.path.join(HERE, "crop_kats.npz"),
        cases=np.array([(h, w, s, i, p) for (h, w, s, i, _, p) in CROP_CASES], dtype=np.int64),
        boxes=np.array([b for (_, _, _, _, b, _) in CROP_CASES], dtype=np.float64),
        ok=np.array(oks), crops=np.stack(crops),
    )

    # 3. 720p clip: log-probs
    sd = synth.make_state_dict(WEIGHT_SEED)
    n, h, w = 12, 720, 1280
    frames = synth.make_frames(n, h, w)
    boxes = synth.make_boxes(n, h, w)
    res = pipeline.run_action_recognition(frames, boxes, sd, mode="cached")
    lit = pipeline.run_action_recognition(frames, boxes, sd, mode="literal", crops_rgb=res["crops_rgb"], frame_nums=[1, 6, 11])
    assert np.abs(lit["logp"] - res["logp"][[0, 5, 10]]).max() < 1e-4
    np.savez_compressed(
        os.path.join(HERE, "clip720_golden.npz"),
        n=n, height=h, width=w, weight_seed=WEIGHT_SEED,
        logp=res["logp"].astype(np.float32), action_id=res["action_id"], confidence=res["confidence"],
        crop_sha=np.array([__import__("hashlib").sha256(res["crops_rgb"].tobytes()).hexdigest()]),
    )

    # 4. 128-frame 1080p plumbing config -> ai_output.yaml
    n, h, w = 128, 1080, 1920
    boxes = synth.make_boxes(n, h, w)
    crops_rgb = np.zeros((n, 2, 128, 128, 3), np.uint8)
    for i in range(n):
        frame = synth.make_frame(i, h, w)
        for p in range(2):
            ok, c = yolo_crop.square_crop(frame, boxes[i, p], 128, padding=30)
            assert ok
            crops_rgb[i, p] = yolo_crop.runner_input_from_crop(c)
    res = pipeline.run_action_recognition(np.zeros((n, 1, 1, 3), np.uint8), boxes, sd, mode="cached", crops_rgb=crops_rgb)
    out = {}
    for p, fighter in enumerate(synth.FIGHTER_NAMES):
        out[fighter] = {}
        for k, f in enumerate(res["frame_nums"]):
            crop = YoloCrop(*boxes[f - 1, p], confidence=1.0, class_id=synth.FIGHTER_CLASS_IDS[p])
            out[fighter][f - 1] = {
                "crop": str(crop),
                "action": ACTIONS[int(res["action_id"][k, p])],
                "predicted_action_confidence": float(np.float32(np.exp(np.float32(res["logp"][k, p].max())))) * 100.0,
            }
    with open(os.path.join(HERE, "ai_output_128.yaml"), "w") as fh:
        yaml.dump(out, fh)
    np.savez_compressed(os.path.join(HERE, "clip1080_128_logp.npz"), logp=res["logp"].astype(np.float32), action_id=res["action_id"])
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
