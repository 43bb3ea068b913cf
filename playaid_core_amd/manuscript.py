"""``manuscript`` CLI plumbing (config 0 of BASELINE.json): game log + optional
``ai_output.yaml`` -> timeline -> per-frame fighter records, on the CPU.

Keeps the reference's entry point and flags (``playaid/manuscript.py:293-385``:
``--video-path --log-path --ai-output-path --frames --skip-graphs
--skip-summaries --show-timer``; ``log_offset`` forced to 5 for ``--video-path``
runs, ``:377``). What the reference does per frame -- draw boxes, bokeh charts,
write an mp4 with ffmpeg (``:111-279``) -- is out of scope (SURVEY.md section 2
rows 10-12); the loop here walks the same timeline and emits the records that
rendering would consume, plus a small summary, so the hand-off between the
MI355X inference path (``ai_runner.py`` -> ``ai_output.yaml``) and the product
CLI can be exercised end to end without a display, ffmpeg or a GPU.

Like the reference, this CLI never calls ``AIRunner`` itself (the ``run_ai``
argument is accepted and ignored there, ``manuscript.py:49,371``); pass
``--run-ai`` to produce ``ai_output.yaml`` first (needs the MI355X).
"""
from __future__ import annotations

import json
import os
from collections import Counter
from typing import Dict, List, Optional

import click

from . import anim_ontology
from .fighter import Fighter
from .timeline import load_ground_truth_from_path, load_timeline_from_ai_output


FighterRecord = Fighter  # round-1 name of the per-fighter record


def update_fighters_from_timeline(frame_number: int, ground_truth: List[Dict], fighters: List[Fighter]):
    """``timeline.py:186-201``: construct on the first call / frame 0, ``Fighter.update`` afterwards."""
    by_id = sorted(ground_truth, key=lambda d: d["fighter_id"])
    if fighters and frame_number != 0:
        for fighter, row in zip(fighters, by_id):
            fighter.update(frame_number, row)
    else:
        fighters.extend(Fighter(frame_num=frame_number, data=row) for row in by_id)
    return fighters


class Manuscript:
    def __init__(self, input_video_path: str, ground_truth_path: str = None, ai_output_path: str = None,
                 start_frame: int = 0, max_frames: int = -1, log_offset: int = 0, run_ai: bool = False, **_ignored):
        self.input_video_path = input_video_path
        self.start_frame = start_frame
        self.timeline = []
        if ground_truth_path:
            self.timeline = load_ground_truth_from_path(ground_truth_path, log_offset=log_offset)
        if ai_output_path:  # overrides, as manuscript.py:101-102
            self.timeline = load_timeline_from_ai_output(ai_output_path)
        self.max_frames = len(self.timeline) if max_frames < 0 else min(max_frames, len(self.timeline))

    def render(self) -> Dict:
        fighters: List[Fighter] = []
        actions = [Counter(), Counter()]
        moves = [0, 0]
        for i in range(self.start_frame, self.max_frames):
            fighters = update_fighters_from_timeline(i, self.timeline[i], fighters)
            for p, f in enumerate(fighters):
                actions[p][f.action] += 1
                moves[p] = f.move_counter
        return {
            "frames": max(self.max_frames - self.start_frame, 0),
            "fighters": [
                {"fighter_id": f.fighter_id, "fighter_name": f.fighter_name, "moves": moves[p],
                 "actions": dict(actions[p]), "last_crop": str(f.crop) if f.crop else None}
                for p, f in enumerate(fighters)
            ],
        }


@click.command()
@click.option("--frames", "-f", default=None, help="Frames in the format start,end. If empty, will use entire video.")
@click.option("--skip-graphs", "-s", is_flag=True, help="Accepted for compatibility (graphs are out of scope)")
@click.option("--skip-summaries", "-c", is_flag=True, help="Accepted for compatibility")
@click.option("--show-timer", "-t", is_flag=True, help="Accepted for compatibility")
@click.option("--video-path", "-p", default=None, help="Path to the input clip (.npz: frames + labels)")
@click.option("--log-path", default=None, help="Path to the input log (JSON lines)")
@click.option("--ai-output-path", "-ai", default=None, help="Path to cached ai output")
@click.option("--run-ai", is_flag=True, help="Run AIRunner on the MI355X first and use its ai_output.yaml")
@click.option("--checkpoint", default=None, help="CNNActionDetector .ckpt for --run-ai")
@click.option("--summary-json", default=None, help="Where to write the summary (default: stdout only)")
@click.option("--params-labels", default=None,
              help="params_labels.csv (motion_kind hex -> param string); default $PLAYAID_PARAMS_LABELS. "
                   "Without it log-derived actions are 'Undefined', as for any hex the table lacks")
def run_manuscript(frames, skip_graphs, skip_summaries, show_timer, video_path, log_path, ai_output_path, run_ai,
                   checkpoint, summary_json, params_labels):
    """Entrypoint to Manuscript"""
    if params_labels or os.environ.get("PLAYAID_PARAMS_LABELS"):
        anim_ontology.load_hex_to_action(params_labels)
    if not video_path:
        print("Must specify --video-path")
        return
    start_frame, end_frame = 0, -1
    if frames:
        start_frame, end_frame = map(int, frames[1:].split(",") if frames[0] == "=" else frames.split(","))
    if run_ai:
        from .ai_runner import AIRunner  # needs the HIP library and a GPU

        runner = AIRunner(video_path, checkpoint_path=checkpoint)
        runner.run_action_recognition()
        runner.write_output()
        ai_output_path = runner.ai_output_file
    m = Manuscript(input_video_path=video_path, ground_truth_path=log_path, ai_output_path=ai_output_path,
                   start_frame=start_frame, max_frames=end_frame, log_offset=5 if log_path else 0, run_ai=run_ai)
    summary = m.render()
    text = json.dumps(summary, indent=1, sort_keys=True)
    if summary_json:
        os.makedirs(os.path.dirname(os.path.abspath(summary_json)), exist_ok=True)
        with open(summary_json, "w") as f:
            f.write(text)
    print(text)
    print("COMPLETED")


if __name__ == "__main__":
    run_manuscript()
