"""Window sampler: host mirror vs the literal oracle restatement
(dataset_utils.py:109-138) and the committed known answers."""
import json
import os

import pytest

from oracle import window as oracle_window
from playaid_core_amd import dataset_utils

HERE = os.path.dirname(os.path.abspath(__file__))


def test_reference_shape_for_defaults():
    # S=7, delta=3: m + {-27,-12,-3,0,3,12,27}, clamped to [1, max_frames-1]
    assert dataset_utils.action_sample_from_frame_middle_out(100, 7, 3, 600, min_frame=1) == [73, 88, 97, 100, 103, 112, 127]
    assert dataset_utils.action_sample_from_frame_middle_out(2, 7, 3, 600, min_frame=1) == [1, 1, 1, 2, 5, 14, 29]
    assert dataset_utils.action_sample_from_frame_middle_out(598, 7, 3, 600, min_frame=1) == [571, 586, 595, 598, 599, 599, 599]


def test_known_answers():
    kats = json.load(open(os.path.join(HERE, "golden", "window_kats.json")))
    assert len(kats) > 100
    for k in kats:
        got = dataset_utils.action_sample_from_frame_middle_out(k["middle"], k["S"], k["delta"], k["max_frames"], min_frame=k["min_frame"])
        assert got == k["expect"], k


@pytest.mark.parametrize("s", [1, 3, 5, 7, 9])
@pytest.mark.parametrize("delta", [1, 2, 3, 5])
def test_mirror_equals_oracle(s, delta):
    for max_frames in (4, 40, 300):
        for m in range(0, max_frames + 3):
            for clamp in (True, False):
                a = oracle_window.action_sample_from_frame_middle_out(m, s, delta, max_frames, min_frame=1, clamp=clamp)
                b = dataset_utils.action_sample_from_frame_middle_out(m, s, delta, max_frames, min_frame=1, clamp=clamp)
                assert a == b


def test_even_window_rejected():
    with pytest.raises(AssertionError):
        dataset_utils.action_sample_from_frame_middle_out(5, 4, 1, 100)
