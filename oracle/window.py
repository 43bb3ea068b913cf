"""ORACLE (test infrastructure, never shipped, never on the product path).

Literal restatement of ``action_sample_from_frame_middle_out``
(``playaid/dataset_utils.py:109-138``), kept in the reference's own branch
structure (including the unreachable ``i == S/2`` branch) so that the host
mirror in ``playaid_core_amd/dataset_utils.py`` -- which is written differently
-- can be checked against it.

Pinning: the reference holds no test or fixture for this function (SURVEY.md
section 4); the known answers in tests/golden/window_kats.json were derived by
hand from the source (S=7, delta=3 -> m+{-27,-12,-3,0,3,12,27} clamped).
"""
import math


def action_sample_from_frame_middle_out(
    middle_frame, num_frames_per_sample, frame_delta, max_frames, min_frame=0, clamp=True
):
    assert num_frames_per_sample % 2 == 1, "num_frames_per_sample must be odd"
    middle_index = math.floor(num_frames_per_sample / 2)
    frame_nums = []
    for i in range(num_frames_per_sample):
        offset = abs(frame_delta * ((middle_index - i) ** 2))
        if i < num_frames_per_sample / 2:
            n = middle_frame - offset
            if clamp:
                n = max(min_frame, n)
        elif i == num_frames_per_sample / 2:
            n = middle_frame
        else:
            n = middle_frame + offset
            if clamp:
                n = min(max_frames - 1, middle_frame + offset)
        frame_nums.append(n)
    return frame_nums
