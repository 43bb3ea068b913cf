"""Window sampler of the hot path (host side, integers only).

Mirrors ``playaid/dataset_utils.py:109-138``: S (odd) frame numbers around a
middle frame with quadratic spacing ``delta * (mid - i)**2``, clamped to
``[min_frame, max_frames - 1]``. The device-side head kernel gathers cached
feature rows with exactly these indices (``window_gather_kernel`` in ``csrc/misc.hip``); this function is
the host mirror used by ``AIRunner`` and by the tests.
"""
from typing import List


def window_offsets(num_frames_per_sample: int, frame_delta: int) -> List[int]:
    """Signed offsets of the S window slots relative to the middle frame."""
    assert num_frames_per_sample % 2 == 1, "num_frames_per_sample must be odd"
    mid = num_frames_per_sample // 2
    return [
        (-1 if i < mid else 1) * abs(frame_delta * (mid - i) ** 2)
        for i in range(num_frames_per_sample)
    ]


def action_sample_from_frame_middle_out(
    middle_frame, num_frames_per_sample, frame_delta, max_frames, min_frame=0, clamp=True
):
    """Same name, arguments and result as the reference function
    (``playaid/dataset_utils.py:109-138``)."""
    assert num_frames_per_sample % 2 == 1, "num_frames_per_sample must be odd"
    frame_nums = []
    for off in window_offsets(num_frames_per_sample, frame_delta):
        n = middle_frame + off
        if clamp:
            # slots before the middle clamp from below, slots after it from
            # above; the middle slot (offset 0) sits in the "before" branch
            n = max(min_frame, n) if off <= 0 else min(max_frames - 1, n)
        frame_nums.append(n)
    return frame_nums
