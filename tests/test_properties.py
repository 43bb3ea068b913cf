"""Property tests (hypothesis) of the host logic: the window sampler mirror against the oracle's
literal restatement, the frame-parallel shard / halo plan, and the label repair invariants."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import window as owindow
from playaid_core_amd import constants
from playaid_core_amd.dataset_utils import action_sample_from_frame_middle_out
from playaid_core_amd.fighter import YoloCrop
from playaid_core_amd.label_cleaning import clean_yolo_labels
from playaid_core_amd.parallel import halo_plan, needed_range, owned_frame_nums, shard_range

A, B = 2, 3
FIGHTERS = [constants.CHAR_LIST[A], constants.CHAR_LIST[B]]


@settings(max_examples=300, deadline=None)
@given(st.integers(1, 400), st.sampled_from([1, 3, 5, 7, 9, 11]), st.integers(1, 5), st.integers(2, 400))
def test_window_mirror_equals_oracle(frame, s, delta, max_frames):
    frame = min(frame, max_frames - 1)
    got = action_sample_from_frame_middle_out(frame, num_frames_per_sample=s, frame_delta=delta, max_frames=max_frames, min_frame=1)
    want = owindow.action_sample_from_frame_middle_out(frame, s, delta, max_frames, min_frame=1)
    assert list(got) == list(want)
    assert len(got) == s and got[s // 2] == frame
    assert all(1 <= f <= max(max_frames - 1, 1) for f in got) and list(got) == sorted(got)


@settings(max_examples=200, deadline=None)
@given(st.integers(2, 5000), st.integers(1, 8), st.integers(0, 60))
def test_shards_cover_the_clip_and_halos_pair_up(n_total, world, reach):
    ranges = [shard_range(n_total, world, r) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n_total
    assert all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
    owned = [owned_frame_nums(n_total, world, r) for r in range(world)]
    nums = [f for lo, hi in owned for f in range(lo, hi)]
    assert nums == list(range(1, n_total))  # every frame number 1..n-1 reported exactly once, in order
    # every send has a matching receive on the peer, and what a rank receives covers what its windows need
    plans = [halo_plan(n_total, world, r, reach) for r in range(world)]
    for r, (recvs, sends) in enumerate(plans):
        for peer, f0, cnt in sends:
            assert (r, f0, cnt) in [(p, a, c) for p, a, c in plans[peer][0]]
        have = set(range(*ranges[r]))
        for peer, f0, cnt in recvs:
            have |= set(range(f0, f0 + cnt))
        lo, hi = owned[r]
        if hi > lo:
            need_lo, need_hi = needed_range(n_total, world, r, reach)
            assert set(range(need_lo, need_hi)) <= have


def _label(present_a, present_b, i):
    lines = []
    if present_a:
        lines.append(str(YoloCrop(0.1 + 0.001 * i, 0.5, 0.1, 0.2, confidence=0.9, class_id=A)))
    if present_b:
        lines.append(str(YoloCrop(0.9 - 0.001 * i, 0.4, 0.12, 0.22, confidence=0.8, class_id=B)))
    return "".join(l + "\n" for l in lines)


@settings(max_examples=200, deadline=None)
@given(st.lists(st.tuples(st.booleans(), st.booleans()), min_size=3, max_size=60))
def test_label_repair_invariants(presence):
    # both fighters detected in frame 1 (the reference asserts otherwise) and somewhere later
    presence = [(True, True)] + presence
    labels = [_label(a, b, i) for i, (a, b) in enumerate(presence)]
    c = clean_yolo_labels(labels, FIGHTERS, len(labels))
    last = [max(i for i, pr in enumerate(presence) if pr[p]) + 1 for p in range(2)]
    assert c.max_frames == max(last)
    for p in range(2):
        for f in range(1, c.max_frames + 1):
            src = c.pixel_frame[f - 1, p]
            if f <= last[p]:
                assert src >= 0  # detected, or a repaired gap
                if presence[f - 1][p]:
                    assert src == f - 1  # its own frame
                elif f < len(labels):
                    assert src == f  # a repaired gap reads VideoCapture position f (one frame late)
            elif f < max(last):
                assert src == c.pixel_frame[last[p] - 1, p]  # tail duplication of the last crop image
            else:
                assert src == -1  # the very last frame of the shorter fighter has no crop
            # the repaired label of a gap frame carries an interpolated box strictly between its neighbours'
            crop = c.label_crop[f - 1][p]
            if f <= last[p]:
                assert crop is not None and crop.class_id == (A, B)[p]
    # repairing repaired labels changes no label text and no box (only the one-frame-late source of the
    # repaired gaps is forgotten: they now look like ordinary detections)
    again = clean_yolo_labels(c.labels, FIGHTERS, len(labels))
    assert again.labels == c.labels
    assert np.array_equal(again.pixel_box, c.pixel_box)


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 90), st.integers(1, 90), st.integers(1, 90), st.integers(1, 90), st.integers(0, 2**31 - 1))
def test_bicubic_restatement_equals_live_pillow_on_random_shapes(h, w, oh, ow, seed):
    """The Pillow half of the oracle is pinned against the live library (the reference's own
    dependency): enlarging, shrinking and mixed, down to 1-pixel images."""
    from PIL import Image

    from oracle import resample as R

    a = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.array(Image.fromarray(a).resize((ow, oh), Image.BICUBIC))
    assert np.array_equal(R.pil_resize_bicubic(a, ow, oh), ref)


@settings(max_examples=40, deadline=None)
@given(st.integers(2, 120), st.integers(2, 120), st.integers(0, 2**31 - 1))
def test_pad_restatement_equals_live_imageops_pad(h, w, seed):
    from PIL import Image, ImageOps

    from oracle import resample as R

    a = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
    d = max(h, w)
    ref = np.array(ImageOps.pad(Image.fromarray(a), (d, d), color="black"))
    assert np.array_equal(R.pil_pad_black(a, (d, d)), ref)
