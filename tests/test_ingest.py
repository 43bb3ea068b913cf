"""Window ingest (no hardware decoder in this image): frames stay in host
memory, only every crop's source slice is uploaded (one 2-D DMA per crop), and the crop stage reads those
packed windows. Crops must be bit-identical to the whole-frame path."""
import numpy as np
import pytest
import torch

from playaid_core_amd import synth

pytestmark = pytest.mark.gpu


def test_window_ingest_equals_whole_frame_ingest(engine):
    n, h, w = 24, 1080, 1920
    frames = synth.make_frames(n, h, w)
    boxes = synth.make_boxes(n, h, w)
    # slices clipped at every frame edge, one fully off-screen, one non-finite, one above the frame (numpy wrap)
    boxes[1, 0] = (0.02, 0.03, 0.16, 0.30)
    boxes[2, 1] = (0.985, 0.97, 0.15, 0.28)
    boxes[3, 0] = (1.6, 0.5, 0.15, 0.3)
    boxes[4, 1] = (np.nan, 0.5, 0.15, 0.3)
    boxes[5, 0] = (0.5, -0.4, 0.15, 0.3)
    boxes[6, 1] = (0.5, 0.5, 256.5 / 1920, 200.5 / 1080)  # integer-scale INTER_AREA path
    ref = engine.infer_clip(frames, boxes, want_crops=True)

    host = torch.from_numpy(frames).pin_memory()
    stage = engine.make_window_stage(n)
    bd = torch.from_numpy(boxes).to(engine.device)
    crops = torch.empty((n, 2, 128, 128, 3), dtype=torch.uint8, device=engine.device)
    status = torch.empty((n, 2), dtype=torch.int32, device=engine.device)
    engine.clip_begin(n)
    engine.upload_crop_windows(host, boxes, stage)
    assert 0 < stage["used"] < frames.nbytes // 4  # a fraction of the 149 MB of whole frames
    engine.preprocess_windows(stage, n, h, w, bd, 0, crops, status)
    engine.backbone_slot(0, n, 0)
    rec, lp = engine.alloc_records(n - 1), engine.alloc_logp(n - 1)
    engine.head_frames(1, n, rec, lp)
    torch.cuda.synchronize()
    assert np.array_equal(status.cpu().numpy(), ref["crop_status"])
    assert list(ref["crop_status"][[3, 4], [0, 1]]) == [1, 2]
    assert np.array_equal(crops.cpu().numpy(), ref["crops_rgb"])
    assert np.array_equal(lp.cpu().numpy(), ref["logp"])
    # the staging is reusable: a second, different batch through the same buffers
    frames2, boxes2 = synth.make_frames(n, h, w, seed=9), synth.make_boxes(n, h, w, first_frame=40)
    ref2 = engine.infer_clip(frames2, boxes2, want_crops=True)
    engine.upload_crop_windows(torch.from_numpy(frames2).pin_memory(), boxes2, stage)
    engine.preprocess_windows(stage, n, h, w, torch.from_numpy(boxes2).to(engine.device), 1, crops, status)
    torch.cuda.synchronize()
    assert np.array_equal(crops.cpu().numpy(), ref2["crops_rgb"])


def test_pageable_host_memory_is_refused(engine):
    """The upload kernel reads the frames from the device: a pageable pointer must come back as a status code,
    never as a GPU page fault."""
    import ctypes as C

    from playaid_core_amd import _lib

    n, h, w = 2, 720, 1280
    frames = torch.from_numpy(synth.make_frames(n, h, w))  # NOT pinned
    stage = engine.make_window_stage(n)
    with pytest.raises(ValueError):
        engine.upload_crop_windows(frames, synth.make_boxes(n, h, w), stage)
    boxes = np.ascontiguousarray(synth.make_boxes(n, h, w))
    used = C.c_size_t(0)
    rc = engine._lib.pa_upload_crop_windows(
        engine._h, C.c_void_p(frames.data_ptr()), n, h, w, boxes.ctypes.data_as(C.c_void_p), 30,
        C.c_void_p(stage["windows"].data_ptr()), stage["windows"].numel(), C.c_void_p(stage["desc_host"].data_ptr()),
        C.c_void_p(stage["desc_dev"].data_ptr()), C.byref(used), None)
    assert rc == _lib.PA_ERR_INVALID_ARG and b"pinned" in engine._lib.pa_last_error(engine._h)
