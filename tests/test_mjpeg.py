"""Row a1 / f2: Motion-JPEG decode. CPU part: the oracle's decoder is pinned byte for byte against the live
libjpeg-turbo behind Pillow, container helpers. GPU part (through the C ABI): pa_mjpeg_decode == the oracle, bit for bit."""
import io
import os

import numpy as np
import pytest

from playaid_core_amd import synth, video

SIZES = [(1080, 1920), (720, 1280), (133, 77), (17, 31), (8, 8), (16, 16), (250, 333)]


def _pil_decode(blob: bytes) -> np.ndarray:
    from PIL import Image

    return np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"))


def _cases():
    """(name, frames_bgr, encoder kwargs) covering sampling, quality, restart markers, optimised tables, odd sizes."""
    out = []
    base = synth.make_frame(5, 1080, 1920)
    for h, w in SIZES:
        fr = base[:h, :w]
        for q in (95, 75, 30):
            for kw in ({}, {"restart_marker_rows": 1}, {"restart_marker_blocks": 5}, {"optimize": True}):
                for ss in (2, 1, 0):
                    if (h, w) in ((1080, 1920), (720, 1280)) and (ss != 2 or q == 30):
                        continue  # the big frames: the path's own configuration only (4:2:0)
                    out.append((f"{h}x{w}-q{q}-ss{ss}-{sorted(kw)}", fr, dict(quality=q, subsampling=ss, **kw)))
    return out


def test_oracle_decoder_is_pinned_to_live_libjpeg_turbo():
    """oracle.jpeg.decode (own marker parser, own bit-serial Huffman decoder in C, own IDCT / up-sampling / colour
    arithmetic) == PIL.Image.open of the same bytes, byte for byte."""
    from oracle import jpeg

    n = 0
    for name, fr, kw in _cases():
        blob = synth.encode_jpeg_frames([fr], **kw)[0]
        assert np.array_equal(jpeg.decode(blob), _pil_decode(blob)), name
        n += 1
    assert n > 150
    grey = synth.encode_jpeg_frames([synth.make_frame(1, 200, 300)[..., 0]], quality=90)[0]
    assert np.array_equal(jpeg.decode(grey), _pil_decode(grey))
    # cv2's channel order
    blob = synth.encode_jpeg_frames([synth.make_frame(2, 64, 96)])[0]
    assert np.array_equal(jpeg.decode_bgr(blob)[..., ::-1], _pil_decode(blob))


def test_oracle_rejects_what_is_not_baseline():
    from PIL import Image

    from oracle import jpeg

    b = io.BytesIO()
    Image.fromarray(synth.make_frame(0, 64, 64)).save(b, "JPEG", progressive=True)
    with pytest.raises(jpeg.JpegError):
        jpeg.decode(b.getvalue())
    with pytest.raises(jpeg.JpegError):
        jpeg.decode(b"\x00\x01\x02\x03")
    good = synth.encode_jpeg_frames([synth.make_frame(0, 64, 64)])[0]
    with pytest.raises(jpeg.JpegError):
        jpeg.decode(good[:100])


def test_containers(tmp_path):
    frames = synth.make_frames(5, 72, 128)
    blobs = synth.encode_jpeg_frames(frames, quality=90)
    raw = np.frombuffer(b"".join(blobs), np.uint8)
    sp = video.split_jpeg_stream(raw)
    assert sp.shape == (5, 2) and all(bytes(raw[a:b]) == blobs[i] for i, (a, b) in enumerate(sp))
    path = str(tmp_path / "clip.avi")
    video.write_avi_mjpeg(path, blobs, 30.0, 128, 72)
    data, off, meta = video.read_avi_mjpeg(path)
    assert meta["fps"] == 30.0 and (meta["height"], meta["width"]) == (72, 128) and meta["handler"] == b"MJPG"
    assert all(bytes(data[a:b]) == blobs[i] for i, (a, b) in enumerate(off))
    assert video.jpeg_frame_size(np.frombuffer(blobs[0], np.uint8)) == (72, 128)
    with pytest.raises(video.VideoError):
        video.read_avi_mjpeg(__file__)
    # a dropped frame (zero-length chunk, what AVI writers emit for a repeated picture) keeps its frame number: it maps to the
    # previous frame's bytes, so the frames behind it stay aligned with cv2's CAP_PROP_POS_FRAMES
    raw_avi = bytearray(open(path, "rb").read())
    k = raw_avi.index(b"00dc", raw_avi.index(b"movi"))
    k = raw_avi.index(b"00dc", k + 4)   # in front of the second frame
    raw_avi[k:k] = b"00dc" + (0).to_bytes(4, "little")
    path2 = str(tmp_path / "dropped.avi")
    open(path2, "wb").write(bytes(raw_avi))
    data2, off2, _ = video.read_avi_mjpeg(path2)
    assert off2.shape == (6, 2) and tuple(off2[1]) == tuple(off2[0])
    assert [bytes(data2[a:b]) for a, b in off2] == [blobs[0], blobs[0]] + blobs[1:]
    # a truncated header is "not opened", not an exception out of the constructor
    open(str(tmp_path / "cut.avi"), "wb").write(bytes(raw_avi[:60]))
    cap = video.VideoCapture(str(tmp_path / "cut.avi"))
    assert not cap.isOpened()


# ---------------------------------------------------------------------------------------------------------------------


@pytest.fixture(scope="module")
def decoder():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = video.MjpegDecoder(max_frames=16, max_height=1080, max_width=1920, max_bytes=32 << 20)
    yield d
    d.close()


def _decode(decoder, blobs, h, w, rgb=False):
    import torch

    data = np.frombuffer(b"".join(blobs), np.uint8)
    ends = np.cumsum([len(b) for b in blobs])
    spans = np.stack([ends - [len(b) for b in blobs], ends], axis=1)
    st = torch.zeros(len(blobs), dtype=torch.int32, device="cuda")
    out = decoder.decode(data, spans, h, w, status=st, rgb=rgb)
    torch.cuda.synchronize()
    return out.cpu().numpy(), st.cpu().numpy()


@pytest.mark.gpu
def test_mjpeg_decode_is_bit_exact(decoder):
    """Every sampling / quality / restart-marker / table variant and odd frame sizes: device frames == oracle (BGR)."""
    from oracle import jpeg

    for name, fr, kw in _cases():
        blob = synth.encode_jpeg_frames([fr], **kw)[0]
        got, st = _decode(decoder, [blob], fr.shape[0], fr.shape[1])
        assert st[0] == 0, name
        assert np.array_equal(got[0], jpeg.decode_bgr(blob)), name
    grey = synth.encode_jpeg_frames([synth.make_frame(1, 200, 300)[..., 0]], quality=90)[0]
    got, st = _decode(decoder, [grey], 200, 300, rgb=True)
    assert st[0] == 0 and np.array_equal(got[0], jpeg.decode(grey))


@pytest.mark.gpu
def test_mjpeg_clip_with_changing_tables(decoder):
    """One call, 12 frames of a 720p clip whose quantisation / Huffman tables and restart interval change from frame to
    frame (a stream is allowed to): every frame equals the oracle; the frames also equal live Pillow."""
    from oracle import jpeg

    h, w = 720, 1280
    frames = synth.make_frames(12, h, w)
    blobs = []
    for i, f in enumerate(frames):
        kw = [dict(quality=95, restart_marker_rows=1), dict(quality=60, optimize=True), dict(quality=85, restart_marker_blocks=7),
              dict(quality=95)][i % 4]
        blobs += synth.encode_jpeg_frames([f], **kw)
    got, st = _decode(decoder, blobs, h, w)
    assert (st == 0).all()
    for i, b in enumerate(blobs):
        assert np.array_equal(got[i], jpeg.decode_bgr(b)), i
        assert np.array_equal(got[i][..., ::-1], _pil_decode(b)), i


@pytest.mark.gpu
def test_mjpeg_groups_streams_and_calls_in_flight():
    """The decoder's concurrency is invisible in the frames: 1 .. 4 frame groups per call (each on a stream of the
    handle's own), spans given out of file order, and calls enqueued back to back on two streams without a
    synchronisation in between (they alternate between the handle's two scratch sets) all give the oracle's frames."""
    import torch

    from oracle import jpeg

    h, w = 360, 640
    frames = synth.make_frames(13, h, w)
    clips = []
    for c in range(2):
        blobs = []
        for i, f in enumerate(frames if c == 0 else frames[::-1]):
            kw = [dict(quality=95), dict(quality=70, restart_marker_blocks=11), dict(quality=90, optimize=True)][(i + c) % 3]
            blobs += synth.encode_jpeg_frames([f], **kw)
        order = np.random.default_rng(5 + c).permutation(len(blobs))       # frame f of the call = file order[f]
        sizes = np.array([len(b) for b in blobs])
        ends = np.cumsum(sizes)
        spans = np.stack([ends - sizes, ends], axis=1)[order]
        data = torch.from_numpy(np.frombuffer(b"".join(blobs), np.uint8).copy()).pin_memory()
        want = np.stack([jpeg.decode_bgr(blobs[j]) for j in order])
        clips.append((data, spans, want))
    dec = video.MjpegDecoder(16, h, w, 8 << 20)
    try:
        for groups in (1, 2, 3, 4):
            dec.set_groups(groups)
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            outs, sts = [], []
            for k in range(6):  # six calls in flight, two clips, two caller streams
                data, spans, _ = clips[k % 2]
                with torch.cuda.stream(streams[k % 2]):
                    st = torch.zeros(13, dtype=torch.int32, device="cuda")
                    outs.append(dec.decode(data, spans, h, w, status=st))
                    sts.append(st)
            torch.cuda.synchronize()
            for k in range(6):
                assert int(sts[k].abs().sum()) == 0, (groups, k)
                assert np.array_equal(outs[k].cpu().numpy(), clips[k % 2][2]), (groups, k)
        with pytest.raises(Exception):
            dec.set_groups(5)
    finally:
        dec.close()


@pytest.mark.gpu
def test_mjpeg_full_size_clip():
    """BASELINE.json configs[1]'s clip as the decoder meets it in ``bench.py``: 64 x 1080p frames at OpenCV's defaults in
    one call. Every frame equals the oracle (all 64: the oracle takes 0.8 s per frame), no frame is flagged, the call equals
    its frames decoded one at a time (batch composition, groups and subsequence sizing do not enter the result), and the
    decoded clip is the source clip up to JPEG's loss."""
    import torch

    from oracle import jpeg

    n, h, w = 64, 1080, 1920
    frames = synth.make_frames(n, h, w)
    blobs = synth.encode_jpeg_frames(frames, quality=95)
    sizes = np.array([len(b) for b in blobs])
    ends = np.cumsum(sizes)
    spans = np.stack([ends - sizes, ends], axis=1)
    data = torch.from_numpy(np.frombuffer(b"".join(blobs), np.uint8).copy()).pin_memory()
    dec = video.MjpegDecoder(n, h, w, int(ends[-1]) + 4096)
    try:
        st = torch.zeros(n, dtype=torch.int32, device="cuda")
        out = dec.decode(data, spans, h, w, status=st)
        torch.cuda.synchronize()
        assert int(st.abs().sum()) == 0
        got = out.cpu().numpy()
        for i in range(n):
            assert np.array_equal(got[i], jpeg.decode_bgr(blobs[i])), i
        one = torch.empty((1, h, w, 3), dtype=torch.uint8, device="cuda")
        for i in (0, 31, 63):
            dec.decode(data, spans[i:i + 1], h, w, out=one)
            torch.cuda.synchronize()
            assert torch.equal(one[0], out[i]), i
        d = got[::8].astype(np.float64) - frames[::8]
        assert 10 * np.log10(255.0 ** 2 / (d * d).mean()) > 28.0
    finally:
        dec.close()


@pytest.mark.gpu
def test_mjpeg_errors_are_reported(decoder):
    from PIL import Image

    from playaid_core_amd import _lib
    from playaid_core_amd.engine import EngineError

    h, w = 64, 96
    good = synth.encode_jpeg_frames([synth.make_frame(0, h, w)], quality=90, restart_marker_blocks=2)[0]
    b = io.BytesIO()
    Image.fromarray(synth.make_frame(0, h, w)).save(b, "JPEG", progressive=True)
    with pytest.raises(EngineError) as ei:  # header the decoder does not take: nothing is enqueued
        _decode(decoder, [good, b.getvalue()], h, w)
    assert ei.value.code == _lib.PA_ERR_INVALID_ARG and "frame 1" in str(ei.value) and "progressive" in str(ei.value)
    with pytest.raises(EngineError) as ei:  # size differs from what the caller said
        _decode(decoder, [good], h, w + 16)
    assert ei.value.code == _lib.PA_ERR_INVALID_ARG
    with pytest.raises(EngineError) as ei:
        _decode(decoder, [good] * 17, h, w)
    assert ei.value.code == _lib.PA_ERR_CAPACITY
    # the SAME frame many times over (spans may repeat): the clean stream is bounded by the sum of the scans, not by the byte
    # range the spans cover -- a handle sized for one copy must refuse, not write past its buffer
    big = synth.encode_jpeg_frames([synth.make_frame(3, 256, 384)], quality=95)[0]
    small = video.MjpegDecoder(max_frames=16, max_height=256, max_width=384, max_bytes=len(big) + 64)
    try:
        got, st = _decode(small, [big], 256, 384)
        assert st[0] == 0
        data = np.frombuffer(big, np.uint8)
        spans = np.array([[0, len(big)]] * 12, np.int64)
        with pytest.raises(EngineError) as ei:
            small.decode(data, spans, 256, 384)
        assert ei.value.code == _lib.PA_ERR_CAPACITY
        two = np.array([[0, len(big)]] * 2, np.int64)   # (two copies still fit the slack or are refused: never a fault)
        try:
            out2 = small.decode(data, two, 256, 384)
            import torch

            torch.cuda.synchronize()
            assert np.array_equal(out2[0].cpu().numpy(), got[0]) and np.array_equal(out2[1].cpu().numpy(), got[0])
        except EngineError as exc:
            assert exc.code == _lib.PA_ERR_CAPACITY
    finally:
        small.close()
    # a scan cut short: restart markers are missing -> status bit 2, no fault, the intact frame beside it is fine
    cut = good[: len(good) // 2]
    got, st = _decode(decoder, [cut, good], h, w)
    assert st[0] & 2 and st[1] == 0
    from oracle import jpeg

    assert np.array_equal(got[1], jpeg.decode_bgr(good))
    # a corrupted entropy segment decodes to garbage or raises a status bit, but stays inside its buffers
    bad = bytearray(good)
    for k in range(len(bad) - 200, len(bad) - 2, 7):
        bad[k] = 0x5A
    got, st = _decode(decoder, [bytes(bad), good], h, w)
    assert st[1] == 0 and np.array_equal(got[1], jpeg.decode_bgr(good))


@pytest.mark.gpu
def test_videocapture_mirror_over_an_avi(tmp_path, decoder):
    """cv2.VideoCapture's call shape (ai_runner.py:153,404-405; manuscript.py:70-86,154-155) over an MJPG .avi."""
    from oracle import jpeg

    n, h, w = 9, 360, 640
    frames = synth.make_frames(n, h, w)
    blobs = synth.encode_jpeg_frames(frames, quality=95)
    path = str(tmp_path / "clip.avi")
    video.write_avi_mjpeg(path, blobs, 60.0, w, h)
    cap = video.VideoCapture(path)
    assert cap.isOpened()
    assert cap.get(video.CAP_PROP_FPS) == 60.0 and int(cap.get(video.CAP_PROP_FRAME_COUNT)) == n
    assert (int(cap.get(video.CAP_PROP_FRAME_HEIGHT)), int(cap.get(video.CAP_PROP_FRAME_WIDTH))) == (h, w)
    cap.set(video.CAP_PROP_POS_FRAMES, 4)
    ok, fr = cap.read()
    assert ok and fr.dtype == np.uint8 and np.array_equal(fr, jpeg.decode_bgr(blobs[4]))
    ok, fr = cap.read()  # the position advanced
    assert ok and np.array_equal(fr, jpeg.decode_bgr(blobs[5]))
    cap.set(video.CAP_PROP_POS_FRAMES, n)
    assert cap.read() == (False, None)
    batch = cap.read_frames(2, 5).cpu().numpy()
    for i in range(5):
        assert np.array_equal(batch[i], jpeg.decode_bgr(blobs[2 + i]))
    cap.release()
    assert not cap.isOpened()
    assert not video.VideoCapture(str(tmp_path / "missing.avi")).isOpened()
    # raw concatenation and image-sequence directory
    raw = str(tmp_path / "clip.mjpeg")
    open(raw, "wb").write(b"".join(blobs))
    cap = video.VideoCapture(raw)
    assert cap.frame_count() == n and np.array_equal(cap.read_frames(8, 1).cpu().numpy()[0], jpeg.decode_bgr(blobs[8]))
    cap.release()
    d = tmp_path / "seq"
    os.makedirs(d)
    for i, b in enumerate(blobs[:3]):
        open(d / f"frame_{i:04d}.jpg", "wb").write(b)
    cap = video.VideoCapture(str(d))
    assert cap.frame_count() == 3 and cap.read()[0]
    cap.release()


@pytest.mark.gpu
def test_airunner_from_a_video_file(tmp_path, state_dict):
    """Rows a1 + b2 together: AIRunner(<video>.avi) -- frames decoded on the device, label files where run_yolo leaves them --
    writes the same ai_output.yaml as the runner fed with the ORACLE's decode of the same JPEG files."""
    import yaml

    from oracle import jpeg
    from playaid_core_amd.ai_runner import AIRunner, ClipSource
    from playaid_core_amd.anim_ontology import MOVE_TO_CLASS_ID
    from playaid_core_amd.cnn_action_detector import CNNActionDetector

    n, h, w = 20, 720, 1280
    synth_clip = ClipSource.synthetic(n, h, w)
    blobs = synth.encode_jpeg_frames(synth_clip.frames, quality=95)
    out_dir = tmp_path / "ai_cache" / "match"
    os.makedirs(out_dir / "labels")
    for i, text in enumerate(synth_clip.labels):
        open(out_dir / "labels" / f"match_{i + 1}.txt", "w").write(text)
    path = str(tmp_path / "match.avi")
    video.write_avi_mjpeg(path, blobs, 60.0, w, h)
    ckpt = str(tmp_path / "seeded.ckpt")
    synth.save_checkpoint(ckpt, seed=1234)
    model = CNNActionDetector.load_from_checkpoint(ckpt, actions=list(MOVE_TO_CLASS_ID.keys()), max_batch_frames=32,
                                                   max_clip_frames=64, max_frame_height=h, max_frame_width=w)
    runner = AIRunner(path, model=model, output_dir=str(out_dir))
    assert runner.video_name == "match" and runner.max_frames == n and runner.clip.frames.is_cuda
    runner.run_action_recognition()
    runner.write_output()
    decoded = np.stack([jpeg.decode_bgr(b) for b in blobs])
    assert np.array_equal(runner.clip.frames.cpu().numpy(), decoded)
    ref = AIRunner(ClipSource(decoded, synth_clip.labels, name="ref"), model=model, output_dir=str(tmp_path / "ref"))
    ref.run_action_recognition()
    ref.write_output()
    a, b = yaml.safe_load(open(runner.ai_output_file)), yaml.safe_load(open(ref.ai_output_file))
    assert a == b and len(a["Joker"]) == n - 1
    # the codec's loss is real: the raw frames give other log-probabilities
    raw = AIRunner(synth_clip, model=model, output_dir=str(tmp_path / "raw"))
    raw.run_action_recognition()
    assert np.abs(raw._results["logp"] - runner._results["logp"]).max() > 1e-4


@pytest.mark.gpu
def test_flat_frames_need_the_exact_mode(decoder):
    """Long runs of identical blocks (black bars) re-synchronise slowly: the enqueued verify passes may not settle, which the
    status word says (bit 8), and the exact mode (verify until nothing changes) decodes the frame bit-exactly."""
    import torch

    from oracle import jpeg

    h, w = 1080, 1920
    flat = np.zeros((h, w, 3), np.uint8)
    flat[: h // 2] = 40
    flat[100:200, 300:900] = synth.make_frame(3, 100, 600)
    blob = synth.encode_jpeg_frames([flat], quality=95)[0]
    want = jpeg.decode_bgr(blob)
    got, st = _decode(decoder, [blob], h, w)
    if st[0] == 0:
        assert np.array_equal(got[0], want)
    else:
        assert st[0] & 8
    decoder.set_sync_rounds(0)
    try:
        got, st = _decode(decoder, [blob], h, w)
        assert st[0] == 0 and np.array_equal(got[0], want) and decoder.last_sync_rounds() >= 1
    finally:
        decoder.set_sync_rounds(video.MjpegDecoder.DEFAULT_SYNC_ROUNDS)
    # the capture mirror does that by itself
    cap = video.VideoCapture([blob])
    ok, fr = cap.read()
    assert ok and np.array_equal(fr, want)
    cap.release()
    del torch
