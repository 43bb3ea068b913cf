"""The C-ABI library builds for gfx950, loads without a GPU and exports every
symbol include/playaid_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from playaid_core_amd import _build, _lib

    _build.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "playaid_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pa_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound(lib):
    from playaid_core_amd import _lib

    declared = _declared_symbols()
    assert "pa_create" in declared and "pa_infer_clip" in declared and len(declared) >= 15
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert set(declared) == bound, set(declared) ^ bound
    for name in declared:
        assert hasattr(lib, name)


def test_version_status_strings_and_blob_size(lib):
    from playaid_core_amd import _lib, synth, weights

    assert lib.pa_abi_version() == _lib.PA_ABI_VERSION
    assert lib.pa_status_string(0) == b"ok"
    assert b"capacity" in lib.pa_status_string(_lib.PA_ERR_CAPACITY)
    blob = weights.pack_state_dict(synth.make_state_dict(), 7, 63)
    assert blob.nbytes == lib.pa_weight_blob_bytes(7, 63)
    assert lib.pa_weight_blob_bytes(5, 10) < blob.nbytes


def test_struct_layouts_match_header():
    from playaid_core_amd import _lib

    assert ctypes.sizeof(_lib.pa_record) == 16
    assert ctypes.sizeof(_lib.pa_config) == 16 * 4
    assert ctypes.sizeof(_lib.pa_kernel_stat) == 72


def test_create_rejects_bad_arguments_without_gpu(lib):
    """Argument validation happens before any device call."""
    from playaid_core_amd import _lib

    cfg = _lib.pa_config()
    cfg.abi_version = 999
    h = ctypes.c_void_p(0)
    buf = (ctypes.c_uint8 * 64)()
    assert lib.pa_create(ctypes.byref(cfg), buf, 64, ctypes.byref(h)) == _lib.PA_ERR_INVALID_ARG
    cfg.abi_version = _lib.PA_ABI_VERSION
    cfg.sequence_length, cfg.num_actions, cfg.num_fighters = 7, 63, 2
    cfg.max_batch_frames, cfg.max_clip_frames, cfg.max_frame_height, cfg.max_frame_width = 8, 64, 720, 1280
    assert lib.pa_create(ctypes.byref(cfg), buf, 64, ctypes.byref(h)) == _lib.PA_ERR_BAD_WEIGHTS
    cfg.sequence_length = 4  # even window
    assert lib.pa_create(ctypes.byref(cfg), buf, 64, ctypes.byref(h)) == _lib.PA_ERR_INVALID_ARG
    cfg.sequence_length, cfg.compute_dtype = 7, 5  # unknown arithmetic type
    assert lib.pa_create(ctypes.byref(cfg), buf, 64, ctypes.byref(h)) == _lib.PA_ERR_INVALID_ARG


def test_side_model_handles_reject_bad_tables_without_gpu(lib):
    """pa_lstm_create / pa_encoder_create / pa_convnet_create check their blobs and tables before any device call."""
    import numpy as np

    from playaid_core_amd import _lib, synth
    from playaid_core_amd.rnn_action_detector import pack_lstm_blob

    h = ctypes.c_void_p(0)
    blob = pack_lstm_blob(synth.make_rnn_state_dict(seed=3, num_actions=9), 9)
    ptr = blob.ctypes.data_as(ctypes.c_void_p)
    assert lib.pa_lstm_create(0, 300, 512, 3, 10, 64, ptr, blob.nbytes, ctypes.byref(h)) == _lib.PA_ERR_BAD_WEIGHTS   # A mismatch
    assert lib.pa_lstm_create(0, 300, 520, 3, 9, 64, ptr, blob.nbytes, ctypes.byref(h)) == _lib.PA_ERR_INVALID_ARG    # hidden > 512
    assert lib.pa_lstm_create(0, 300, 512, 3, 9, 0, ptr, blob.nbytes, ctypes.byref(h)) == _lib.PA_ERR_INVALID_ARG
    assert lib.pa_encoder_create(0, 2048, 247, 7, 9, 8, 3, 2048, 63, 64, ptr, blob.nbytes, ctypes.byref(h)) == _lib.PA_ERR_BAD_WEIGHTS
    assert lib.pa_encoder_create(0, 2048, 247, 7, 9, 4, 3, 2048, 63, 64, ptr, blob.nbytes, ctypes.byref(h)) == _lib.PA_ERR_INVALID_ARG  # 64-wide heads
    assert lib.pa_encoder_blob_bytes(2048, 247, 7, 9, 3, 2048, 63) > 4 * 3 * (3 * 256 * 256 + 2 * 256 * 2048)

    def conv(**kw):
        d = dict(kind=0, cin=64, cout=64, ksize=3, stride=1, in_hw=32, in_buf=0, in_pad=1, out_buf=1, out_pad=0, res_buf=-1, relu=1,
                 w_off=0, b_off=64 * 9 * 64)
        d.update(kw)
        return d

    def create(descs, bufs, n_weights=64 * 9 * 64 + 64):
        arr = (_lib.pa_conv_desc * len(descs))()
        for i, d in enumerate(descs):
            for k, v in d.items():
                setattr(arr[i], k, v)
        w = np.zeros(n_weights, np.float32)
        hh = ctypes.c_void_p(0)
        rc = lib.pa_convnet_create(0, arr, len(descs), (ctypes.c_int64 * len(bufs))(*bufs), len(bufs), w.ctypes.data_as(ctypes.c_void_p),
                                   w.size, 4, ctypes.byref(hh))
        msg = lib.pa_convnet_last_error(hh).decode() if hh else ""
        if hh:
            lib.pa_convnet_destroy(hh)
        return rc, msg

    big = 34 * 34 * 64
    rc, msg = create([conv(cin=48)], [big, big])
    assert rc == _lib.PA_ERR_INVALID_ARG and "unsupported convolution" in msg          # cin % 32
    rc, msg = create([conv(in_pad=0)], [big, big])
    assert rc == _lib.PA_ERR_INVALID_ARG                                               # a 3x3 needs a border
    rc, msg = create([conv()], [big, big], n_weights=100)
    assert rc == _lib.PA_ERR_BAD_WEIGHTS and "outside the blob" in msg
    rc, msg = create([conv()], [1000, big])
    assert rc == _lib.PA_ERR_INVALID_ARG and "too small" in msg
    rc, msg = create([conv(), conv(in_hw=16, out_buf=1)], [big, big])
    assert rc == _lib.PA_ERR_INVALID_ARG and "two geometries" in msg                   # bordered buffer 0 reused at 16 x 16
    rc, msg = create([conv(kind=7)], [big, big])
    assert rc == _lib.PA_ERR_INVALID_ARG and "unknown kind" in msg


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from playaid_core_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryError):
        _lib.load()


def test_product_path_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "playaid_core_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_jpeg_header_probe_on_the_host(lib):
    """pa_mjpeg_probe is pure host code (the marker parser of pa_mjpeg_decode): sizes, sampling, restart interval of real
    files; every truncation of a file and random corruptions are refused or parsed, never read out of bounds (this test is
    what scripts/asan_host.sh runs against the sanitizer build)."""
    import io

    import numpy as np
    from PIL import Image

    from playaid_core_amd import _lib, synth

    def probe(blob):
        info = (ctypes.c_int32 * 8)()
        why = ctypes.create_string_buffer(128)
        # an exact-size heap buffer: under scripts/asan_host.sh any read past the file's last byte hits a red zone
        buf = np.frombuffer(blob if blob else b"\0", np.uint8).copy()
        rc = lib.pa_mjpeg_probe(buf.ctypes.data_as(ctypes.c_void_p), len(blob), info, why, 128)
        return rc, list(info), why.value.decode()

    fr = synth.make_frame(1, 90, 130)
    blob = synth.encode_jpeg_frames([fr], quality=90, restart_marker_blocks=3)[0]
    rc, info, why = probe(blob)
    assert rc == 0 and info[:6] == [90, 130, 3, 2, 2, 3] and why == "" and blob[info[6] - 2 - 12:info[6] - 12][:1] != b""
    rc, info, _ = probe(synth.encode_jpeg_frames([fr], subsampling=0)[0])
    assert rc == 0 and info[3:6] == [1, 1, 0]
    rc, info, _ = probe(synth.encode_jpeg_frames([fr[..., 0]])[0])
    assert rc == 0 and info[2] == 1
    b = io.BytesIO()
    Image.fromarray(fr).save(b, "JPEG", progressive=True)
    rc, _, why = probe(b.getvalue())
    assert rc == _lib.PA_ERR_INVALID_ARG and "progressive" in why
    for cut in range(0, info[6] + 40, 1):  # every truncation of the header
        rc, _, why = probe(blob[:cut])
        assert rc in (0, _lib.PA_ERR_INVALID_ARG) and (rc == 0 or why)
    rng = np.random.default_rng(3)
    for _ in range(300):  # corrupted headers
        bad = bytearray(blob[:700])
        for k in rng.integers(2, len(bad), 6):
            bad[k] = int(rng.integers(0, 256))
        rc, _, _ = probe(bytes(bad))
        assert rc in (0, _lib.PA_ERR_INVALID_ARG)
