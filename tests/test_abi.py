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
