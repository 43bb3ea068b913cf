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
    assert ctypes.sizeof(_lib.pa_kernel_stat) == 80


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
    # bench.py: ONE import, inside cpu_baseline(); __graft_entry__.py: inside smoke(); scripts/ are measurement helpers run by hand
    import ast

    for name, func in (("bench.py", "cpu_baseline"), ("__graft_entry__.py", "smoke")):
        tree = ast.parse(open(os.path.join(ROOT, name)).read())
        inside = set()
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef) and node.name == func:
                inside = {id(n) for n in ast.walk(node)}
        hits = [n for n in ast.walk(tree) if (isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle") or
                (isinstance(n, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in n.names))]
        assert hits and all(id(n) in inside for n in hits), f"{name}: the oracle may only be imported inside {func}()"


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


def test_round5_host_planners_reject_bad_tables_and_stay_in_bounds(lib):
    """Host-side planning code added in rounds 4-5, driven without a GPU (and under ASan / UBSan by scripts/asan_host.sh):
    ``pa_detector_create``'s layer-table validation with truncated / inconsistent tables, ``pa_detector_forward_timed`` with a
    short result buffer, the Winograd filter transform writing exactly its ``pa_wino_weight_floats`` floats for every layout, and
    the argument checks of the round's new entries."""
    import numpy as np

    from playaid_core_amd import _lib, synth
    from playaid_core_amd.yolov5 import build_yolov5s_table

    layers, bufs, weights, rows = build_yolov5s_table(synth.make_yolov5s_state_dict(), (384, 640), 6)

    def create(mutate=None, n_layers=None, n_weights=None, bufs_=None, net=(384, 640), nc=6, max_images=2):
        arr = (_lib.pa_net_layer * len(layers))(*layers)
        arr = (_lib.pa_net_layer * len(layers)).from_buffer_copy(bytes(arr))   # a private copy: the table object is reused
        if mutate:
            mutate(arr)
        b = list(bufs_ if bufs_ is not None else bufs)
        w = np.ascontiguousarray(weights[: n_weights if n_weights is not None else len(weights)], dtype=np.float32)
        hh = ctypes.c_void_p(0)
        rc = lib.pa_detector_create(0, arr, n_layers if n_layers is not None else len(layers), (ctypes.c_int64 * len(b))(*b), len(b),
                                    w.ctypes.data_as(ctypes.c_void_p), w.size, max_images, net[0], net[1], nc, ctypes.byref(hh))
        msg = lib.pa_detector_last_error(hh).decode() if hh else ""
        return rc, msg, hh

    def gone(hh):
        if hh:
            lib.pa_detector_destroy(hh)

    first_conv = next(i for i, l in enumerate(layers) if l.kind == 0)
    first_3x3 = next(i for i, l in enumerate(layers) if l.kind == 0 and l.ksize == 3 and l.stride == 1)
    cases = [
        (dict(n_weights=len(weights) // 2), "weights outside the blob"),                                   # a truncated blob
        (dict(n_layers=first_conv + 3), "no decode layer"),                                                 # a truncated table
        (dict(mutate=lambda a: setattr(a[first_conv], "cin", 48)), "unsupported convolution"),
        (dict(mutate=lambda a: setattr(a[first_conv], "in_coff", 16)), "slice outside its buffer"),        # slices start at multiples of 32
        (dict(mutate=lambda a: setattr(a[first_3x3], "out_cstride", a[first_3x3].cout - 4)), "slice outside its buffer"),
        (dict(mutate=lambda a: setattr(a[first_conv], "in_buf", 99)), "slice outside its buffer"),
        (dict(mutate=lambda a: setattr(a[0], "cout", 64)), "bad stem"),
        (dict(mutate=lambda a: setattr(a[first_conv], "kind", 9)), "unknown kind"),
        (dict(bufs_=[max(1, v // 8) for v in bufs]), "bad stem"),                                           # buffers too small: the first layer's slice already
    ]
    for kw, want in cases:
        rc, msg, hh = create(**kw)
        gone(hh)
        assert rc == _lib.PA_ERR_INVALID_ARG and want in msg, (kw.keys(), rc, msg)
    assert create(net=(380, 640))[0] == _lib.PA_ERR_INVALID_ARG and create(nc=0)[0] == _lib.PA_ERR_INVALID_ARG
    # a valid table passes the validation and stops at the device (none here) -- with the handle handed back for the error text;
    # a timed forward with too short a result buffer is refused before anything is enqueued
    rc, msg, hh = create()
    assert rc in (_lib.PA_OK, _lib.PA_ERR_NO_DEVICE, _lib.PA_ERR_HIP), (rc, msg)
    if hh:
        us = np.zeros(4, np.float32)
        z = ctypes.c_void_p(0)
        assert lib.pa_detector_forward_timed(hh, z, 1, 720, 1280, z, z, us.ctypes.data_as(ctypes.c_void_p), 4) == _lib.PA_ERR_INVALID_ARG
        assert lib.pa_detector_forward_timed(hh, z, 1, 720, 1280, z, z, z, 1000) == _lib.PA_ERR_INVALID_ARG
    gone(hh)
    # the Winograd filter transform: every layout the engine uses, into buffers of exactly the advertised size
    rng = np.random.default_rng(1)
    for cin, cout, bn in ((8, 32, 32), (32, 32, 32), (64, 64, 64), (64, 64, 32), (128, 128, 64), (256, 256, 32), (24, 96, 32)):
        w = rng.standard_normal((cout, 3, 3, cin)).astype(np.float32)
        n = lib.pa_wino_weight_floats(cin, cout)
        assert n == 16 * cin * cout
        ug = np.full(n, np.nan, np.float32)
        assert lib.pa_wino_transform_weights(w.ctypes.data_as(ctypes.c_void_p), cin, cout, bn, ug.ctypes.data_as(ctypes.c_void_p)) == 0
        assert np.isfinite(ug).all(), "every float of the layout is written exactly once"
        # the centre position of every filter is (g00 + g01 + ... ) / 4-type sums; a cheap invariant: total energy is preserved up to G's norm
        assert abs(float(np.abs(ug).sum())) > 0
    z = ctypes.c_void_p(0)
    assert lib.pa_wino_channels_per_workgroup(48, 10) == 0 and lib.pa_wino_channels_per_workgroup(64, 0) == 0
    assert lib.pa_wino_conv3x3(z, z, z, z, z, 1, 8, 8, 8, 32, 32, 8, 32, 1, 0, 0, z) == _lib.PA_ERR_INVALID_ARG
    # (host-side checks of the launcher, reached with non-null but never dereferenced addresses: pixel pitches that are not whole
    # groups of four floats, bases that are not 16-byte aligned, a workgroup width the layer does not have)
    a16, a4 = ctypes.c_void_p(4096), ctypes.c_void_p(4100)
    assert lib.pa_wino_conv3x3(a16, a16, z, z, a16, 1, 8, 8, 8, 32, 32, 10, 32, 1, 0, 0, z) == _lib.PA_ERR_INVALID_ARG
    assert lib.pa_wino_conv3x3(a4, a16, z, z, a16, 1, 8, 8, 8, 32, 32, 8, 32, 1, 0, 0, z) == _lib.PA_ERR_INVALID_ARG
    assert lib.pa_wino_conv3x3(a16, a16, z, z, a16, 1, 8, 8, 8, 32, 64, 8, 32, 1, 0, 0, z) == _lib.PA_ERR_INVALID_ARG
    assert lib.pa_wino_conv3x3(a16, a16, z, z, a16, 1, 6, 8, 8, 32, 32, 8, 32, 1, 0, 0, z) == _lib.PA_ERR_INVALID_ARG
    # the split-K form (ABI 10) wants its scratch: a null or misaligned slab, null tickets or none of them is refused before any launch,
    # and the shape checks of the plain form apply to it as well
    sk = lambda slab, nf, tk, nt, ips=8: lib.pa_wino_conv3x3_splitk(a16, a16, z, z, a16, 1, 8, 8, 8, 32, 32, ips, 32, 1, 0, 0, slab, nf, tk, nt, z)
    assert sk(z, 1 << 20, a16, 16) == _lib.PA_ERR_INVALID_ARG
    assert sk(a4, 1 << 20, a16, 16) == _lib.PA_ERR_INVALID_ARG
    assert sk(a16, 1 << 20, z, 16) == _lib.PA_ERR_INVALID_ARG
    assert sk(a16, 1 << 20, a16, 0) == _lib.PA_ERR_INVALID_ARG
    assert sk(a16, 1 << 20, a16, 16, ips=10) == _lib.PA_ERR_INVALID_ARG
    # this round's engine entries refuse a null engine / null tables before touching anything
    assert lib.pa_detector_plan(z, z, z, z, z, 4, z, z, z, z, z, z, z) == _lib.PA_ERR_INVALID_ARG
    assert lib.pa_detector_plan_desc(z, z, z, 4, 2, 100, z, 0, 0, z) == _lib.PA_ERR_INVALID_ARG
    assert lib.pa_square_crops_src(z, z, 1, 720, 1280, z, z, 1, 30, 0, z, z, z) == _lib.PA_ERR_INVALID_ARG


def _device_disassembly(obj_name):
    """gfx950 disassembly of one in-tree object (None when the LLVM tools of the ROCm image are not there)."""
    import glob
    import shutil
    import subprocess
    import tempfile

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    obj = os.path.join(ROOT, "playaid_core_amd", "csrc", obj_name)
    if not (os.path.exists(objdump) and os.path.exists(obj)):
        return None
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(obj, os.path.join(td, obj_name))
        subprocess.run([objdump, "--offloading", obj_name], cwd=td, check=True, capture_output=True)
        dev = glob.glob(os.path.join(td, obj_name + ".*gfx950"))
        assert len(dev) == 1, dev
        return subprocess.run([objdump, "-d", dev[0]], check=True, capture_output=True, text=True).stdout


def test_wino_object_uses_m0_only_for_its_lds_dma(lib):
    """csrc/wino.hip writes M0 inside an inline-assembly LDS-DMA and cannot declare it (hipcc rejects "m0" on a clobber list):
    the static check ADVICE round 5 asked for instead -- in the built object every mention of M0 is that statement's own
    `s_mov_b32 m0, ...`, and nothing indexes registers through it (no movrel: the accumulator arrays are fully unrolled)."""
    asm = _device_disassembly("wino.o")
    if asm is None:
        pytest.skip("no llvm-objdump / object in this environment")
    uses = [ln.split("//")[0].split() for ln in asm.splitlines() if re.search(r"\bm0\b", ln.split("//")[0])]
    assert len(uses) > 100 and all(u[0] == "s_mov_b32" and u[1].rstrip(",") == "m0" for u in uses), [u for u in uses if u[0] != "s_mov_b32"][:5]
    assert not re.search(r"movrel|s_sendmsg\b(?!.*MSG_DEALLOC)|lds_direct", asm)


def test_pgemm_store_count_matches_the_counted_waits(lib):
    """ADVICE round 5 (csrc/pigemm.hip): the persistent GEMM's waits are `s_waitcnt vmcnt(NLD + k * NST)` with NST = the 16-byte
    stores a wave issues per tile -- MI * 4, five times that with the fused up-sampled copy -- a number the kernel states as a
    constant while the stores themselves are ordinary C++. A compiler that split, merged or duplicated them would make the waits
    under-wait silently. The build check: in the object every pgemm kernel holds exactly NST `global_store_dwordx4`."""
    asm = _device_disassembly("pigemm.o")
    if asm is None:
        pytest.skip("no llvm-objdump / object in this environment")
    kernels = re.split(r"\n[0-9a-f]+ <(_ZN2pa12pgemm_kernel[^>]*)>:\n", asm)
    found = {}
    for name, body in zip(kernels[1::2], kernels[2::2]):
        m = re.match(r"_ZN2pa12pgemm_kernelILi(\d+)ELi(\d+)ELb([01])ELb([01])ELb([01])EEEvNS_10GemmParamsE", name)
        assert m, name
        bm, bn, _silu, _stamp, up = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))
        wm = 2 if bn == 64 else 4
        nst = (bm // wm // 32) * 4 * (5 if up else 1)
        body = body.split("s_endpgm")[0]
        found[name] = (len(re.findall(r"\bglobal_store_dwordx4\b", body)), nst)
    assert len(found) >= 6, sorted(found)
    for name, (count, nst) in found.items():
        assert count == nst, f"{name}: {count} 16-byte stores in the object, the waits count {nst} per tile"


def test_psgemm_copy_count_matches_the_loaders_waits(lib):
    """csrc/psgemm.hip: the loader waves wait with `s_waitcnt vmcnt(k * NLD)`, NLD = 4 activation + BN * 3 / 64 weight LDS-DMA
    instructions per k-step (BN = 32: 2 weight pieces, padding included). Every `issue()` must therefore be exactly NLD
    `buffer_load_dwordx4 ... lds`: NSTAGE inlined copies of it in the prologue and one in the loop."""
    asm = _device_disassembly("psgemm.o")
    if asm is None:
        pytest.skip("no llvm-objdump / object in this environment")
    kernels = re.split(r"\n[0-9a-f]+ <(_ZN2pa13psgemm_kernel[^>]*)>:\n", asm)
    seen = 0
    for name, body in zip(kernels[1::2], kernels[2::2]):
        m = re.match(r"_ZN2pa13psgemm_kernelILi(\d+)ELi(\d+)ELi(\d)ELb([01])EEE", name)
        assert m, name
        bn, nstage = int(m.group(1)), int(m.group(2))
        nld = 4 + (8 if bn == 32 else bn * 3 // 16) // 4
        body = "\n".join(body.split("s_endpgm")[:-1])   # (up to the kernel's last exit: the loader waves return early, the consumers at the end)
        count = len(re.findall(r"\bbuffer_load_dwordx4\b[^\n]*\blds\b", body))
        assert count == nld * (nstage + 1), f"{name}: {count} LDS-DMA instructions, expected {nld} x ({nstage} + 1)"
        seen += 1
    assert seen >= 12
