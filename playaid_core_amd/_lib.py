"""ctypes binding of ``libplayaid_hip.so`` (the C ABI in include/playaid_hip.h).

The product path has no CPU fallback: if the HIP library is missing or does
not load, importing the engine raises ``HipLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PA_LIB_PATH: load a differently built library (e.g. the ablation build used by scripts/)
LIB_PATH = os.environ.get("PA_LIB_PATH") or os.path.join(HERE, "libplayaid_hip.so")

PA_ABI_VERSION = 11
PA_DTYPE_F32 = 0
PA_DTYPE_BF16 = 1
PA_DTYPE_EMULATED_F32 = 2
DTYPES = {"f32": PA_DTYPE_F32, "bf16": PA_DTYPE_BF16, "emulated_f32": PA_DTYPE_EMULATED_F32}
PA_WEIGHT_MAGIC = 0x31574150
PA_LSTM_MAGIC = 0x314C4150
PA_ENCODER_MAGIC = 0x31454150
PA_FEATURE_STRIDE = 1024

PA_OK = 0
PA_ERR_INVALID_ARG = -1
PA_ERR_HIP = -2
PA_ERR_BAD_WEIGHTS = -3
PA_ERR_CAPACITY = -4
PA_ERR_NO_DEVICE = -5
PA_ERR_NOT_READY = -6

PA_CROP_OK = 0
PA_CROP_EMPTY = 1
PA_CROP_BAD_BOX = 2
PA_CROP_UPSCALE = 3
PA_CROP_FILTER_TOO_WIDE = 4
PA_CROP_BAD_FRAME = 5


class HipLibraryError(RuntimeError):
    pass


class pa_config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("device_id", C.c_int32),
        ("sequence_length", C.c_int32),
        ("frame_delta", C.c_int32),
        ("num_actions", C.c_int32),
        ("num_fighters", C.c_int32),
        ("crop_padding", C.c_int32),
        ("max_batch_frames", C.c_int32),
        ("max_clip_frames", C.c_int32),
        ("max_frame_height", C.c_int32),
        ("max_frame_width", C.c_int32),
        ("fighter_class_ids", C.c_int32 * 4),
        ("compute_dtype", C.c_int32),
    ]


class pa_record(C.Structure):
    _fields_ = [
        ("char_id", C.c_int32),
        ("action_id", C.c_int32),
        ("prob", C.c_float),
        ("status", C.c_int32),
    ]


class pa_crop_image(C.Structure):
    _fields_ = [("offset", C.c_int64), ("height", C.c_int32), ("width", C.c_int32)]


class pa_crop_window(C.Structure):
    _fields_ = [("offset", C.c_int64), ("pitch", C.c_int32), ("rows", C.c_int32), ("src_offset", C.c_int64),
                ("src_pitch", C.c_int32), ("row_bytes", C.c_int32)]


class pa_conv_desc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "cin", "cout", "ksize", "stride", "in_hw", "in_buf", "in_pad", "out_buf", "out_pad",
                                        "res_buf", "relu")] + [("w_off", C.c_int64), ("b_off", C.c_int64)]


class pa_net_layer(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "cin", "cout", "ksize", "stride", "in_h", "in_w", "in_buf", "in_coff", "in_cstride", "in_pad",
                                        "out_buf", "out_coff", "out_cstride", "out_pad", "res_buf", "res_coff", "act", "res_after",
                                        "reserved")] + [("w_off", C.c_int64), ("b_off", C.c_int64), ("aux", C.c_float * 8)]


class pa_kernel_stat(C.Structure):
    _fields_ = [
        ("name", C.c_char * 48),
        ("launches", C.c_int32),
        ("total_ms", C.c_float),
        ("flops", C.c_double),
        ("bytes", C.c_double),
        ("flops_executed", C.c_double),
    ]


# every symbol include/playaid_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("pa_create", C.c_int, [C.POINTER(pa_config), _P, C.c_size_t, C.POINTER(_P)]),
    ("pa_create_from_arena", C.c_int, [C.POINTER(pa_config), _P, C.c_size_t, C.POINTER(_P)]),
    ("pa_weights_arena_bytes", C.c_size_t, [_P]),
    ("pa_weights_export", C.c_int, [_P, _P, C.c_size_t, _P]),
    ("pa_destroy", None, [_P]),
    ("pa_last_error", C.c_char_p, [_P]),
    ("pa_status_string", C.c_char_p, [C.c_int]),
    ("pa_abi_version", C.c_int, []),
    ("pa_weight_blob_bytes", C.c_size_t, [C.c_int, C.c_int]),
    ("pa_infer_windows", C.c_int, [_P, _P, C.c_int32, _P, _P]),
    ("pa_square_crops", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    ("pa_runner_inputs", C.c_int, [_P, _P, C.c_size_t, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    ("pa_backbone_crop_images", C.c_int, [_P, _P, C.c_size_t, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    ("pa_detect_postprocess", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_uint32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    ("pa_detector_create", C.c_int, [C.c_int32, C.POINTER(pa_net_layer), C.c_int32, C.POINTER(C.c_int64), C.c_int32, _P, C.c_size_t,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("pa_detector_create_dtype", C.c_int, [C.c_int32, C.POINTER(pa_net_layer), C.c_int32, C.POINTER(C.c_int64), C.c_int32, _P, C.c_size_t,
                                           C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("pa_detector_destroy", None, [_P]),
    ("pa_detector_last_error", C.c_char_p, [_P]),
    ("pa_detector_rows", C.c_int, [_P]),
    ("pa_detector_forward", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    ("pa_detector_forward_timed", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32]),
    ("pa_clean_detections", C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    ("pa_save_one_box_crops", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P,
                                        C.c_size_t, _P, _P]),
    ("pa_project_boxes", C.c_int, [_P, _P, C.c_int32, _P, _P]),
    ("pa_clip_begin", C.c_int, [_P, C.c_int32]),
    ("pa_clip_begin_batch", C.c_int, [_P, C.c_int32, C.c_int32]),
    ("pa_backbone_frames", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P]),
    ("pa_preprocess_frames", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P]),
    ("pa_backbone_slot", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P]),
    ("pa_upload_crop_windows", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_size_t, _P, _P,
                                         C.POINTER(C.c_size_t), _P]),
    ("pa_preprocess_windows", C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P]),
    ("pa_backbone_frames_indexed", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    ("pa_clip_mark_ready", C.c_int, [_P, _P, C.c_int32]),
    ("pa_device_errors", C.c_int, [_P, C.POINTER(C.c_int32), _P]),
    ("pa_backbone_frames_src", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    ("pa_head_frames", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P]),
    ("pa_infer_clip", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P]),
    ("pa_features_export", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    ("pa_features_import", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    ("pa_profile_enable", C.c_int, [_P, C.c_int32]),
    ("pa_profile_read", C.c_int, [_P, C.POINTER(pa_kernel_stat), C.c_int32, C.POINTER(C.c_int32)]),
    ("pa_set_crop_jpeg_quality", C.c_int, [_P, C.c_int32]),
    ("pa_stream_spin", C.c_int, [_P, C.c_int32, _P]),
    ("pa_stream_gate", C.c_int, [_P, C.c_int32, _P]),
    ("pa_stream_gate_open", C.c_int, [_P]),
    ("pa_stream_sync", C.c_int, [_P, _P]),
    ("pa_backbone_windows", C.c_int, [_P, _P, C.c_int32, _P, _P]),
    ("pa_lstm_blob_bytes", C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ("pa_lstm_create", C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_size_t, C.POINTER(_P)]),
    ("pa_lstm_destroy", None, [_P]),
    ("pa_lstm_last_error", C.c_char_p, [_P]),
    ("pa_lstm_last_status", C.c_int, [_P]),
    ("pa_lstm_forward", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    ("pa_convnet_create", C.c_int, [C.c_int32, C.POINTER(pa_conv_desc), C.c_int32, C.POINTER(C.c_int64), C.c_int32, _P, C.c_size_t,
                                    C.c_int32, C.POINTER(_P)]),
    ("pa_convnet_create_dtype", C.c_int, [C.c_int32, C.POINTER(pa_conv_desc), C.c_int32, C.POINTER(C.c_int64), C.c_int32, _P, C.c_size_t,
                                          C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("pa_convnet_destroy", None, [_P]),
    ("pa_convnet_last_error", C.c_char_p, [_P]),
    ("pa_convnet_forward", C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P]),
    ("pa_detector_plan", C.c_int, [_P] * 5 + [C.c_int32] + [_P] * 7),
    ("pa_detector_plan_desc", C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_longlong, _P, C.c_int32, C.c_longlong, _P]),
    ("pa_square_crops_src", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    ("pa_wino_weight_floats", C.c_size_t, [C.c_int32, C.c_int32]),
    ("pa_wino_channels_per_workgroup", C.c_int, [C.c_int32, C.c_int64]),
    ("pa_wino_transform_weights", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P]),
    ("pa_wino_conv3x3", C.c_int, [_P, _P, _P, _P, _P] + [C.c_int32] * 11 + [_P]),
    ("pa_conv_weight_bytes", C.c_size_t, [C.c_int32] * 5),
    ("pa_conv_pack_weights", C.c_int, [_P] + [C.c_int32] * 5 + [_P]),
    ("pa_conv2d", C.c_int, [_P, _P, _P, _P, _P] + [C.c_int32] * 14 + [_P]),
    ("pa_wino_conv3x3_splitk", C.c_int, [_P, _P, _P, _P, _P] + [C.c_int32] * 11 + [_P, C.c_size_t, _P, C.c_int32, _P]),
    ("pa_crop_resize_width", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, _P, C.c_int32,
                                       C.POINTER(C.c_int32), _P]),
    ("pa_encoder_blob_bytes", C.c_size_t, [C.c_int32] * 7),
    ("pa_encoder_create", C.c_int, [C.c_int32] * 10 + [_P, C.c_size_t, C.POINTER(_P)]),
    ("pa_encoder_destroy", None, [_P]),
    ("pa_encoder_last_error", C.c_char_p, [_P]),
    ("pa_encoder_forward", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    ("pa_mjpeg_create", C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_size_t, C.POINTER(_P)]),
    ("pa_mjpeg_destroy", None, [_P]),
    ("pa_mjpeg_last_error", C.c_char_p, [_P]),
    ("pa_mjpeg_probe", C.c_int, [_P, C.c_size_t, C.POINTER(C.c_int32), C.c_char_p, C.c_size_t]),
    ("pa_mjpeg_set_sync_rounds", C.c_int, [_P, C.c_int32]),
    ("pa_mjpeg_set_groups", C.c_int, [_P, C.c_int32]),
    ("pa_mjpeg_last_sync_rounds", C.c_int, [_P]),
    ("pa_mjpeg_debug_counters", C.c_int, [_P]),
    ("pa_mjpeg_decode", C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
]

_lib = None


def load() -> C.CDLL:
    """Load the shared library and bind every exported symbol (fails loudly)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64.so.7; import it FIRST so that this
    # library's NEEDED libamdhip64.so.7 resolves to that same runtime instance.
    # Two HIP/HSA runtimes in one process cannot both open the device, and
    # streams / device pointers are only meaningful inside one of them.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -m playaid_core_amd._build` "
            "(hipcc, --offload-arch=gfx950). There is no CPU fallback for this path."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the box
        raise HipLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, restype, argtypes in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.pa_abi_version() != PA_ABI_VERSION:
        raise HipLibraryError("libplayaid_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib
