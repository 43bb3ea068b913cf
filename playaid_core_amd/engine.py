"""Python handle on the HIP engine: device memory and streams come from
PyTorch-ROCm, every computation goes through the C ABI
(include/playaid_hip.h). There is no eager / CPU fallback in this module.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Mapping, Optional, Tuple

import numpy as np
import torch

from . import _lib, constants
from .weights import infer_geometry, pack_state_dict


class EngineError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"playaid_hip error {code}: {msg}")
        self.code = code


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class WeightsArena:
    """The folded, re-laid-out weights of an engine as one device byte tensor
    (``pa_weights_export``): what rank 0 broadcasts over RCCL so that the other ranks build
    their engines with ``pa_create_from_arena`` -- no host copy, no second BatchNorm fold."""

    def __init__(self, data: torch.Tensor, sequence_length: int, num_actions: int, compute_dtype: str):
        self.data = data
        self.sequence_length = sequence_length
        self.num_actions = num_actions
        self.compute_dtype = compute_dtype


class Engine:
    """One engine per device (``pa_create`` / ``pa_destroy``)."""

    def __init__(
        self,
        state_dict: Mapping,
        device: str = "cuda:0",
        num_fighters: int = 2,
        frame_delta: int = constants.FRAME_DELTA,
        crop_padding: int = constants.CROP_PADDING,
        max_batch_frames: int = 64,
        max_clip_frames: int = 8192,
        max_frame_height: int = 1080,
        max_frame_width: int = 1920,
        fighter_class_ids: Tuple[int, ...] = (2, 3),
        compute_dtype: str = "f32",
    ):
        """``compute_dtype="bf16"`` selects the bf16 conv path (BASELINE.json configs[2]): 3x3
        convolutions on bf16 activations / weights with fp32 accumulation, everything else fp32.
        It is outside the 1e-4 parity bar of the default fp32 path."""
        if compute_dtype not in _lib.DTYPES:
            raise ValueError("compute_dtype must be 'f32', 'bf16' or 'emulated_f32'")
        self.compute_dtype = compute_dtype
        self._lib = _lib.load()  # raises HipLibraryError when the .so is missing
        self._weights = state_dict
        self._ctor_kwargs = dict(
            device=device, num_fighters=num_fighters, frame_delta=frame_delta, crop_padding=crop_padding,
            max_batch_frames=max_batch_frames, max_clip_frames=max_clip_frames, max_frame_height=max_frame_height,
            max_frame_width=max_frame_width, fighter_class_ids=tuple(fighter_class_ids),
            compute_dtype=compute_dtype,
        )
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("no HIP device visible to PyTorch-ROCm; this path has no CPU fallback")
        self.device = torch.device(device)
        arena = state_dict if isinstance(state_dict, WeightsArena) else None
        if arena is not None:  # prepared weights received from another rank (pa_create_from_arena)
            self.S, self.A = arena.sequence_length, arena.num_actions
            if arena.compute_dtype != compute_dtype:
                raise ValueError(f"weight arena was prepared for {arena.compute_dtype}, engine asked for {compute_dtype}")
        elif isinstance(state_dict, np.ndarray):  # already-packed blob
            hdr = state_dict[:32].view(np.int32)
            self.S, self.A = int(hdr[2]), int(hdr[3])
        else:
            self.S, self.A = infer_geometry(state_dict)
        self.F = num_fighters
        self.max_batch_frames = max_batch_frames
        self.max_clip_frames = max_clip_frames
        cfg = _lib.pa_config()
        cfg.abi_version = _lib.PA_ABI_VERSION
        cfg.device_id = self.device.index or 0
        cfg.sequence_length = self.S
        cfg.frame_delta = frame_delta
        cfg.num_actions = self.A
        cfg.num_fighters = num_fighters
        cfg.crop_padding = crop_padding
        cfg.max_batch_frames = max_batch_frames
        cfg.max_clip_frames = max_clip_frames
        cfg.max_frame_height = max_frame_height
        cfg.max_frame_width = max_frame_width
        ids = list(fighter_class_ids) + [0] * (4 - len(fighter_class_ids))
        for i in range(4):
            cfg.fighter_class_ids[i] = ids[i]
        cfg.compute_dtype = _lib.DTYPES[compute_dtype]
        self.cfg = cfg
        self._h = C.c_void_p(0)
        torch.cuda.set_device(self.device)
        if arena is not None:
            data = arena.data.to(self.device).contiguous()
            torch.cuda.synchronize(self.device)  # the arena is complete before the library copies it
            rc = self._lib.pa_create_from_arena(C.byref(cfg), _ptr(data), data.numel(), C.byref(self._h))
        else:
            if isinstance(state_dict, np.ndarray):
                blob = np.ascontiguousarray(state_dict, dtype=np.uint8)
            else:
                blob = pack_state_dict(state_dict, self.S, self.A)
            rc = self._lib.pa_create(C.byref(cfg), blob.ctypes.data_as(C.c_void_p), blob.nbytes, C.byref(self._h))
        if rc != _lib.PA_OK:
            msg = self._lib.pa_last_error(self._h).decode() if self._h else self._lib.pa_status_string(rc).decode()
            if self._h:
                self._lib.pa_destroy(self._h)
                self._h = C.c_void_p(0)
            raise EngineError(rc, msg)

    def clone(self, **overrides) -> "Engine":
        """A second engine on the same device with the same (already folded) weights: its own activation buffers,
        feature cache and scratch. Built from this engine's weight arena (``pa_create_from_arena``): no second fold."""
        kw = dict(self._ctor_kwargs)
        kw.update(overrides)
        return Engine(self.weights_arena(), **kw)

    def reconfigured(self, **overrides) -> "Engine":
        """A new engine on the same device and weights with some geometry changed."""
        kw = dict(self._ctor_kwargs)
        kw.update(overrides)
        return Engine(self._weights, **kw)

    # -- plumbing ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.pa_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_crop_jpeg_quality(self, quality: int):
        """``quality`` 1..100: every crop the engine cuts also goes through a baseline-JPEG write + read at that quality
        (the reference writes every crop with ``cv2.imwrite`` and reads it back, ``ai_runner.py:420,446``; OpenCV's
        default is 95); 0 switches it off (default: crops are the exact resampler output)."""
        self._check(self._lib.pa_set_crop_jpeg_quality(self._h, int(quality)))

    def stream_spin(self, microseconds: int, stream: "torch.cuda.Stream"):
        """Keep ``stream`` busy for that long with one spinning thread (``pa_stream_spin``): the concurrency probe of
        ``parallel.ClipLanes``."""
        self._check(self._lib.pa_stream_spin(self._h, int(microseconds), C.c_void_p(stream.cuda_stream)))

    def stream_gate(self, max_microseconds: int, stream: "torch.cuda.Stream"):
        """Keep ``stream`` busy until ``stream_gate_open()`` (or ``max_microseconds``, the safety bound): ``pa_stream_gate``."""
        self._check(self._lib.pa_stream_gate(self._h, int(max_microseconds), C.c_void_p(stream.cuda_stream)))

    def stream_gate_open(self):
        self._check(self._lib.pa_stream_gate_open(self._h))

    def _check(self, rc: int):
        if rc != _lib.PA_OK:
            raise EngineError(rc, self._lib.pa_last_error(self._h).decode())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, dtype) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype)

    def weights_arena(self) -> WeightsArena:
        n = int(self._lib.pa_weights_arena_bytes(self._h))
        out = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._check(self._lib.pa_weights_export(self._h, _ptr(out), n, self._stream()))
        return WeightsArena(out, self.S, self.A, self.compute_dtype)

    def check_device_errors(self):
        """Synchronises the current stream; raises EngineError when ``backbone_frames_indexed`` was
        handed a frame id outside the clip since the last check (the kernel skipped it)."""
        bad = C.c_int32(0)
        self._check(self._lib.pa_device_errors(self._h, C.byref(bad), self._stream()))

    # -- b1: operator ------------------------------------------------------
    def infer_windows(self, x: torch.Tensor) -> torch.Tensor:
        """x float32[B,S,3,128,128] -> logp float32[B,A] (on the engine's device)."""
        if x.dim() != 5 or x.shape[1] != self.S or tuple(x.shape[2:]) != (3, 128, 128):
            raise ValueError(f"expected [B,{self.S},3,128,128], got {tuple(x.shape)}")
        xd = self._dev(x, torch.float32)
        out = torch.empty((xd.shape[0], self.A), dtype=torch.float32, device=self.device)
        self._check(self._lib.pa_infer_windows(self._h, _ptr(xd), xd.shape[0], _ptr(out), self._stream()))
        return out

    # -- f4: damage HUD crops -------------------------------------------------
    def crop_resize_width(self, frames, rects, out_w: int = 256):
        """``YoloCrop.crop_img`` + ``imutils.resize(width=out_w)`` (``ai_runner.py:114,556-571``) for up to four pixel
        rectangles ``(x1, y1, x2, y2)`` per frame: frames uint8[n,H,W,3] -> list of uint8[n, oh_j, out_w, 3] on the
        host, one array per rectangle, channel order kept."""
        fd = self._dev(frames, torch.uint8)
        n, h, w, _ = fd.shape
        rects = [tuple(int(v) for v in r) for r in rects]
        if not 1 <= len(rects) <= 4:
            raise ValueError("1..4 rectangles per call")
        cap = 1
        for (x1, y1, x2, y2) in rects:
            if x2 > x1 and y2 > y1:
                cap = max(cap, int((y2 - y1) * (out_w / float(x2 - x1))))
        out = torch.zeros((n, len(rects), cap, out_w, 3), dtype=torch.uint8, device=self.device)
        flat = (C.c_int32 * (4 * len(rects)))(*[v for r in rects for v in r])
        oh = (C.c_int32 * len(rects))()
        self._check(self._lib.pa_crop_resize_width(self._h, _ptr(fd), n, h, w, flat, len(rects), out_w, _ptr(out), cap, oh, self._stream()))
        torch.cuda.synchronize(self.device)
        host = out.cpu().numpy()
        return [np.ascontiguousarray(host[:, j, : oh[j]]) for j in range(len(rects))]

    # -- a6: crops -----------------------------------------------------------
    def square_crops(self, frames, boxes, padding: int = constants.CROP_PADDING, swap_rb: bool = False):
        """frames uint8[n,H,W,3], boxes float64[n,F',4] -> (crops uint8[n,F',128,128,3], status int32[n,F']) on host."""
        fd = self._dev(frames, torch.uint8)
        bd = self._dev(boxes, torch.float64)
        n, h, w, _ = fd.shape
        nf = bd.shape[1]
        if nf != self.F:
            # run fighter slots through F-wide calls
            pad = torch.zeros((n, self.F, 4), dtype=torch.float64, device=self.device)
            pad[:, :nf] = bd[:, : self.F]
            bd_use = pad
        else:
            bd_use = bd
        crops = torch.empty((n, self.F, 128, 128, 3), dtype=torch.uint8, device=self.device)
        status = torch.empty((n, self.F), dtype=torch.int32, device=self.device)
        self._check(
            self._lib.pa_square_crops(
                self._h, _ptr(fd), n, h, w, _ptr(bd_use), padding, int(swap_rb), _ptr(crops), _ptr(status), self._stream()
            )
        )
        torch.cuda.synchronize(self.device)
        return crops[:, :nf].cpu().numpy(), status[:, :nf].cpu().numpy()

    def square_crops_device(self, frames_dev: torch.Tensor, boxes_dev: torch.Tensor, out: torch.Tensor,
                            padding: int = constants.CROP_PADDING, swap_rb: bool = False) -> torch.Tensor:
        """One ``square_crop`` per frame, everything on the device and nothing waited for: frames uint8[k,H,W,3], boxes
        float64[k,4] -> ``out`` uint8[k,128,128,3] (written in place); returns the status int32[k] (device)."""
        if frames_dev.dim() != 4 or frames_dev.shape[3] != 3 or frames_dev.dtype != torch.uint8 or not frames_dev.is_cuda \
                or not frames_dev.is_contiguous():
            raise ValueError("square_crops_device: frames are a contiguous uint8[k, H, W, 3] device tensor")
        k, h, w, _ = frames_dev.shape
        if tuple(boxes_dev.shape) != (k, 4) or boxes_dev.dtype != torch.float64 or not boxes_dev.is_cuda:
            raise ValueError("square_crops_device: boxes are float64[k, 4] on the device")
        if tuple(out.shape) != (k, 128, 128, 3) or out.dtype != torch.uint8 or not out.is_cuda or not out.is_contiguous():
            raise ValueError("square_crops_device: out is a contiguous uint8[k, 128, 128, 3] device tensor")
        status = torch.empty((k,), dtype=torch.int32, device=self.device)
        step = self.max_batch_frames
        for i0 in range(0, k, step):
            cnt = min(step, k - i0)
            bd = boxes_dev[i0:i0 + cnt, None, :].expand(-1, self.F, -1).contiguous()
            crops = torch.empty((cnt, self.F, 128, 128, 3), dtype=torch.uint8, device=self.device)
            st = torch.empty((cnt, self.F), dtype=torch.int32, device=self.device)
            self._check(self._lib.pa_square_crops(self._h, _ptr(frames_dev[i0:]), cnt, h, w, _ptr(bd), padding, int(swap_rb), _ptr(crops),
                                                  _ptr(st), self._stream()))
            out[i0:i0 + cnt].copy_(crops[:, 0])
            status[i0:i0 + cnt].copy_(st[:, 0])
        return status

    # -- a5: crop images -> runner inputs ------------------------------------
    def _pack_crop_images(self, images):
        """list of uint8[h_i, w_i, 3] -> (device byte tensor, device descriptor tensor int64[n, 2])."""
        flat, desc, off = [], np.zeros((len(images), 2), dtype=np.int64), 0
        for i, im in enumerate(images):
            im = np.ascontiguousarray(im, dtype=np.uint8)
            if im.ndim != 3 or im.shape[2] != 3:
                raise ValueError("crop images are uint8[h, w, 3]")
            desc[i, 0] = off
            desc[i, 1] = (im.shape[1] << 32) | im.shape[0]  # int32 height, int32 width (little endian struct)
            flat.append(im.reshape(-1))
            off += (im.size + 15) & ~15
            flat.append(np.zeros(((im.size + 15) & ~15) - im.size, np.uint8))
        buf = torch.from_numpy(np.concatenate(flat) if flat else np.zeros(16, np.uint8)).to(self.device)
        return buf, torch.from_numpy(desc).to(self.device)

    def runner_inputs(self, images, swap_rb: bool = True):
        """``AIRunner.get_action_recognition_input_for_frame``'s per-frame preprocessing
        (``ai_runner.py:446-459``) for a list of BGR crop images of any size ->
        (uint8[n,128,128,3] RGB, status int32[n]) on the host."""
        out_c, out_s = [], []
        step = self.max_batch_frames * self.F
        for i0 in range(0, len(images), step):
            chunk = images[i0 : i0 + step]
            buf, desc = self._pack_crop_images(chunk)
            out = torch.empty((len(chunk), 128, 128, 3), dtype=torch.uint8, device=self.device)
            st = torch.empty((len(chunk),), dtype=torch.int32, device=self.device)
            self._check(self._lib.pa_runner_inputs(self._h, _ptr(buf), buf.numel(), _ptr(desc), len(chunk), int(swap_rb),
                                                   _ptr(out), _ptr(st), self._stream()))
            torch.cuda.synchronize(self.device)
            out_c.append(out.cpu().numpy())
            out_s.append(st.cpu().numpy())
        return np.concatenate(out_c), np.concatenate(out_s)

    def infer_clip_from_crop_images(self, images, want_crops: bool = False):
        """A clip given as crop images, ``images[frame][fighter]`` uint8[h, w, 3] BGR (the reference's on-disk
        hand-off between YOLOv5 and the runner): same result dict as ``infer_clip``."""
        n = len(images)
        records = self.alloc_records(n - 1)
        logp = torch.empty((n - 1, self.F, self.A), dtype=torch.float32, device=self.device)
        status = torch.empty((n, self.F), dtype=torch.int32, device=self.device)
        crops = torch.empty((n, self.F, 128, 128, 3), dtype=torch.uint8, device=self.device) if want_crops else None
        self.clip_begin(n)
        step = self.max_batch_frames
        for f0 in range(0, n, step):
            chunk = [im for fr in images[f0 : f0 + step] for im in fr]
            cnt = len(chunk) // self.F
            buf, desc = self._pack_crop_images(chunk)
            self._check(self._lib.pa_backbone_crop_images(self._h, _ptr(buf), buf.numel(), _ptr(desc), cnt, f0,
                                                          _ptr(crops[f0 : f0 + cnt]) if want_crops else C.c_void_p(0),
                                                          _ptr(status[f0 : f0 + cnt]), self._stream()))
            torch.cuda.synchronize(self.device)  # buf / desc are freed when this iteration ends
        self.head_frames(1, n, records, logp)
        torch.cuda.synchronize(self.device)
        out = self.decode_records(records)
        out["logp"] = logp.cpu().numpy()
        out["crop_status"] = status.cpu().numpy()
        if want_crops:
            out["crops_rgb"] = crops.cpu().numpy()
        return out

    def infer_clip_from_packed_crop_images(self, images: torch.Tensor, desc: torch.Tensor, n: int, want_crops: bool = False,
                                           device_results: bool = False):
        """``infer_clip_from_crop_images`` for crop images that are already on the device (``save_one_box_crops``):
        packed bytes + descriptors int64[n * F, 2] in (frame, fighter) order. ``device_results``: only enqueue, and return
        the device tensors (records, logp, crop_status, crops_rgb) without waiting for them."""
        records = self.alloc_records(n - 1)
        logp = torch.empty((n - 1, self.F, self.A), dtype=torch.float32, device=self.device)
        status = torch.empty((n, self.F), dtype=torch.int32, device=self.device)
        crops = torch.empty((n, self.F, 128, 128, 3), dtype=torch.uint8, device=self.device) if want_crops else None
        self.clip_begin(n)
        step = self.max_batch_frames
        for f0 in range(0, n, step):
            cnt = min(step, n - f0)
            self._check(self._lib.pa_backbone_crop_images(self._h, _ptr(images), images.numel(), _ptr(desc[f0 * self.F:]), cnt, f0,
                                                          _ptr(crops[f0 : f0 + cnt]) if want_crops else C.c_void_p(0),
                                                          _ptr(status[f0 : f0 + cnt]), self._stream()))
        self.head_frames(1, n, records, logp)
        if device_results:
            return {"records": records, "logp": logp, "crop_status": status, "crops_rgb": crops}
        torch.cuda.synchronize(self.device)
        out = self.decode_records(records)
        out["logp"] = logp.cpu().numpy()
        out["crop_status"] = status.cpu().numpy()
        if want_crops:
            out["crops_rgb"] = crops.cpu().numpy()
        return out

    # -- f1: detection post-processing ------------------------------------------
    def detect_postprocess(self, pred, net_hw, img_hw, conf_thres: float = 0.25, iou_thres: float = 0.45,
                           classes=(2, 3), max_det: int = 2):
        """pred float32[n, rows, 5 + nc] (decoded head rows) -> (dets float32[n, max_det, 6], counts int32[n]) on
        the device; dets rows are ``cls cx cy w h conf`` in label-file order."""
        pd = self._dev(pred, torch.float32)
        n, rows, width = pd.shape
        mask = 0
        for c in classes:
            mask |= 1 << int(c)
        dets = torch.empty((n, max_det, 6), dtype=torch.float32, device=self.device)
        counts = torch.empty((n,), dtype=torch.int32, device=self.device)
        self._check(self._lib.pa_detect_postprocess(self._h, _ptr(pd), n, rows, width - 5, conf_thres, iou_thres, mask, max_det,
                                                    int(net_hw[0]), int(net_hw[1]), int(img_hw[0]), int(img_hw[1]), _ptr(dets),
                                                    _ptr(counts), self._stream()))
        return dets, counts

    def clean_detections(self, dets: torch.Tensor, counts: torch.Tensor, n_decoded_frames: int):
        """``clean_yolo_crops`` (``ai_runner.py:226-424``) on the detection table, on the device (``pa_clean_detections``)
        -> dict of device tensors: labels float64[n,F,6], pixel_frame int32[n,F], pixel_box float64[n,F,4], crop_kind
        int32[n,F], crop_row float32[n,F,6], info int32[4] (max_frames, error code, error frame, duplicates resolved)."""
        n, md = dets.shape[0], dets.shape[1]
        dev = self.device
        out = {
            "labels": torch.empty((n, self.F, 6), dtype=torch.float64, device=dev),
            "pixel_frame": torch.empty((n, self.F), dtype=torch.int32, device=dev),
            "pixel_box": torch.empty((n, self.F, 4), dtype=torch.float64, device=dev),
            "crop_kind": torch.empty((n, self.F), dtype=torch.int32, device=dev),
            "crop_row": torch.empty((n, self.F, 6), dtype=torch.float32, device=dev),
            "info": torch.zeros(4, dtype=torch.int32, device=dev),
        }
        self._check(self._lib.pa_clean_detections(self._h, _ptr(dets), _ptr(counts), n, md, int(n_decoded_frames), _ptr(out["labels"]),
                                                  _ptr(out["pixel_frame"]), _ptr(out["pixel_box"]), _ptr(out["crop_kind"]),
                                                  _ptr(out["crop_row"]), _ptr(out["info"]), self._stream()))
        return out

    def detector_plan(self, tab, n_rows: int):
        """The crop hand-off's inputs from ``clean_detections``' tables, one launch (``pa_detector_plan``) -> dict of device
        tensors: det_index / src_own int32[n_rows, F] (``save_one_box_crops``' arguments), rep_entry / rep_src int32[n_rows * F],
        rep_boxes float64[n_rows * F, 4] (the square-crop repairs in entry order, padded to a whole number of F) and words
        int32[5] (``info`` + the number of repairs)."""
        dev, F = self.device, self.F
        out = {
            "det_index": torch.empty((n_rows, F), dtype=torch.int32, device=dev),
            "src_own": torch.empty((n_rows, F), dtype=torch.int32, device=dev),
            "rep_entry": torch.empty((n_rows * F,), dtype=torch.int32, device=dev),
            "rep_boxes": torch.empty((n_rows * F, 4), dtype=torch.float64, device=dev),
            "rep_src": torch.empty((n_rows * F,), dtype=torch.int32, device=dev),
            "words": torch.empty((5,), dtype=torch.int32, device=dev),
        }
        self._check(self._lib.pa_detector_plan(self._h, _ptr(tab["pixel_frame"]), _ptr(tab["pixel_box"]), _ptr(tab["crop_kind"]), _ptr(tab["info"]),
                                               n_rows, _ptr(out["det_index"]), _ptr(out["src_own"]), _ptr(out["rep_entry"]), _ptr(out["rep_boxes"]),
                                               _ptr(out["rep_src"]), _ptr(out["words"]), self._stream()))
        return out

    def detector_plan_desc(self, desc: torch.Tensor, crop_kind: torch.Tensor, n_frames: int, step_frames: int, region_bytes: int,
                           rep_entry: torch.Tensor, n_rep: int, rep_base: int):
        """Descriptors of a clip's crop images after the per-chunk packing (``pa_detector_plan_desc``), in place."""
        if desc.dtype != torch.int64 or not desc.is_cuda or not desc.is_contiguous() or desc.numel() < n_frames * self.F * 2:
            raise ValueError(f"detector_plan_desc: desc is a contiguous int64[{n_frames * self.F}, 2] device tensor")
        if crop_kind.dtype != torch.int32 or crop_kind.numel() < n_frames * self.F or rep_entry.numel() < n_rep:
            raise ValueError("detector_plan_desc: tables shorter than the clip")
        self._check(self._lib.pa_detector_plan_desc(self._h, _ptr(desc), _ptr(crop_kind), n_frames, step_frames, int(region_bytes), _ptr(rep_entry),
                                                    n_rep, int(rep_base), self._stream()))

    def square_crops_src_device(self, frames_dev: torch.Tensor, boxes_dev: torch.Tensor, src_dev: torch.Tensor, k: int, out: torch.Tensor,
                                padding: int = constants.CROP_PADDING, swap_rb: bool = False) -> torch.Tensor:
        """``k`` square crops (k a multiple of F), crop i cut from ``frames_dev[src_dev[i]]`` with ``boxes_dev[i]`` (float64[k, 4], int32[k],
        device) -> ``out`` uint8[k, 128, 128, 3] written in place; returns the status int32[k] (device). Nothing is waited for."""
        if frames_dev.dim() != 4 or frames_dev.shape[3] != 3 or frames_dev.dtype != torch.uint8 or not frames_dev.is_cuda \
                or not frames_dev.is_contiguous():
            raise ValueError("square_crops_src_device: frames are a contiguous uint8[n, H, W, 3] device tensor")
        n_src, h, w, _ = frames_dev.shape
        F = self.F
        if k < F or k % F or boxes_dev.dtype != torch.float64 or boxes_dev.numel() < k * 4 or src_dev.dtype != torch.int32 or src_dev.numel() < k:
            raise ValueError("square_crops_src_device: k is a multiple of the fighter count; boxes float64[k, 4], src int32[k]")
        if out.dtype != torch.uint8 or not out.is_cuda or not out.is_contiguous() or out.numel() < k * 128 * 128 * 3:
            raise ValueError("square_crops_src_device: out is a contiguous uint8[k, 128, 128, 3] device tensor")
        status = torch.empty((k,), dtype=torch.int32, device=self.device)
        step = self.max_batch_frames * F
        for i0 in range(0, k, step):
            cnt = min(step, k - i0)
            self._check(self._lib.pa_square_crops_src(self._h, _ptr(frames_dev), n_src, h, w, _ptr(boxes_dev[i0:]), _ptr(src_dev[i0:]), cnt // F,
                                                      padding, int(swap_rb), _ptr(out.view(-1)[i0 * 49152:]), _ptr(status[i0:]), self._stream()))
        return status

    def save_one_box_crops(self, frames_dev: torch.Tensor, dets: torch.Tensor, counts: torch.Tensor, det_index=None,
                           jpeg_quality: int = 95, images: torch.Tensor = None, desc: torch.Tensor = None, src_frame=None):
        """``detect.py --save-crop`` + ``cv2.imread`` of every crop (``ai_runner.py:208,445-446``) on the device: frames
        uint8[n,H,W,3] + the detections of ``detect_postprocess`` -> (packed BGR crop images uint8[...] device, descriptors
        int64[n * F, 2] device = what ``pa_runner_inputs`` / ``pa_backbone_crop_images`` take). ``det_index`` int32[n, F]:
        which detection each fighter's crop comes from (-1 none); None = the first of the fighter's class in label order.
        ``src_frame`` int32[n, F]: the frame each crop's pixels are cut from when that is not its own (``n = dets.shape[0]``)."""
        n_src, h, w, _ = frames_dev.shape
        n = dets.shape[0]
        sf = self._dev(src_frame, torch.int32) if src_frame is not None else None
        if images is None:
            images = torch.empty(n * self.F * min(h * w, 1 << 20) * 3 + 64, dtype=torch.uint8, device=self.device)
        elif images.dtype != torch.uint8 or not images.is_cuda or not images.is_contiguous():
            raise ValueError("save_one_box_crops: images is a contiguous uint8 device tensor (its size is the capacity)")
        if desc is None:
            desc = torch.zeros((n * self.F, 2), dtype=torch.int64, device=self.device)
        elif desc.dtype != torch.int64 or not desc.is_cuda or not desc.is_contiguous() or desc.numel() < n * self.F * 2:
            raise ValueError(f"save_one_box_crops: desc is a contiguous int64[{n * self.F}, 2] device tensor")
        if dets.dtype != torch.float32 or counts.dtype != torch.int32 or counts.numel() < n or not dets.is_contiguous():
            raise ValueError("save_one_box_crops: dets float32[n, max_det, 6] / counts int32[n] as detect_postprocess writes them")
        di = self._dev(det_index, torch.int32) if det_index is not None else None
        for name, t in (("det_index", di), ("src_frame", sf)):
            if t is not None and t.numel() < n * self.F:
                raise ValueError(f"save_one_box_crops: {name} is int32[{n}, {self.F}]")
        self._check(self._lib.pa_save_one_box_crops(self._h, _ptr(frames_dev), n_src, h, w, _ptr(dets), _ptr(counts), dets.shape[1],
                                                    _ptr(di), _ptr(sf), n, int(jpeg_quality), _ptr(images), images.numel(), _ptr(desc),
                                                    self._stream()))
        return images, desc

    @staticmethod
    def unpack_crop_images(images: torch.Tensor, desc: torch.Tensor):
        """(packed images, descriptors) -> list of uint8[h, w, 3] numpy arrays (None where height == 0), host side."""
        d = desc.cpu().numpy()
        buf = images.cpu().numpy()
        out = []
        for off, hw in d:
            hh, ww = int(hw) & 0xFFFFFFFF, int(hw) >> 32
            out.append(buf[off: off + hh * ww * 3].reshape(hh, ww, 3).copy() if hh and ww else None)
        return out

    # -- boxes from the game log ----------------------------------------------
    def project_boxes(self, log_rows) -> torch.Tensor:
        """log_rows float64[..., 9] (pos_x,pos_y,cam xyz,target xyz,fov deg) -> boxes float64[..., 4] on device."""
        ld = self._dev(log_rows, torch.float64)
        out = torch.empty(ld.shape[:-1] + (4,), dtype=torch.float64, device=self.device)
        self._check(self._lib.pa_project_boxes(self._h, _ptr(ld), ld.numel() // 9, _ptr(out), self._stream()))
        return out

    # -- b2: clip ------------------------------------------------------------
    def clip_begin(self, clip_frames: int, batch_of: int = 0):
        """``batch_of`` = n > 0: the ``clip_frames`` frames are n independent clips of ``clip_frames // n`` frames
        each (``pa_clip_begin_batch``): windows never cross from one clip into the next."""
        if batch_of:
            if clip_frames % batch_of:
                raise ValueError("the batch must hold whole clips")
            self._check(self._lib.pa_clip_begin_batch(self._h, batch_of, clip_frames // batch_of))
        else:
            self._check(self._lib.pa_clip_begin(self._h, clip_frames))

    def backbone_frames(self, frames_dev: torch.Tensor, boxes_dev: torch.Tensor, frame0: int, crops_rgb=None, status=None):
        n, h, w, _ = frames_dev.shape
        self._check(
            self._lib.pa_backbone_frames(
                self._h, _ptr(frames_dev), n, h, w, _ptr(boxes_dev), frame0, _ptr(crops_rgb), _ptr(status), self._stream()
            )
        )

    def backbone_frames_src(self, frames_dev: torch.Tensor, boxes_dev: torch.Tensor, src_dev: torch.Tensor, frame0: int,
                            crops_rgb=None, status=None):
        """Crop (i, p) of clip frame frame0 + i is cut from frames_dev[src_dev[i, p]] (repaired label gaps)."""
        n_src, h, w, _ = frames_dev.shape
        n = boxes_dev.shape[0]
        self._check(
            self._lib.pa_backbone_frames_src(
                self._h, _ptr(frames_dev), n_src, h, w, _ptr(boxes_dev), _ptr(src_dev), n, frame0, _ptr(crops_rgb),
                _ptr(status), self._stream()
            )
        )

    def preprocess_frames(self, frames_dev: torch.Tensor, boxes_dev: torch.Tensor, slot: int, crops_rgb=None, status=None):
        """Crop stage only, into model-input buffer `slot` (0/1), on the current stream."""
        n, h, w, _ = frames_dev.shape
        self._check(
            self._lib.pa_preprocess_frames(
                self._h, _ptr(frames_dev), n, h, w, _ptr(boxes_dev), slot, _ptr(crops_rgb), _ptr(status), self._stream()
            )
        )

    def backbone_slot(self, slot: int, n: int, frame0: int):
        """Backbone over the crops in buffer `slot` -> feature cache rows of frames frame0.., current stream."""
        self._check(self._lib.pa_backbone_slot(self._h, slot, n, frame0, self._stream()))

    supports_pipelining = True

    # -- ingest from host memory: only the crops' slices cross PCIe ------------------
    def make_window_stage(self, n_frames: int, bytes_per_crop: int = 1 << 20):
        """Reusable staging for ``upload_crop_windows``: a device window buffer, a pinned host and a device
        descriptor array for ``n_frames`` frames."""
        nc = n_frames * self.F
        return {
            "windows": torch.empty(nc * bytes_per_crop + 16, dtype=torch.uint8, device=self.device),
            "desc_host": torch.zeros((nc, 4), dtype=torch.int64).pin_memory(),  # pa_crop_window is 32 bytes
            "desc_dev": torch.zeros((nc, 4), dtype=torch.int64, device=self.device),
            "used": 0,
        }

    def upload_crop_windows(self, frames_host: torch.Tensor, boxes_host, stage, padding: int = None):
        """frames_host: uint8[n,H,W,3] CPU tensor in PINNED memory (the upload kernel reads it from the device);
        boxes_host: float64[n,F,4] numpy. Enqueues the descriptor copy and the slice-upload kernel on the current stream."""
        if not frames_host.is_pinned():
            raise ValueError("frames_host must be pinned host memory (tensor.pin_memory())")
        n, h, w, _ = frames_host.shape
        bx = np.ascontiguousarray(boxes_host, dtype=np.float64)
        used = C.c_size_t(0)
        self._check(self._lib.pa_upload_crop_windows(
            self._h, C.c_void_p(frames_host.data_ptr()), n, h, w, bx.ctypes.data_as(C.c_void_p),
            self.cfg.crop_padding if padding is None else padding, _ptr(stage["windows"]), stage["windows"].numel(),
            C.c_void_p(stage["desc_host"].data_ptr()), _ptr(stage["desc_dev"]), C.byref(used), self._stream()))
        stage["used"] = int(used.value)
        return stage

    def preprocess_windows(self, stage, n: int, height: int, width: int, boxes_dev: torch.Tensor, slot: int, crops_rgb=None, status=None):
        self._check(self._lib.pa_preprocess_windows(self._h, _ptr(stage["windows"]), _ptr(stage["desc_dev"]), n, height, width,
                                                    _ptr(boxes_dev), slot, _ptr(crops_rgb), _ptr(status), self._stream()))

    def backbone_frames_indexed(self, frames_dev, boxes_dev, ids_dev, crops_rgb=None, status=None):
        """Crop + backbone for frames whose clip positions are given by ids_dev (int32, device)."""
        n, h, w, _ = frames_dev.shape
        self._check(
            self._lib.pa_backbone_frames_indexed(
                self._h, _ptr(frames_dev), n, h, w, _ptr(boxes_dev), _ptr(ids_dev), _ptr(crops_rgb), _ptr(status), self._stream()
            )
        )

    def clip_mark_ready(self, ids_host):
        ids = np.ascontiguousarray(ids_host, dtype=np.int32)
        self._check(self._lib.pa_clip_mark_ready(self._h, ids.ctypes.data_as(C.c_void_p), len(ids)))

    def head_frames(self, lo: int, hi: int, records: torch.Tensor, logp: Optional[torch.Tensor]):
        self._check(self._lib.pa_head_frames(self._h, lo, hi, _ptr(records), _ptr(logp), self._stream()))

    def infer_clip_device(self, frames_dev, boxes_dev, records, logp=None, crops_rgb=None, status=None):
        """Enqueue a whole clip (all tensors already on the device)."""
        n, h, w, _ = frames_dev.shape
        self._check(
            self._lib.pa_infer_clip(
                self._h, _ptr(frames_dev), n, h, w, _ptr(boxes_dev), _ptr(records), _ptr(logp), _ptr(crops_rgb),
                _ptr(status), self._stream(),
            )
        )

    def alloc_records(self, count: int) -> torch.Tensor:
        return torch.zeros((count, self.F, 4), dtype=torch.int32, device=self.device)

    def alloc_logp(self, count: int) -> torch.Tensor:
        return torch.zeros((count, self.F, self.A), dtype=torch.float32, device=self.device)

    def features_buffer(self, n: int) -> torch.Tensor:
        return torch.empty((n, self.F, _lib.PA_FEATURE_STRIDE), dtype=torch.float32, device=self.device)

    @staticmethod
    def decode_records(records: torch.Tensor) -> Dict[str, np.ndarray]:
        r = records.cpu().numpy()
        return {
            "char_id": r[..., 0].copy(),
            "action_id": r[..., 1].copy(),
            "prob": r[..., 2].copy().view(np.float32),
            "status": r[..., 3].copy(),
        }

    def infer_clip(self, frames, boxes, want_crops: bool = False, src=None):
        """Host convenience: upload, run, download. -> dict with logp[n-1,F,A],
        action_id, prob, char_id, status[n,F] (and crops_rgb[n,F,128,128,3]).
        ``src`` (int[n,F], optional): the frame each crop is cut from when that is not its own
        (``pa_backbone_frames_src``); the clip then has ``boxes.shape[0]`` frames."""
        fd = self._dev(frames, torch.uint8)
        bd = self._dev(boxes, torch.float64)
        n = bd.shape[0] if src is not None else fd.shape[0]
        records = self.alloc_records(n - 1)
        logp = torch.empty((n - 1, self.F, self.A), dtype=torch.float32, device=self.device)
        status = torch.empty((n, self.F), dtype=torch.int32, device=self.device)
        crops = torch.empty((n, self.F, 128, 128, 3), dtype=torch.uint8, device=self.device) if want_crops else None
        if src is None:
            self.infer_clip_device(fd, bd, records, logp, crops, status)
        else:
            sd = self._dev(src, torch.int32)
            self.clip_begin(n)
            step = self.max_batch_frames
            for f0 in range(0, n, step):
                cnt = min(step, n - f0)
                self.backbone_frames_src(fd, bd[f0 : f0 + cnt], sd[f0 : f0 + cnt], f0,
                                         crops[f0 : f0 + cnt] if want_crops else None, status[f0 : f0 + cnt])
            self.head_frames(1, n, records, logp)
        torch.cuda.synchronize(self.device)
        out = self.decode_records(records)
        out["logp"] = logp.cpu().numpy()
        out["crop_status"] = status.cpu().numpy()
        if want_crops:
            out["crops_rgb"] = crops.cpu().numpy()
        return out

    def features_export(self, frame0: int, n: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty((n, self.F, _lib.PA_FEATURE_STRIDE), dtype=torch.float32, device=self.device)
        elif (out.dtype != torch.float32 or out.device.type != self.device.type or not out.is_contiguous()
              or (self.device.index is not None and out.device.index != self.device.index)
              or out.numel() < n * self.F * _lib.PA_FEATURE_STRIDE):
            # (the library writes n * F * PA_FEATURE_STRIDE floats through a raw pointer: a wrong buffer is a wild write)
            raise ValueError(f"features_export: out must be a contiguous float32 tensor on {self.device} with at least "
                             f"{n * self.F * _lib.PA_FEATURE_STRIDE} elements")
        self._check(self._lib.pa_features_export(self._h, frame0, n, _ptr(out), self._stream()))
        return out

    def features_import(self, frame0: int, feats: torch.Tensor):
        feats = feats.to(self.device, torch.float32).contiguous()
        self._check(self._lib.pa_features_import(self._h, frame0, feats.shape[0], _ptr(feats), self._stream()))

    # -- measurement -----------------------------------------------------------
    def profile_enable(self, on: bool):
        self._check(self._lib.pa_profile_enable(self._h, int(on)))

    def profile_read(self) -> List[Dict]:
        arr = (_lib.pa_kernel_stat * 64)()
        n = C.c_int32(0)
        self._check(self._lib.pa_profile_read(self._h, arr, 64, C.byref(n)))
        return [
            {
                "name": arr[i].name.decode(),
                "launches": arr[i].launches,
                "total_ms": arr[i].total_ms,
                "flops": arr[i].flops,
                "bytes": arr[i].bytes,
                "flops_executed": arr[i].flops_executed,
            }
            for i in range(n.value)
        ]


_default: Optional[Engine] = None


def default_engine() -> Engine:
    """Process-wide engine on cuda:0 with seeded synthetic weights, for the
    ``YoloCrop.square_crop`` mirror (which needs no trained weights)."""
    global _default
    if _default is None:
        from . import synth

        _default = Engine(synth.make_state_dict(), max_batch_frames=8, max_clip_frames=64)
    return _default
