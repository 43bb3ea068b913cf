"""Video ingest for the runner: Motion-JPEG decode on the MI355X (rows a1 / f2 of SURVEY.md section 8).

The reference opens its input with ``cv2.VideoCapture(path)`` and pulls single frames with
``set(cv2.CAP_PROP_POS_FRAMES, j)`` + ``read()`` (``playaid/ai_runner.py:153,404-405,558-559``;
``playaid/manuscript.py:70-86,154-155``; ``playaid/timeline.py:283-322``). This module keeps that call shape
(``VideoCapture`` below: same method names, property ids and return conventions) for the containers whose frames are
baseline JPEG files -- ``.avi`` with an ``MJPG`` stream, a raw concatenation of JPEG files (``.mjpeg`` / ``.mjpg``), or a
directory / list of ``.jpg`` files (OpenCV's image-sequence capture) -- and adds the batched form the hot path wants:
``read_frames(j0, n)`` decodes n frames in one ``pa_mjpeg_decode`` call into HBM, where the crop stage reads them.

Pixels are libjpeg-turbo's (``oracle/jpeg.py::decode`` is pinned to it byte for byte), i.e. what OpenCV's JPEG reader
delivers for image sequences and ``imread``. For ``.avi`` OpenCV normally runs FFmpeg's own MJPEG decoder, whose IDCT and
chroma up-sampling differ from libjpeg's in the last bit; neither FFmpeg nor cv2 exists in this image, so that variant is
not restated.

There is no CPU fallback: decoding needs the HIP library and a GPU (``HipLibraryError`` otherwise). Container parsing
(RIFF chunks, SOI / EOI scanning) is host byte bookkeeping, as in the reference's OpenCV.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np

# the cv2 property ids the reference uses (manuscript.py:71-86, timeline.py:290-295, ai_runner.py:404)
CAP_PROP_POS_FRAMES = 1
CAP_PROP_FRAME_WIDTH = 3
CAP_PROP_FRAME_HEIGHT = 4
CAP_PROP_FPS = 5
CAP_PROP_FRAME_COUNT = 7


class VideoError(RuntimeError):
    pass


# ---------------------------------------------------------------------------------------------------------------------
# containers
# ---------------------------------------------------------------------------------------------------------------------

def split_jpeg_stream(data: np.ndarray) -> np.ndarray:
    """Byte spans ``int64[n, 2]`` of the JPEG files in a raw concatenation (``.mjpeg``): frame f is
    ``data[sp[f, 0]:sp[f, 1]]``. A file starts at ``FF D8 FF`` and ends with the first ``FF D9`` after its SOS that is
    followed by the next ``FF D8`` or the end of the data (inside entropy-coded data ``FF`` is always followed by ``00``
    or ``D0..D7``, so ``FF D9`` cannot occur there; APPn thumbnails may hold one, hence the look-ahead)."""
    d = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    ff = np.flatnonzero(d[:-1] == 0xFF)
    nxt = d[ff + 1]
    soi = ff[nxt == 0xD8]
    soi = soi[(soi + 2 < d.size)]
    soi = soi[d[soi + 2] == 0xFF]
    eoi = ff[nxt == 0xD9] + 2
    starts: List[int] = []
    ends: List[int] = []
    pos = 0
    si = 0
    while si < len(soi):
        s = int(soi[si])
        if s < pos:
            si += 1
            continue
        # the EOI that is directly followed by another SOI (or ends the data)
        cand = eoi[eoi > s]
        end = None
        for e in cand:
            e = int(e)
            if e >= d.size or (e + 1 < d.size and d[e] == 0xFF and d[e + 1] == 0xD8):
                end = e
                break
        if end is None:
            end = int(cand[-1]) if len(cand) else d.size
        starts.append(s)
        ends.append(end)
        pos = end
        si += 1
    if not starts:
        raise VideoError("no JPEG files in the stream")
    return np.stack([np.array(starts, dtype=np.int64), np.array(ends, dtype=np.int64)], axis=1)


def read_avi_mjpeg(path: str) -> Tuple[np.ndarray, np.ndarray, dict]:
    """RIFF / AVI with one ``MJPG`` video stream -> (file bytes uint8[...], offsets int64[n, 2] = (start, end) of every
    frame's JPEG file inside those bytes, meta dict(fps, width, height))."""
    data = np.fromfile(path, dtype=np.uint8)
    if data.size < 12 or bytes(data[:4]) != b"RIFF" or bytes(data[8:12]) != b"AVI ":
        raise VideoError(f"{path}: not a RIFF AVI file")
    meta = {"fps": 0.0, "width": 0, "height": 0, "handler": b""}
    frames: List[Tuple[int, int]] = []

    def walk(lo: int, hi: int, in_movi: bool):
        p = lo
        while p + 8 <= hi:
            cid = bytes(data[p:p + 4])
            size = int(struct.unpack_from("<I", data, p + 4)[0])
            body = p + 8
            if cid == b"LIST":
                kind = bytes(data[body:body + 4])
                walk(body + 4, min(body + size, hi), kind == b"movi" or in_movi)
            elif cid == b"avih" and size >= 40:
                us, = struct.unpack_from("<I", data, body)
                w, h = struct.unpack_from("<II", data, body + 32)
                meta["width"], meta["height"] = int(w), int(h)
                if us:
                    meta["fps"] = 1e6 / us
            elif cid == b"strh" and size >= 32 and bytes(data[body:body + 4]) == b"vids":
                meta["handler"] = bytes(data[body + 4:body + 8])
                scale, rate = struct.unpack_from("<II", data, body + 20)
                if scale:
                    meta["fps"] = rate / scale
            elif in_movi and cid[2:4] in (b"dc", b"db"):
                # a zero-length chunk is a dropped frame: the player repeats the previous picture, and the frame numbers
                # (cv2's CAP_PROP_POS_FRAMES, which the label files are keyed on) keep counting
                if size > 0:
                    frames.append((body, body + size))
                elif frames:
                    frames.append(frames[-1])
            p = body + size + (size & 1)

    walk(12, data.size, False)
    if not frames:
        raise VideoError(f"{path}: no video chunks")
    off = np.array(frames, dtype=np.int64)
    first = data[off[0, 0]:off[0, 0] + 3]
    if bytes(first) != b"\xff\xd8\xff":
        raise VideoError(f"{path}: video stream is not Motion-JPEG (handler {meta['handler']!r}); only MJPG is decoded here")
    return data, off, meta


def write_avi_mjpeg(path: str, jpeg_frames: Sequence[bytes], fps: float, width: int, height: int) -> None:
    """Minimal RIFF / AVI writer (one ``MJPG`` stream + ``idx1``) for synthetic clips: the container OpenCV's
    ``VideoWriter(fourcc="MJPG")`` produces, without the optional OpenDML super-index."""
    n = len(jpeg_frames)
    movi = bytearray(b"movi")
    idx = bytearray()
    for jf in jpeg_frames:
        idx += struct.pack("<4sIII", b"00dc", 0x10, len(movi), len(jf))
        movi += b"00dc" + struct.pack("<I", len(jf)) + jf + (b"\0" if len(jf) & 1 else b"")
    biggest = max((len(j) for j in jpeg_frames), default=0)
    scale, rate = 1000, int(round(fps * 1000))
    avih = struct.pack("<IIIIIIIIIIIIII", int(round(1e6 / fps)), biggest * int(round(fps)), 0, 0x10, n, 0, 1, biggest, width, height, 0, 0, 0, 0)
    strh = struct.pack("<4s4sIHHIIIIIIIIhhhh", b"vids", b"MJPG", 0, 0, 0, 0, scale, rate, 0, n, biggest, 0xFFFFFFFF, 0, 0, 0, width, height)
    strf = struct.pack("<IiiHH4sIiiII", 40, width, height, 1, 24, b"MJPG", width * height * 3, 0, 0, 0, 0)

    def chunk(cid: bytes, body: bytes) -> bytes:
        return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")

    strl = b"LIST" + struct.pack("<I", 4 + len(chunk(b"strh", strh)) + len(chunk(b"strf", strf))) + b"strl" + chunk(b"strh", strh) + chunk(b"strf", strf)
    hdrl_body = b"hdrl" + chunk(b"avih", avih) + strl
    hdrl = b"LIST" + struct.pack("<I", len(hdrl_body)) + hdrl_body
    movi_l = b"LIST" + struct.pack("<I", len(movi)) + bytes(movi)
    body = b"AVI " + hdrl + movi_l + chunk(b"idx1", bytes(idx))
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def jpeg_frame_size(buf: np.ndarray) -> Tuple[int, int]:
    """(height, width) from the first SOF0 / SOF1 segment of a JPEG file."""
    d = np.ascontiguousarray(buf, dtype=np.uint8)
    p = 2
    while p + 9 < d.size:
        if d[p] != 0xFF:
            break
        m = int(d[p + 1])
        if m == 0xFF:
            p += 1
            continue
        seg = (int(d[p + 2]) << 8) | int(d[p + 3])
        if m in (0xC0, 0xC1, 0xC2):
            return (int(d[p + 5]) << 8) | int(d[p + 6]), (int(d[p + 7]) << 8) | int(d[p + 8])
        if m == 0xDA:
            break
        p += 2 + seg
    raise VideoError("no frame header (SOF) in the JPEG file")


# ---------------------------------------------------------------------------------------------------------------------
# the decoder handle
# ---------------------------------------------------------------------------------------------------------------------

class MjpegDecoder:
    """``pa_mjpeg_create`` / ``pa_mjpeg_decode``: compressed frames in (pinned) host memory -> uint8[n,H,W,3] in HBM."""

    DEFAULT_SYNC_ROUNDS = 8

    def __init__(self, max_frames: int, max_height: int, max_width: int, max_bytes: int, device: str = "cuda:0"):
        import torch

        from . import _lib

        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("no HIP device visible to PyTorch-ROCm; Motion-JPEG decode has no CPU fallback")
        self.device = torch.device(device)
        self.max_frames, self.max_height, self.max_width, self.max_bytes = max_frames, max_height, max_width, int(max_bytes)
        self._h = C.c_void_p(0)
        torch.cuda.set_device(self.device)
        rc = self._lib.pa_mjpeg_create(self.device.index or 0, max_frames, max_height, max_width, int(max_bytes), C.byref(self._h))
        if rc != _lib.PA_OK:
            msg = self._lib.pa_mjpeg_last_error(self._h).decode() if self._h else self._lib.pa_status_string(rc).decode()
            self.close()
            from .engine import EngineError

            raise EngineError(rc, msg)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pa_mjpeg_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_sync_rounds(self, rounds: int):
        """Verify passes of the self-synchronising entropy decoder: 1..16 enqueued blindly (default 8; a pass over a frame
        that has settled returns at once; a frame that has not settled gets status bit 8), 0 = repeat until settled (one
        stream synchronisation per pass)."""
        from . import _lib
        from .engine import EngineError

        rc = self._lib.pa_mjpeg_set_sync_rounds(self._h, int(rounds))
        if rc != _lib.PA_OK:
            raise EngineError(rc, "sync rounds must be 0..16")

    def last_sync_rounds(self) -> int:
        return int(self._lib.pa_mjpeg_last_sync_rounds(self._h))

    def set_groups(self, groups: int):
        """Frame groups per call, 1..4 (default 2), each decoded on a stream of the handle's own and joined to the
        current stream at the end. 1 = everything on the current stream: what a caller that runs several decoders on
        several streams itself (``bench.py``'s decode_inclusive) wants."""
        from . import _lib
        from .engine import EngineError

        rc = self._lib.pa_mjpeg_set_groups(self._h, int(groups))
        if rc != _lib.PA_OK:
            raise EngineError(rc, "groups must be 1..4")

    def decode(self, data, spans, height: int, width: int, out=None, status=None, rgb: bool = False):
        """``data``: the compressed bytes, a uint8 numpy array or CPU torch tensor (pin it for an asynchronous copy);
        ``spans`` int64[n, 2] = (start, end) of every frame's JPEG file in ``data``, or int64[n + 1] offsets of
        back-to-back frames. Enqueues on the current stream and returns the device tensor uint8[n, H, W, 3] (BGR unless
        ``rgb``). ``status`` (optional int32[n] device tensor) receives the per-frame error bits."""
        import torch

        from . import _lib
        from .engine import EngineError

        off = np.asarray(spans, dtype=np.int64)
        if off.ndim == 1:
            off = np.stack([off[:-1], off[1:]], axis=1)
        off = np.ascontiguousarray(off)
        n = off.shape[0]
        if isinstance(data, torch.Tensor):
            assert data.dtype == torch.uint8 and not data.is_cuda and data.is_contiguous()
            ptr, nbytes = data.data_ptr(), data.numel()
        else:
            data = np.ascontiguousarray(data, dtype=np.uint8)
            ptr, nbytes = data.ctypes.data, data.size
        if n < 1 or off.min() < 0 or off.max() > nbytes:
            raise ValueError("frame spans do not fit the data")
        if out is None:
            out = torch.empty((n, height, width, 3), dtype=torch.uint8, device=self.device)
        assert out.is_cuda and out.is_contiguous() and tuple(out.shape) == (n, height, width, 3)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = self._lib.pa_mjpeg_decode(self._h, C.c_void_p(ptr), off.ctypes.data_as(C.c_void_p), n, height, width, int(rgb),
                                       C.c_void_p(out.data_ptr()), C.c_void_p(status.data_ptr()) if status is not None else C.c_void_p(0),
                                       stream)
        if rc != _lib.PA_OK:
            raise EngineError(rc, self._lib.pa_mjpeg_last_error(self._h).decode())
        # the copy is asynchronous and the handle keeps TWO calls in flight (two scratch sets): the host bytes of the call
        # before this one may still be crossing the link, so the last two calls' buffers stay referenced
        self._keep = (getattr(self, "_keep", (None,))[-1], (data, off))
        return out


# ---------------------------------------------------------------------------------------------------------------------
# cv2.VideoCapture's call shape
# ---------------------------------------------------------------------------------------------------------------------

class VideoCapture:
    """The subset of ``cv2.VideoCapture`` the reference uses, over Motion-JPEG, decoded on the device.

    ``VideoCapture(path)`` with ``path`` an ``.avi`` (MJPG), a raw ``.mjpeg`` / ``.mjpg`` concatenation, a directory of
    ``.jpg`` files (sorted by name, like OpenCV's image-sequence reader), or an in-memory list of JPEG byte strings."""

    def __init__(self, source, device: str = "cuda:0", batch_frames: int = 64, fps: float = 30.0):
        import torch

        self._torch = torch
        self._pos = 0
        self._opened = False
        self.fps = float(fps)
        self._device = device
        self._batch = batch_frames
        self._dec: Optional[MjpegDecoder] = None
        try:
            if isinstance(source, (list, tuple)):
                blobs = [bytes(b) for b in source]
            elif os.path.isdir(source):
                names = sorted(f for f in os.listdir(source) if f.lower().endswith((".jpg", ".jpeg")))
                blobs = [open(os.path.join(source, f), "rb").read() for f in names]
            elif str(source).lower().endswith(".avi"):
                data, off2, meta = read_avi_mjpeg(source)
                self.fps = meta["fps"] or self.fps
                blobs = None
                self._data = torch.from_numpy(data)
                self._spans = off2
            else:
                data = np.fromfile(source, dtype=np.uint8)
                blobs = None
                self._data = torch.from_numpy(data)
                self._spans = split_jpeg_stream(data)
            if blobs is not None:
                if not blobs:
                    raise VideoError("no frames")
                sizes = np.array([len(b) for b in blobs], dtype=np.int64)
                ends = np.cumsum(sizes)
                self._data = torch.from_numpy(np.frombuffer(b"".join(blobs), dtype=np.uint8).copy())
                self._spans = np.stack([ends - sizes, ends], axis=1)
            first = self._data[int(self._spans[0, 0]):int(self._spans[0, 1])].numpy()
            self.height, self.width = jpeg_frame_size(first)
            if torch.cuda.is_available():
                self._data = self._data.pin_memory()  # asynchronous host -> device copies
            self._opened = True
        except (OSError, VideoError, ValueError, struct.error):
            self._opened = False

    # -- cv2's interface ---------------------------------------------------------
    def isOpened(self) -> bool:
        return self._opened

    def get(self, prop: int) -> float:
        if not self._opened:
            return 0.0
        return {
            CAP_PROP_POS_FRAMES: float(self._pos),
            CAP_PROP_FRAME_WIDTH: float(self.width),
            CAP_PROP_FRAME_HEIGHT: float(self.height),
            CAP_PROP_FPS: float(self.fps),
            CAP_PROP_FRAME_COUNT: float(len(self._spans)),
        }.get(prop, 0.0)

    def set(self, prop: int, value) -> bool:
        if prop == CAP_PROP_POS_FRAMES and self._opened:
            self._pos = int(value)
            return True
        return False

    def read(self):
        """-> ``(ok, uint8[H, W, 3] BGR numpy)`` for the frame at the current position, which then advances; ``(False,
        None)`` past the end (``ai_runner.py:405-415`` handles exactly that)."""
        if not self._opened or not 0 <= self._pos < len(self._spans):
            return False, None
        st = self._torch.zeros(1, dtype=self._torch.int32, device=self._device)
        fr = self.read_frames(self._pos, 1, status=st)
        if int(st[0]) & 8:  # decoder states not settled in the enqueued passes (long flat areas): decode exactly
            fr = self.read_frames(self._pos, 1, status=st, exact=True)
        self._torch.cuda.synchronize(fr.device)
        self._pos += 1
        if int(st[0]):
            return False, None  # a frame that does not decode reads as a failed read (ai_runner.py:405-415 copes with it)
        return True, fr[0].cpu().numpy()

    def release(self):
        if self._dec is not None:
            self._dec.close()
            self._dec = None
        self._opened = False

    # -- the batched form ----------------------------------------------------------
    def frame_count(self) -> int:
        return len(self._spans)

    def _decoder(self, n: int, nbytes: int) -> MjpegDecoder:
        d = self._dec
        if d is None or d.max_frames < n or d.max_bytes < nbytes:
            if d is not None:
                self._torch.cuda.synchronize()
                d.close()
            per_frame = int((self._spans[:, 1] - self._spans[:, 0]).max())
            cap_n = max(n, self._batch)
            self._dec = MjpegDecoder(cap_n, self.height, self.width, max(nbytes, cap_n * per_frame) + 4096, self._device)
        return self._dec

    def read_frames(self, j0: int, n: int, out=None, status=None, rgb: bool = False, exact: bool = False):
        """Frames ``j0 .. j0 + n - 1`` -> device tensor uint8[n, H, W, 3] (BGR), enqueued on the current stream.
        ``exact``: run the entropy decoder's verify passes until nothing changes (synchronises the stream; for frames that
        came back with status bit 8)."""
        if not self._opened:
            raise VideoError("capture is not open")
        if not (0 <= j0 and j0 + n <= len(self._spans) and n >= 1):
            raise IndexError(f"frames {j0}..{j0 + n - 1} outside the stream's {len(self._spans)}")
        sp = self._spans[j0:j0 + n]
        dec = self._decoder(n, int(sp[:, 1].max() - sp[:, 0].min()))
        if exact:
            dec.set_sync_rounds(0)
        try:
            return dec.decode(self._data, sp, self.height, self.width, out=out, status=status, rgb=rgb)
        finally:
            if exact:
                dec.set_sync_rounds(MjpegDecoder.DEFAULT_SYNC_ROUNDS)
