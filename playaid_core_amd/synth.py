"""Seeded synthetic inputs for the action-recognition hot path.

Everything here is derived from a build-owned 32-bit integer hash (not
``torch.manual_seed``, which is not stable across versions/devices), so the
same seeds give bit-identical frames, boxes and weights in this container, on
the GPU box and inside the CPU oracle.

What is synthesised, and the reference shape it stands in for:

* frames   -- ``uint8[N,H,W,3]`` BGR, HWC: what ``cv2.VideoCapture.read`` hands
  to the reference (``playaid/ai_runner.py:404-405``).
* boxes    -- normalised ``cx cy w h`` per (frame, fighter): the content of the
  YOLO label lines ``"cls cx cy w h conf"`` (``playaid/ai_runner.py:53-71``).
* weights  -- a Lightning-layout ``state_dict`` for ``CNNActionDetector``
  (``playaid/models/cnn_action_detector.py:14-27``; torchvision resnet18 keys).
* stub log -- JSON-lines game log with every key ``Fighter.set_from_json``
  indexes (``playaid/fighter.py:458-555``), for the manuscript plumbing config.
"""
from __future__ import annotations

import json
import math
from typing import Dict, List, Tuple

import numpy as np

from . import constants

_M32 = np.uint64(0xFFFFFFFF)


def hash_u32(idx, seed: int) -> np.ndarray:
    """lowbias32-style avalanche of (idx, seed) -> uint32. Vectorised, exact."""
    x = (np.asarray(idx, dtype=np.uint64) + np.uint64((seed * 0x9E3779B1) & 0xFFFFFFFF)) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def uniform(shape, seed: int, lo: float, hi: float) -> np.ndarray:
    """float32 array of U[lo, hi) from the integer hash (24 random bits each)."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = hash_u32(np.arange(n, dtype=np.uint64), seed)
    u = (h >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)
    return (lo + u * (hi - lo)).astype(np.float32).reshape(shape)


def _name_seed(name: str, seed: int) -> int:
    s = seed & 0xFFFFFFFF
    for ch in name.encode():
        s = (s * 16777619) ^ ch
        s &= 0xFFFFFFFF
    return s


# ----------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------

def resnet18_param_shapes() -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of torchvision-0.15.2 ``resnet18`` parameters and buffers,
    in module order (the graph the reference instantiates at
    ``playaid/models/cnn_action_detector.py:16``)."""
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def bn(prefix, c):
        out.extend(
            [
                (prefix + ".weight", (c,)),
                (prefix + ".bias", (c,)),
                (prefix + ".running_mean", (c,)),
                (prefix + ".running_var", (c,)),
            ]
        )

    out.append(("conv1.weight", (64, 3, 7, 7)))
    bn("bn1", 64)
    cin = 64
    for li, cout in enumerate([64, 128, 256, 512], start=1):
        for b in range(2):
            p = f"layer{li}.{b}"
            c_in = cin if b == 0 else cout
            out.append((p + ".conv1.weight", (cout, c_in, 3, 3)))
            bn(p + ".bn1", cout)
            out.append((p + ".conv2.weight", (cout, cout, 3, 3)))
            bn(p + ".bn2", cout)
            if b == 0 and li > 1:
                out.append((p + ".downsample.0.weight", (cout, c_in, 1, 1)))
                bn(p + ".downsample.1", cout)
        cin = cout
    out.append(("fc.weight", (1000, 512)))
    out.append(("fc.bias", (1000,)))
    return out


def make_state_dict(seed: int = 1234, num_actions: int = 63, sequence_length: int = 7) -> Dict[str, np.ndarray]:
    """Seeded fp32 weights in the Lightning ``state_dict`` key layout
    (``model.cnn2d.*``, ``model.cnn1d.0.*``, ``model.classifier.{0,2}.*``;
    SURVEY.md section 8b). conv/linear ~ U(+-sqrt(3/fan_in)); BN gamma in
    U(0.5,1.5), beta/mean in U(-0.1,0.1), var in U(0.5,1.5)."""
    sd: Dict[str, np.ndarray] = {}

    def dense(key, shape):
        fan_in = int(np.prod(shape[1:]))
        a = math.sqrt(3.0 / fan_in)
        sd[key] = uniform(shape, _name_seed(key, seed), -a, a)

    for key, shape in resnet18_param_shapes():
        full = "model.cnn2d." + key
        if key.endswith("running_var") or (key.endswith(".weight") and len(shape) == 1):
            sd[full] = uniform(shape, _name_seed(full, seed), 0.5, 1.5)
        elif len(shape) == 1 and not key.startswith("fc"):
            sd[full] = uniform(shape, _name_seed(full, seed), -0.1, 0.1)
        elif key == "fc.bias":
            sd[full] = uniform(shape, _name_seed(full, seed), -0.05, 0.05)
        else:
            dense(full, shape)
    dense("model.cnn1d.0.weight", (512, 1000, sequence_length))
    sd["model.cnn1d.0.bias"] = uniform((512,), _name_seed("model.cnn1d.0.bias", seed), -0.05, 0.05)
    dense("model.classifier.0.weight", (128, 512))
    sd["model.classifier.0.bias"] = uniform((128,), _name_seed("model.classifier.0.bias", seed), -0.05, 0.05)
    # last layer a little wider so the 63 logits are well separated (argmax
    # stable under 1e-4 noise)
    fan_in = 128
    a = 4.0 * math.sqrt(3.0 / fan_in)
    sd["model.classifier.2.weight"] = uniform(
        (num_actions, 128), _name_seed("model.classifier.2.weight", seed), -a, a
    )
    sd["model.classifier.2.bias"] = uniform(
        (num_actions,), _name_seed("model.classifier.2.bias", seed), -0.5, 0.5
    )
    return sd


def save_checkpoint(path: str, seed: int = 1234, num_actions: int = 63, sequence_length: int = 7) -> None:
    """Write a Lightning-1.6.5-shaped ``.ckpt`` (``state_dict`` +
    ``hyper_parameters``) holding the seeded weights, so
    ``CNNActionDetector.load_from_checkpoint`` can be exercised offline."""
    import torch

    sd = {k: torch.from_numpy(v.copy()) for k, v in make_state_dict(seed, num_actions, sequence_length).items()}
    torch.save(
        {
            "state_dict": sd,
            "hyper_parameters": {
                "batch_size": 64,
                "sequence_length": sequence_length,
                "learning_rate": 2e-4,
                "num_samples": 1024,
                "freeze_encoder": False,
            },
            "pytorch-lightning_version": "1.6.5",
            "epoch": 0,
            "global_step": 0,
        },
        path,
    )


# ----------------------------------------------------------------------------
# frames + boxes
# ----------------------------------------------------------------------------

def _tri(t: int, period: int) -> float:
    """Integer triangle wave in [0,1] (exact rational, no libm)."""
    t %= 2 * period
    return (t if t <= period else 2 * period - t) / period


def fighter_box(frame_idx: int, fighter: int, height: int, width: int) -> Tuple[float, float, float, float]:
    """Normalised (cx, cy, w, h) of synthetic fighter ``fighter`` (0/1) at
    0-based frame ``frame_idx``. Boxes are about 275x315 px at 1080p (the
    projected fighter box of ``playaid/fighter.py:507-526`` scaled from 720p),
    wander so that some square crops clip the frame edge, and stay >= 128 px on
    the long side (decimation branch of INTER_AREA only)."""
    f = frame_idx
    if fighter == 0:
        cx = 0.04 + 0.92 * _tri(7 * f + 13, 97)
        cy = 0.10 + 0.82 * _tri(5 * f + 40, 61)
        ws = 0.90 + 0.25 * _tri(3 * f, 37)
        hs = 0.92 + 0.20 * _tri(2 * f + 5, 29)
    else:
        cx = 0.04 + 0.92 * _tri(11 * f + 70, 113)
        cy = 0.10 + 0.82 * _tri(3 * f + 9, 43)
        ws = 0.88 + 0.30 * _tri(5 * f + 11, 41)
        hs = 0.90 + 0.24 * _tri(4 * f + 2, 31)
    w = (275.0 / 1920.0) * ws
    h = (315.0 / 1080.0) * hs
    return (cx, cy, w, h)


def make_boxes(n: int, height: int, width: int, first_frame: int = 0) -> np.ndarray:
    """float64[n,2,4] normalised boxes for frames first_frame .. first_frame+n-1."""
    out = np.zeros((n, 2, 4), dtype=np.float64)
    for i in range(n):
        for p in range(2):
            out[i, p] = fighter_box(first_frame + i, p, height, width)
    return out


FIGHTER_CLASS_IDS = (2, 3)  # Pikachu, Joker: the "--classes 2 3" of ai_runner.py:215-217
FIGHTER_NAMES = tuple(constants.CHAR_LIST[c] for c in FIGHTER_CLASS_IDS)


def make_frame(frame_idx: int, height: int, width: int, seed: int = 7) -> np.ndarray:
    """One ``uint8[H,W,3]`` BGR frame: smooth gradient + hash noise + two
    textured fighter rectangles at ``fighter_box`` positions."""
    ys = np.arange(height, dtype=np.uint64)[:, None]
    xs = np.arange(width, dtype=np.uint64)[None, :]
    pix = ys * np.uint64(width) + xs  # [H,W]
    img = np.empty((height, width, 3), dtype=np.int32)
    g0 = (xs * np.uint64(200) // np.uint64(width)).astype(np.int32)
    g1 = (ys * np.uint64(200) // np.uint64(height)).astype(np.int32)
    g2 = (((xs + ys + np.uint64(3 * frame_idx)) % np.uint64(512)) * np.uint64(200) // np.uint64(512)).astype(np.int32)
    for c, g in enumerate((g0, g1, g2)):
        noise = hash_u32(pix * np.uint64(3) + np.uint64(c), seed * 7919 + frame_idx) & np.uint32(31)
        img[:, :, c] = g + noise.astype(np.int32)
    for p in range(2):
        cx, cy, w, h = fighter_box(frame_idx, p, height, width)
        x0 = int((cx - w / 2) * width)
        x1 = int((cx + w / 2) * width)
        y0 = int((cy - h / 2) * height)
        y1 = int((cy + h / 2) * height)
        x0c, x1c = max(x0, 0), min(x1, width)
        y0c, y1c = max(y0, 0), min(y1, height)
        if x1c <= x0c or y1c <= y0c:
            continue
        yy = np.arange(y0c, y1c, dtype=np.int64)[:, None] - y0
        xx = np.arange(x0c, x1c, dtype=np.int64)[None, :] - x0
        # 12x12 px blocks whose colours change every 4 frames (an "animation")
        blk = ((yy // 12) * 64 + (xx // 12)).astype(np.uint64)
        phase = frame_idx // 4
        for c in range(3):
            base = hash_u32(blk * np.uint64(3) + np.uint64(c), seed * 131 + p * 17 + phase * 1009) & np.uint32(127)
            fine = hash_u32((yy * 4096 + xx).astype(np.uint64) * np.uint64(3) + np.uint64(c), seed + p) & np.uint32(15)
            val = 96 + base.astype(np.int32) + fine.astype(np.int32)
            img[y0c:y1c, x0c:x1c, c] = val
    return np.clip(img, 0, 255).astype(np.uint8)


def make_frames(n: int, height: int, width: int, seed: int = 7, first_frame: int = 0) -> np.ndarray:
    out = np.empty((n, height, width, 3), dtype=np.uint8)
    for i in range(n):
        out[i] = make_frame(first_frame + i, height, width, seed)
    return out


# ----------------------------------------------------------------------------
# stub game log (manuscript plumbing config)
# ----------------------------------------------------------------------------

def make_stub_log(path: str, n_frames: int) -> None:
    """JSON-lines log, 2 lines per frame, with every key that
    ``Fighter.set_from_json`` indexes (``playaid/fighter.py:461-477,485,492-493,554``)
    and ``num_frames_left`` decreasing by one per frame so the gap fill of
    ``playaid/timeline.py:249-255`` is inert."""
    with open(path, "w") as f:
        for i in range(n_frames):
            for p in range(2):
                rec = {
                    "pos_x": -50.0 + 100.0 * p + 0.25 * i,
                    "pos_y": 0.25 * (i % 8),
                    "damage": 0.5 * (i // 16),
                    "facing": 1.0 if p == 0 else -1.0,
                    "fighter_id": p,
                    "motion_kind": 19292652517,
                    "num_frames_left": 25200 - i,
                    "shield_size": 50.0,
                    "status_kind": 0,
                    "stock_count": 3,
                    "attack_connected": False,
                    "stage_id": 86,
                    "fighter_name": 8 if p == 0 else 82,
                    "camera_position": {"x": 0.0, "y": 15.0, "z": 150.0},
                    "camera_target_position": {"x": 0.0, "y": 11.0, "z": 0.0},
                    "hitstun_left": 0.0,
                    "can_act": True,
                }
                f.write(json.dumps(rec) + "\n")


# ----------------------------------------------------------------------------
# the same frames, generated where they will be consumed
# ----------------------------------------------------------------------------

def _hash_u32_torch(idx, seed: int):
    """``hash_u32`` on an int64 torch tensor (values < 2**32; every product stays below 2**63)."""
    m = 0xFFFFFFFF
    x = (idx + ((seed * 0x9E3779B1) & m)) & m
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & m
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & m
    return x ^ (x >> 16)


def make_frames_torch(n: int, height: int, width: int, seed: int = 7, first_frame: int = 0, device="cuda", out=None,
                      noise_mask: int = 31, fine_mask: int = 15):
    """``make_frames`` computed with torch ops on ``device``: bit-identical frames (same integer
    hash, same rectangles) without synthesising and uploading gigabytes on the host -- the
    8192-frame clip of BASELINE.json configs[3] is 51 GB of raw BGR.

    ``noise_mask`` / ``fine_mask``: amplitude masks of the per-pixel noise on the background / inside the fighters. The
    defaults are ``make_frames``' (five bits of white noise per sample: the hardest content a JPEG coder can meet, 1.05 MB
    per 1080p frame at quality 95); 3 / 3 gives camera-like content (0.3-0.4 MB per frame) for the decode measurements."""
    import torch

    dev = torch.device(device)
    frames = out if out is not None else torch.empty((n, height, width, 3), dtype=torch.uint8, device=dev)
    ys = torch.arange(height, dtype=torch.int64, device=dev)[:, None]
    xs = torch.arange(width, dtype=torch.int64, device=dev)[None, :]
    pix3 = (ys * width + xs) * 3
    g0 = (xs * 200 // width).expand(height, width)
    g1 = (ys * 200 // height).expand(height, width)
    for i in range(n):
        f = first_frame + i
        g2 = ((xs + ys + 3 * f) % 512) * 200 // 512
        img = torch.empty((height, width, 3), dtype=torch.int64, device=dev)
        for c, g in enumerate((g0, g1, g2)):
            img[:, :, c] = g + (_hash_u32_torch(pix3 + c, seed * 7919 + f) & noise_mask)
        for p in range(2):
            cx, cy, w, h = fighter_box(f, p, height, width)
            x0, x1 = int((cx - w / 2) * width), int((cx + w / 2) * width)
            y0, y1 = int((cy - h / 2) * height), int((cy + h / 2) * height)
            x0c, x1c, y0c, y1c = max(x0, 0), min(x1, width), max(y0, 0), min(y1, height)
            if x1c <= x0c or y1c <= y0c:
                continue
            yy = torch.arange(y0c, y1c, dtype=torch.int64, device=dev)[:, None] - y0
            xx = torch.arange(x0c, x1c, dtype=torch.int64, device=dev)[None, :] - x0
            blk3 = ((yy // 12) * 64 + (xx // 12)) * 3
            fine3 = (yy * 4096 + xx) * 3
            phase = f // 4
            for c in range(3):
                base = _hash_u32_torch(blk3 + c, seed * 131 + p * 17 + phase * 1009) & 127
                fine = _hash_u32_torch(fine3 + c, seed + p) & fine_mask
                img[y0c:y1c, x0c:x1c, c] = 96 + base + fine
        frames[i] = img.clamp_(0, 255).to(torch.uint8)
    return frames


def make_rnn_state_dict(seed: int = 4321, num_actions: int = 63) -> Dict[str, np.ndarray]:
    """Seeded fp32 weights in the key layout of the reference's ``RNNActionDetector``
    (``playaid/models/rnn_action_detector.py:55-65``): ``resnet.*`` (torchvision resnet18 whose ``fc`` is
    ``nn.Sequential(nn.Linear(512, 300))`` -> ``resnet.fc.0.*``), ``lstm.{weight,bias}_{ih,hh}_l{0,1,2}``
    (``nn.LSTM(300, 512, num_layers=3)``), ``action_decoder.{0,2}.*``. Same distributions as
    ``make_state_dict``; LSTM tensors ~ U(+-1/sqrt(512)) like ``nn.LSTM.reset_parameters``."""
    sd: Dict[str, np.ndarray] = {}

    def dense(key, shape):
        a = math.sqrt(3.0 / int(np.prod(shape[1:])))
        sd[key] = uniform(shape, _name_seed(key, seed), -a, a)

    for key, shape in resnet18_param_shapes():
        if key.startswith("fc."):
            continue
        full = "resnet." + key
        if key.endswith("running_var") or (key.endswith(".weight") and len(shape) == 1):
            sd[full] = uniform(shape, _name_seed(full, seed), 0.5, 1.5)
        elif len(shape) == 1:
            sd[full] = uniform(shape, _name_seed(full, seed), -0.1, 0.1)
        else:
            dense(full, shape)
    dense("resnet.fc.0.weight", (300, 512))
    sd["resnet.fc.0.bias"] = uniform((300,), _name_seed("resnet.fc.0.bias", seed), -0.05, 0.05)
    k = 1.0 / math.sqrt(512.0)
    for layer in range(3):
        in_dim = 300 if layer == 0 else 512
        for name, shape in (("weight_ih", (2048, in_dim)), ("weight_hh", (2048, 512)), ("bias_ih", (2048,)), ("bias_hh", (2048,))):
            key = f"lstm.{name}_l{layer}"
            sd[key] = uniform(shape, _name_seed(key, seed), -k, k)
    dense("action_decoder.0.weight", (128, 512))
    sd["action_decoder.0.bias"] = uniform((128,), _name_seed("action_decoder.0.bias", seed), -0.05, 0.05)
    a = 4.0 * math.sqrt(3.0 / 128)
    sd["action_decoder.2.weight"] = uniform((num_actions, 128), _name_seed("action_decoder.2.weight", seed), -a, a)
    sd["action_decoder.2.bias"] = uniform((num_actions,), _name_seed("action_decoder.2.bias", seed), -0.5, 0.5)
    return sd


def make_resformer_state_dict(seed: int = 2468, num_actions: int = 63, sequence_length: int = 7) -> Dict[str, np.ndarray]:
    """Seeded fp32 weights in the key layout of the reference's ``ResnetTransformerDetector``
    (``playaid/models/resnet_transformer_detector.py:26-66,127``): ``model.resnet.*`` (timm ``resnet50`` without
    classifier), ``model.resnet_ffn``, the ``model.freq_encoding`` buffer, ``model.transformer.layers.{0,1,2}.*``
    (``nn.TransformerEncoderLayer(256, 8)``: ``self_attn.in_proj_*``, ``self_attn.out_proj``, ``linear1``,
    ``linear2``, ``norm1``, ``norm2``), ``model.classifier``; also the tensors the forward never reads
    (``model.encoder_layer.*``, ``model.resnet_classifier``) because a real checkpoint holds them. The last
    BatchNorm of every bottleneck gets a small gamma so that 16 residual additions keep the activations O(1)."""
    from .resnet_transformer_detector import D_MODEL, FF_DIM, HIDDEN_DIM, NUM_LAYERS, resnet50_param_shapes, time_encoding

    sd: Dict[str, np.ndarray] = {}

    def dense(key, shape, gain=1.0):
        a = gain * math.sqrt(3.0 / int(np.prod(shape[1:])))
        sd[key] = uniform(shape, _name_seed(key, seed), -a, a)

    def small(key, shape, a=0.05):
        sd[key] = uniform(shape, _name_seed(key, seed), -a, a)

    for key, shape in resnet50_param_shapes():
        full = "model.resnet." + key
        if key.endswith("running_var"):
            sd[full] = uniform(shape, _name_seed(full, seed), 0.5, 1.5)
        elif key.endswith(".weight") and len(shape) == 1:
            lo, hi = (0.1, 0.3) if ".bn3." in key else (0.5, 1.5)
            sd[full] = uniform(shape, _name_seed(full, seed), lo, hi)
        elif len(shape) == 1:
            sd[full] = uniform(shape, _name_seed(full, seed), -0.1, 0.1)
        else:
            dense(full, shape, gain=math.sqrt(2.0))
    dense("model.resnet_ffn.weight", (HIDDEN_DIM, 2048))
    small("model.resnet_ffn.bias", (HIDDEN_DIM,))
    sd["model.freq_encoding"] = time_encoding(sequence_length)

    def encoder_layer(prefix):
        dense(prefix + "self_attn.in_proj_weight", (3 * D_MODEL, D_MODEL))
        small(prefix + "self_attn.in_proj_bias", (3 * D_MODEL,))
        dense(prefix + "self_attn.out_proj.weight", (D_MODEL, D_MODEL))
        small(prefix + "self_attn.out_proj.bias", (D_MODEL,))
        dense(prefix + "linear1.weight", (FF_DIM, D_MODEL))
        small(prefix + "linear1.bias", (FF_DIM,))
        dense(prefix + "linear2.weight", (D_MODEL, FF_DIM))
        small(prefix + "linear2.bias", (D_MODEL,))
        for nm in ("norm1", "norm2"):
            sd[prefix + nm + ".weight"] = uniform((D_MODEL,), _name_seed(prefix + nm + ".weight", seed), 0.8, 1.2)
            small(prefix + nm + ".bias", (D_MODEL,), 0.1)

    encoder_layer("model.encoder_layer.")
    for layer in range(NUM_LAYERS):
        encoder_layer(f"model.transformer.layers.{layer}.")
    dense("model.resnet_classifier.weight", (num_actions, HIDDEN_DIM))
    small("model.resnet_classifier.bias", (num_actions,))
    dense("model.classifier.weight", (num_actions, D_MODEL), gain=4.0)
    small("model.classifier.bias", (num_actions,), 0.5)
    return sd


# ----------------------------------------------------------------------------
# compressed clips (Motion-JPEG) for the decode path
# ----------------------------------------------------------------------------

def encode_jpeg_frames(frames_bgr, quality: int = 95, subsampling: int = 2, restart_marker_blocks: int = 0,
                       restart_marker_rows: int = 0, optimize: bool = False) -> List[bytes]:
    """uint8[n,H,W,3] BGR frames (or a list of them; uint8[H,W] = grey) -> one baseline JPEG file per frame, written by
    the libjpeg-turbo behind Pillow with the settings ``cv2.imwrite`` / ``cv2.VideoWriter("MJPG")`` use (quality 95,
    4:2:0 = ``subsampling`` 2). Synthetic-data helper for tests and ``bench.py`` only: encoding is not part of the
    accelerated path (the reference only ever DECODES video, ``ai_runner.py:153``)."""
    import io

    from PIL import Image

    out = []
    for f in frames_bgr:
        f = np.asarray(f)
        im = Image.fromarray(f if f.ndim == 2 else np.ascontiguousarray(f[..., ::-1]))
        kw = {}
        if restart_marker_blocks:
            kw["restart_marker_blocks"] = restart_marker_blocks
        if restart_marker_rows:
            kw["restart_marker_rows"] = restart_marker_rows
        if f.ndim == 3:
            kw["subsampling"] = subsampling
        b = io.BytesIO()
        im.save(b, "JPEG", quality=quality, optimize=optimize, **kw)
        out.append(b.getvalue())
    return out


# ----------------------------------------------------------------------------
# a YOLOv5s checkpoint's state dict (the detector the reference shells out to)
# ----------------------------------------------------------------------------

def make_yolov5s_state_dict(seed: int = 1357, nc: int = 6) -> Dict[str, np.ndarray]:
    """Seeded fp32 weights in the key layout of an ultralytics/yolov5 v7.0 ``yolov5s`` checkpoint (``models/yolov5s.yaml``:
    ``model.<i>.conv.weight``, ``model.<i>.bn.{weight,bias,running_mean,running_var}``, C3's ``cv1 / cv2 / cv3 / m.<j>.cv1|cv2``,
    SPPF's ``cv1 / cv2``, ``model.24.m.<k>.{weight,bias}``, ``model.24.anchors`` in stride units). ``nc`` = 6: the
    reference's ``CHAR_LIST`` (``constants.py:51``; ``--classes 2 3`` picks two of them). Same distributions as
    ``make_state_dict``; the Detect biases as ``Model._initialize_biases`` leaves them (objectness ~ 8 / (640 / s)^2)."""
    sd: Dict[str, np.ndarray] = {}
    c = (32, 64, 128, 256, 512)

    def conv(prefix, c1, c2, k):
        fan = c1 * k * k
        a = math.sqrt(3.0 / fan) * 1.5   # SiLU shrinks the signal: this gain keeps 60 layers alive without saturating the head
        sd[prefix + ".conv.weight"] = uniform((c2, c1, k, k), _name_seed(prefix + ".conv.weight", seed), -a, a)
        sd[prefix + ".bn.weight"] = uniform((c2,), _name_seed(prefix + ".bn.weight", seed), 0.5, 1.5)
        sd[prefix + ".bn.bias"] = uniform((c2,), _name_seed(prefix + ".bn.bias", seed), -0.1, 0.1)
        sd[prefix + ".bn.running_mean"] = uniform((c2,), _name_seed(prefix + ".bn.running_mean", seed), -0.1, 0.1)
        sd[prefix + ".bn.running_var"] = uniform((c2,), _name_seed(prefix + ".bn.running_var", seed), 0.5, 1.5)

    def c3(prefix, c1, c2, n):
        c_ = c2 // 2
        conv(prefix + ".cv1", c1, c_, 1)
        conv(prefix + ".cv2", c1, c_, 1)
        conv(prefix + ".cv3", 2 * c_, c2, 1)
        for j in range(n):
            conv(f"{prefix}.m.{j}.cv1", c_, c_, 1)
            conv(f"{prefix}.m.{j}.cv2", c_, c_, 3)

    conv("model.0", 3, c[0], 6)
    conv("model.1", c[0], c[1], 3)
    c3("model.2", c[1], c[1], 1)
    conv("model.3", c[1], c[2], 3)
    c3("model.4", c[2], c[2], 2)
    conv("model.5", c[2], c[3], 3)
    c3("model.6", c[3], c[3], 3)
    conv("model.7", c[3], c[4], 3)
    c3("model.8", c[4], c[4], 1)
    conv("model.9.cv1", c[4], c[3], 1)
    conv("model.9.cv2", 4 * c[3], c[4], 1)
    conv("model.10", c[4], c[3], 1)
    c3("model.13", 2 * c[3], c[3], 1)
    conv("model.14", c[3], c[2], 1)
    c3("model.17", 2 * c[2], c[2], 1)
    conv("model.18", c[2], c[2], 3)
    c3("model.20", 2 * c[2], c[3], 1)
    conv("model.21", c[3], c[3], 3)
    c3("model.23", 2 * c[3], c[4], 1)
    no = 5 + nc
    for i, (ch, s) in enumerate(((c[2], 8), (c[3], 16), (c[4], 32))):
        a = math.sqrt(3.0 / ch)
        sd[f"model.24.m.{i}.weight"] = uniform((3 * no, ch, 1, 1), _name_seed(f"model.24.m.{i}.weight", seed), -a, a)
        b = uniform((3, no), _name_seed(f"model.24.m.{i}.bias", seed), -0.05, 0.05)
        b[:, 4] += math.log(8 / (640 / s) ** 2)
        b[:, 5:] += math.log(0.6 / (nc - 0.99999))
        sd[f"model.24.m.{i}.bias"] = b.reshape(-1)
    anchors = np.array([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]], np.float32).reshape(3, 3, 2)
    sd["model.24.anchors"] = anchors / np.array([8, 16, 32], np.float32)[:, None, None]
    return sd
