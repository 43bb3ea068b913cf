"""``ResnetTransformerDetector`` -- the reference's second alternative model.

Mirror of ``playaid/models/resnet_transformer_detector.py:26-141`` for inference: a timm ``resnet50`` without
classifier (2048 pooled features per frame, ``:37``), ``Linear(2048, 247)`` (``:41``), the 9-value time encoding of the
frame's position in the window appended (``:18-23,43-49,77-80``) -> 256, three post-norm
``nn.TransformerEncoderLayer(d_model=256, nhead=8)`` (``:53-60``), ``Linear(256, A)`` (``:65``), ``log_softmax`` over
the actions (``:141``); ``model(x)`` with ``x: float32[B,S,3,128,128]`` -> ``float32[B,S,A]``. As with the LSTM model the
encoder is built without ``batch_first``, so attention runs ACROSS THE WINDOWS of a call (dimension 0) for each
frame slot; that is reproduced as is.

The ResNet-50 runs on the engine's fp32 convolution kernels through a layer table (``pa_convnet_*``,
``csrc/convnet.hip``): BatchNorm folded in fp64 here, 1x1 convolutions as GEMMs on the im2col engine, the stride-1
3x3 ones on the patch-resident kernel, the stem on ``stem_pool_kernel``. The encoder and the classifier run in
``csrc/transformer.hip`` (``pa_encoder_*``). No PyTorch fallback; training hooks (``:143-260``) are out of scope.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Mapping, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .engine import EngineError, _ptr

HIDDEN_DIM, NUM_FREQ, NUM_HEADS, NUM_LAYERS, FF_DIM = 247, 4, 8, 3, 2048
D_MODEL = HIDDEN_DIM + 1 + 2 * NUM_FREQ
RESNET50_BLOCKS = (3, 4, 6, 3)
BN_EPS = 1e-5


def _np(v) -> np.ndarray:
    return v if isinstance(v, np.ndarray) else v.detach().cpu().numpy()


def resnet50_param_shapes() -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of a timm / torchvision ``resnet50`` without its classifier (bottleneck blocks [3, 4, 6, 3],
    stride on the 3x3 convolution), in module order."""
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def bn(prefix, c):
        out.extend([(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)), (prefix + ".running_var", (c,))])

    out.append(("conv1.weight", (64, 3, 7, 7)))
    bn("bn1", 64)
    cin = 64
    for li, (width, blocks) in enumerate(zip((64, 128, 256, 512), RESNET50_BLOCKS), start=1):
        for b in range(blocks):
            p = f"layer{li}.{b}"
            out.append((p + ".conv1.weight", (width, cin, 1, 1)))
            bn(p + ".bn1", width)
            out.append((p + ".conv2.weight", (width, width, 3, 3)))
            bn(p + ".bn2", width)
            out.append((p + ".conv3.weight", (4 * width, width, 1, 1)))
            bn(p + ".bn3", 4 * width)
            if b == 0:
                out.append((p + ".downsample.0.weight", (4 * width, cin, 1, 1)))
                bn(p + ".downsample.1", 4 * width)
            cin = 4 * width
    return out


def time_encoding(sequence_length: int) -> np.ndarray:
    """``time_encoding(torch.linspace(0, 1, S).reshape(-1, 1), 4)`` (``:18-23,43-49``) -> float32[S, 9]:
    x, then cos(pi x 2^i), sin(pi x 2^i) for i = 0..3, computed by torch in fp32 exactly as the reference does."""
    x = torch.linspace(0, 1, sequence_length).reshape(-1, 1)
    out = [x]
    for i in range(NUM_FREQ):
        out.extend((torch.cos(np.pi * x * (2 ** i)), torch.sin(np.pi * x * (2 ** i))))
    return torch.cat(out, dim=1).numpy().astype(np.float32)


class _Table:
    """Builds the ``pa_conv_desc`` table, the buffer plan and the folded weight blob of a ResNet-50."""

    def __init__(self):
        self.descs: List[dict] = []
        self.buf_floats: List[int] = []
        self.weights: List[np.ndarray] = []
        self.n_weights = 0

    def buffer(self, floats_per_crop: int) -> int:
        self.buf_floats.append(int(floats_per_crop))
        return len(self.buf_floats) - 1

    def put(self, a: np.ndarray) -> int:
        off = self.n_weights
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        pad = (-a.size) % 64  # keep every tensor 256-byte aligned
        self.weights.append(a)
        if pad:
            self.weights.append(np.zeros(pad, np.float32))
        self.n_weights += a.size + pad
        return off

    def fold(self, sd: Mapping, conv_key: str, bn_key: str):
        w = _np(sd[conv_key + ".weight"]).astype(np.float64)
        g, b = _np(sd[bn_key + ".weight"]).astype(np.float64), _np(sd[bn_key + ".bias"]).astype(np.float64)
        m, v = _np(sd[bn_key + ".running_mean"]).astype(np.float64), _np(sd[bn_key + ".running_var"]).astype(np.float64)
        scale = g / np.sqrt(v + BN_EPS)
        return w * scale[:, None, None, None], b - m * scale

    def conv(self, sd, conv_key, bn_key, cin, cout, k, stride, in_hw, in_buf, in_pad, out_buf, out_pad, res_buf, relu):
        w, bias = self.fold(sd, conv_key, bn_key)
        assert w.shape == (cout, cin, k, k), (conv_key, w.shape)
        self.descs.append(dict(kind=0, cin=cin, cout=cout, ksize=k, stride=stride, in_hw=in_hw, in_buf=in_buf, in_pad=in_pad,
                               out_buf=out_buf, out_pad=out_pad, res_buf=res_buf, relu=int(relu),
                               w_off=self.put(w.transpose(0, 2, 3, 1)), b_off=self.put(bias)))


def build_resnet50_table(state_dict: Mapping, prefix: str = "model.resnet."):
    """-> (descs, buf_floats_per_crop, weights float32, feature_dim). Buffers: 0 stem output (bordered), 1 / 2 block
    outputs (ping-pong), 3 the 3x3 convolution's output, 4 the downsample branch, then one bordered buffer per
    (map size, width) a 3x3 convolution reads, last the pooled 2048-vector."""
    sd = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
    for key, shape in resnet50_param_shapes():
        if key not in sd:
            raise KeyError(f"state_dict is missing {prefix}{key}")
        if tuple(_np(sd[key]).shape) != tuple(shape):
            raise ValueError(f"{prefix}{key}: expected shape {tuple(shape)}, got {tuple(_np(sd[key]).shape)}")
    t = _Table()
    stem_out = t.buffer(34 * 34 * 64)
    ping, pong = t.buffer(32 * 32 * 256), t.buffer(32 * 32 * 256)
    t2, ds = t.buffer(32 * 32 * 64), t.buffer(32 * 32 * 256)
    bordered: Dict[Tuple[int, int], int] = {}
    # stem: [64][7 ky][8 px][4 ch]
    w, bias = t.fold(sd, "conv1", "bn1")
    stem = np.zeros((64, 7, 8, 4), np.float64)
    stem[:, :, :7, :3] = w.transpose(0, 2, 3, 1)
    t.descs.append(dict(kind=1, cin=3, cout=64, ksize=7, stride=2, in_hw=128, in_buf=0, in_pad=3, out_buf=stem_out, out_pad=1,
                        res_buf=-1, relu=1, w_off=t.put(stem), b_off=t.put(bias)))
    cur, cur_pad, cin, hw = stem_out, 1, 64, 32
    for li, (width, blocks) in enumerate(zip((64, 128, 256, 512), RESNET50_BLOCKS), start=1):
        for b in range(blocks):
            p = f"layer{li}.{b}"
            stride = 2 if (b == 0 and li > 1) else 1
            out_hw = hw // stride
            key = (hw, width)
            if key not in bordered:
                bordered[key] = t.buffer((hw + 2) * (hw + 2) * width)
            t1 = bordered[key]
            nxt = ping if cur != ping else pong
            t.conv(sd, p + ".conv1", p + ".bn1", cin, width, 1, 1, hw, cur, cur_pad, t1, 1, -1, True)
            t.conv(sd, p + ".conv2", p + ".bn2", width, width, 3, stride, hw, t1, 1, t2, 0, -1, True)
            if b == 0:
                t.conv(sd, p + ".downsample.0", p + ".downsample.1", cin, 4 * width, 1, stride, hw, cur, cur_pad, ds, 0, -1, False)
                res = ds
            else:
                res = cur
            t.conv(sd, p + ".conv3", p + ".bn3", width, 4 * width, 1, 1, out_hw, t2, 0, nxt, 0, res, True)
            cur, cur_pad, cin, hw = nxt, 0, 4 * width, out_hw
    pooled = t.buffer(cin)
    t.descs.append(dict(kind=2, cin=cin, cout=cin, ksize=1, stride=1, in_hw=hw, in_buf=cur, in_pad=0, out_buf=pooled, out_pad=0,
                        res_buf=-1, relu=0, w_off=0, b_off=0))
    return t.descs, t.buf_floats, np.concatenate(t.weights), cin


class ConvNet:
    """A layer table on the device (``pa_convnet_create`` / ``pa_convnet_forward``)."""

    def __init__(self, descs, buf_floats, weights: np.ndarray, out_floats: int, device: str = "cuda:0", max_crops: int = 64,
                 compute_dtype: str = "f32"):
        self._lib = _lib.load()
        if compute_dtype not in ("f32", "emulated_f32"):
            raise ValueError("compute_dtype must be 'f32' or 'emulated_f32'")
        self.compute_dtype = compute_dtype
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("no HIP device visible to PyTorch-ROCm; this path has no CPU fallback")
        self.device = torch.device(device)
        self.max_crops = max_crops
        self.out_floats = out_floats
        arr = (_lib.pa_conv_desc * len(descs))()
        for i, d in enumerate(descs):
            for k, v in d.items():
                setattr(arr[i], k, int(v))
        bufs = (C.c_int64 * len(buf_floats))(*buf_floats)
        weights = np.ascontiguousarray(weights, dtype=np.float32)
        h = C.c_void_p()
        rc = self._lib.pa_convnet_create_dtype(self.device.index or 0, arr, len(descs), bufs, len(buf_floats),
                                               weights.ctypes.data_as(C.c_void_p), weights.size, max_crops, _lib.DTYPES[compute_dtype], C.byref(h))
        self._h = h
        if rc != 0:
            msg = self._lib.pa_convnet_last_error(h).decode() if h else "bad argument"
            self.close()
            raise EngineError(rc, msg)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pa_convnet_destroy(self._h)
            self._h = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x float32[n,3,128,128] -> float32[n, out_floats] on the device (groups of max_crops)."""
        xd = x.to(self.device, torch.float32).contiguous()
        n = int(xd.shape[0])
        out = torch.empty((n, self.out_floats), dtype=torch.float32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        for c0 in range(0, n, self.max_crops):
            m = min(self.max_crops, n - c0)
            rc = self._lib.pa_convnet_forward(self._h, _ptr(xd[c0:]), m, _ptr(out[c0:]), self.out_floats, stream)
            if rc != 0:
                raise EngineError(rc, self._lib.pa_convnet_last_error(self._h).decode())
        return out


def pack_encoder_blob(state_dict: Mapping, num_actions: int, sequence_length: int) -> np.ndarray:
    """-> uint8 blob of ``pa_encoder_create`` (layout: include/playaid_hip.h)."""
    enc_dim = 1 + 2 * NUM_FREQ
    hdr = np.zeros(16, np.int32)
    hdr[:10] = [_lib.PA_ENCODER_MAGIC, 1, 2048, HIDDEN_DIM, sequence_length, enc_dim, NUM_HEADS, NUM_LAYERS, FF_DIM, num_actions]
    parts = [hdr.view(np.uint8)]

    def take(key, shape):
        if key not in state_dict:
            raise KeyError(f"state_dict is missing {key}")
        a = np.ascontiguousarray(_np(state_dict[key]), dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{key}: expected shape {tuple(shape)}, got {tuple(a.shape)}")
        parts.append(a.reshape(-1).view(np.uint8))

    take("model.resnet_ffn.weight", (HIDDEN_DIM, 2048))
    take("model.resnet_ffn.bias", (HIDDEN_DIM,))
    take("model.freq_encoding", (sequence_length, enc_dim))
    for layer in range(NUM_LAYERS):
        p = f"model.transformer.layers.{layer}."
        take(p + "self_attn.in_proj_weight", (3 * D_MODEL, D_MODEL))
        take(p + "self_attn.in_proj_bias", (3 * D_MODEL,))
        take(p + "self_attn.out_proj.weight", (D_MODEL, D_MODEL))
        take(p + "self_attn.out_proj.bias", (D_MODEL,))
        take(p + "linear1.weight", (FF_DIM, D_MODEL))
        take(p + "linear1.bias", (FF_DIM,))
        take(p + "linear2.weight", (D_MODEL, FF_DIM))
        take(p + "linear2.bias", (D_MODEL,))
        for nm in ("norm1", "norm2"):
            take(p + nm + ".weight", (D_MODEL,))
            take(p + nm + ".bias", (D_MODEL,))
    take("model.classifier.weight", (num_actions, D_MODEL))
    take("model.classifier.bias", (num_actions,))
    return np.concatenate(parts)


class ResnetTransformerDetector:
    def __init__(
        self,
        actions: List[str],
        batch_size: int = 64,
        sequence_length: int = 4,
        learning_rate: float = 2e-4,
        num_samples: int = 1024,
        freeze_encoder=False,
        state_dict: Optional[Mapping] = None,
        device: str = "cuda:0",
        max_rows: int = 448,
        **kwargs,
    ):
        if state_dict is None:
            raise ValueError("ResnetTransformerDetector needs weights: use load_from_checkpoint() or pass state_dict= "
                             "(timm's pretrained download is not available offline)")
        a = int(_np(state_dict["model.classifier.weight"]).shape[0])
        if a != len(actions):
            raise ValueError(f"checkpoint has {a} action logits but {len(actions)} actions were given")
        s = int(_np(state_dict["model.freq_encoding"]).shape[0])
        if s != sequence_length:
            raise ValueError(f"checkpoint encodes {s} frame slots but sequence_length={sequence_length}")
        self.actions = list(actions)
        self.num_actions = a
        self.sequence_length = s
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.num_samples = num_samples
        self.dataset_kwargs = kwargs
        self.training = False
        self.max_rows = max_rows
        self._lib = _lib.load()
        descs, bufs, weights, feat_dim = build_resnet50_table(state_dict)
        # (compute_dtype: beyond the reference's arguments -- "emulated_f32" = pa_convnet_create_dtype(PA_DTYPE_EMULATED_F32); never the default)
        self._net = ConvNet(descs, bufs, weights, feat_dim, device=device, max_crops=min(max_rows, 64),
                            compute_dtype=kwargs.get("compute_dtype", "f32"))
        self.device = self._net.device
        blob = pack_encoder_blob(state_dict, a, s)
        enc_dim = 1 + 2 * NUM_FREQ
        assert blob.nbytes == self._lib.pa_encoder_blob_bytes(2048, HIDDEN_DIM, s, enc_dim, NUM_LAYERS, FF_DIM, a)
        h = C.c_void_p()
        rc = self._lib.pa_encoder_create(self.device.index or 0, 2048, HIDDEN_DIM, s, enc_dim, NUM_HEADS, NUM_LAYERS, FF_DIM, a, max_rows,
                                         blob.ctypes.data_as(C.c_void_p), blob.nbytes, C.byref(h))
        self._h = h
        if rc != 0:
            msg = self._lib.pa_encoder_last_error(h).decode() if h else "bad argument"
            self.close()
            raise EngineError(rc, msg)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, map_location=None, **kwargs):
        """Lightning ``.ckpt`` (``state_dict`` + ``hyper_parameters``; keyword arguments override the saved ones,
        ``action_detector.py:55``)."""
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        if "state_dict" not in ckpt:
            raise KeyError(f"{checkpoint_path} has no 'state_dict' (not a Lightning checkpoint)")
        hparams = dict(ckpt.get("hyper_parameters", {}) or {})
        hparams.update(kwargs)
        return cls(state_dict=ckpt["state_dict"], **hparams)

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("training is out of scope for the MI355X inference path")
        return self

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pa_encoder_destroy(self._h)
            self._h = None
        if getattr(self, "_net", None) is not None:
            self._net.close()
            self._net = None

    def forward(self, frames: torch.Tensor) -> torch.Tensor:
        """frames [batch_size, frames_per_sequence, channel, height, width] -> log-probabilities [batch, frames, A]."""
        if frames.dim() != 5 or tuple(frames.shape[2:]) != (3, 128, 128) or frames.shape[1] != self.sequence_length:
            raise ValueError(f"expected [B,{self.sequence_length},3,128,128], got {tuple(frames.shape)}")
        b, s = int(frames.shape[0]), int(frames.shape[1])
        if b * s > self.max_rows:
            raise ValueError(f"{b} x {s} rows exceed max_rows={self.max_rows}")
        feats = self._net.forward(frames.reshape(b * s, 3, 128, 128))
        out = torch.empty((b, s, self.num_actions), dtype=torch.float32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = self._lib.pa_encoder_forward(self._h, _ptr(feats), feats.shape[1], b, s, _ptr(out), stream)
        if rc != 0:
            raise EngineError(rc, self._lib.pa_encoder_last_error(self._h).decode())
        return out if frames.is_cuda else out.cpu()

    __call__ = forward
