"""``RNNActionDetector`` -- the reference's alternative temporal model.

Mirror of ``playaid/models/rnn_action_detector.py:14-95`` for inference: torchvision ``resnet18`` with
``fc = Sequential(Linear(512, 300))`` per frame, ``nn.LSTM(300, 512, num_layers=3)``, ``Linear(512, 128)`` + ReLU,
``Linear(128, A)``, ``log_softmax``; ``model(x)`` with ``x: float32[B,S,3,128,128]`` -> ``float32[B*S, A]``, one row
per (window, frame) (``:74-95``). The reference gives the LSTM a ``[B, S, 300]`` tensor without ``batch_first``, so
the recurrence runs over the WINDOWS of a call and the frames of a window are its batch; that is kept as is.

The backbone runs on the engine's convolution kernels (``pa_backbone_windows``: the fc rows are zero-padded from 300
to the engine's 1000, the Conv1d head of the engine is unused), the recurrent head and decoder in
``csrc/lstm.hip`` (``pa_lstm_forward``). No PyTorch fallback; training hooks (``:97-260``) are out of scope.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Mapping, Optional

import numpy as np
import torch

from . import _lib
from .engine import Engine, EngineError, _ptr
from .synth import resnet18_param_shapes

INPUT_DIM, HIDDEN_DIM, NUM_LAYERS = 300, 512, 3


def _np(v) -> np.ndarray:
    return v if isinstance(v, np.ndarray) else v.detach().cpu().numpy()


def backbone_state_dict(state_dict: Mapping) -> Dict[str, np.ndarray]:
    """``resnet.*`` tensors re-keyed for ``Engine`` (``model.cnn2d.*``): fc zero-padded 300 -> 1000 rows, a zero
    one-tap Conv1d head and classifier (never run)."""
    out: Dict[str, np.ndarray] = {}
    for key, shape in resnet18_param_shapes():
        if key.startswith("fc."):
            continue
        a = np.ascontiguousarray(_np(state_dict["resnet." + key]), dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"resnet.{key}: expected shape {tuple(shape)}, got {tuple(a.shape)}")
        out["model.cnn2d." + key] = a
    w = np.ascontiguousarray(_np(state_dict["resnet.fc.0.weight"]), dtype=np.float32)
    b = np.ascontiguousarray(_np(state_dict["resnet.fc.0.bias"]), dtype=np.float32)
    if w.shape != (INPUT_DIM, 512) or b.shape != (INPUT_DIM,):
        raise ValueError(f"resnet.fc.0: expected (300, 512) / (300,), got {w.shape} / {b.shape}")
    fw = np.zeros((1000, 512), dtype=np.float32)
    fb = np.zeros((1000,), dtype=np.float32)
    fw[:INPUT_DIM] = w
    fb[:INPUT_DIM] = b
    out["model.cnn2d.fc.weight"] = fw
    out["model.cnn2d.fc.bias"] = fb
    out["model.cnn1d.0.weight"] = np.zeros((512, 1000, 1), dtype=np.float32)
    out["model.cnn1d.0.bias"] = np.zeros((512,), dtype=np.float32)
    out["model.classifier.0.weight"] = np.zeros((128, 512), dtype=np.float32)
    out["model.classifier.0.bias"] = np.zeros((128,), dtype=np.float32)
    out["model.classifier.2.weight"] = np.zeros((1, 128), dtype=np.float32)
    out["model.classifier.2.bias"] = np.zeros((1,), dtype=np.float32)
    return out


def pack_lstm_blob(state_dict: Mapping, num_actions: int) -> np.ndarray:
    """-> uint8 blob of ``pa_lstm_create`` (layout: include/playaid_hip.h)."""
    parts = [np.array([_lib.PA_LSTM_MAGIC, 1, INPUT_DIM, HIDDEN_DIM, NUM_LAYERS, num_actions, 0, 0], dtype=np.int32).view(np.uint8)]

    def take(key, shape):
        if key not in state_dict:
            raise KeyError(f"state_dict is missing {key}")
        a = np.ascontiguousarray(_np(state_dict[key]), dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{key}: expected shape {tuple(shape)}, got {tuple(a.shape)}")
        parts.append(a.reshape(-1).view(np.uint8))

    for layer in range(NUM_LAYERS):
        in_dim = INPUT_DIM if layer == 0 else HIDDEN_DIM
        take(f"lstm.weight_ih_l{layer}", (4 * HIDDEN_DIM, in_dim))
        take(f"lstm.weight_hh_l{layer}", (4 * HIDDEN_DIM, HIDDEN_DIM))
        take(f"lstm.bias_ih_l{layer}", (4 * HIDDEN_DIM,))
        take(f"lstm.bias_hh_l{layer}", (4 * HIDDEN_DIM,))
    take("action_decoder.0.weight", (128, HIDDEN_DIM))
    take("action_decoder.0.bias", (128,))
    take("action_decoder.2.weight", (num_actions, 128))
    take("action_decoder.2.bias", (num_actions,))
    return np.concatenate(parts)


class RNNActionDetector:
    def __init__(
        self,
        fighter_name: str,
        actions: List[str],
        batch_size: int = 8,
        learning_rate: float = 2e-4,
        num_samples: int = 1024,
        freeze_encoder=False,
        state_dict: Optional[Mapping] = None,
        device: str = "cuda:0",
        max_rows: int = 1024,
        **kwargs,
    ):
        if state_dict is None:
            raise ValueError("RNNActionDetector needs weights: use load_from_checkpoint() or pass state_dict=")
        a = int(_np(state_dict["action_decoder.2.weight"]).shape[0])
        if a != len(actions):
            raise ValueError(f"checkpoint has {a} action logits but {len(actions)} actions were given")
        self.fighter_name = fighter_name
        self.actions = list(actions)
        self.num_actions = a
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.num_samples = num_samples
        self.dataset_kwargs = kwargs
        self.training = False
        self.max_rows = max_rows
        # (compute_dtype: beyond the reference's arguments -- "emulated_f32" puts the backbone's stride-2 openers and branch GEMMs on the
        # emulated-fp32 kernels, PA_DTYPE_EMULATED_F32; never the default)
        self._engine = Engine(backbone_state_dict(state_dict), device=device, num_fighters=1, frame_delta=1,
                              max_batch_frames=min(max_rows, 128), max_clip_frames=1, compute_dtype=kwargs.get("compute_dtype", "f32"))
        self._lib = self._engine._lib
        blob = pack_lstm_blob(state_dict, a)
        assert blob.nbytes == self._lib.pa_lstm_blob_bytes(INPUT_DIM, HIDDEN_DIM, NUM_LAYERS, a)
        h = C.c_void_p()
        rc = self._lib.pa_lstm_create(self._engine.device.index or 0, INPUT_DIM, HIDDEN_DIM, NUM_LAYERS, a, max_rows,
                                      blob.ctypes.data_as(C.c_void_p), blob.nbytes, C.byref(h))
        self._h = h
        if rc != 0:
            msg = self._lib.pa_lstm_last_error(h).decode() if h else "bad argument"
            self.close()
            raise EngineError(rc, msg)
        self._feats = torch.zeros((max_rows, _lib.PA_FEATURE_STRIDE), dtype=torch.float32, device=self._engine.device)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, map_location=None, **kwargs):
        """Lightning ``.ckpt`` (``state_dict`` + ``hyper_parameters``; keyword arguments override the saved ones,
        as in ``visualizations/rnn_action_detector_vis.py:79-84``)."""
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
            self._lib.pa_lstm_destroy(self._h)
            self._h = None
        if getattr(self, "_engine", None) is not None:
            self._engine.close()
            self._engine = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [batch_size, frames_per_sequence, channel, height, width] -> log-probabilities [batch * frames, A]."""
        if x.dim() != 5 or tuple(x.shape[2:]) != (3, 128, 128):
            raise ValueError(f"expected [B,S,3,128,128], got {tuple(x.shape)}")
        b, s = int(x.shape[0]), int(x.shape[1])
        if b * s > self.max_rows:
            raise ValueError(f"{b} x {s} rows exceed max_rows={self.max_rows}")
        eng = self._engine
        xd = eng._dev(x, torch.float32)
        out = torch.empty((b * s, self.num_actions), dtype=torch.float32, device=eng.device)
        eng._check(self._lib.pa_backbone_windows(eng._h, _ptr(xd), b * s, _ptr(self._feats), eng._stream()))
        rc = self._lib.pa_lstm_forward(self._h, _ptr(self._feats), _lib.PA_FEATURE_STRIDE, b, s, _ptr(out), eng._stream())
        if rc != 0:
            raise EngineError(rc, self._lib.pa_lstm_last_error(self._h).decode())
        if x.is_cuda:
            return out  # (enqueued only: should the per-layer kernel's barrier ever give up, these rows are NaN; check() says why)
        res = out.cpu()  # synchronises
        self.check()
        return res

    def check(self):
        """After the stream of a ``forward`` has been synchronised: raises if the per-layer LSTM kernel's grid barrier
        timed out in that call (``pa_lstm_last_status``)."""
        rc = self._lib.pa_lstm_last_status(self._h)
        if rc != 0:
            raise EngineError(rc, self._lib.pa_lstm_last_error(self._h).decode())

    __call__ = forward
