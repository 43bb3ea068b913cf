"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of ``RNNActionDetector.forward`` (``playaid/models/rnn_action_detector.py:74-95``): the resnet18
of ``oracle/cnn.py`` with ``fc = Linear(512, 300)`` (``:55-56``), then the LIVE ``torch.nn.LSTM(300, 512,
num_layers=3)`` CPU kernels loaded with the state dict's tensors (``:63``), the decoder (``:65-67``) and
``log_softmax(dim=1)``. As in the reference the LSTM gets ``[B, S, 300]`` without ``batch_first`` (``:88-90``):
time runs over B, the S frames are the batch. ``lstm_literal`` restates one layer-by-layer recurrence in numpy
(torch's documented gate equations, order i, f, g, o) to check that reading of the live module.

Pinning: "parity unpinned" -- the reference holds no vectors or checkpoint for this model.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .cnn import _t, resnet18_features


def _lstm_module(sd: Dict, dtype) -> torch.nn.LSTM:
    m = torch.nn.LSTM(input_size=300, hidden_size=512, num_layers=3).to(dtype)
    with torch.no_grad():
        for layer in range(3):
            for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                getattr(m, f"{name}_l{layer}").copy_(_t(sd, f"lstm.{name}_l{layer}", dtype))
    return m.eval()


def features(x: torch.Tensor, sd: Dict) -> torch.Tensor:
    """x[N,3,128,128] -> [N,300]: resnet18 with the replaced fc."""
    view = dict(sd)
    view["resnet.fc.weight"] = sd["resnet.fc.0.weight"]
    view["resnet.fc.bias"] = sd["resnet.fc.0.bias"]
    return resnet18_features(x, view, prefix="resnet.")


def head(feats: torch.Tensor, sd: Dict) -> torch.Tensor:
    """feats[B,S,300] -> logp[B*S,A] (``:88-95``)."""
    dt = feats.dtype
    b, s, _ = feats.shape
    y, _ = _lstm_module(sd, dt)(feats, None)
    y = y.reshape(b * s, -1)
    y = F.relu(F.linear(y, _t(sd, "action_decoder.0.weight", dt), _t(sd, "action_decoder.0.bias", dt)))
    y = F.linear(y, _t(sd, "action_decoder.2.weight", dt), _t(sd, "action_decoder.2.bias", dt))
    return F.log_softmax(y, dim=1)


def forward(x: torch.Tensor, sd: Dict) -> torch.Tensor:
    b, s, c, h, w = x.shape
    with torch.no_grad():
        return head(features(x.reshape(b * s, c, h, w), sd).view(b, s, -1), sd)


def lstm_literal(feats: np.ndarray, sd: Dict) -> np.ndarray:
    """feats float64[L,N,300] -> top-layer outputs float64[L,N,512]; plain loops over time and layers."""
    def sig(v):
        return 1.0 / (1.0 + np.exp(-v))

    x = np.asarray(feats, dtype=np.float64)
    for layer in range(3):
        w_ih = np.asarray(sd[f"lstm.weight_ih_l{layer}"], dtype=np.float64)
        w_hh = np.asarray(sd[f"lstm.weight_hh_l{layer}"], dtype=np.float64)
        b_ih = np.asarray(sd[f"lstm.bias_ih_l{layer}"], dtype=np.float64)
        b_hh = np.asarray(sd[f"lstm.bias_hh_l{layer}"], dtype=np.float64)
        h = np.zeros((x.shape[1], 512))
        c = np.zeros((x.shape[1], 512))
        ys = []
        for t in range(x.shape[0]):
            g = x[t] @ w_ih.T + b_ih + h @ w_hh.T + b_hh
            i, f, gg, o = sig(g[:, :512]), sig(g[:, 512:1024]), np.tanh(g[:, 1024:1536]), sig(g[:, 1536:])
            c = f * c + i * gg
            h = o * np.tanh(c)
            ys.append(h)
        x = np.stack(ys)
    return x
