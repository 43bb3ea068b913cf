"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of ``ResnetTransformerDetector.forward`` (``playaid/models/resnet_transformer_detector.py:65-93,
136-141``). The backbone is a timm ``resnet50(num_classes=0)`` (``:37``): timm is not vendored in /root/reference and
not installed here, so its published architecture is restated with ``torch.nn.functional`` CPU ops -- bottleneck
blocks [3, 4, 6, 3], expansion 4, stride on the 3x3 convolution, eval-mode BatchNorm eps 1e-5, 3x3/2 max-pool,
global average pool, identical to torchvision's ``resnet50`` v1.5 graph. The head uses the LIVE
``torch.nn.TransformerEncoder`` CPU module (``:53-60``: ``TransformerEncoderLayer(d_model=256, nhead=8)`` defaults:
dim_feedforward 2048, ReLU, post-norm, NOT batch_first) loaded with the state dict's tensors, fed ``[B, S, 256]`` as the
reference does, so dimension 0 -- the windows -- is the sequence. ``encoder_literal`` restates the encoder in numpy
from torch's documented equations to check that reading.

Pinning: "parity unpinned" -- the reference holds no vectors or checkpoint for this model.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .cnn import _bn, _t

BLOCKS = (3, 4, 6, 3)


def resnet50_features(x: torch.Tensor, sd: Dict, prefix: str = "model.resnet.") -> torch.Tensor:
    """x[N,3,H,W] -> pooled features [N,2048]."""
    dt = x.dtype
    p = prefix
    x = F.conv2d(x, _t(sd, p + "conv1.weight", dt), None, stride=2, padding=3)
    x = F.relu(_bn(x, sd, p + "bn1", dt))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, blocks in enumerate(BLOCKS, start=1):
        for b in range(blocks):
            q = f"{p}layer{li}.{b}"
            stride = 2 if (b == 0 and li > 1) else 1
            identity = x
            out = F.relu(_bn(F.conv2d(x, _t(sd, q + ".conv1.weight", dt)), sd, q + ".bn1", dt))
            out = F.relu(_bn(F.conv2d(out, _t(sd, q + ".conv2.weight", dt), None, stride=stride, padding=1), sd, q + ".bn2", dt))
            out = _bn(F.conv2d(out, _t(sd, q + ".conv3.weight", dt)), sd, q + ".bn3", dt)
            if (q + ".downsample.0.weight") in sd:
                identity = _bn(F.conv2d(x, _t(sd, q + ".downsample.0.weight", dt), None, stride=stride), sd, q + ".downsample.1", dt)
            x = F.relu(out + identity)
    return F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)


def _encoder(sd: Dict, dtype) -> torch.nn.TransformerEncoder:
    layer = torch.nn.TransformerEncoderLayer(d_model=256, nhead=8)
    enc = torch.nn.TransformerEncoder(layer, num_layers=3, enable_nested_tensor=False).to(dtype)
    with torch.no_grad():
        for name, prm in enc.named_parameters():
            prm.copy_(_t(sd, "model.transformer." + name, dtype))
    return enc.eval()


def head(feats: torch.Tensor, sd: Dict) -> torch.Tensor:
    """pooled features [B,S,2048] -> log-probabilities [B,S,A] (``:74-93,141``)."""
    dt = feats.dtype
    b, s, _ = feats.shape
    y = F.linear(feats, _t(sd, "model.resnet_ffn.weight", dt), _t(sd, "model.resnet_ffn.bias", dt))
    enc = _t(sd, "model.freq_encoding", dt)
    y = torch.cat((y, enc.unsqueeze(0).expand(b, -1, -1)), dim=2)
    y = _encoder(sd, dt)(y)
    y = F.linear(y, _t(sd, "model.classifier.weight", dt), _t(sd, "model.classifier.bias", dt))
    return F.log_softmax(y, dim=2)


def forward(x: torch.Tensor, sd: Dict) -> torch.Tensor:
    b, s, c, h, w = x.shape
    with torch.no_grad():
        return head(resnet50_features(x.reshape(b * s, c, h, w), sd).view(b, s, -1), sd)


def encoder_literal(x: np.ndarray, sd: Dict) -> np.ndarray:
    """x float64[L,N,256] -> float64[L,N,256]: three post-norm layers, attention over L for each n and head."""
    x = np.asarray(x, dtype=np.float64)

    def ln(v, g, b):
        mu = v.mean(-1, keepdims=True)
        var = ((v - mu) ** 2).mean(-1, keepdims=True)
        return (v - mu) / np.sqrt(var + 1e-5) * g + b

    def prm(layer, name):
        return np.asarray(sd[f"model.transformer.layers.{layer}.{name}"], dtype=np.float64)

    L, N, D = x.shape
    H, hd = 8, D // 8
    for layer in range(3):
        qkv = x @ prm(layer, "self_attn.in_proj_weight").T + prm(layer, "self_attn.in_proj_bias")
        q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(L, N, H, hd) for i in range(3))
        att = np.einsum("inhd,jnhd->nhij", q, k) / np.sqrt(hd)
        att = np.exp(att - att.max(-1, keepdims=True))
        att = att / att.sum(-1, keepdims=True)
        o = np.einsum("nhij,jnhd->inhd", att, v).reshape(L, N, D)
        o = o @ prm(layer, "self_attn.out_proj.weight").T + prm(layer, "self_attn.out_proj.bias")
        x = ln(x + o, prm(layer, "norm1.weight"), prm(layer, "norm1.bias"))
        f = np.maximum(x @ prm(layer, "linear1.weight").T + prm(layer, "linear1.bias"), 0.0)
        f = f @ prm(layer, "linear2.weight").T + prm(layer, "linear2.bias")
        x = ln(x + f, prm(layer, "norm2.weight"), prm(layer, "norm2.bias"))
    return x
