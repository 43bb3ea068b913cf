"""Weight hand-over: Lightning ``state_dict`` -> the flat fp32 blob that
``pa_create`` consumes (layout documented in include/playaid_hip.h).

The blob is a straight concatenation of the reference's own tensors in
state-dict order; BatchNorm folding and kernel-specific re-layout happen in
C++ inside the library. Key layout: SURVEY.md section 8b
(``model.cnn2d.*`` torchvision resnet18, ``model.cnn1d.0.*``,
``model.classifier.{0,2}.*``; ``playaid/models/cnn_action_detector.py:14-27``).
"""
from __future__ import annotations

from typing import Dict, Mapping

import numpy as np

from . import _lib
from .synth import resnet18_param_shapes


def _as_np(v) -> np.ndarray:
    if isinstance(v, np.ndarray):
        return v
    return v.detach().cpu().numpy()  # torch tensor


def blob_key_order(sequence_length: int, num_actions: int):
    keys = []
    for key, shape in resnet18_param_shapes():
        keys.append(("model.cnn2d." + key, shape))
    keys.append(("model.cnn1d.0.weight", (512, 1000, sequence_length)))
    keys.append(("model.cnn1d.0.bias", (512,)))
    keys.append(("model.classifier.0.weight", (128, 512)))
    keys.append(("model.classifier.0.bias", (128,)))
    keys.append(("model.classifier.2.weight", (num_actions, 128)))
    keys.append(("model.classifier.2.bias", (num_actions,)))
    return keys


def pack_state_dict(state_dict: Mapping, sequence_length: int, num_actions: int) -> np.ndarray:
    """-> uint8 blob. Raises KeyError / ValueError on a state dict that is not
    a ``CNNActionDetector`` (the reference would fail in ``load_state_dict``)."""
    parts = [np.array([_lib.PA_WEIGHT_MAGIC, 1, sequence_length, num_actions, 0, 0, 0, 0], dtype=np.int32).view(np.uint8)]
    for key, shape in blob_key_order(sequence_length, num_actions):
        if key not in state_dict:
            raise KeyError(f"state_dict is missing {key}")
        a = np.ascontiguousarray(_as_np(state_dict[key]), dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{key}: expected shape {tuple(shape)}, got {tuple(a.shape)}")
        parts.append(a.reshape(-1).view(np.uint8))
    return np.concatenate(parts)


def infer_geometry(state_dict: Mapping):
    """(sequence_length, num_actions) read off the head tensors."""
    s = int(_as_np(state_dict["model.cnn1d.0.weight"]).shape[2])
    a = int(_as_np(state_dict["model.classifier.2.weight"]).shape[0])
    return s, a
