"""``CNNActionDetector`` -- the operator boundary (b1 in SURVEY.md section 8b).

Mirror of ``playaid/models/cnn_action_detector.py:46-92`` for inference:
``load_from_checkpoint(path, actions=[...])``, ``.eval()``, ``model(x)`` with
``x: float32[B,S,3,128,128]`` in [0,1] -> ``float32[B,len(actions)]``
log-probabilities, ``.actions``. The arithmetic (ResNet-18 per frame, Conv1d
over the S features, MLP, log_softmax) runs in the HIP library through
``pa_infer_windows``; there is no PyTorch fallback. Training hooks of the
reference (``:94-207``) are out of scope.
"""
from __future__ import annotations

from typing import List, Mapping, Optional

import torch

from .engine import Engine
from .weights import infer_geometry


class CNNActionDetector:
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
        **kwargs,
    ):
        if state_dict is None:
            raise ValueError(
                "CNNActionDetector needs weights: use load_from_checkpoint() or pass state_dict= "
                "(the reference's resnet18(pretrained=True) download is not available offline)"
            )
        s, a = infer_geometry(state_dict)
        if a != len(actions):
            raise ValueError(f"checkpoint has {a} action logits but {len(actions)} actions were given")
        self.actions = list(actions)
        self.num_actions = len(self.actions)
        self.sequence_length = s
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.num_samples = num_samples
        self.dataset_kwargs = kwargs
        self.training = False
        # (beyond the reference's arguments: the engine's capacities and, never by default, the arithmetic of the fp32 path --
        # compute_dtype="emulated_f32", PA_DTYPE_EMULATED_F32)
        self._engine = Engine(state_dict, device=device, **{k: v for k, v in kwargs.items() if k.startswith("max_") or k == "compute_dtype"})

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, map_location=None, **kwargs):
        """Lightning-1.6.5 ``.ckpt``: ``state_dict`` (keys ``model.cnn2d.*``,
        ``model.cnn1d.0.*``, ``model.classifier.{0,2}.*``) plus
        ``hyper_parameters``; keyword arguments override the saved ones like
        ``LightningModule.load_from_checkpoint`` (``ai_runner.py:164-167``)."""
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        if "state_dict" not in ckpt:
            raise KeyError(f"{checkpoint_path} has no 'state_dict' (not a Lightning checkpoint)")
        hparams = dict(ckpt.get("hyper_parameters", {}) or {})
        hparams.update(kwargs)
        if "actions" not in hparams:
            raise TypeError("load_from_checkpoint() missing 'actions' (ai_runner.py:166 passes list(MOVE_TO_CLASS_ID))")
        return cls(state_dict=ckpt["state_dict"], **hparams)

    # -- nn.Module-like surface -------------------------------------------------
    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("training is out of scope for the MI355X inference path")
        return self

    @property
    def engine(self) -> Engine:
        return self._engine

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [batch, frames_per_sequence, channel, height, width] -> log-probabilities."""
        out = self._engine.infer_windows(x)
        return out if x.is_cuda else out.cpu()

    __call__ = forward
