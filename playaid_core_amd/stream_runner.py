"""Mixed-resolution streams (BASELINE.json configs[4]): 1080p and 720p frames
interleaved in one clip, dynamic batch bucketing, hipGraph-captured steady state.

Every frame becomes two 128x128 crops whatever its resolution, so the feature
cache and the temporal head are resolution-agnostic; only the crop stage cares.
Frames are bucketed by (H, W); each bucket is fed through fixed-size batches
whose "crop + backbone + scatter-into-cache" launch sequence is captured ONCE
per bucket shape into a hipGraph (torch.cuda.CUDAGraph around the C-ABI call
``pa_backbone_frames_indexed``) and replayed with new buffer contents. A short
last batch is padded by repeating its last frame (the scatter is idempotent).
The head runs once the whole clip is cached.

The reference has no counterpart (it reads one video file of one resolution,
``playaid/ai_runner.py:153``); the results equal running its per-frame crop at
each frame's own resolution followed by the same windows.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch


class _Bucket:
    def __init__(self, engine, shape: Tuple[int, int], batch: int):
        h, w = shape
        dev = engine.device
        self.frames = torch.zeros((batch, h, w, 3), dtype=torch.uint8, device=dev)
        self.boxes = torch.zeros((batch, engine.F, 4), dtype=torch.float64, device=dev)
        self.ids = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.crops = torch.zeros((batch, engine.F, 128, 128, 3), dtype=torch.uint8, device=dev)
        self.status = torch.zeros((batch, engine.F), dtype=torch.int32, device=dev)
        self.graph = None


class MixedResolutionRunner:
    def __init__(self, engine, batch_frames: int = 16, use_graphs: bool = True):
        if batch_frames > engine.max_batch_frames:
            raise ValueError("batch_frames exceeds the engine's max_batch_frames")
        self.engine = engine
        self.batch = batch_frames
        self.use_graphs = use_graphs
        self.buckets: Dict[Tuple[int, int], _Bucket] = {}
        self.replays = 0
        self.captures = 0

    def _step(self, b: _Bucket):
        self.engine.backbone_frames_indexed(b.frames, b.boxes, b.ids, b.crops, b.status)

    def _run_bucket_batch(self, b: _Bucket):
        if not self.use_graphs:
            self._step(b)
            return
        if b.graph is None:
            # warm-up on a side stream (first-call lazy work must not be captured), then capture
            s = torch.cuda.Stream(self.engine.device)
            s.wait_stream(torch.cuda.current_stream(self.engine.device))
            with torch.cuda.stream(s):
                self._step(b)
            torch.cuda.current_stream(self.engine.device).wait_stream(s)
            b.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(b.graph):
                self._step(b)
            self.captures += 1
        b.graph.replay()
        self.replays += 1

    def run(self, frames: Sequence[np.ndarray], boxes: np.ndarray, want_crops: bool = False) -> Dict[str, np.ndarray]:
        """frames: list of uint8[H_i, W_i, 3] BGR (any mix of resolutions <= the engine's
        maximum); boxes: float64[N, F, 4]. Returns the same dict as ``Engine.infer_clip``."""
        eng = self.engine
        n = len(frames)
        assert boxes.shape[0] == n
        eng.clip_begin(n)
        order: Dict[Tuple[int, int], List[int]] = {}
        for i, f in enumerate(frames):
            order.setdefault((f.shape[0], f.shape[1]), []).append(i)
        crops_out = np.zeros((n, eng.F, 128, 128, 3), np.uint8) if want_crops else None
        status_out = np.zeros((n, eng.F), np.int32)
        for shape, idx in order.items():
            b = self.buckets.get(shape)
            if b is None:
                b = self.buckets[shape] = _Bucket(eng, shape, self.batch)
            for k in range(0, len(idx), self.batch):
                chunk = idx[k : k + self.batch]
                padded = chunk + [chunk[-1]] * (self.batch - len(chunk))
                b.frames.copy_(torch.from_numpy(np.stack([frames[i] for i in padded])))
                b.boxes.copy_(torch.from_numpy(boxes[padded]))
                b.ids.copy_(torch.tensor(padded, dtype=torch.int32))
                self._run_bucket_batch(b)
                eng.clip_mark_ready(chunk)
                st = b.status.cpu().numpy()  # synchronises; also fences the buffers before their reuse
                status_out[chunk] = st[: len(chunk)]
                if want_crops:
                    crops_out[chunk] = b.crops.cpu().numpy()[: len(chunk)]
        records = eng.alloc_records(n - 1)
        logp = eng.alloc_logp(n - 1)
        eng.head_frames(1, n, records, logp)
        torch.cuda.synchronize(eng.device)
        out = eng.decode_records(records)
        out["logp"] = logp.cpu().numpy()
        out["crop_status"] = status_out
        if want_crops:
            out["crops_rgb"] = crops_out
        return out
