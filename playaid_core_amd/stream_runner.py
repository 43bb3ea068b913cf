"""Mixed-resolution streams (BASELINE.json configs[4]): 1080p and 720p frames
interleaved in one clip, dynamic batch bucketing, hipGraph-captured steady state.

Every frame becomes two 128x128 crops whatever its resolution, so the feature
cache and the temporal head are resolution-agnostic; only the crop stage cares.
Frames are bucketed by (H, W); each bucket is fed through fixed-size batches
whose "crop + backbone + scatter-into-cache" launch sequence is captured ONCE
per (bucket shape, staging slot) into a hipGraph (torch.cuda.CUDAGraph around the
C-ABI call ``pa_backbone_frames_indexed``) and replayed with new buffer
contents. Each bucket has two staging slots: while the graph of one slot runs,
the next batch is copied into the other (pinned host staging, asynchronous
copies), so the host never waits for the device inside the clip; per-crop status
words and crops are read back once at the end. A short last batch is padded by
repeating its last frame (the scatter is idempotent). The head runs once the
whole clip is cached.

The reference has no counterpart (it reads one video file of one resolution,
``playaid/ai_runner.py:153``); the results equal running its per-frame crop at
each frame's own resolution followed by the same windows.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch


class _Slot:
    def __init__(self, engine, shape: Tuple[int, int], batch: int):
        h, w = shape
        dev = engine.device
        self.frames = torch.zeros((batch, h, w, 3), dtype=torch.uint8, device=dev)
        self.boxes = torch.zeros((batch, engine.F, 4), dtype=torch.float64, device=dev)
        self.ids = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.crops = torch.zeros((batch, engine.F, 128, 128, 3), dtype=torch.uint8, device=dev)
        self.status = torch.zeros((batch, engine.F), dtype=torch.int32, device=dev)
        # pinned host staging so that the uploads are asynchronous
        self.h_frames = torch.zeros((batch, h, w, 3), dtype=torch.uint8).pin_memory()
        self.h_boxes = torch.zeros((batch, engine.F, 4), dtype=torch.float64).pin_memory()
        self.h_ids = torch.zeros((batch,), dtype=torch.int32).pin_memory()
        self.graph = None
        self.done = torch.cuda.Event()  # recorded after the slot's last replay + read-back copies
        self.busy = False


class _Bucket:
    def __init__(self, engine, shape, batch):
        self.slots = [_Slot(engine, shape, batch), _Slot(engine, shape, batch)]
        self.next = 0


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

    def _step(self, b: _Slot):
        self.engine.backbone_frames_indexed(b.frames, b.boxes, b.ids, b.crops, b.status)

    def _launch(self, b: _Slot):
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
        status_dev = torch.zeros((n, eng.F), dtype=torch.int32, device=eng.device)
        crops_dev = torch.zeros((n, eng.F, 128, 128, 3), dtype=torch.uint8, device=eng.device) if want_crops else None
        for shape, idx in order.items():
            bk = self.buckets.get(shape)
            if bk is None:
                bk = self.buckets[shape] = _Bucket(eng, shape, self.batch)
            for k in range(0, len(idx), self.batch):
                chunk = idx[k : k + self.batch]
                padded = chunk + [chunk[-1]] * (self.batch - len(chunk))
                # ids are validated on the host BEFORE anything reaches the device (PA_ERR_CAPACITY
                # for an id outside the clip); the kernel's own range check is the second line
                eng.clip_mark_ready(chunk)
                b = bk.slots[bk.next]
                bk.next ^= 1
                if b.busy:
                    b.done.synchronize()  # only when this slot's previous batch is still in flight
                for j, i in enumerate(padded):
                    b.h_frames[j].copy_(torch.from_numpy(frames[i]))
                b.h_boxes.copy_(torch.from_numpy(boxes[padded]))
                b.h_ids.copy_(torch.tensor(padded, dtype=torch.int32))
                b.frames.copy_(b.h_frames, non_blocking=True)
                b.boxes.copy_(b.h_boxes, non_blocking=True)
                b.ids.copy_(b.h_ids, non_blocking=True)
                self._launch(b)
                sel = torch.tensor(chunk, dtype=torch.int64, device=eng.device)
                status_dev.index_copy_(0, sel, b.status[: len(chunk)])
                if want_crops:
                    crops_dev.index_copy_(0, sel, b.crops[: len(chunk)])
                b.done.record(torch.cuda.current_stream(eng.device))
                b.busy = True
        records = eng.alloc_records(n - 1)
        logp = eng.alloc_logp(n - 1)
        eng.head_frames(1, n, records, logp)
        eng.check_device_errors()  # synchronises; raises if the device skipped a frame id
        for bk in self.buckets.values():
            for b in bk.slots:
                b.busy = False
        out = eng.decode_records(records)
        out["logp"] = logp.cpu().numpy()
        out["crop_status"] = status_dev.cpu().numpy()
        if want_crops:
            out["crops_rgb"] = crops_dev.cpu().numpy()
        return out

    def run_resident(self, buckets: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]], n: int,
                     records: torch.Tensor, logp: torch.Tensor):
        """Steady-state form for throughput measurement: the clip's frames are already in HBM,
        grouped per resolution as ``{(H, W): (frames[m,H,W,3], boxes[m,F,4], ids int32[m])}`` with
        ``m`` a multiple of the batch size; every batch is one device copy into the graph's static
        buffers and one graph replay. Nothing synchronises; results land in ``records`` / ``logp``."""
        eng = self.engine
        eng.clip_begin(n)
        for shape, (fr, bx, ids) in buckets.items():
            bk = self.buckets.get(shape)
            if bk is None:
                bk = self.buckets[shape] = _Bucket(eng, shape, self.batch)
            assert fr.shape[0] % self.batch == 0
            for k in range(0, fr.shape[0], self.batch):
                b = bk.slots[bk.next]
                bk.next ^= 1
                b.frames.copy_(fr[k : k + self.batch])
                b.boxes.copy_(bx[k : k + self.batch])
                b.ids.copy_(ids[k : k + self.batch])
                self._launch(b)
        eng.clip_mark_ready(list(range(n)))
        eng.head_frames(1, n, records, logp)
