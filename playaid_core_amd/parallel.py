"""Frame-parallel sharding of one clip over the GPUs of a node (SURVEY.md
section 8e). The reference has no multi-process inference at all; this is the
MI355X-side design:

* one process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI);
* the clip's frames are split into contiguous ranges, one per rank; every
  rank crops + runs the backbone on its own frames only (eval-mode BatchNorm
  makes crops independent, ``playaid/ai_runner.py:168``);
* the only data-path exchange is the *halo*: a window reaches
  ``delta * (S//2)**2`` = 27 frames past a shard edge
  (``playaid/dataset_utils.py:122-136``), so each rank receives the cached
  1000-d feature rows (4 KB each) of those frames from the neighbouring
  rank(s) -- point-to-point send/recv, 216 KB per edge;
* result records are gathered on rank 0.

The weight blob is broadcast once at start-up (``broadcast_blob``).
The class only needs the engine's clip interface, so the CPU tests drive it
with an oracle-backed stand-in over gloo.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous 0-based frame range [lo, hi) owned by ``rank``."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def owned_frame_nums(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """1-based frame numbers [lo, hi) this rank labels: its frames, within
    ``range(1, max_frames)`` (``ai_runner.py:508``). Frame number f is frame
    index f-1."""
    lo, hi = shard_range(n_total, world, rank)
    return max(lo + 1, 1), min(hi + 1, n_total)


def needed_range(n_total: int, world: int, rank: int, reach: int) -> Tuple[int, int]:
    """0-based frame indices [lo, hi) whose features the rank's windows read."""
    f_lo, f_hi = owned_frame_nums(n_total, world, rank)
    if f_hi <= f_lo:
        return 0, 0
    first = max(1, f_lo - reach)
    last = min(n_total - 1, f_hi - 1 + reach)
    return first - 1, last


def halo_plan(n_total: int, world: int, rank: int, reach: int):
    """-> (recvs, sends): lists of (peer, frame0, count). Deterministic on
    every rank, so sends and receives pair up without negotiation."""
    recvs, sends = [], []
    my_lo, my_hi = shard_range(n_total, world, rank)
    need_lo, need_hi = needed_range(n_total, world, rank, reach)
    for peer in range(world):
        if peer == rank:
            continue
        p_lo, p_hi = shard_range(n_total, world, peer)
        a, b = max(need_lo, p_lo), min(need_hi, p_hi)
        if b > a:
            recvs.append((peer, a, b - a))
        pn_lo, pn_hi = needed_range(n_total, world, peer, reach)
        a, b = max(pn_lo, my_lo), min(pn_hi, my_hi)
        if b > a:
            sends.append((peer, a, b - a))
    return recvs, sends


def _host_staged(group=None) -> bool:
    """True when the process group cannot move device tensors (gloo): collectives then go
    through host copies. Only used to rehearse the multi-rank path on boxes with fewer GPUs
    than ranks; the product backend is "nccl" (RCCL), which takes device tensors directly."""
    return dist.get_backend(group) != "nccl"


def broadcast_blob(blob: Optional[np.ndarray], nbytes: int, device: torch.device, group=None) -> np.ndarray:
    """One broadcast of the weight blob (61.4 MB fp32) from rank 0."""
    t = torch.empty(nbytes, dtype=torch.uint8, device="cpu" if _host_staged(group) else device)
    if dist.get_rank(group) == 0:
        t.copy_(torch.from_numpy(blob))
    dist.broadcast(t, src=0, group=group)
    return t.cpu().numpy()


class FrameParallelClip:
    def __init__(self, engine, sequence_length: int, frame_delta: int, group=None):
        self.engine = engine
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.reach = abs(frame_delta) * (sequence_length // 2) ** 2
        # two-stream software pipeline (see backbone_shard)
        self._pre_stream = None
        self._pre_done = None
        self._slot_free = None
        self._slot_used = [False, False]
        self._slot = 0
        self._out = None
        self._pack = None

    def backbone_shard(self, frames_local, boxes_local, lo: int, pipeline: bool = False):
        """Crop + backbone for this rank's frames, in chunks of the engine's batch size.

        ``pipeline=True`` (HIP engine only) runs the crop stage on a second stream into
        alternating model-input slots, so the VALU/LDS-bound crop kernels of chunk k+1 --
        or of the next call -- overlap the MFMA-bound backbone of chunk k. The caller
        guarantees that ``frames_local`` / ``boxes_local`` are complete in HBM before the
        call (the crop stream does not wait for the caller's stream)."""
        eng = self.engine
        step = eng.max_batch_frames
        n = frames_local.shape[0]
        if not (pipeline and getattr(eng, "supports_pipelining", False)):
            for f0 in range(0, n, step):
                eng.backbone_frames(frames_local[f0 : f0 + step], boxes_local[f0 : f0 + step], lo + f0)
            return
        main = torch.cuda.current_stream(eng.device)
        if self._pre_stream is None:
            self._pre_stream = torch.cuda.Stream(eng.device)
            self._pre_done = [torch.cuda.Event(), torch.cuda.Event()]
            self._slot_free = [torch.cuda.Event(), torch.cuda.Event()]
        pre = self._pre_stream
        for f0 in range(0, n, step):
            cnt = min(step, n - f0)
            slot = self._slot
            self._slot ^= 1
            with torch.cuda.stream(pre):
                if self._slot_used[slot]:
                    pre.wait_event(self._slot_free[slot])  # the backbone that read this slot is done
                eng.preprocess_frames(frames_local[f0 : f0 + cnt], boxes_local[f0 : f0 + cnt], slot)
                self._pre_done[slot].record(pre)
            main.wait_event(self._pre_done[slot])
            eng.backbone_slot(slot, cnt, lo + f0)
            self._slot_free[slot].record(main)
            self._slot_used[slot] = True

    def exchange_halo(self, n_total: int):
        if self.world == 1:
            return
        recvs, sends = halo_plan(n_total, self.world, self.rank, self.reach)
        staged = _host_staged(self.group)
        ops, bufs = [], []
        for peer, f0, cnt in sends:
            t = self.engine.features_export(f0, cnt)
            ops.append(dist.P2POp(dist.isend, t.cpu() if staged else t, peer, self.group))
        for peer, f0, cnt in recvs:
            t = self.engine.features_buffer(cnt)
            if staged:
                t = torch.empty(t.shape, dtype=t.dtype)
            bufs.append((f0, t))
            ops.append(dist.P2POp(dist.irecv, t, peer, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for f0, t in bufs:
            self.engine.features_import(f0, t)

    def run(self, frames_local, boxes_local, n_total: int, gather: bool = True, pipeline: bool = False):
        """frames_local / boxes_local: this rank's shard (device tensors for the
        HIP engine). Returns on rank 0 (or every rank when ``gather`` is False:
        the local part) ``(records int32[count,F,4], logp float32[count,F,A])``
        for frame numbers 1..n_total-1 in order."""
        eng = self.engine
        lo, hi = shard_range(n_total, self.world, self.rank)
        assert frames_local.shape[0] == hi - lo, "shard size mismatch"
        eng.clip_begin(n_total)
        self.backbone_shard(frames_local, boxes_local, lo, pipeline=pipeline)
        self.exchange_halo(n_total)
        f_lo, f_hi = owned_frame_nums(n_total, self.world, self.rank)
        count = max(f_hi - f_lo, 0)
        # result buffers are kept between calls (a fresh torch.zeros costs a fill kernel each)
        if self._out is None or self._out[0].shape[0] < max(count, 1):
            self._out = (eng.alloc_records(max(count, 1)), eng.alloc_logp(max(count, 1)))
        records, logp = self._out
        if count > 0:
            eng.head_frames(f_lo, f_hi, records, logp)
        if not gather or self.world == 1:
            return records[:count], logp[:count]
        # ONE equal-size all_gather of padded shards: records (as int32 bit patterns) and
        # log-probs travel in the same float32 buffer [cap, F, 4 + A]; trimmed on the way out
        counts = [
            max(owned_frame_nums(n_total, self.world, r)[1] - owned_frame_nums(n_total, self.world, r)[0], 0)
            for r in range(self.world)
        ]
        cap = max(max(counts), 1)
        A = logp.shape[-1]
        staged = _host_staged(self.group)
        if self._pack is None or self._pack[0].shape[0] != cap or self._pack[1].shape[0] != self.world:
            mine = torch.zeros((cap, eng.F, 4 + A), dtype=torch.float32, device=records.device)
            everyone = torch.zeros((self.world, cap, eng.F, 4 + A), dtype=torch.float32, device="cpu" if staged else records.device)
            self._pack = (mine, everyone)
        mine, everyone = self._pack
        mine[:count, :, :4].view(torch.int32).copy_(records[:count])
        mine[:count, :, 4:].copy_(logp[:count])
        if staged:
            parts = [torch.empty((cap, eng.F, 4 + A), dtype=torch.float32) for _ in range(self.world)]
            dist.all_gather(parts, mine.cpu(), group=self.group)
            everyone = torch.stack(parts)
        else:
            try:
                dist.all_gather_into_tensor(everyone, mine, group=self.group)
            except (RuntimeError, NotImplementedError, AttributeError):
                # a backend without the single-tensor form: the list form moves the same bytes
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=self.group)
                everyone = torch.stack(parts)
        rec = torch.cat([everyone[r, : counts[r], :, :4] for r in range(self.world)]).view(torch.int32)
        lp = torch.cat([everyone[r, : counts[r], :, 4:] for r in range(self.world)])
        return rec, lp
