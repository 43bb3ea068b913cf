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
  rank(s) -- point-to-point send/recv, 216 KB per edge. The exchange is posted
  as soon as the backbone of the shard is enqueued and the head of the
  *interior* frames (whose windows stay inside the shard) runs underneath it;
  only the <= 2 x 27 edge frames wait for the halo;
* result records are gathered with one equal-size all-gather.

The weights cross xGMI once at start-up, already folded and laid out
(``broadcast_engine``: rank 0 builds its engine, exports the device weight
arena, one RCCL broadcast, the other ranks adopt it with a device copy).
The class only needs the engine's clip interface, so the CPU tests drive it
with an oracle-backed stand-in over gloo.
"""
from __future__ import annotations

import time
from typing import List, Optional, Tuple

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


def interior_frame_nums(n_total: int, world: int, rank: int, reach: int) -> Tuple[int, int]:
    """Sub-range [lo, hi) of ``owned_frame_nums`` whose windows only touch frames this rank owns
    (sampler clamps included: ``max(1, f - reach)`` .. ``min(max_frames - 1, f + reach)``)."""
    f_lo, f_hi = owned_frame_nums(n_total, world, rank)
    own_lo, own_hi = shard_range(n_total, world, rank)  # cached frame numbers own_lo + 1 .. own_hi
    # a window of frame f reads frame numbers max(1, f - reach) .. min(n_total - 1, f + reach); both
    # bounds are monotone in f, so the interior is one run
    lo = f_lo if own_lo == 0 else max(f_lo, own_lo + 1 + reach)
    hi = f_hi if own_hi >= n_total - 1 else min(f_hi, own_hi - reach + 1)
    return (lo, hi) if hi > lo else (f_lo, f_lo)


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


_DTYPE_CODES = {"f32": 0, "bf16": 1, "emulated_f32": 2}   # (== _lib.DTYPES; kept here so that the gloo tests' CPU stand-in needs no library)


def _host_staged(group=None) -> bool:
    """True when the process group cannot move device tensors (gloo): collectives then go
    through host copies. Only used to rehearse the multi-rank path on boxes with fewer GPUs
    than ranks; the product backend is "nccl" (RCCL), which takes device tensors directly."""
    return dist.get_backend(group) != "nccl"


def broadcast_engine(make_engine, weights, device: torch.device, group=None):
    """Engines for every rank from weights only rank 0 holds.

    ``make_engine(w)`` builds an ``Engine`` on this rank's device from ``w``; rank 0 calls it with
    ``weights`` (state dict or packed blob), exports the prepared device arena and broadcasts it
    (61-64 MB fp32, once); the other ranks call it with the received ``WeightsArena``. With RCCL
    the arena never leaves HBM; a gloo rehearsal stages it through the host."""
    from .engine import WeightsArena

    rank = dist.get_rank(group)
    staged = _host_staged(group)
    meta = torch.zeros(4, dtype=torch.int64, device="cpu" if staged else device)
    eng = None
    if rank == 0:
        eng = make_engine(weights)
        arena = eng.weights_arena()
        meta = torch.tensor([arena.data.numel(), arena.sequence_length, arena.num_actions, _DTYPE_CODES[arena.compute_dtype]],
                            dtype=torch.int64, device=meta.device)
    dist.broadcast(meta, src=0, group=group)
    nbytes, S, A, dtype_code = (int(v) for v in meta.tolist())
    if rank == 0:
        buf = arena.data.cpu() if staged else arena.data
    else:
        buf = torch.empty(nbytes, dtype=torch.uint8, device="cpu" if staged else device)
    dist.broadcast(buf, src=0, group=group)
    if rank != 0:
        eng = make_engine(WeightsArena(buf, S, A, {v: k for k, v in _DTYPE_CODES.items()}[dtype_code]))
    return eng


class FrameParallelClip:
    def __init__(self, engine, sequence_length: int, frame_delta: int, group=None, collectives_at_world_one: bool = False):
        self.engine = engine
        self.group = group
        # a one-rank process group normally skips the record gather (nothing to collect); True runs it anyway, so that a
        # one-GPU box can execute the collective code path of the backend (tests: RCCL at world size 1)
        self.collectives_at_world_one = collectives_at_world_one
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.reach = abs(frame_delta) * (sequence_length // 2) ** 2
        # how results travel is fixed here, once, from what the backend can do -- never by catching a
        # failed collective (a rank that falls back alone would desynchronise the others)
        self._staged = self.distributed and _host_staged(group)
        self._gather_into_tensor = self.distributed and not self._staged and hasattr(dist, "all_gather_into_tensor")
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

    def post_halo(self, n_total: int):
        """Start the halo exchange; returns ``(requests, recv_buffers)`` for ``finish_halo``.
        With RCCL the transfers run on the communicator's own stream behind the backbone that
        produced the exported rows, so whatever the caller enqueues next overlaps them."""
        if self.world == 1:
            return [], []
        recvs, sends = halo_plan(n_total, self.world, self.rank, self.reach)
        ops, bufs = [], []
        # the halo buffers of a clip shape are allocated once and reused by every pass (the exchange of pass k has been
        # waited for in finish_halo before pass k + 1 exports into them again)
        cache = self.__dict__.setdefault("_halo_bufs", {})
        for peer, f0, cnt in sends:
            key = ("s", peer, f0, cnt)
            if key not in cache:
                cache[key] = self.engine.features_buffer(cnt)
            t = (self.engine.features_export(f0, cnt, out=cache[key]) if getattr(self.engine, "supports_pipelining", False)
                 else self.engine.features_export(f0, cnt))   # (the CPU stand-in of the gloo tests allocates its own)
            ops.append(dist.P2POp(dist.isend, t.cpu() if self._staged else t, peer, self.group))
        for peer, f0, cnt in recvs:
            key = ("r", peer, f0, cnt)
            if key not in cache:
                cache[key] = self.engine.features_buffer(cnt)
            t = cache[key]
            if self._staged:
                t = torch.empty(t.shape, dtype=t.dtype)
            bufs.append((f0, t))
            ops.append(dist.P2POp(dist.irecv, t, peer, self.group))
        return (dist.batch_isend_irecv(ops) if ops else []), bufs

    def finish_halo(self, pending):
        reqs, bufs = pending
        for req in reqs:
            req.wait()
        for f0, t in bufs:
            self.engine.features_import(f0, t)

    def exchange_halo(self, n_total: int):
        self.finish_halo(self.post_halo(n_total))

    def run(self, frames_local, boxes_local, n_total: int, gather: bool = True, pipeline: bool = False,
            reuse_buffers: bool = False, batch_of: int = 0):
        """frames_local / boxes_local: this rank's shard (device tensors for the HIP engine).
        Returns on every rank ``(records int32[count,F,4], logp float32[count,F,A])`` for frame
        numbers 1..n_total-1 in order (``gather=False`` or one rank: the local part only).

        ``batch_of`` = k: the frames are k independent clips of ``n_total // k`` frames (see ``Engine.clip_begin``);
        clip c's rows are ``[c * L : c * L + L - 1]`` of the result, row ``c * L + L - 1`` belongs to no clip.

        Result tensors: the local part lives in buffers this object keeps between calls. With
        ``reuse_buffers=False`` (default) the caller gets its own copies; ``reuse_buffers=True``
        (bench loops) returns views that the next ``run`` overwrites in place."""
        eng = self.engine
        lo, hi = shard_range(n_total, self.world, self.rank)
        assert frames_local.shape[0] == hi - lo, "shard size mismatch"
        if batch_of:
            # n_total frames = batch_of independent clips (one rank only: a clip is not cut across ranks); rows whose
            # frame number is a multiple of the clip length belong to no clip
            assert self.world == 1, "clip batches are a single-GPU throughput mode"
            eng.clip_begin(n_total, batch_of=batch_of)
        else:
            eng.clip_begin(n_total)
        self.backbone_shard(frames_local, boxes_local, lo, pipeline=pipeline)
        pending = self.post_halo(n_total)
        f_lo, f_hi = owned_frame_nums(n_total, self.world, self.rank)
        count = max(f_hi - f_lo, 0)
        # result buffers are kept between calls (a fresh torch.zeros costs a fill kernel each)
        if self._out is None or self._out[0].shape[0] < max(count, 1):
            self._out = (eng.alloc_records(max(count, 1)), eng.alloc_logp(max(count, 1)))
        records, logp = self._out

        def head(a, b):
            if b > a:
                eng.head_frames(a, b, records[a - f_lo :], logp[a - f_lo :])

        if self.world == 1:
            head(f_lo, f_hi)
        else:
            i_lo, i_hi = interior_frame_nums(n_total, self.world, self.rank, self.reach)
            head(i_lo, i_hi)             # under the exchange
            self.finish_halo(pending)
            head(f_lo, i_lo)             # the edge frames need the neighbours' rows
            head(i_hi, f_hi)
        if not gather or (self.world == 1 and not (self.collectives_at_world_one and self.distributed)):
            if reuse_buffers:
                return records[:count], logp[:count]
            return records[:count].clone(), logp[:count].clone()
        # ONE equal-size all_gather of padded shards: records (as int32 bit patterns) and
        # log-probs travel in the same float32 buffer [cap, F, 4 + A]; trimmed on the way out
        counts = [
            max(owned_frame_nums(n_total, self.world, r)[1] - owned_frame_nums(n_total, self.world, r)[0], 0)
            for r in range(self.world)
        ]
        cap = max(max(counts), 1)
        A = logp.shape[-1]
        if self._pack is None or self._pack[0].shape[0] != cap or self._pack[1].shape[0] != self.world:
            mine = torch.zeros((cap, eng.F, 4 + A), dtype=torch.float32, device=records.device)
            everyone = torch.zeros((self.world, cap, eng.F, 4 + A), dtype=torch.float32,
                                   device="cpu" if self._staged else records.device)
            self._pack = (mine, everyone)
        mine, everyone = self._pack
        mine[:count, :, :4].view(torch.int32).copy_(records[:count])
        mine[:count, :, 4:].copy_(logp[:count])
        if self._gather_into_tensor:
            dist.all_gather_into_tensor(everyone, mine, group=self.group)
        else:
            src = mine.cpu() if self._staged else mine
            parts = [torch.empty_like(src) for _ in range(self.world)]
            dist.all_gather(parts, src, group=self.group)
            everyone = torch.stack(parts)
        rec = torch.cat([everyone[r, : counts[r], :, :4] for r in range(self.world)]).view(torch.int32)
        lp = torch.cat([everyone[r, : counts[r], :, 4:] for r in range(self.world)])
        return rec, lp


def _concurrent_streams(engine, count: int, tries: int = 16, allow_sharing: bool = True):
    """``count`` HIP streams that really run side by side. The HIP runtime multiplexes streams onto a few hardware
    queues (four by default) and two streams that land on the same queue execute strictly in turn: measured on
    MI355X, torch's first and second pool streams shared one (two lanes then ran at the one-lane rate, 43.0 k
    frames/s against 47.9 k for any pair on distinct queues). So each candidate is probed against the streams already
    taken with two ~1 ms single-thread spin kernels: overlapping pairs finish in one kernel time, serialised ones
    in two. Candidates that fail are dropped (they stay in torch's pool)."""
    device = engine.device
    with torch.cuda.device(device):
        def elapsed(streams):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for st in streams:
                engine.stream_spin(1000, st)     # ~1 ms single-thread kernel (pa_stream_spin)
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        taken = []
        last = None
        for _ in range(tries):
            if len(taken) == count:
                break
            cand = last = torch.cuda.Stream()
            elapsed([cand])                      # first use binds the stream to its queue
            one = min(elapsed([cand]) for _ in range(2))
            if all(min(elapsed([cand, st]) for _ in range(2)) < 1.5 * one for st in taken):
                taken.append(cand)
        while allow_sharing and len(taken) < count:   # more lanes than queues: accept sharing
            taken.append(last if last is not None else torch.cuda.Stream())
            last = None
        return taken


class ClipLanes:
    """Throughput mode for a stream of INDEPENDENT clips on one GPU: ``lanes`` engines (clones of the first: own
    activation buffers and feature cache, shared nothing but the device) each with its own HIP stream; clips are
    dealt to the lanes in turn. One clip's launch ramps, prologues, epilogues and the tail of every kernel --
    about 8 us per convolution launch at 128 crops, where the second workgroup of each CU finishes alone --
    then run underneath another clip's steady state (measured on the headline shape: 43.6 k -> 48.0 k frames/s with
    two lanes, three lanes add nothing). The reference processes one window at a time and has no counterpart.

    ``submit`` only enqueues; results of a lane are valid once that lane's stream (or the device) is synchronised
    and are overwritten by the lane's next clip."""

    def __init__(self, engine, sequence_length: int, frame_delta: int, lanes: int = 2, gate_timeout_us: int = 20000):
        self.engines = [engine] + [engine.clone() for _ in range(max(lanes, 1) - 1)]
        self.runners = [FrameParallelClip(e, sequence_length, frame_delta) for e in self.engines]
        # a few more mutually concurrent streams than lanes: calibrate() picks among them
        self._candidates = _concurrent_streams(engine, max(len(self.engines), 4), allow_sharing=False)
        self.streams = _concurrent_streams(engine, len(self.engines)) if len(self._candidates) < len(self.engines) \
            else self._candidates[: len(self.engines)]
        self.calibration = None
        self._next = 0
        self.gate_timeout_us = gate_timeout_us
        self._gate_pending = 0
        self._helper = torch.cuda.Stream(engine.device)
        self._go = torch.cuda.Event()
        self._cold = True

    def calibrate(self, frames, boxes, n_total: int, clips: int = 12, batch_of: int = 0):
        """Pick the lanes' streams by measurement on the caller's own clip shape: every combination of the candidate
        streams (they passed the spin-kernel probe, i.e. sit on distinct hardware queues at that moment; the runtime
        may re-draw the mapping later) is timed over ``clips`` clips from an aligned start, the fastest is kept.
        A pair that lands on ONE queue after all shows up here at the one-lane rate (43-45 k frames/s against 47 k).
        -> {combination: frames/s}."""
        import itertools

        n = len(self.engines)
        if n < 2 or len(self._candidates) <= n:
            return {}
        rates = {}
        # warm-up: the first pair timed used to come out 6-8 % slow whichever pair it was (12 clips right after the engines were
        # made: clocks, first-touch of the clones' buffers) -- the "queue lottery" of round 5's records was this, the same streams run
        # the timed region at the calibrated rate (profiles/r06_hw_queues.txt)
        self.streams = self._candidates[:n]
        self._cold = True
        for k in range(24):
            self.submit(frames, boxes, n_total, batch_of)
        torch.cuda.synchronize(self.engines[0].device)
        for combo in itertools.combinations(range(len(self._candidates)), n):
            self.streams = [self._candidates[i] for i in combo]
            self._cold = True
            for k in range(2 * n):
                self.submit(frames, boxes, n_total, batch_of)
            torch.cuda.synchronize(self.engines[0].device)
            self._cold = True
            t0 = time.perf_counter()
            for k in range(clips):
                self.submit(frames, boxes, n_total, batch_of)
            torch.cuda.synchronize(self.engines[0].device)
            rates[combo] = n_total * clips / (time.perf_counter() - t0)
        best = max(rates, key=rates.get)
        self.streams = [self._candidates[i] for i in best]
        self._cold = True
        self.calibration = {"picked": list(best), "rates": {",".join(map(str, c)): round(v, 1) for c, v in rates.items()}}
        return rates

    def _aligned_start(self):
        """Hold every lane behind ONE gate while the caller enqueues the first clip of each lane, so that the lanes then
        START TOGETHER. The offset between the lanes decides the rate and, once running, stays what the start gave it:
        measured on the headline shape with the start offset set on the GPU, 47.7-47.9 k frames/s for offsets within
        +-0.3 ms of aligned (also one whole clip period later), 45.0-45.1 k for anything between 0.35 and 1.2 ms -- one
        lane's crop / stem kernels (80 KB of LDS per workgroup) then fall into the other lane's convolution layers for
        good. Left to the host, the offset is the time it takes to enqueue one clip (~0.3-0.4 ms): right on the edge,
        which made identical runs land on either rate.

        The gate is a one-thread kernel on a helper stream that polls a word of coherent pinned memory
        (``pa_stream_gate``); ``submit`` opens it as soon as every lane has its first clip enqueued -- no guess of how
        long the host will take (round 2 held the lanes behind a spin kernel of 1-4 ms sized from recent submit times).
        Should the caller stop submitting before that, ``synchronize`` / ``idle`` open it, and the kernel gives up by
        itself after ``gate_timeout_us``."""
        self.engines[0].stream_gate(self.gate_timeout_us, self._helper)
        self._go.record(self._helper)
        for st in self.streams:
            st.wait_event(self._go)
        self._gate_pending = len(self.engines)

    def _gate_progress(self, force: bool = False):
        if self._gate_pending > 0:
            self._gate_pending = 0 if force else self._gate_pending - 1
            if self._gate_pending == 0:
                self.engines[0].stream_gate_open()

    def idle(self):
        """Tell the lanes that the device has drained (the caller synchronised it): the next clips start aligned."""
        self._gate_progress(force=True)
        self._cold = True

    def submit(self, frames, boxes, n_total: int, batch_of: int = 0):
        """Enqueue one clip (``batch_of`` = k: k independent clips concatenated, see ``FrameParallelClip.run``) on the
        next lane -> (lane index, records view, logp view)."""
        if self._cold and len(self.engines) > 1:
            self._aligned_start()
        self._cold = False
        lane = self._next
        self._next = (self._next + 1) % len(self.engines)
        with torch.cuda.stream(self.streams[lane]):
            rec, lp = self.runners[lane].run(frames, boxes, n_total, gather=False, pipeline=False, reuse_buffers=True,
                                             batch_of=batch_of)
        self._gate_progress()  # the last lane's first clip is enqueued: open the gate
        return lane, rec, lp

    def synchronize(self):
        self._gate_progress(force=True)
        for st in self.streams:
            st.synchronize()
        self._cold = True

    def close(self):
        for e in self.engines[1:]:
            e.close()
