"""Frame-parallel sharding (playaid_core_amd/parallel.py) on CPU: world_size 2
and 3 over gloo, driven by an oracle-backed stand-in for the HIP engine. Checks
the shard arithmetic, the halo exchange and the result gather against a
single-process run of the same clip."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from playaid_core_amd import parallel, synth


def test_shard_and_halo_plans_cover_every_window():
    for n_total, world in [(60, 2), (61, 3), (128, 8), (40, 4), (30, 8), (70, 2), (8192, 8), (300, 3), (56, 1)]:
        owned = []
        for r in range(world):
            lo, hi = parallel.owned_frame_nums(n_total, world, r)
            owned += list(range(lo, hi))
            need_lo, need_hi = parallel.needed_range(n_total, world, r, 27)
            have = set(range(*parallel.shard_range(n_total, world, r)))
            for peer, f0, cnt in parallel.halo_plan(n_total, world, r, 27)[0]:
                have |= set(range(f0, f0 + cnt))
            assert set(range(need_lo, need_hi)) <= have
        assert owned == list(range(1, n_total))  # range(1, max_frames), ai_runner.py:508
        # sends and receives pair up
        sends = {(r, p, f0, c) for r in range(world) for (p, f0, c) in parallel.halo_plan(n_total, world, r, 27)[1]}
        recvs = {(p, r, f0, c) for r in range(world) for (p, f0, c) in parallel.halo_plan(n_total, world, r, 27)[0]}
        assert sends == recvs
        # interior frames: exactly those whose window stays inside the rank's own shard
        for r in range(world):
            own = set(range(*parallel.shard_range(n_total, world, r)))
            f_lo, f_hi = parallel.owned_frame_nums(n_total, world, r)
            brute = [f for f in range(f_lo, f_hi)
                     if set(range(max(1, f - 27) - 1, min(n_total - 1, f + 27))) <= own]
            i_lo, i_hi = parallel.interior_frame_nums(n_total, world, r, 27)
            assert list(range(i_lo, i_hi)) == brute, (n_total, world, r)


class OracleEngine:
    """CPU stand-in with the engine's clip interface (tests only)."""

    def __init__(self, sd, S=7, delta=3, F=2, A=63, max_batch_frames=16):
        from oracle import cnn  # noqa: F401

        self.sd, self.S, self.delta, self.F, self.A = sd, S, delta, F, A
        self.max_batch_frames = max_batch_frames

    def clip_begin(self, n):
        self.n = n
        self.cache = torch.zeros((n, self.F, 1024))
        self.ready = np.zeros(n, bool)

    def backbone_frames(self, frames, boxes, frame0):
        from oracle import cnn, pipeline

        crops, ok = pipeline.crops_for_clip(frames.numpy(), boxes.numpy())
        assert ok.all()
        x = torch.from_numpy(crops.reshape(-1, 128, 128, 3)).permute(0, 3, 1, 2).float() / 255.0
        with torch.no_grad():
            f = cnn.resnet18_features(x, self.sd)
        n = frames.shape[0]
        self.cache[frame0 : frame0 + n, :, :1000] = f.view(n, self.F, 1000)
        self.ready[frame0 : frame0 + n] = True

    def features_export(self, f0, n):
        assert self.ready[f0 : f0 + n].all()
        return self.cache[f0 : f0 + n].clone()

    def features_buffer(self, n):
        return torch.empty((n, self.F, 1024))

    def features_import(self, f0, t):
        self.cache[f0 : f0 + t.shape[0]] = t
        self.ready[f0 : f0 + t.shape[0]] = True

    def alloc_records(self, c):
        return torch.zeros((c, self.F, 4), dtype=torch.int32)

    def alloc_logp(self, c):
        return torch.zeros((c, self.F, self.A))

    def head_frames(self, lo, hi, records, logp):
        import torch.nn.functional as Fn

        from oracle import cnn, window

        for k, f in enumerate(range(lo, hi)):
            idx = window.action_sample_from_frame_middle_out(f, self.S, self.delta, self.n, min_frame=1)
            for p in range(self.F):
                assert all(self.ready[j - 1] for j in idx), (f, idx)
                wf = torch.stack([self.cache[j - 1, p, :1000] for j in idx])[None]
                with torch.no_grad():
                    lp = Fn.log_softmax(cnn.head_logits(wf, self.sd), dim=1)[0]
                logp[k, p] = lp
                records[k, p, 1] = int(lp.argmax())


def _worker(rank, world, port, n_total, h, w, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = synth.make_state_dict(1234)
    lo, hi = parallel.shard_range(n_total, world, rank)
    frames = torch.from_numpy(synth.make_frames(hi - lo, h, w, first_frame=lo))
    boxes = torch.from_numpy(synth.make_boxes(hi - lo, h, w, first_frame=lo))
    runner = parallel.FrameParallelClip(OracleEngine(sd), 7, 3)
    rec, lp = runner.run(frames, boxes, n_total, gather=True)
    if rank == 0:
        np.savez(out_path, action=rec[..., 1].numpy(), logp=lp.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_total", [(2, 70), (3, 34)])
def test_frame_parallel_equals_single_process(world, n_total, tmp_path, state_dict):
    """(2, 70): each rank has interior frames (head under the halo exchange) and edge frames;
    (3, 34): shards shorter than the 27-frame reach, multi-peer halos, no interior at all."""
    from oracle import pipeline

    h, w = 720, 1280
    out = str(tmp_path / "par.npz")
    mp.spawn(_worker, args=(world, _free_port(), n_total, h, w, out), nprocs=world, join=True)
    got = np.load(out)
    ref = pipeline.run_action_recognition(synth.make_frames(n_total, h, w), synth.make_boxes(n_total, h, w), state_dict, mode="cached")
    assert got["logp"].shape == (n_total - 1, 2, 63)
    assert np.abs(got["logp"] - ref["logp"]).max() < 1e-5
    assert np.array_equal(got["action"], ref["action_id"])
