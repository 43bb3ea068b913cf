"""Probe: clips alternating over NE engines (each its own activation buffers and stream) on one GPU.
Independent clips in flight overlap one clip's launch ramps, prologues, epilogues and tails with another's
steady state; round 2 measured 43.6 k -> 47.7 k frames/s at NE = 2 on the headline shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from playaid_core_amd import synth
from playaid_core_amd.engine import Engine
from playaid_core_amd.parallel import FrameParallelClip

dt_name = sys.argv[1] if len(sys.argv) > 1 else "f32"
n, h, w = (64, 1080, 1920) if dt_name == "f32" else (256, 720, 1280)
sd = synth.make_state_dict(seed=1234)
frames = synth.make_frames_torch(n, h, w, device="cuda")
boxes = torch.from_numpy(synth.make_boxes(n, h, w)).cuda()
for NE, pipe in ((1, True), (2, True), (2, False), (3, False)):
    engs = [Engine(sd, max_batch_frames=n, max_clip_frames=max(n, 64), max_frame_height=h, max_frame_width=w, compute_dtype=dt_name) for _ in range(NE)]
    runners = [FrameParallelClip(e, 7, 3) for e in engs]
    streams = [torch.cuda.Stream() for _ in range(NE)]

    def step(i):
        with torch.cuda.stream(streams[i % NE]):
            runners[i % NE].run(frames, boxes, n, gather=True, pipeline=pipe, reuse_buffers=True)

    for i in range(30):
        step(i)
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(dt_name, NE, "engines, crop pipeline", pipe, ":", round(n * K / dt, 1), "frames/s", round(dt / K * 1e3, 4), "ms/clip", flush=True)
    for e in engs:
        e.close()
