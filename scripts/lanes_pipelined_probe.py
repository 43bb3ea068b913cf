"""Experiment: ClipLanes with each lane's crop stage on its own side stream (FrameParallelClip's two-stream pipeline inside
every lane: four streams) against the committed lanes (crop stage and backbone in stream order). Headline shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from playaid_core_amd import synth
from playaid_core_amd.engine import Engine
from playaid_core_amd.parallel import ClipLanes

n = 64
eng = Engine(synth.make_state_dict(), max_batch_frames=n, max_clip_frames=n)
frames = torch.from_numpy(synth.make_frames(8, 1080, 1920)).cuda().repeat(n // 8, 1, 1, 1).contiguous()
boxes = torch.from_numpy(synth.make_boxes(n, 1080, 1920)).cuda()
lanes = ClipLanes(eng, 7, 3, lanes=int(os.environ.get("LANES", "2")))


def submit(pipeline):
    if lanes._cold and len(lanes.engines) > 1:
        lanes._aligned_start()
    lanes._cold = False
    lane = lanes._next
    lanes._next = (lanes._next + 1) % len(lanes.engines)
    with torch.cuda.stream(lanes.streams[lane]):
        lanes.runners[lane].run(frames, boxes, n, gather=False, pipeline=pipeline, reuse_buffers=True)
    lanes._gate_progress()


def rate(pipeline, clips=240):
    lanes._cold = True
    for _ in range(8):
        submit(pipeline)
    torch.cuda.synchronize()
    lanes.idle()
    t0 = time.perf_counter()
    for _ in range(clips):
        submit(pipeline)
    torch.cuda.synchronize()
    lanes.idle()
    return n * clips / (time.perf_counter() - t0)


lanes.calibrate(frames, boxes, n)
for p in (False, True, False, True):
    print(f"pipeline={p}:", " ".join(f"{rate(p):.0f}" for _ in range(3)), flush=True)
lanes.close()
eng.close()
