"""bf16 conv path (configs[2]) against the fp32 engine: error on the log-probs and step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from playaid_core_amd import synth
from playaid_core_amd.engine import Engine

sd = synth.make_state_dict(seed=1234)
n, h, w = 48, 720, 1280
frames, boxes = synth.make_frames(n, h, w), synth.make_boxes(n, h, w)
e32 = Engine(sd, max_batch_frames=64, max_clip_frames=64, max_frame_height=h, max_frame_width=w)
a = e32.infer_clip(frames, boxes)
e32.close()
e16 = Engine(sd, max_batch_frames=64, max_clip_frames=64, max_frame_height=h, max_frame_width=w, compute_dtype="bf16")
b = e16.infer_clip(frames, boxes)
d = np.abs(a["logp"] - b["logp"])
top2 = np.sort(a["logp"], axis=-1)[..., -2:]
margin = top2[..., 1] - top2[..., 0]
agree = a["action_id"] == b["action_id"]
print("max |dlogp|", d.max(), "mean", d.mean(), "argmax agreement", agree.mean(), "min margin where disagree", margin[~agree].max() if (~agree).any() else None)
print("prob diff max", np.abs(a["prob"] - b["prob"]).max())
e16.close()
for dt in ("f32", "bf16"):
    n = 64
    eng = Engine(sd, max_batch_frames=n, max_clip_frames=64, compute_dtype=dt)
    fr = torch.from_numpy(synth.make_frames(8, 1080, 1920)).cuda().repeat(8, 1, 1, 1).contiguous()
    bx = torch.from_numpy(synth.make_boxes(n, 1080, 1920)).cuda()
    rec = eng.alloc_records(n - 1)
    for _ in range(3):
        eng.infer_clip_device(fr, bx, rec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.infer_clip_device(fr, bx, rec)
    torch.cuda.synchronize()
    print(dt, "ms/step", (time.perf_counter() - t0) * 100)
    eng.profile_enable(True)
    for _ in range(5):
        eng.infer_clip_device(fr, bx, rec)
    for s in eng.profile_read():
        print(f"   {s['name']:24s} {s['total_ms'] / 5 * 1000:9.1f} us")
    eng.close()
