"""Per-layer HIP-event timings of one 64x1080p step (PA_PROFILE_LAYERS=1)."""
import os, sys
os.environ["PA_PROFILE_LAYERS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from playaid_core_amd import synth
from playaid_core_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd = synth.make_state_dict()
eng = Engine(sd, max_batch_frames=n, max_clip_frames=max(n, 64), compute_dtype=os.environ.get("PA_DTYPE", "f32"))
frames = torch.from_numpy(synth.make_frames(8, 1080, 1920)).cuda().repeat((n + 7) // 8, 1, 1, 1)[:n].contiguous()
boxes = torch.from_numpy(synth.make_boxes(n, 1080, 1920)).cuda()
rec = eng.alloc_records(n - 1)
for _ in range(3):
    eng.infer_clip_device(frames, boxes, rec)
torch.cuda.synchronize()
eng.profile_enable(True)
K = 10
for _ in range(K):
    eng.infer_clip_device(frames, boxes, rec)
st = eng.profile_read()
tot = sum(s["total_ms"] for s in st)
print(f"total kernel ms/step {tot / K:.3f}")
for s in st:
    tf = s["flops"] / (s["total_ms"] * 1e-3) / 1e12 if s["flops"] else 0
    print(f"{s['name']:28s} {s['total_ms'] / K * 1000:9.1f} us  {tf:7.1f} TF  {s['bytes'] / (s['total_ms'] * 1e-3) / 1e9:8.1f} GB/s")
