import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from playaid_core_amd import synth
from playaid_core_amd.engine import Engine
from oracle import yolo_crop
h, w = 1080, 1920
n = 6
frames = synth.make_frames(n, h, w, seed=11)
boxes = synth.make_boxes(n, h, w)
boxes[0, 0] = (0.03, 0.05, 0.16, 0.30)
boxes[0, 1] = (0.97, 0.96, 0.15, 0.28)
boxes[1, 0] = (1.6, 0.5, 0.15, 0.3)
boxes[1, 1] = (0.5, 0.5, 0.30, 0.20)
boxes[2, 0] = (0.5, 0.5, 256.5 / w, 200.5 / h)
boxes[2, 1] = (0.4, 0.6, 128.5 / w, 100.5 / h)
boxes[3, 0] = (0.5, 0.5, 384.5 / w, 300.5 / h)
boxes[3, 1] = (0.5, -0.4, 0.15, 0.3)
eng = Engine(synth.make_state_dict(), max_batch_frames=8, max_clip_frames=64)
PAD = int(os.environ.get("PAD", "30"))
boxes[4, 0] = (0.5, 0.5, 300.5 / w, 200.5 / h)   # even d: with PAD=0 the slice is already d x d
boxes[4, 1] = (0.25, 0.5, 256.5 / w, 200.5 / h)
boxes[5, 0] = (0.3, 0.3, 128.5 / w, 100.5 / h)
crops, status = eng.square_crops(frames, boxes, padding=PAD)
for i in range(n):
    for p in range(2):
        ok, ref = yolo_crop.square_crop(frames[i], boxes[i, p], 128, padding=PAD)
        if not ok:
            print(i, p, "ref fail, status", status[i, p]); continue
        d = np.abs(crops[i, p].astype(int) - ref.astype(int))
        bad_rows = np.nonzero(d.max(axis=(1, 2)))[0]
        bad_cols = np.nonzero(d.max(axis=(0, 2)))[0]
        print(i, p, "status", status[i, p], "maxdiff", d.max(), "bad rows", bad_rows[:10], len(bad_rows), "bad cols", bad_cols[:10], len(bad_cols))
